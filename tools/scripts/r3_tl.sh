#!/bin/bash
# timelines of kernel variants: tools/scripts/r3_tl.sh "<lib names>" [layer]
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
for n in $1; do
  FTCF_PERSIST_TS=gpurun_out/r3/ts_$n.bin FTCF_LIB_NAME=libftcf_$n.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n %.1f tok/s' % d['value'])"
  python tools/ps_timeline.py gpurun_out/r3/ts_$n.bin ${2:-20} > gpurun_out/r3/tl_$n.txt
  rm -f gpurun_out/r3/ts_$n.bin
done
