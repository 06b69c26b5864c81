#!/bin/bash
# round-4 baseline on this box: TP=1 / fake-TP 2 / fake-TP 8 tok/s + per-layer stamps of each
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  launch %.1f us  %s" % (d["value"], (d["roofline"].get("avg_launch_us") or 0), d["tensor_parallel"]["decode_path"]))'
for tp in 0 2 8; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp 2>gpurun_out/r4/err_$tp.log | python -c "$pp" | sed "s/^/faketp=$tp /"
  FTCF_PERSIST_TS=gpurun_out/r4/ts_$tp.bin python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 --fake-tp $tp 2>/dev/null | python -c "$pp" | sed "s/^/faketp=$tp (stamps) /"
  python tools/ps_timeline.py gpurun_out/r4/ts_$tp.bin 20 > gpurun_out/r4/tl_base_tp$tp.txt
  rm -f gpurun_out/r4/ts_$tp.bin
done
cat gpurun_out/r4/tl_base_tp8.txt
