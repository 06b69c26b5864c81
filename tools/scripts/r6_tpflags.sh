#!/bin/bash
# round 6: sweep of the persistent kernel's compile-time variants on the TENSOR-PARALLEL instantiation (one rank's shard, --fake-tp)
# usage: r6_tpflags.sh "<lib> <lib> ..." "<tp> <tp> ..." [outdir] ; libs are lib/libftcf_<lib>.so
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${3:-r6_tpflags}; mkdir -p $O
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0)))'
for tp in $2; do for lib in $1; do
  v=$(FTCF_LIB_NAME=libftcf_$lib.so timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --fake-tp $tp 2>$O/err_${lib}_$tp.txt | python -c "$pp" 2>&1 | tail -1)
  echo "tp$tp $lib : $v" | tee -a $O/sweep.txt
done; done
if [ -n "$TL" ]; then for tp in $2; do for lib in $TL; do
  FTCF_PERSIST_TS=$O/ts.bin FTCF_LIB_NAME=libftcf_$lib.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 --fake-tp $tp >/dev/null 2>&1
  python tools/ps_timeline.py $O/ts.bin 20 > $O/tl_${lib}_tp$tp.txt; rm -f $O/ts.bin
done; done; fi
