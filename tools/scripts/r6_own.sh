#!/bin/bash
# round 6: P3 in the own-group layout against the K-piece form (FTCF_PERSIST_OWN=0) on one box: parity tests that run the 13B shapes,
# then A/B of the headline and of one rank's shards, timelines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_own}; mkdir -p $O
if [ -z "$NOTEST" ]; then
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "oracle or tensor_parallel_one_row or full_size_properties or long_context" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_tp_local.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3
fi
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us  %s" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0), d["tensor_parallel"]["decode_path"]))'
for rep in 1 2; do for tp in 0 2 4 8; do for own in 1 0; do
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  v=$(FTCF_PERSIST_OWN=$own $EXTRA timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc $tpflag 2>$O/err.txt | python -c "$pp" 2>&1 | tail -1)
  echo "tp$tp own=$own : $v" | tee -a $O/ab.txt
done; done; done
for tp in 0 8; do for own in 1 0; do
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  FTCF_PERSIST_OWN=$own FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 $tpflag >/dev/null 2>&1
  python tools/ps_timeline.py $O/ts.bin 20 > $O/tl_tp${tp}_own$own.txt; rm -f $O/ts.bin
done; done
tail -n 3 $O/tl_*.txt
