#!/bin/bash
# round 6: the rows kernel (256-in / 128-out at 16 / 8 / 4 rows) -- product library against variant libraries, repetitions; rows tests first
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_rows}; mkdir -p $O
[ -z "$NOTEST" ] && timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_fullsize.py -x -q -m gpu -k "rows or bs16" 2>&1 | tail -2
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0)))'
for rep in $(seq 1 ${REPS:-3}); do for lib in "" $LIBS; do for bs in ${BSS:-16 8}; do
  v=$(FTCF_LIB_NAME=libftcf${lib:+_$lib}.so timeout 300 python bench.py --batch $bs --prompt-len 256 --output-len 128 --steps 60 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc 2>$O/err.txt | python -c "$pp" 2>&1 | tail -1)
  echo "bs$bs ${lib:-product} : $v" | tee -a $O/ab.txt
done; done; done
if [ -n "$TL" ]; then
  FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 64 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 >/dev/null 2>&1
  python tools/rows_timeline.py $O/ts.bin 20 > $O/timeline_rows_bs16.txt; rm -f $O/ts.bin; cat $O/timeline_rows_bs16.txt | head -40
fi
