#!/bin/bash
# round 5: variant libraries of the rows kernel at bs = 16: r5_rows_var.sh <outdir> <lib or -> ... ("-" = the product library)
O=gpurun_out/${1:-r5_rows_var}; mkdir -p $O; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset FTCF_LIB_NAME; else export FTCF_LIB_NAME=libftcf_$v.so; fi
  timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/b_$v.json 2> $O/b_$v.err
  python -c "import sys,json; d=json.loads(open('$O/b_$v.json').read()); print('$v', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us')" || tail -3 $O/b_$v.err
  if [ -n "$TS" ]; then
    FTCF_PERSIST_TS=$O/ts_$v.bin timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 64 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-pmc > /dev/null 2> $O/ts_$v.err
    python tools/rows_timeline.py $O/ts_$v.bin 20 > $O/timeline_$v.txt; tail -14 $O/timeline_$v.txt | head -3
  fi
done
