#!/bin/bash
# round 5: one rank's shard of TP 8 at bs = 16 with / without the rows kernel (timing aid)
O=gpurun_out/${1:-r5_faketp}; mkdir -p $O
for v in 1 0; do
  FTCF_ROWS=$v timeout 300 python bench.py --fake-tp 8 --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/b_rows$v.json 2> $O/b_rows$v.err
  python -c "import sys,json; d=json.loads(open('$O/b_rows$v.json').read()); print('fake-tp 8 bs 16 rows=$v', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step')" || tail -3 $O/b_rows$v.err
done
