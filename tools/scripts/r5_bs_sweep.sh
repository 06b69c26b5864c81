#!/bin/bash
# round 5: batch sizes 3..16 on the rows kernel and on the general path (256-in / 64-out), and top_k = 50 next to greedy at bs = 1
O=gpurun_out/${1:-r5_bs_sweep}; mkdir -p $O
for b in 3 4 8 12 16; do
  for v in 1 0; do
    FTCF_ROWS=$v timeout 300 python bench.py --batch $b --prompt-len 256 --output-len 64 --steps 24 --warmup 4 --no-cpu-baseline --no-e2e --no-pmc > $O/b${b}_rows$v.json 2> $O/b${b}_rows$v.err
    python -c "import sys,json; d=json.loads(open('$O/b${b}_rows$v.json').read()); print('bs $b rows=$v', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step')" || tail -3 $O/b${b}_rows$v.err
  done
done
for k in 1 50; do
  timeout 300 python bench.py --top-k $k --steps 200 --warmup 8 --no-cpu-baseline --no-e2e --no-pmc > $O/topk$k.json 2> $O/topk$k.err
  python -c "import sys,json; d=json.loads(open('$O/topk$k.json').read()); print('bs 1 top_k $k', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step')" || tail -3 $O/topk$k.err
done
