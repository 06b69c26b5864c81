#!/bin/bash
O=gpurun_out/${1:-r5_small_tiles}; mkdir -p $O
for dt in int8 fp16; do for st in 160 161 241 321; do
  FTCF_GEMM_SMALL_TILES=$st timeout 600 python tools/bench_prefill.py --lens 768,1024,1536,2048 --dtype $dt --reps 3 2>/dev/null | grep prompt_len | python -c "
import sys,json
print('$dt small_tiles=$st', ' '.join(f\"{json.loads(l)['prompt_len']}:{json.loads(l)['prefill_ms']:.2f}\" for l in sys.stdin))" | tee -a $O/sweep.txt
done; done
