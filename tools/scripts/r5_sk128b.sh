#!/bin/bash
O=gpurun_out/${1:-r5_sk128b}; mkdir -p $O
for dt in fp16 int8; do
for cfg in "100000 320" "100000 640" "64 640"; do
  set -- $cfg
  FTCF_GEMM_SK128_MIN_M=$1 FTCF_GEMM_SPLITK_MAX_M=$2 timeout 600 python tools/bench_prefill.py --lens 160,192,224,256,288,320,352,384,448,512,640 --dtype $dt --reps 3 2>/dev/null | grep prompt_len | python -c "
import sys,json
print('$dt sk128_min=$1 max_m=$2', ' '.join(f\"{json.loads(l)['prompt_len']}:{json.loads(l)['prefill_ms']:.2f}\" for l in sys.stdin))" | tee -a $O/sweep.txt
done; done
