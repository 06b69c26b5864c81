#!/bin/bash
O=gpurun_out/${1:-r5_sk128c}; mkdir -p $O
for dt in fp16 int8; do
for cfg in "-1" "1024"; do
  if [ "$cfg" = "-1" ]; then unset FTCF_GEMM_SPLITK_MAX_M; else export FTCF_GEMM_SPLITK_MAX_M=$cfg; export FTCF_GEMM_SK128_MIN_M=192; fi
  timeout 600 python tools/bench_prefill.py --lens 65,128,256,384,512,640,768,1024 --dtype $dt --reps 3 2>/dev/null | grep prompt_len | python -c "
import sys,json
print('$dt max_m=$cfg', ' '.join(f\"{json.loads(l)['prompt_len']}:{json.loads(l)['prefill_ms']:.2f}\" for l in sys.stdin))" | tee -a $O/sweep.txt
  unset FTCF_GEMM_SK128_MIN_M
done; done
