#!/bin/bash
O=gpurun_out/${1:-r5_lonesk}; mkdir -p $O
for v in 0 2 0 2 4; do
  FTCF_GEMM_LONE_SK=$v timeout 600 python tools/bench_prefill.py --lens 896,1024 --dtype int8 --reps 3 2>/dev/null | grep prompt_len | python -c "
import sys,json
print('int8 lone_sk=$v', ' '.join(f\"{json.loads(l)['prompt_len']}:{json.loads(l)['prefill_ms']:.2f}\" for l in sys.stdin))" | tee -a $O/sweep.txt
done
FTCF_GEMM_LONE_SK=2 timeout 900 python -m pytest tests/test_gpu_headline_shapes.py -q -m gpu -k "gemm_at or prompt" 2>&1 | grep -v "^\[FT\]" | tail -5 | tee $O/pytest.log
