#!/bin/bash
O=gpurun_out/${1:-r5_ks2}; mkdir -p $O
for dt in int8 fp16; do for v in 0 1 0 1; do
  FTCF_GEMM_KS2=$v timeout 600 python tools/bench_prefill.py --lens 896,1024 --dtype $dt --reps 3 2>/dev/null | grep prompt_len | python -c "
import sys,json
print('$dt ks2=$v', ' '.join(f\"{json.loads(l)['prompt_len']}:{json.loads(l)['prefill_ms']:.2f}\" for l in sys.stdin))" | tee -a $O/sweep.txt
done; done
FTCF_GEMM_KS2=1 timeout 900 python -m pytest tests/test_gpu_headline_shapes.py -q -m gpu -k "gemm_at or prompt" 2>&1 | grep -v "^\[FT\]" | tail -5 | tee $O/pytest.log
