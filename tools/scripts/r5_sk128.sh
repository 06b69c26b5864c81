#!/bin/bash
# round 5: split-K tiled GEMM with 128-row tiles (FTCF_GEMM_SK128) against 64-row tiles on short prompt phases
O=gpurun_out/${1:-r5_sk128}; mkdir -p $O
for sk in 0 1; do
  FTCF_GEMM_SK128=$sk timeout 600 python tools/bench_prefill.py --lens 65,96,128,160,192,256,320 --dtype int8 2>/dev/null | grep prompt_len | sed "s/^/sk128=$sk /" | tee -a $O/sweep.txt
done
for mm in 320 640; do
  FTCF_GEMM_SPLITK_MAX_M=$mm timeout 600 python tools/bench_prefill.py --lens 384,512,640 --dtype int8 2>/dev/null | grep prompt_len | sed "s/^/sk128=1 max_m=$mm /" | tee -a $O/sweep.txt
done
for sk in 0 1; do
  FTCF_GEMM_SK128=$sk timeout 600 python tools/bench_prefill.py --lens 65,128,256,320 --dtype fp16 2>/dev/null | grep prompt_len | sed "s/^/sk128=$sk /" | tee -a $O/sweep.txt
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline_shapes.py -q -m gpu -k "gemm or prompt or fuzz" 2>&1 | grep -v "^\[FT\]" | tail -8 | tee $O/pytest.log
