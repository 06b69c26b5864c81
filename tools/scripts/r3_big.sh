#!/bin/bash
g() { python tools/bench_gemm.py --m $1 $2 2>/dev/null| python -c "
import sys,json
print('m=$1 $2', ' '.join('%s %.1f'%(json.loads(l)['gemm'],json.loads(l)['us']) for l in sys.stdin))"; }
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
for rep in 1 2; do for d in 2 0; do
  echo "== big deep $d"; export FTCF_GEMM_BIG_DEEP=$d
  g 1024; g 2048; g 1024 --fp16
  timeout 300 python tools/bench_prefill.py --reps 3 --lens 384,512,1024,2048,4096 2>&1 | grep prompt_len | cut -c1-95
  timeout 300 python tools/bench_prefill.py --dtype fp16 --reps 3 --lens 512,1024,2048 2>&1 | grep prompt_len | cut -c1-95
done; done
