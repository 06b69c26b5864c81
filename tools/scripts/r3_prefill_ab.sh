#!/bin/bash
# usage: r3_prefill_ab.sh "<lib names>" [lens] [gemm m list]
g() { python tools/bench_gemm.py --m $1 2>/dev/null| python -c "
import sys,json
print('m=$1', ' '.join('%s %.1f'%(json.loads(l)['gemm'],json.loads(l)['us']) for l in sys.stdin))"; }
for rep in 1 2; do for l in $1; do
  echo "== $l"
  export FTCF_LIB_NAME=$l
  timeout 300 python tools/bench_prefill.py --reps 3 --lens ${2:-65,128,256} 2>&1 | grep prompt_len | cut -c1-95
  for m in ${3:-128}; do g $m; done
done; done
