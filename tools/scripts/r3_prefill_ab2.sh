#!/bin/bash
# usage: r3_prefill_ab2.sh "<lib:splitk ...>" [lens]
for rep in 1 2; do for x in $1; do
  l=${x%%:*}; sk=${x##*:}
  echo "== $l splitk $sk"
  FTCF_GEMM_SPLITK=$sk FTCF_LIB_NAME=$l timeout 300 python tools/bench_prefill.py --reps 3 --lens ${2:-65,128,256} 2>&1 | grep prompt_len | cut -c1-95
done; done
