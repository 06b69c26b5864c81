#!/bin/bash
O=gpurun_out/${1:-r5_sk_target}; mkdir -p $O
run() {  # name, lens, dtype, env...
  name=$1; lens=$2; dt=$3; shift 3
  env "$@" timeout 600 python tools/bench_prefill.py --lens $lens --dtype $dt --reps 3 2>/dev/null | grep prompt_len | python -c "
import sys,json
print('$dt $name', ' '.join(f\"{json.loads(l)['prompt_len']}:{json.loads(l)['prefill_ms']:.2f}\" for l in sys.stdin))" | tee -a $O/sweep.txt
}
L1=65,128,192,256,320
L2=512,640,768,1024,2048
run "default" $L1 int8 X=1
run "target384" $L1 int8 FTCF_GEMM_SPLITK=384
run "target512" $L1 int8 FTCF_GEMM_SPLITK=512
run "default" $L2 int8 X=1
run "target512_max2048" $L2 int8 FTCF_GEMM_SPLITK=512 FTCF_GEMM_SPLITK_MAX_M=2048
run "target384_max2048" $L2 int8 FTCF_GEMM_SPLITK=384 FTCF_GEMM_SPLITK_MAX_M=2048
run "default" $L2 fp16 X=1
run "target512_max2048" $L2 fp16 FTCF_GEMM_SPLITK=512 FTCF_GEMM_SPLITK_MAX_M=2048
