#!/bin/bash
# SQ breakdown of the prompt-phase kernels (prompt attention, tiled GEMM) from one rocprofv3 --pmc pass over a 1024-token prompt phase
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_prefill; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/p -o p -- python $GRAFT_REPO_ROOT/tools/bench_prefill.py --lens 1024 --reps 2 > $O/log 2>&1
echo "rc=$?"; tail -3 $O/log
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O/p -name "*results.db" | head -1) > $O/pmc.txt 2>&1
grep -E "context_attention|gemm_tiled|qkv_bias|residual_dual" $O/pmc.txt | cut -c1-400
find $O -name "*.db" -delete
