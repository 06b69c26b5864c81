#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/a -o p -- python $R/tools/bench_gemm.py --reps 5 --m ${M:-128} > $O/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --kernel-trace -d $O/b -o p -- python $R/tools/bench_gemm.py --reps 5 --m ${M:-128} > $O/b.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA --kernel-trace -d $O/c -o p -- python $R/tools/bench_gemm.py --reps 5 --m ${M:-128} > $O/c.log 2>&1
for x in a b c; do python $R/tools/pmc_summary.py $(find $O/$x -name "*results.db" | head -1) > $O/pmc_$x.txt 2>&1; grep gemm_tiled $O/pmc_$x.txt; tail -3 $O/$x.log; done
find $O -name "*.db" -delete
