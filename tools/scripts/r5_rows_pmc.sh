#!/bin/bash
# round 5: SQ + fetch counters of the rows kernel at bs = 16 (two rocprofv3 --pmc passes, kernel trace only)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_rows_pmc}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --batch 16 --prompt-len 256 --output-len 48 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/p1 -o p -- $B > $O/log1 2>&1
echo "rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O/p1 -name "*results.db" | head -1) > $O/pmc_sq.txt 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/p2 -o p -- $B > $O/log2 2>&1
echo "rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O/p2 -name "*results.db" | head -1) > $O/pmc_fetch.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $O/p3 -o p -- $B > $O/log3 2>&1
echo "rc=$?"
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O/p3 -name "*results.db" | head -1) > $O/pmc_sq2.txt 2>&1
grep -h "decode_rows" $O/pmc_sq.txt $O/pmc_fetch.txt $O/pmc_sq2.txt | cut -c1-700
find $O -name "*.db" -delete
