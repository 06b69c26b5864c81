#!/bin/bash
# kernel-trace of the default bench with the separate k_lm_head + k_greedy_decode (FTCF_LM_GREEDY=0)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lmg0
mkdir -p $O
FTCF_LM_GREEDY=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -9 | cut -c1-140
find $O -name "*.db" -delete
