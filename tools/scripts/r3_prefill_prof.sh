#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3pf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/tools/bench_prefill.py --lens ${1:-128} --reps 4 > $O/out.txt 2> $O/err.txt
python $R/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -40
find $O -name "*.db" -delete
