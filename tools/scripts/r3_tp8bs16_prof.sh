#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tp8bs16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/bench.py --fake-tp ${1:-8} --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 > $O/bench.json 2> $O/trace.err
python $R/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -16
python $R/tools/prof_timeline.py $(find $O/trace -name "*results.db" | head -1) 12000 16
find $O -name "*.db" -delete
