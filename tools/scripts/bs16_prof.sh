#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bs16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp

timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 > $O/bench.json 2> $O/trace.err
echo "trace rc=$?"; cut -c1-250 $O/bench.json
python $R/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -30
find $O -name "*.db" -delete
