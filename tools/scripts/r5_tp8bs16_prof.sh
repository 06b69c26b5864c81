#!/bin/bash
# round 5: kernel trace of one rank's shard of TP 8 at 16 rows on the final tree (pair all-reduce: five launches per layer)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_tp8bs16}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr -o r -- python $R/bench.py --fake-tp 8 --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/bench.json 2> $O/tr.err
python $R/tools/prof_summary.py $(find $O/tr -name "*results.db" | head -1) $O/kernel_stats.txt | head -12 | cut -c1-150
find $O -name "*.db" -delete
