#!/bin/bash
# kernel timeline of a few decode tokens of the headline request (gaps between the per-token kernels)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3tok
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 > $O/bench.json 2> $O/err.txt
DB=$(find $O/trace -name "*results.db" | head -1)
python $R/tools/prof_timeline.py $DB 3000 24
find $O -name "*.db" -delete
