#!/bin/bash
# round 5: kernel trace of the overlapped batched TP decode on one TP 8 shard (32 rows): which streams ran what, and how much of it concurrently
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5_ovtrace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
FTCF_DECODE_OVERLAP=$v timeout 600 rocprofv3 --kernel-trace -d $O/tr$v -o r -- python $R/bench.py --fake-tp 8 --batch 32 --prompt-len 256 --output-len 64 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-pmc > /dev/null 2> $O/tr$v.err
python $R/tools/prof_overlap.py $(find $O/tr$v -name "*results.db" | head -1) | sed "s#$O/##" | tee $O/overlap$v.txt
done
find $O -name "*.db" -delete
