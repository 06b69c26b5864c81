#!/bin/bash
# round 5: prompt-phase time by prompt length (int8 + fp16), and the kernel breakdown of a 128-token prompt
O=gpurun_out/${1:-r5_prefill}; mkdir -p $O
timeout 600 python tools/bench_prefill.py --lens 17,33,48,64,65,96,128,160,192,256,320,384,512,1024,2048 --dtype int8 2>/dev/null | grep prompt_len > $O/sweep_int8.txt
cat $O/sweep_int8.txt
timeout 600 python tools/bench_prefill.py --lens 65,128,256,512,1024 --dtype fp16 2>/dev/null | grep prompt_len > $O/sweep_fp16.txt
cat $O/sweep_fp16.txt
cd /tmp && export TMPDIR=/tmp
for S in 128 256; do
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace$S -o r -- python $GRAFT_REPO_ROOT/tools/bench_prefill.py --lens $S --dtype int8 --reps 8 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $GRAFT_REPO_ROOT/$O/trace$S -name "*.db" | head -1) 2>/dev/null | head -16 | cut -c1-160 > $GRAFT_REPO_ROOT/$O/kernels_S$S.txt
cat $GRAFT_REPO_ROOT/$O/kernels_S$S.txt
rm -rf $GRAFT_REPO_ROOT/$O/trace$S
done
