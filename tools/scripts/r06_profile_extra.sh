#!/bin/bash
# round 6, second batch: wave-cycle counters of the layer kernel in both layouts of its out-proj / FFN2 stage, kernel traces of one
# rank's shards of TP 2 / 4 (own-group layout) and of bs 16
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06extra; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for own in 1 0; do
  FTCF_PERSIST_OWN=$own timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc_own$own -o p -- python $R/bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-pmc --profile-steps 0 > /dev/null 2> $O/pmc_own$own.err
  python $R/tools/pmc_summary.py $(find $O/pmc_own$own -name "*results.db" | head -1) > $O/pmc_sq_own$own.txt 2>&1
  grep -h persistent $O/pmc_sq_own$own.txt | cut -c1-400
done
for tp in 2 4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_tp$tp -o r -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --fake-tp $tp > /dev/null 2> $O/trace_tp$tp.err
  python $R/tools/prof_summary.py $(find $O/trace_tp$tp -name "*results.db" | head -1) $O/kernel_stats_faketp$tp.txt | head -6
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace16 -o r -- python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > /dev/null 2> $O/trace16.err
python $R/tools/prof_summary.py $(find $O/trace16 -name "*results.db" | head -1) $O/kernel_stats_bs16.txt | head -6
find $O -name "*.db" -delete
ls $O
