#!/bin/bash
# round 6 measurement batch: default bench, driver-style bench, kernel-trace stats of the same command, PMC pass; the own-group layout
# against the K-piece form (FTCF_PERSIST_OWN=0) on the headline and on one rank's shards of TP 2 / 4 / 8; bs = 16 on the rows kernel;
# fp16; timelines; prompt-phase sweep
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "driver-style rc=$?"; cut -c1-300 $O/bench_driver.json
FTCF_PERSIST_OWN=0 timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_own0.json 2> /dev/null
echo "driver-style, K pieces:"; cut -c1-200 $O/bench_driver_own0.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err
echo "trace rc=$?"
python $R/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -8
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --profile-steps 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
python $R/tools/pmc_summary.py $(find $O/pmc_fetch -name "*results.db" | head -1) > $O/pmc_fetch.txt 2>&1
grep persistent $O/pmc_fetch.txt | cut -c1-200
find $O -name "*.db" -delete
cd $R
# one rank's shard of TP 2 / 4 / 8 (kernel-side scaling without xGMI): default (auto) and both layouts forced
for tp in 2 4 8; do for own in 2 0 1; do
  FTCF_PERSIST_OWN=$own timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp > $O/bench_faketp${tp}_own$own.json 2>/dev/null
  python -c "import json,sys; d=json.load(open('$O/bench_faketp${tp}_own$own.json')); print('faketp $tp own $own: %.1f tok/s, launch %.1f us' % (d['value'], d['roofline']['avg_launch_us']))"
done; done
timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp 8 > $O/bench_faketp8_bs16.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_faketp8_bs16.json')); print('faketp 8 bs16: %.3f ms per step' % d['ms_per_step'])"
# bs = 16 (BASELINE config 5's single-GPU regime), fp16 weights (config 2)
timeout 600 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_bs16.json 2> $O/bench_bs16.err
cut -c1-200 $O/bench_bs16.json
timeout 600 python bench.py --dtype fp16 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
cut -c1-160 $O/bench_fp16.json
# timelines (layer 20): headline in both layouts, TP 2 / 8 shards
OUT=r06final bash tools/scripts/r6_tl.sh "own:0:FTCF_PERSIST_OWN=1 pieces:0:FTCF_PERSIST_OWN=0 own:2:FTCF_PERSIST_OWN=1 pieces:2:FTCF_PERSIST_OWN=0 own:8:FTCF_PERSIST_OWN=1 pieces:8:FTCF_PERSIST_OWN=0" > /dev/null
# prompt phase by prompt length
timeout 600 python tools/bench_prefill.py --lens 17,33,64,65,128,192,256,320,384,512,768,1024,2048 --dtype int8 2>/dev/null | grep prompt_len > $O/prefill_sweep_int8.txt
cut -c1-120 $O/prefill_sweep_int8.txt
ls $O
