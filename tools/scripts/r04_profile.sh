#!/bin/bash
# measurement batch: default bench, driver-style bench, kernel-trace stats of the same command, PMC passes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "driver-style rc=$?"; cut -c1-300 $O/bench_driver.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --profile-steps 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --profile-steps 0 > $O/pmc_write.json 2> $O/pmc_write.err
echo "write rc=$?"
python $R/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -14
python $R/tools/pmc_summary.py $(find $O/pmc_fetch -name "*results.db" | head -1) > $O/pmc_fetch.txt 2>&1
python $R/tools/pmc_summary.py $(find $O/pmc_write -name "*results.db" | head -1) > $O/pmc_write.txt 2>&1
grep persistent $O/pmc_fetch.txt $O/pmc_write.txt
# fp16 and bs=16 bench lines (BASELINE configs 2 and 5's single-GPU regime)
timeout 600 python $R/bench.py --dtype fp16 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
timeout 600 python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_bs16.json 2> $O/bench_bs16.err
cut -c1-200 $O/bench_fp16.json; cut -c1-200 $O/bench_bs16.json
find $O -name "*.db" -size +40M -delete
find $O -name "*.db" -delete
ls -la $O
# one rank's shard of TP 2 / 4 / 8 (kernel-side scaling without xGMI), bs 1 and bs 16
cd $R
for tp in 2 4 8; do
  timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp > $O/bench_faketp$tp.json 2>/dev/null
  python -c "import json,sys; d=json.load(open('$O/bench_faketp$tp.json')); print('faketp $tp: %.1f tok/s, launch %.1f us' % (d['value'], d['roofline']['avg_launch_us']))"
done
timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp 8 > $O/bench_faketp8_bs16.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_faketp8_bs16.json')); print('faketp 8 bs16: %.3f ms per step' % d['ms_per_step'])"
for tp in 0 8; do
  FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 --fake-tp $tp >/dev/null 2>&1
  python tools/ps_timeline.py $O/ts.bin 20 > $O/timeline_tp$tp.txt; rm -f $O/ts.bin
done
ls -la $O
