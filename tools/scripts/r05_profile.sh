#!/bin/bash
# round 5 measurement batch: default bench, driver-style bench, kernel-trace stats of the same command, PMC passes; bs = 16 on the rows
# kernel with its trace and counters; fp16; top_k = 50; one rank's shards of TP 2 / 4 / 8
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "driver-style rc=$?"; cut -c1-300 $O/bench_driver.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err
echo "trace rc=$?"
python $R/tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | head -8
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --profile-steps 0 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
python $R/tools/pmc_summary.py $(find $O/pmc_fetch -name "*results.db" | head -1) > $O/pmc_fetch.txt 2>&1
grep persistent $O/pmc_fetch.txt | cut -c1-200
find $O -name "*.db" -delete
# bs = 16 (BASELINE config 5's single-GPU regime): bench line, kernel trace, counters
timeout 600 python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_bs16.json 2> $O/bench_bs16.err
cut -c1-200 $O/bench_bs16.json
FTCF_ROWS=0 timeout 600 python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > $O/bench_bs16_general.json 2> /dev/null
cut -c1-120 $O/bench_bs16_general.json
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace16 -o r -- python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > /dev/null 2> $O/trace16.err
python $R/tools/prof_summary.py $(find $O/trace16 -name "*results.db" | head -1) $O/kernel_stats_bs16.txt | head -8
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc16 -o p -- python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-e2e --no-cpu-baseline --profile-steps 0 > /dev/null 2> $O/pmc16.err
python $R/tools/pmc_summary.py $(find $O/pmc16 -name "*results.db" | head -1) > $O/pmc_fetch_bs16.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d $O/pmc16s -o p -- python $R/bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-e2e --no-cpu-baseline --profile-steps 0 > /dev/null 2> $O/pmc16s.err
python $R/tools/pmc_summary.py $(find $O/pmc16s -name "*results.db" | head -1) > $O/pmc_sq_bs16.txt 2>&1
grep -h decode_rows $O/pmc_fetch_bs16.txt $O/pmc_sq_bs16.txt | cut -c1-400
find $O -name "*.db" -delete
# fp16 weights (config 2), the harness default top_k = 50 next to greedy
timeout 600 python $R/bench.py --dtype fp16 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
cut -c1-160 $O/bench_fp16.json
for k in 1 50; do
  timeout 300 python $R/bench.py --top-k $k --steps 200 --warmup 8 --no-cpu-baseline --no-e2e --no-pmc > $O/bench_topk$k.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/bench_topk$k.json')); print('top_k $k: %.1f tok/s %.4f ms' % (d['value'], d['ms_per_step']))"
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/tracek -o r -- python $R/bench.py --top-k 50 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > /dev/null 2> $O/tracek.err
python $R/tools/prof_summary.py $(find $O/tracek -name "*results.db" | head -1) $O/kernel_stats_topk50.txt | head -6
find $O -name "*.db" -delete
# one rank's shard of TP 2 / 4 / 8 (kernel-side scaling without xGMI), bs 1 and bs 16
cd $R
for tp in 2 4 8; do
  timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp > $O/bench_faketp$tp.json 2>/dev/null
  python -c "import json,sys; d=json.load(open('$O/bench_faketp$tp.json')); print('faketp $tp: %.1f tok/s, launch %.1f us' % (d['value'], d['roofline']['avg_launch_us']))"
done
timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp 8 > $O/bench_faketp8_bs16.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_faketp8_bs16.json')); print('faketp 8 bs16: %.3f ms per step' % d['ms_per_step'])"
FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 >/dev/null 2>&1
python tools/ps_timeline.py $O/ts.bin 20 > $O/timeline_tp1.txt; rm -f $O/ts.bin
FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 64 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 >/dev/null 2>&1
python tools/rows_timeline.py $O/ts.bin 20 > $O/timeline_rows_bs16.txt; rm -f $O/ts.bin
ls -la $O
# prompt phase by prompt length (int8, fp16), batched TP decode with / without the two-micro-batch overlap on one rank's shard
timeout 600 python tools/bench_prefill.py --lens 17,33,48,64,65,96,128,160,192,256,320,384,512,640,768,1024,2048 --dtype int8 2>/dev/null | grep prompt_len > $O/prefill_sweep_int8.txt
timeout 600 python tools/bench_prefill.py --lens 65,128,192,256,320,384,512,640,768,1024 --dtype fp16 2>/dev/null | grep prompt_len > $O/prefill_sweep_fp16.txt
cat $O/prefill_sweep_int8.txt $O/prefill_sweep_fp16.txt | cut -c1-120
for bs in 16 32; do for v in 0 1; do
  FTCF_DECODE_OVERLAP=$v timeout 300 python bench.py --fake-tp 8 --batch $bs --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/bench_faketp8_bs${bs}_overlap$v.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/bench_faketp8_bs${bs}_overlap$v.json')); print('faketp 8 bs $bs overlap $v: %.3f ms per step' % d['ms_per_step'], d['tensor_parallel']['decode_overlap'])"
done; done
