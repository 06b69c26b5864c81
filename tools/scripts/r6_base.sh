#!/bin/bash
# round 6 baseline of the tree: headline driver-style, one rank's shards of TP 2 / 4 / 8 (+ bs 16), timelines TP 1 / TP 8, bs 16, smoke
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_base}; mkdir -p $O
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-400 $O/bench_driver.json
for tp in 2 4 8; do
  timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp > $O/bench_faketp$tp.json 2>/dev/null
  python -c "import json,sys; d=json.load(open('$O/bench_faketp$tp.json')); print('faketp $tp: %.1f tok/s, launch %.1f us' % (d['value'], d['roofline']['avg_launch_us']))"
done
timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp 8 > $O/bench_faketp8_bs16.json 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_faketp8_bs16.json')); print('faketp 8 bs16: %.3f ms per step' % d['ms_per_step'])"
timeout 600 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > $O/bench_bs16.json 2> $O/bench_bs16.err; cut -c1-200 $O/bench_bs16.json
FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 >/dev/null 2>&1
python tools/ps_timeline.py $O/ts.bin 20 > $O/timeline_tp1.txt; rm -f $O/ts.bin
for tp in 2 8; do
FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 --fake-tp $tp >/dev/null 2>&1
python tools/ps_timeline.py $O/ts.bin 20 > $O/timeline_faketp$tp.txt; rm -f $O/ts.bin
done
tail -4 $O/timeline_tp1.txt $O/timeline_faketp8.txt
