#!/bin/bash
O=gpurun_out/${1:-r5_fake_ar3}; mkdir -p $O
for bs in 32 16; do for us in 0 20 40; do for cfg in "0 0 0" "0 1 0" "1 0 0" "1 1 1"; do
  set -- $cfg
  FTCF_FAKE_AR_US=$us FTCF_DECODE_OVERLAP=$1 FTCF_TP_GRAPH=$2 FTCF_DECODE_OVERLAP_GRAPH=$3 timeout 300 python bench.py --fake-tp 8 --batch $bs --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc 2>$O/err.txt > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('fake-tp 8 bs $bs all-reduce ${us} us overlap=$1 graph=$2: %.3f ms per step' % d['ms_per_step'])" 2>/dev/null | tee -a $O/sweep.txt || { echo "bs $bs us $us cfg $cfg FAILED"; tail -3 $O/err.txt; }
done; done; done
