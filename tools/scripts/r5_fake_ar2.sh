#!/bin/bash
O=gpurun_out/${1:-r5_fake_ar2}; mkdir -p $O
for q in 4 8 16; do for us in 20 40; do for v in 0 1; do
  GPU_MAX_HW_QUEUES=$q FTCF_FAKE_AR_US=$us FTCF_DECODE_OVERLAP=$v timeout 300 python bench.py --fake-tp 8 --batch 32 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc 2>/dev/null > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('GPU_MAX_HW_QUEUES=$q fake-tp 8 bs 32 all-reduce ${us} us overlap=$v: %.3f ms per step' % d['ms_per_step'])" | tee -a $O/sweep.txt
done; done; done
