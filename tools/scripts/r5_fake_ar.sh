#!/bin/bash
# round 5: what the overlapped batched TP decode HIDES, modelled on one GPU: one rank's shard of TP 8 with a kernel of N microseconds
# in place of every per-layer all-reduce (FTCF_FAKE_AR_US; the peers are missing), in line (FTCF_DECODE_OVERLAP=0) and overlapped (1)
O=gpurun_out/${1:-r5_fake_ar}; mkdir -p $O
for bs in 32 16 24; do for us in 0 10 20 40; do for v in 0 1; do
  FTCF_FAKE_AR_US=$us FTCF_DECODE_OVERLAP=$v timeout 300 python bench.py --fake-tp 8 --batch $bs --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc 2>/dev/null > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('fake-tp 8 bs $bs all-reduce ${us} us overlap=$v: %.3f ms per step' % d['ms_per_step'])" | tee -a $O/sweep.txt
done; done; done
