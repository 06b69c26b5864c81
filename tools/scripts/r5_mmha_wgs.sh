#!/bin/bash
O=gpurun_out/${1:-r5_mmha_wgs}; mkdir -p $O
for tp in 8 2; do for wg in 640 80 160 320 1280; do
  FTCF_MMHA_WGS=$wg timeout 300 python bench.py --fake-tp $tp --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc 2>/dev/null > $O/b.json
  python -c "import json; d=json.load(open('$O/b.json')); print('fake-tp $tp bs16 FTCF_MMHA_WGS=$wg: %.4f ms per step' % d['ms_per_step'])" | tee -a $O/sweep.txt
done; done
