#!/bin/bash
for b in 24 32 48 64; do for mr in 64 16; do
  echo "== batch $b smallm_max_rows $mr"
  FTCF_SMALLM_MAX_ROWS=$mr timeout 300 python bench.py --batch $b --prompt-len 128 --output-len 48 --steps 24 --warmup 4 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('prefill_ms'))"
done; done
