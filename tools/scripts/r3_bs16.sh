#!/bin/bash
# usage: r3_bs16.sh "ENV=VAL,ENV2=VAL ..." (one config per word; '-' = defaults)
for rep in 1 2; do for cfg in $1; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then IFS=','; for kv in $cfg; do export "$kv"; done; fi
    timeout 300 python bench.py --batch ${BATCH:-16} --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))" )
done; done
