#!/bin/bash
# control-wave shares on a tensor-parallel shard: r3_cs_tp.sh <tp> "<cs1:cs3 ...>"
for rep in 1 2; do for c in $2; do
  c1=${c%%:*}; c3=${c##*:}
  v=$(FTCF_PERSIST_CS1=$c1 FTCF_PERSIST_CS3=$c3 python bench.py --fake-tp $1 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f tok/s %.4f ms' % (d['value'], d['ms_per_step']))")
  echo "fake-tp $1 cs $c1,$c3 : $v"
done; done
