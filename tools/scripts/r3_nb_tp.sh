#!/bin/bash
# persistent kernel workgroup count on a tensor-parallel shard: r3_nb_tp.sh <tp> "<nb ...>"
for rep in 1 2; do for nb in $2; do
  v=$(FTCF_PERSIST_NB=$nb python bench.py --fake-tp $1 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.1f tok/s %.4f ms %s' % (d['value'], d['ms_per_step'], d['tensor_parallel']['decode_path']))")
  echo "fake-tp $1 NB $nb : $v"
done; done
