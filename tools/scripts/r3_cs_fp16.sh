#!/bin/bash
# control-wave share sweep (P1, P3) for a dtype / batch: r3_cs_fp16.sh <dtype> <batch> "<cs1:cs3 ...>"
for rep in 1 2; do for c in $3; do
  c1=${c%%:*}; c3=${c##*:}
  v=$(FTCF_PERSIST_CS1=$c1 FTCF_PERSIST_CS3=$c3 python bench.py --dtype $1 --batch $2 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f tok/s  launch %.1f us %s' % (d['value'], d['roofline']['avg_launch_us'], d['tensor_parallel']['decode_path']))")
  echo "$1 bs$2 cs $c1,$c3 : $v"
done; done
