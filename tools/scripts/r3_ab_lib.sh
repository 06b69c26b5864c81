#!/bin/bash
# A/B of whole libraries on the headline bench: r3_ab_lib.sh "<lib file names>"
for rep in 1 2 3; do for l in $1; do
  v=$(FTCF_LIB_NAME=$l python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%.2f tok/s  %.4f ms  launch %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))")
  echo "$l : $v"
done; done
