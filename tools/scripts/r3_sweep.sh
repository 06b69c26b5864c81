#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
lib=$1
cat > /tmp/pp.py <<'P'
import sys, json
d = json.loads(sys.stdin.read())
print("%.1f tok/s  launch %.1f us" % (d["value"], d["roofline"]["avg_launch_us"]))
P
for cs1 in 10 12 14; do for cs3 in 6 8 10 12; do
  v=$(FTCF_PERSIST_CS1=$cs1 FTCF_PERSIST_CS3=$cs3 FTCF_LIB_NAME=libftcf_$lib.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python /tmp/pp.py)
  echo "cs1=$cs1 cs3=$cs3 $v"
done; done
