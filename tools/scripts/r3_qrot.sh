#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
cat > /tmp/pp.py <<'P'
import sys, json
d = json.loads(sys.stdin.read())
print("%.1f tok/s  launch %.1f us" % (d["value"], d["roofline"]["avg_launch_us"]))
P
for lib in libftcf.so libftcf_fp3.so; do for q in 0 1 2 3; do
  v=$(FTCF_PERSIST_QROT=$q FTCF_LIB_NAME=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python /tmp/pp.py)
  echo "$lib qrot=$q $v"
done; done
