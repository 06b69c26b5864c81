#!/bin/bash
# A/B of kernel variants on one box: tools/scripts/r3_ab.sh "<lib names>" [env assignments...]
R=$GRAFT_REPO_ROOT; cd $R
libs=$1; shift
cat > /tmp/pp.py <<'P'
import sys, json
d = json.loads(sys.stdin.read())
print("%.1f tok/s  launch %.1f us  %s" % (d["value"], d["roofline"]["avg_launch_us"], d["tensor_parallel"]["decode_path"]))
P
for rep in 1 2; do for n in $libs; do
  v=$(env "$@" FTCF_LIB_NAME=libftcf_$n.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python /tmp/pp.py)
  echo "$n $* : $v"
done; done
