#!/bin/bash
# round 6: control-wave shares (cs1:cs3) x kernel variants on one rank's shard.  usage: r6_tpcs.sh "<lib ...>" "<tp ...>" "<cs1:cs3 ...>" [outdir]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${4:-r6_tpcs}; mkdir -p $O
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0)))'
for tp in $2; do for lib in $1; do for c in $3; do
  c1=${c%%:*}; c3=${c##*:}
  v=$(FTCF_PERSIST_CS1=$c1 FTCF_PERSIST_CS3=$c3 FTCF_LIB_NAME=libftcf_$lib.so timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --fake-tp $tp 2>/dev/null | python -c "$pp" 2>&1 | tail -1)
  echo "tp$tp $lib cs $c1:$c3 : $v" | tee -a $O/sweep.txt
done; done; done
