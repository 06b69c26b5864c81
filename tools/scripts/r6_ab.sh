#!/bin/bash
# round 6 A/B on one box: CFGS="<name>:<ENV=V>:<ENV=V> ..." [TPS="0 2 4 8"] [TL="<name>:<tp>:<ENV=V> ..."] r6_ab.sh -- every configuration as
# the headline request (tp 0) or as one rank's shard of a TP = n job (bench.py --fake-tp n), two repetitions; TL adds timelines (r6_tl.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_ab}; mkdir -p $O
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0)))'
run() { local name=$1 tp=$2; shift 2
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  v=$(env "$@" timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc $tpflag 2>$O/err.txt | python -c "$pp" 2>&1 | tail -1)
  echo "tp$tp $name : $v" | tee -a $O/ab.txt; }
for rep in 1 2; do for tp in ${TPS:-0 2 4 8}; do
  for c in $CFGS; do
    name=$(echo $c | cut -d: -f1); envs=$(echo $c | cut -d: -f2- | tr ':' ' ')
    run $name $tp $envs
  done
done; done
[ -n "$TL" ] && OUT=${OUT:-r6_ab} bash tools/scripts/r6_tl.sh "$TL" > /dev/null && for f in $O/tl_*.txt; do echo "== $f"; head -19 $f | tail -17; done
