#!/bin/bash
# round 6: own-group layout, second form (LDS copy of the layer table): product library own=1 / own=0, variant libraries, control shares
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_own2}; mkdir -p $O
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0)))'
run() { # name tp env...
  local name=$1 tp=$2; shift 2
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  v=$(env "$@" timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc $tpflag 2>$O/err.txt | python -c "$pp" 2>&1 | tail -1)
  echo "tp$tp $name : $v" | tee -a $O/ab.txt
}
tl() { # name tp env...
  local name=$1 tp=$2; shift 2
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  env "$@" FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 $tpflag >/dev/null 2>&1
  python tools/ps_timeline.py $O/ts.bin 20 > $O/tl_tp${tp}_$name.txt; rm -f $O/ts.bin
}
for rep in 1 2; do for tp in 0 2 4 8; do
  run own0 $tp FTCF_PERSIST_OWN=0
  run own1 $tp FTCF_PERSIST_OWN=1
  run own1cs7 $tp FTCF_PERSIST_CS3=7
  run v5ce_cs7 $tp FTCF_LIB_NAME=libftcf_v5ce.so FTCF_PERSIST_CS3=7
  run v5ce_cs10 $tp FTCF_LIB_NAME=libftcf_v5ce.so
done; done
tl own1 0 FTCF_PERSIST_OWN=1
tl own1 8 FTCF_PERSIST_OWN=1
tl v5ce_cs7 0 FTCF_LIB_NAME=libftcf_v5ce.so FTCF_PERSIST_CS3=7
tail -n 3 $O/tl_*.txt
