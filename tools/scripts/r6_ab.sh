#!/bin/bash
# round 6 A/B on one box: r6_ab.sh "<cfg> <cfg> ..." [reps] where cfg = lib[:ENV=V[:ENV=V...]] (lib "" = the product library);
# FAKE_TP=<n> runs every cfg as one rank's shard of a TP = n job; TL=1 adds a timeline (layer 20) per cfg
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_ab}; mkdir -p $O
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us  %s" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0), d["tensor_parallel"]["decode_path"]))'
tpflag=""; [ -n "$FAKE_TP" ] && tpflag="--fake-tp $FAKE_TP"
for rep in $(seq 1 ${2:-2}); do for c in $1; do
  lib=${c%%:*}; envs=$(echo "${c#*:}" | tr ':' ' '); [ "$envs" = "$c" ] && envs=""
  v=$(env $envs FTCF_LIB_NAME=libftcf${lib:+_$lib}.so timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc $tpflag 2>$O/err.txt | python -c "$pp" 2>&1 | tail -1)
  echo "${FAKE_TP:+tp$FAKE_TP }$c : $v" | tee -a $O/ab.txt
done; done
if [ -n "$TL" ]; then for c in $1; do
  lib=${c%%:*}; envs=$(echo "${c#*:}" | tr ':' ' '); [ "$envs" = "$c" ] && envs=""
  n=$(echo $c | tr ':=' '__')
  env $envs FTCF_PERSIST_TS=$O/ts.bin FTCF_LIB_NAME=libftcf${lib:+_$lib}.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 $tpflag >/dev/null 2>&1
  python tools/ps_timeline.py $O/ts.bin 20 > $O/tl_${n:-product}${FAKE_TP:+_tp$FAKE_TP}.txt; rm -f $O/ts.bin
done; fi
