#!/bin/bash
# A/B on one box: tools/scripts/r4_ab.sh "<cfg> <cfg> ..." where cfg = lib[:ENV=V[:ENV=V...]] ; two interleaved passes; then a timeline of
# every cfg (layer 20).  FAKE_TP=<n> in the environment runs every cfg as one rank's shard of a TP = n job.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  launch %.1f us  %s" % (d["value"], (d["roofline"].get("avg_launch_us") or 0), d["tensor_parallel"]["decode_path"]))'
tpflag=""; [ -n "$FAKE_TP" ] && tpflag="--fake-tp $FAKE_TP"
for rep in 1 2; do for c in $1; do
  lib=${c%%:*}; envs=$(echo "${c#*:}" | tr ':' ' '); [ "$envs" = "$c" ] && envs=""
  v=$(env $envs FTCF_LIB_NAME=libftcf${lib:+_$lib}.so timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e $tpflag 2>/dev/null | python -c "$pp")
  echo "$c : $v"
done; done
if [ -z "$NO_TL" ]; then for c in $1; do
  lib=${c%%:*}; envs=$(echo "${c#*:}" | tr ':' ' '); [ "$envs" = "$c" ] && envs=""
  n=$(echo $c | tr ':=' '__')
  env $envs FTCF_PERSIST_TS=gpurun_out/r4/ts.bin FTCF_LIB_NAME=libftcf${lib:+_$lib}.so timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 $tpflag >/dev/null 2>&1
  python tools/ps_timeline.py gpurun_out/r4/ts.bin 20 > gpurun_out/r4/tl_$n${FAKE_TP:+_tp$FAKE_TP}.txt; rm -f gpurun_out/r4/ts.bin
done; fi
