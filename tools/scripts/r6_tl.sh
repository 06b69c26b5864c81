#!/bin/bash
# timelines: r6_tl.sh "<name:tp:ENV=V:ENV=V ...> ..." (layer 20)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_tl}; mkdir -p $O
for c in $1; do
  name=$(echo $c | cut -d: -f1); tp=$(echo $c | cut -d: -f2); envs=$(echo $c | cut -d: -f3- | tr ':' ' ')
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  env $envs FTCF_PERSIST_TS=$O/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc --profile-steps 0 $tpflag >/dev/null 2>&1
  python tools/ps_timeline.py $O/ts.bin 20 > $O/tl_${name}_tp$tp.txt; rm -f $O/ts.bin
  echo "== $c"; sed -n '12,19p' $O/tl_${name}_tp$tp.txt
done
