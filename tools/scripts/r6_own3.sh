#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-r6_own3}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "variants or tiny_fp16_greedy or mid_model" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "oracle or tensor_parallel_one_row" 2>&1 | tail -3
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  step %.1f us  launch %.1f us" % (d["value"], d["ms_per_step"]*1000, (d["roofline"].get("avg_launch_us") or 0)))'
run() { local name=$1 tp=$2; shift 2
  tpflag=""; [ $tp -gt 0 ] && tpflag="--fake-tp $tp"
  v=$(env "$@" timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc $tpflag 2>$O/err.txt | python -c "$pp" 2>&1 | tail -1)
  echo "tp$tp $name : $v" | tee -a $O/ab.txt; }
for rep in 1 2; do for tp in 0 2 4 8; do
  run own0 $tp FTCF_PERSIST_OWN=0
  run own1 $tp FTCF_PERSIST_OWN=1
  run own1cs6 $tp FTCF_PERSIST_CS3=6
  run nosmid $tp FTCF_LIB_NAME=libftcf_nosmid.so
done; done
OUT=${OUT:-r6_own3} bash tools/scripts/r6_tl.sh "own1:0:X=1 own1:8:X=1" > /dev/null
tail -n 22 $O/tl_own1_tp0.txt $O/tl_own1_tp8.txt
