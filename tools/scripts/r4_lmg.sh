#!/bin/bash
# A/B of the LM head + greedy decode in one launch (FTCF_LM_GREEDY) against k_lm_head + k_greedy_decode, and the parity tests
# of the greedy paths.  Usage: gpurun -- bash tools/scripts/r4_lmg.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/lmg
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/lmg/tests.txt
for rep in 1 2; do
  for v in 0 1; do
    FTCF_LM_GREEDY=$v timeout 600 python bench.py --steps 160 --warmup 5 > gpurun_out/lmg/bench_v${v}_r${rep}.json 2> gpurun_out/lmg/bench_v${v}_r${rep}.err
  done
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/lmg/trace" -o r -- python "$GRAFT_REPO_ROOT/bench.py" --steps 64 --warmup 5 > "$GRAFT_REPO_ROOT/gpurun_out/lmg/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/lmg/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 2), round(d["ms_per_step"] * 1000, 1), round(d["roofline"]["avg_launch_us"], 1))
    except Exception as e:
        print(f, "FAILED", e)
PY
python tools/prof_summary.py $(find gpurun_out/lmg/trace -name "*results.db" | head -1) gpurun_out/lmg/kernel_stats.txt | head -12
find gpurun_out/lmg -name "*.db" -delete
cat gpurun_out/lmg/tests.txt
