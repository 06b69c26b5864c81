#!/bin/bash
# the whole -m gpu suite, then the three bench lines (TP = 1, one rank's shard of TP = 2 / TP = 8)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
timeout 3300 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 > gpurun_out/r4/pytest_gpu.log
tail -5 gpurun_out/r4/pytest_gpu.log
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  launch %.1f us  %s" % (d["value"], (d["roofline"].get("avg_launch_us") or 0), d["tensor_parallel"]["decode_path"]))'
for tp in 0 2 8; do
  timeout 300 python bench.py --steps 120 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp 2>/dev/null | python -c "$pp" | sed "s/^/faketp=$tp /"
done
