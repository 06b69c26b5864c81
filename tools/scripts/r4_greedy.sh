#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
timeout 3000 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fuzz.py tests/test_gpu_beam.py tests/test_gpu_batcher.py tests/test_gpu_fp32.py tests/test_gpu_th_modules.py tests/test_gpu_cli.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r4/pytest_greedy.log
tail -6 gpurun_out/r4/pytest_greedy.log
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  launch %.1f us  ms/step %.4f" % (d["value"], (d["roofline"].get("avg_launch_us") or 0), d["ms_per_step"]))'
for rep in 1 2; do for g in 0 1; do
  FTCF_GREEDY_FUSED=$g timeout 300 python bench.py --steps 160 --warmup 8 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "$pp" | sed "s/^/tp1 fused=$g /"
done; done
for tp in 2 8; do for g in 0 1; do
  FTCF_GREEDY_FUSED=$g timeout 300 python bench.py --steps 160 --warmup 8 --no-cpu-baseline --no-e2e --fake-tp $tp 2>/dev/null | python -c "$pp" | sed "s/^/faketp=$tp fused=$g /"
done; done
