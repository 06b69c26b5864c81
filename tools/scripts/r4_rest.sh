#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
timeout 3300 python -m pytest tests/test_gpu_logger.py tests/test_gpu_th_modules.py tests/test_gpu_cli.py tests/test_gpu_tp_local.py tests/test_gpu_tp_process.py "tests/test_gpu_engine.py::test_multi_token_graphs_equal_single_token_graphs" tests/test_gpu_engine.py::test_begin_step_finish_equals_forward -q -m gpu --durations=8 2>&1 | tail -60 > gpurun_out/r4/pytest_rest.log
tail -30 gpurun_out/r4/pytest_rest.log
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  launch %.1f us  ms/step %.4f" % (d["value"], (d["roofline"].get("avg_launch_us") or 0), d["ms_per_step"]))'
for gt in 1 8 1 8; do
  FTCF_GRAPH_TOKENS=$gt timeout 300 python bench.py --steps 160 --warmup 8 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "$pp" | sed "s/^/graph_tokens=$gt /"
done
