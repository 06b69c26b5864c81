#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
timeout 3300 python -m pytest tests/test_gpu_tp_process.py tests/test_gpu_tp_local.py tests/test_gpu_batcher.py tests/test_gpu_cli.py "tests/test_gpu_fullsize.py::test_full_size_tensor_parallel_one_row" -q -m gpu -x --durations=8 2>&1 | tail -40 > gpurun_out/r4/pytest_tp.log
tail -25 gpurun_out/r4/pytest_tp.log
