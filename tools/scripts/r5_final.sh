#!/bin/bash
# round-end check: the whole -m gpu suite with its slowest tests, smoke(), then the measurement batch of profiles/
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -q -m gpu --durations=25 2>&1 | tail -45 > gpurun_out/r5/pytest_gpu.log
tail -34 gpurun_out/r5/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
bash tools/scripts/r05_profile.sh 2>&1 | tail -60
