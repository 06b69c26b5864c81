#!/bin/bash
# round-end check: the whole -m gpu suite, smoke(), then the measurement batch of profiles/ (tools/scripts/r04_profile.sh)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
timeout 3000 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -30 > gpurun_out/r4/pytest_gpu.log
tail -6 gpurun_out/r4/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
bash tools/scripts/r04_profile.sh 2>&1 | tail -40
