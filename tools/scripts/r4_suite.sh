#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
timeout 3400 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -40 > gpurun_out/r4/pytest_suite.log
tail -6 gpurun_out/r4/pytest_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
