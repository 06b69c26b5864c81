#!/bin/bash
# round 5: the batcher on the rows kernel -- its tests, then tools/bench_batcher.py with and without it
O=gpurun_out/${1:-r5_batcher}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_batcher.py tests/test_gpu_rows.py -q -m gpu -x 2>&1 | tail -15
for v in 1 0; do
  FTCF_BATCHER_ROWS=$v timeout 600 python tools/bench_batcher.py --page 64 > $O/bench_rows$v.log 2>&1; echo "rows=$v rc=$?"; tail -4 $O/bench_rows$v.log
done
