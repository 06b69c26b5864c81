#!/bin/bash
# prompt attention with the heaviest query blocks split along their keys (FTCF_CTX_SPLIT): parity, then prompt-phase time by length
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_headline_shapes.py tests/test_gpu_engine.py -q -m gpu 2>&1 | tail -3
for v in 0 1 0 1; do
  echo "== split=$v"; FTCF_CTX_SPLIT=$v timeout 600 python tools/bench_prefill.py --lens 512,1024,2048 2>/dev/null | grep prompt_len | cut -c1-120
done
