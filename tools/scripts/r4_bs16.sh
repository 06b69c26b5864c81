#!/bin/bash
# bs = 16: the transposing LM-head GEMM (FTCF_LMHEAD_TR) and the wave-per-row greedy tail -- parity tests, then step time A/B and a trace
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -m gpu -k "lm_head or fuzz or bs16 or batch or ragged or rows or engine" 2>&1 | tail -3
for v in 0 1 0 1; do
  FTCF_LMHEAD_TR=$v timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tr=$v', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step')"
done
bash tools/scripts/bs16_prof.sh 2>&1 | grep -E "nk_f32out|greedy|SmallmGroup|Mmha" | cut -c30-150
