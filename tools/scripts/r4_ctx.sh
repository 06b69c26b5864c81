#!/bin/bash
# prompt attention with pipelined K/V tiles: parity tests, then the kernel's time inside the default bench (kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ctx; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_headline_shapes.py -q -m gpu -x 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
timeout 600 python tools/bench_prefill.py --lens 128,512,1024,2048 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/trace" -o r -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/bench.json" 2> "$GRAFT_REPO_ROOT/$O/prof.log"
cd "$GRAFT_REPO_ROOT"
python tools/prof_summary.py $(find $O/trace -name "*results.db" | head -1) $O/kernel_stats.txt | grep -E "attention_mfma|persist|lm_head|kernel " | cut -c1-150
find $O -name "*.db" -delete
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('prefill_ms', d['prefill_ms'], 'mfma_frac', d['prefill_mfma_frac'], 'tok/s', d['value'])"
