#!/bin/bash
# A/B of the second form of the persistent kernel (FTCF_PERSIST_A4) on one box, then the parity tests on it
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4
pp='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f tok/s  launch %.1f us  %s" % (d["value"], (d["roofline"].get("avg_launch_us") or 0), d["tensor_parallel"]["decode_path"]))'
for rep in 1 2; do for a4 in 0 1; do
  FTCF_PERSIST_A4=$a4 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>gpurun_out/r4/err_a4_$a4.log | python -c "$pp" | sed "s/^/a4=$a4 /"
done; done
for tp in 2 8; do for a4 in 0 1; do
  FTCF_PERSIST_A4=$a4 FTCF_PERSIST_A4_MAX_TP=8 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --fake-tp $tp 2>gpurun_out/r4/err_a4_${a4}_tp$tp.log | python -c "$pp" | sed "s/^/faketp=$tp a4=$a4 /"
done; done
FTCF_PERSIST_TS=gpurun_out/r4/ts.bin timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "$pp" | sed "s/^/a4=1 (stamps) /"
python tools/ps_timeline.py gpurun_out/r4/ts.bin 20 > gpurun_out/r4/tl_a4_tp0.txt; rm -f gpurun_out/r4/ts.bin
cat gpurun_out/r4/tl_a4_tp0.txt
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_tp_local.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15
