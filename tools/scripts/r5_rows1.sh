#!/bin/bash
# round 5: first run of the rows kernel -- its tests, then bs = 16 with it and without it
O=gpurun_out/r5_rows1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
for v in 1 0; do
  FTCF_ROWS=$v timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/bench_rows$v.json 2> $O/bench_rows$v.err
  echo "rows=$v rc $?"; python -c "import sys,json; d=json.loads(open('$O/bench_rows$v.json').read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step', d.get('roofline'))" || tail -5 $O/bench_rows$v.err
done
