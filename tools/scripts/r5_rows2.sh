#!/bin/bash
# round 5: rows kernel -- tests, bs = 16 with / without it, in-kernel stamps of one short run
O=gpurun_out/${1:-r5_rows2}; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_rows.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
fi
for v in ${VARIANTS:-1 0}; do
  FTCF_ROWS=$v timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/bench_rows$v.json 2> $O/bench_rows$v.err
  echo "rows=$v rc $?"; python -c "import sys,json; d=json.loads(open('$O/bench_rows$v.json').read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step', d.get('roofline'))" || tail -5 $O/bench_rows$v.err
done
FTCF_PERSIST_TS=$O/rows_ts.bin timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 64 --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-pmc > $O/bench_ts.json 2> $O/bench_ts.err
python tools/rows_timeline.py $O/rows_ts.bin 20 | tee $O/timeline.txt

