#!/bin/bash
# round 5: K shares of the streamer waves (FTCF_ROWS_WS) at bs = 16
O=gpurun_out/${1:-r5_rows_ws}; mkdir -p $O
run() {
  FTCF_ROWS_WS=$1 timeout 300 python bench.py --batch 16 --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/b_$1.json 2> $O/b_$1.err
  python -c "import sys,json; d=json.loads(open('$O/b_$1.json').read()); print('$1', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step', round(d['roofline']['avg_launch_us'],1), 'us')" || tail -3 $O/b_$1.err
}
for w in "$@"; do [ "$w" = "$1" ] && continue; run $w; done
