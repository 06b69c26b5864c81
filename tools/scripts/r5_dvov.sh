#!/bin/bash
# round 5: batched TP decode with the layer's all-reduce on the side stream (two micro-batches): parity, then one rank's
# shard of TP 8 / TP 2 at bs = 16 and bs = 32 with FTCF_DECODE_OVERLAP = 0 / 1 (timing aid: the peers are missing, the
# 1-rank ncclAllReduce is launched but carries nothing -- this measures what the ping-pong COSTS, not what it hides)
O=gpurun_out/${1:-r5_dvov}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tp_local.py tests/test_gpu_batcher.py -q -m gpu -k "overlapped_all_reduce or tensor_parallel_batchers" 2>&1 | grep -v "^\[FT\]" | tail -60 | tee $O/pytest.log
for tp in 8 2; do
for bs in 16 32; do
for v in 0 1; do for g in 0 1; do
  FTCF_TP_GRAPH=$g FTCF_DECODE_OVERLAP=$v timeout 300 python bench.py --fake-tp $tp --batch $bs --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/b_tp${tp}_bs${bs}_ov${v}_g$g.json 2> $O/b_tp${tp}_bs${bs}_ov${v}_g$g.err
  python -c "import sys,json; d=json.loads(open('$O/b_tp${tp}_bs${bs}_ov${v}_g$g.json').read()); print('fake-tp $tp bs $bs overlap=$v graph=$g', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step', d['tensor_parallel'].get('decode_overlap'))" || tail -3 $O/b_tp${tp}_bs${bs}_ov${v}_g$g.err
done; done; done; done
