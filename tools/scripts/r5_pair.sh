#!/bin/bash
# round 5: tensor-parallel batched decode with attn | ffn all-reduced as one message and the residual inside the next LN pass
# (FTCF_TP_PAIR_AR=1, default) against the reference's order (0): the TP suites, then one rank's shard at bs 16 / 32
O=gpurun_out/${1:-r5_pair}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_tp_local.py tests/test_gpu_tp_process.py tests/test_gpu_batcher.py tests/test_gpu_fullsize.py tests/test_gpu_cli.py -q -m gpu -x -k "tensor_parallel or tp or overlap or process or ranks" --deselect tests/test_gpu_tp_process.py::test_13b_tp2_shards_between_two_processes 2>&1 | grep -v "^\[FT\]" | tail -30 | tee $O/pytest.log
for tp in 8 2; do
for bs in 16 32 4; do
for pr in 0 1; do for v in 0 1; do
  FTCF_TP_PAIR_AR=$pr FTCF_DECODE_OVERLAP=$v timeout 300 python bench.py --fake-tp $tp --batch $bs --prompt-len 256 --output-len 128 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-pmc > $O/b_tp${tp}_bs${bs}_pair${pr}_ov${v}.json 2> $O/b_tp${tp}_bs${bs}_pair${pr}_ov${v}.err
  python -c "import sys,json; d=json.loads(open('$O/b_tp${tp}_bs${bs}_pair${pr}_ov${v}.json').read()); print('fake-tp $tp bs $bs pair=$pr overlap=$v', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step', d['tensor_parallel']['decode_overlap']['ran_in_the_last_request'])" || tail -3 $O/b_tp${tp}_bs${bs}_pair${pr}_ov${v}.err
done; done; done; done
