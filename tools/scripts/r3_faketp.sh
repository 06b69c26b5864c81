#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
for tp in 2 8; do for ov in 0 1; do
  FTCF_PREFILL_OVERLAP=$ov python bench.py --fake-tp $tp --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --profile-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fake-tp $tp overlap $ov: prefill %.2f ms (wall %.2f), decode %.1f tok/s, path %s' % (d['prefill_ms'], d['prefill_wall_ms'], d['value'], d['tensor_parallel']['decode_path']))"
done; done
