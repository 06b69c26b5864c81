#!/bin/bash
# parity smoke (13B layer shape, 1024-token prompt, int8, vs the oracle) of each variant library, then r4_ab.sh on the same list
R=$GRAFT_REPO_ROOT; cd $R
for c in $1; do
  lib=${c%%:*}; envs=$(echo "${c#*:}" | tr ':' ' '); [ "$envs" = "$c" ] && envs=""
  echo "== parity $c"
  env $envs FTCF_LIB_NAME=libftcf${lib:+_$lib}.so timeout 600 python -m pytest "tests/test_gpu_headline_shapes.py::test_1024_token_prompt_at_the_13b_layer_shape_against_the_oracle[int8]" -q -m gpu -s 2>&1 | grep -E "passed|failed|error|worst" | tail -3
done
bash tools/scripts/r4_ab.sh "$1"
