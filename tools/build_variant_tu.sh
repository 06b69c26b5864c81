#!/bin/bash
# Like build_variant.sh for any translation unit: builds lib/libftcf_<name>.so with extra -D flags on <tu>.hip.
# usage: tools/build_variant_tu.sh <name> <tu> "<flags>"   ; run with FTCF_LIB_NAME=libftcf_<name>.so
set -e
name=$1; tu=$2; flags=$3
cd "$(dirname "$0")/../fastertransformer4codefuse_amd/csrc"
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wall -Wno-unused-function -I../../include $flags \
  -c $tu.hip -o build/var/${tu}_$name.o
objs=$(ls build/*.o | grep -v "$tu.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fopenmp -o ../lib/libftcf_$name.so build/var/${tu}_$name.o $objs -L/opt/rocm/lib -lrccl -lroctx64 -Wl,-rpath,/opt/rocm/lib
echo "built libftcf_$name.so"
