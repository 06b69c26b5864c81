#!/bin/bash
# Kernel-variant experiments on the second form of the persistent kernel: builds fastertransformer4codefuse_amd/lib/libftcf_<name>.so
# with extra -D flags on kernels_persist4.hip (headline instantiation only: -DPS_ONLY_ONE), the other objects are the product build's.
# usage: tools/build_variant4.sh <name> "<flags>"   ; run with FTCF_LIB_NAME=libftcf_<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../fastertransformer4codefuse_amd/csrc"
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wall -Wno-unused-function -I../../include -DPS_ONLY_ONE $flags \
  -I. -I../../tools/experiments -c ../../tools/experiments/kernels_persist4.hip -o build/var/kp4_$name.o -Rpass-analysis=kernel-resource-usage 2> build/var/kp4_$name.log || { tail -30 build/var/kp4_$name.log; exit 1; }
grep -E "VGPRs Spill|ScratchSize" build/var/kp4_$name.log | sed "s/.*remark: *//;s/\[-Rpass[^]]*\]//" | paste - -
objs=$(ls build/*.o | grep -v kernels_persist4.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fopenmp -o ../lib/libftcf_$name.so build/var/kp4_$name.o $objs -L/opt/rocm/lib -lrccl -lroctx64 -Wl,-rpath,/opt/rocm/lib
echo "built libftcf_$name.so"
