#!/bin/bash
# Kernel-variant experiments: builds fastertransformer4codefuse_amd/lib/libftcf_<name>.so with extra -D flags on the
# persistent-kernel translation unit (headline instantiation only: -DPS_ONLY_ONE), the other objects are the product build's.
# usage: tools/build_variant.sh <name> "<flags>"   ; run with FTCF_LIB_NAME=libftcf_<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../fastertransformer4codefuse_amd/csrc"
mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wall -Wno-unused-function -I../../include -DPS_ONLY_ONE $flags \
  -c kernels_persist.hip -o build/var/kp_$name.o -Rpass-analysis=kernel-resource-usage 2> build/var/kp_$name.log || { tail -30 build/var/kp_$name.log; exit 1; }
grep -E "Function Name|VGPRs Spill" build/var/kp_$name.log | sed "s/.*remark: *//;s/\[-Rpass[^]]*\]//;s/_ZN4ftcf19k_decode_persistentI//;s/EEEvNSt.*//" | paste - - 
objs=$(ls build/*.o | grep -v kernels_persist.hip.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fopenmp -o ../lib/libftcf_$name.so build/var/kp_$name.o $objs -L/opt/rocm/lib -lrccl -lroctx64 -Wl,-rpath,/opt/rocm/lib
echo "built libftcf_$name.so"
