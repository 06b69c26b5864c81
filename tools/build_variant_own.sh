#!/bin/bash
# Kernel-variant experiments on the own-group layout: builds lib/libftcf_<name>.so with extra -D flags on the two translation units
# that instantiate it (kernels_persist_own.hip, kernels_persist_tp_own.hip; 13B int8 one-row form only: -DPS_ONLY_ONE).
# usage: tools/build_variant_own.sh <name> "<flags>"   ; run with FTCF_LIB_NAME=libftcf_<name>.so
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../fastertransformer4codefuse_amd/csrc"
mkdir -p build/var
for tu in kernels_persist_own kernels_persist_tp_own; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wall -Wno-unused-function -I../../include -DPS_ONLY_ONE $flags \
  -c $tu.hip -o build/var/${tu}_$name.o -Rpass-analysis=kernel-resource-usage 2> build/var/${tu}_$name.log || { tail -30 build/var/${tu}_$name.log; exit 1; } &
done
wait
cat build/var/kernels_persist_own_$name.log build/var/kernels_persist_tp_own_$name.log | grep -E "Function Name|VGPRs Spill" | sed "s/.*remark: *//;s/\[-Rpass[^]]*\]//;s/_ZN4ftcf19k_decode_persistentI//;s/EEEvNSt.*//" | paste - -
objs=$(ls build/*.o | grep -v "kernels_persist_own.hip.o\|kernels_persist_tp_own.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fopenmp -o ../lib/libftcf_$name.so build/var/kernels_persist_own_$name.o build/var/kernels_persist_tp_own_$name.o $objs -L/opt/rocm/lib -lrccl -lroctx64 -Wl,-rpath,/opt/rocm/lib
echo "built libftcf_$name.so"
