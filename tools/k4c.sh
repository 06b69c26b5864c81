#!/bin/bash
# quick compile of the headline instantiation of k_decode_persistent4 with extra flags; prints spills: tools/k4c.sh "<flags>"
mkdir -p /tmp/k4 && cd /tmp/k4 && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wno-unused-function -I/root/repo/include -I/root/repo/fastertransformer4codefuse_amd/csrc -DPS_ONLY_ONE $1 -c /root/repo/tools/experiments/kernels_persist4.hip -o k4.o -save-temps -Rpass-analysis=kernel-resource-usage 2> k4.log; grep -E "error|VGPRs Spill|SGPRs Spill|ScratchSize" k4.log | sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - -
