#!/bin/bash
# compile one instantiation of the rows kernel and print its resource usage (extra flags: "$@")
cd /root/repo/fastertransformer4codefuse_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fopenmp -Wall -Wno-unused-function -I../../include -DRW_FEW -Rpass-analysis=kernel-resource-usage "$@" -c kernels_rows.hip -o /tmp/rows_few.o 2>&1 | grep -i "error\|warning:\|SGPRs\|VGPRs\|Scratch\|Occupancy" | head -30
