#!/bin/bash
# Builds the compiled Python extension modules lib/libth_gptneox.so and lib/libth_common.so (pybind11 + TorchScript class
# over libftcf.so).  Plain g++ against the installed torch's headers and libraries; no HIP code in these files.
set -e
cd "$(dirname "$0")/.."
TI=$(python -c 'import torch, os; print(os.path.dirname(torch.__file__))')
PYI=$(python -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
ABI=$(python -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')
COMMON="-O2 -std=c++17 -fPIC -shared -D_GLIBCXX_USE_CXX11_ABI=$ABI -I../../include -I$TI/include -I$TI/include/torch/csrc/api/include -I$PYI"
LINK="-L../lib -lftcf -L$TI/lib -lc10 -ltorch_cpu -ltorch -ltorch_python -Wl,-rpath,\$ORIGIN -Wl,-rpath,$TI/lib"
g++ $COMMON -D__HIP_PLATFORM_AMD__ -DUSE_ROCM -I/opt/rocm/include th_op/th_gptneox.cc -o ../lib/libth_gptneox.so $LINK -lc10_hip -ltorch_hip &
g++ $COMMON th_op/th_common.cc -o ../lib/libth_common.so $LINK &
wait
ls -la ../lib/libth_gptneox.so ../lib/libth_common.so
