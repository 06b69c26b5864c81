"""Drop-in module for the reference's pybind11 extension `libth_common` (th_op/common/WeightOnlyQuantOps.cc:344-349)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fastertransformer4codefuse_amd.gptneox_op import (  # noqa: E402,F401
    symmetric_quantize_last_axis_of_batched_matrix_int8)
