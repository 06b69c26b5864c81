"""Drop-in module for the reference's pybind11 extension `libth_gptneox` (th_op/gptneox/GptNeoXOp.cc:190-212).

`codefuse_example.py` does `sys.path.append(lib_path); import libth_gptneox; libth_gptneox.GptNeoXOp(...)`
(codefuse_example.py:468-470): point `--lib_path` at this directory.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp  # noqa: E402,F401
