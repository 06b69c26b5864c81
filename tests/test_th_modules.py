"""CPU: the COMPILED extension modules (csrc/th_op: pybind11 `libth_gptneox` / `libth_common` + the TorchScript class) load the
way the reference harness loads them -- `sys.path.append(lib_path); import libth_gptneox` (codefuse_example.py:468-470) --
export the reference's names (th_op/gptneox/GptNeoXOp.cc:190-236, th_op/common/WeightOnlyQuantOps.cc:344-356), share the
host quantiser with the ctypes path bit for bit, and fail loudly without a GPU."""
import os
import sys

import pytest
import torch

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastertransformer4codefuse_amd", "lib")


@pytest.fixture(scope="module")
def mods():
    if LIB not in sys.path:
        sys.path.append(LIB)
    import libth_common
    import libth_gptneox
    return libth_gptneox, libth_common


def test_the_compiled_modules_are_the_ones_on_lib_path(mods):
    g, c = mods
    # (no Python stand-ins sit next to them any more: the directory holds the compiled modules only)
    assert not [f for f in os.listdir(os.path.dirname(g.__file__)) if f.startswith("libth_") and f.endswith(".py")]
    assert g.__file__.endswith(".so") and c.__file__.endswith(".so"), (g.__file__, c.__file__)
    assert g.compiled and c.compiled
    assert hasattr(g, "GptNeoXOp") and hasattr(g.GptNeoXOp, "forward")
    assert hasattr(c, "symmetric_quantize_last_axis_of_batched_matrix_int8")


def test_torchscript_class_and_op_are_registered(mods):
    assert torch.classes.FasterTransformer.GptNeoXOp is not None  # GptNeoXOp.cc:213-236
    w = (torch.randn(128, 64) * 0.05).half()
    q, s = torch.ops.fastertransformer.symmetric_quantize_last_axis_of_batched_matrix_int8(w)
    assert q.dtype == torch.int8 and q.shape == w.shape and s.shape == (64,)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_compiled_quantiser_equals_the_ctypes_path(mods, dtype):
    from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as ref
    g = torch.Generator().manual_seed(5)
    for shape in ((128, 48), (3, 64, 32)):
        w = (torch.randn(*shape, generator=g) * 0.03).to(dtype)
        q, s = mods[1].symmetric_quantize_last_axis_of_batched_matrix_int8(w)
        q0, s0 = ref(w)
        assert torch.equal(q, q0) and torch.equal(s.float(), s0.float()) and s.dtype == dtype
    with pytest.raises(RuntimeError):
        mods[1].symmetric_quantize_last_axis_of_batched_matrix_int8(torch.zeros(4, dtype=torch.float16))  # not 2-D / 3-D
    with pytest.raises(RuntimeError):
        mods[1].symmetric_quantize_last_axis_of_batched_matrix_int8(torch.zeros((64, 16), dtype=torch.int32))


def test_compiled_op_fails_loudly_without_device_tensors(mods):
    # (CHECK_TH_CUDA of the reference, th_utils.h:32-49: there is no CPU fallback)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        mods[0].GptNeoXOp(None, 0, 4, 64, 1024, 2, 512, 32, 0, 2, 1, 1, 0, 1024, True, [torch.zeros(4, dtype=torch.float16)], [], [])
    with pytest.raises(RuntimeError):
        torch.classes.FasterTransformer.GptNeoXOp(4, 64, 1024, 2, 512, 32, 0, 2, 1, 1, 0, 1024, True,
                                                  [torch.zeros(4, dtype=torch.float16)], [], [])
