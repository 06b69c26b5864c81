"""-m gpu: the compiled `libth_gptneox.GptNeoXOp` (pybind11) and `torch.classes.FasterTransformer.GptNeoXOp` (TorchScript),
imported exactly as codefuse_example.py:468-470 imports the reference's module, reproduce the golden tokens and the ctypes
op's outputs bit for bit (same engine underneath)."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.helpers import load_tiny

pytestmark = pytest.mark.gpu
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastertransformer4codefuse_amd", "lib")


@pytest.fixture(scope="module")
def env():
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    sys.path.append(LIB)  # codefuse_example.py:468
    import libth_gptneox
    assert libth_gptneox.__file__.endswith(".so") and libth_gptneox.compiled
    from tests import gpu_helpers
    return libth_gptneox, gpu_helpers


@pytest.mark.parametrize("int8_mode", [0, 1])
def test_compiled_op_reproduces_the_golden_tokens(env, int8_mode):
    mod, gh = env
    cfg, w, z = load_tiny()
    op = gh.make_op(cfg, w, int8_mode=int8_mode, op_class=mod.GptNeoXOp)
    r = gh.run_op(op, z["prompt"][None, :], [16], 8, cfg["vocab_size"], top_k=1)
    ref = gh.run_op(gh.make_op(cfg, w, int8_mode=int8_mode), z["prompt"][None, :], [16], 8, cfg["vocab_size"], top_k=1)
    if int8_mode == 0:
        assert r["output_ids"][0, 16:].tolist() == z["hf_tokens"].tolist()  # HF golden (tests/golden/make_golden.py)
    assert op.stats()["decode_path"] == 1
    for k in ("output_ids", "sequence_lengths", "cum_log_probs", "logits"):
        assert np.array_equal(r[k], ref[k]), k  # one engine behind both bindings


def test_compiled_op_runtime_arguments_and_callback(env):
    mod, gh = env
    cfg, w, z = load_tiny()
    seen = []
    kw = dict(top_k=4, top_p=0.9, temperature=0.8, repetition_penalty=1.2, random_seed=1234)
    op = gh.make_op(cfg, w, op_class=mod.GptNeoXOp)
    r = gh.run_op(op, z["prompt"][None, :], [16], 8, cfg["vocab_size"], callback=lambda d: seen.append(d), **kw)
    ref = gh.run_op(gh.make_op(cfg, w), z["prompt"][None, :], [16], 8, cfg["vocab_size"], **kw)
    assert np.array_equal(r["output_ids"], ref["output_ids"]) and np.array_equal(r["cum_log_probs"], ref["cum_log_probs"])
    # pybind_callback_utils.cc:79-103: one dict per step but the last, last_tokens / idxs as [batch][beam] lists
    assert len(seen) == 7 and all(set(d) == {"last_tokens", "idxs"} for d in seen)
    assert [d["last_tokens"][0][0] for d in seen] == r["output_ids"][0, 16:23].tolist()
    with pytest.raises(RuntimeError):
        op.forward(torch.zeros((1, 4), dtype=torch.int64, device="cuda"), torch.tensor([4], dtype=torch.int32, device="cuda"), 2)


def test_torchscript_class_runs_the_engine(env):
    mod, gh = env
    cfg, w, z = load_tiny()
    captured = {}

    class Capture:  # make_op assembles the reference-order tensor lists; hand them to the TorchScript constructor
        def __init__(self, comm, rank, *a):
            captured["args"] = a

    gh.make_op(cfg, w, op_class=Capture)
    op = torch.classes.FasterTransformer.GptNeoXOp(*captured["args"])
    ids = torch.from_numpy(np.ascontiguousarray(z["prompt"][None, :], dtype=np.int32)).cuda()
    lens = torch.tensor([16], dtype=torch.int32, device="cuda")
    out = op.forward(ids, lens, 8, 1, torch.tensor([1], dtype=torch.int32), None, None, None, None, None, None, None, None, 1)
    torch.cuda.synchronize()
    assert out[0][0, 0, 16:].cpu().tolist() == z["hf_tokens"].tolist()
    assert out[1].cpu().tolist() == [[24]] and out[2].shape == (1, 1)
