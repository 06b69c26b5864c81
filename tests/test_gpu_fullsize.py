"""-m gpu: the engine at BASELINE.json's full size (CodeFuse-13B shape, int8 weight-only, 1024-token prompt), where the
oracle cannot follow: size-independent properties -- determinism, the two bs=1 decode paths against each other, a batch of
identical rows against a single row."""
import argparse
import sys
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full():
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    sys.path.insert(0, ROOT)
    import bench
    a = argparse.Namespace(layers=40, heads=40, head_dim=128, inter=20480, vocab=100864, rotary=32, dtype="int8")
    weights, int8_w, scales = bench.synth_weights(a, 1, torch.device("cuda", 0))
    return a, weights, int8_w, scales


def _op(full):
    from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp
    a, weights, int8_w, scales = full
    return GptNeoXOp(None, 0, a.heads, a.head_dim, a.inter, a.layers, a.vocab, a.rotary, 0, 2, 1, 1, 1, 2048, True, weights,
                     int8_w, scales)


def _run(op, ids, out, V):
    B, S = ids.shape
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    dbg = torch.zeros((out, B, V), dtype=torch.float32, device="cuda")
    o = op.forward(ids, lens, out, 1, torch.tensor([1], dtype=torch.int32), _debug_logits=dbg)
    torch.cuda.synchronize()
    return o[0][:, 0].cpu().numpy(), dbg.cpu().numpy()


def test_full_size_properties(full, monkeypatch):
    a = full[0]
    V, S, out = a.vocab, 1024, 6
    g = torch.Generator().manual_seed(42)
    ids1 = torch.randint(3, V, (1, S), generator=g, dtype=torch.int32).cuda()
    op = _op(full)
    t1, l1 = _run(op, ids1, out, V)
    assert op.stats()["decode_path"] == 1
    t2, l2 = _run(op, ids1, out, V)
    assert np.array_equal(t1, t2) and np.array_equal(l1, l2)  # same request, same engine: bit-identical
    # two identical rows (2-row instantiation of the persistent kernel) against the single row
    tb, lb = _run(op, ids1.repeat(2, 1), out, V)
    assert np.array_equal(tb[0], tb[1]) and np.array_equal(lb[:, 0], lb[:, 1])
    scale = np.abs(l1).max()
    for t in range(out):
        assert np.abs(lb[t, 0] - l1[t, 0]).max() <= 0.03 * scale
        if tb[0, S + t] != t1[0, S + t]:
            break
    del op
    # the per-stage launches (what one row runs under tensor parallelism) against the persistent kernel
    monkeypatch.setenv("FTCF_PERSIST", "0")
    op0 = _op(full)
    t0, l0 = _run(op0, ids1, out, V)
    assert op0.stats()["decode_path"] == 0
    for t in range(out):
        assert np.abs(l0[t, 0] - l1[t, 0]).max() <= 0.03 * scale
        if t0[0, S + t] != t1[0, S + t]:
            top2 = np.sort(l1[t, 0])[-2:]
            assert top2[1] - top2[0] <= 0.03 * scale, "paths disagree without a near tie"
            break
    # prompt tokens come back unchanged, generated ids are in range
    assert np.array_equal(t1[0, :S], ids1.cpu().numpy()[0]) and (t1[0, S:] >= 0).all() and (t1[0, S:] < V).all()
