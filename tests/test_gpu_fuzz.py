"""-m gpu: randomly drawn model shapes / batch / prompt / beam configurations through GptNeoXOp against the oracle.  Covers
the path selection (persistent kernel, per-stage launches, general path with split-K / burst / tiled GEMMs), odd sizes
(one head, inter sizes that are not powers of two, rotary 0 / partial / full, one-token prompts, ragged batches) and rows
that finish on end_id."""
import os

import numpy as np
import pytest
import torch  # noqa: F401

from oracle import oracle as orc
from tests.helpers import quantize_layers, random_model, weight_list_to_layers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gh():
    from tests import gpu_helpers
    from fastertransformer4codefuse_amd import capi
    capi.require_gpu()
    return gpu_helpers


def _draw(rng):
    dh = int(rng.choice([64, 128]))
    nh = int(rng.choice([1, 2, 3, 4, 5, 8]))
    inter = int(rng.choice([64, 128, 192, 256, 512, 1024])) * int(rng.choice([1, 2, 3]))
    cfg = dict(head_num=nh, size_per_head=dh, inter_size=inter, num_layer=int(rng.choice([1, 2, 3])),
               vocab_size=(int(rng.choice([64, 128, 200, 1000, 2048])) + 7) // 8 * 8, rotary_dim=int(rng.choice([0, 16, 32, dh])),
               start_id=0, end_id=2)
    B = int(rng.choice([1, 2, 3, 4, 5, 9, 16, 17, 33]))
    K = int(rng.choice([1, 1, 1, 2, 3]))
    return cfg, (min(B, 5) if K > 1 else B), int(rng.choice([1, 2, 7, 33, 70])), int(rng.choice([3, 9])), int(rng.choice([0, 1])), K


@pytest.mark.parametrize("seed", range(16))
def test_random_configuration_follows_the_oracle(gh, seed):
    rng = np.random.RandomState(1000 + seed)
    cfg, B, S, out, int8, K = _draw(rng)
    V = cfg["vocab_size"]
    w = random_model(cfg, seed=seed, std=0.05)
    layers, glob = weight_list_to_layers(cfg, w)
    if int8:
        layers = quantize_layers(layers)
    lens = rng.randint(1, S + 1, size=B).astype(np.int32)
    lens[0] = S
    ids = np.full((B, S), cfg["end_id"], dtype=np.int32)
    for b in range(B):
        ids[b, :lens[b]] = rng.randint(3, V, size=lens[b])
    op = gh.make_op(cfg, w, int8_mode=int8)
    m = orc.Model(dict(cfg, fp16=1, int8_mode=int8), layers, glob)
    what = f"{cfg} B={B} S={S} out={out} int8={int8} K={K}"
    if K > 1:
        from tests.test_gpu_beam import _replay
        r = gh.run_op_beam(op, ids, lens, out, V, K, return_logits=True)
        p_ids, p_len, p_cum = _replay(cfg, ids, lens, out, K, r["logits"], orc.BeamParams(B))
        assert r["output_ids"].tolist() == p_ids.tolist(), what  # exact given the GPU's own logits
        np.testing.assert_allclose(r["cum_log_probs"], p_cum, rtol=1e-4, atol=1e-3, err_msg=what)
        o = m.generate_beam(ids, lens, out, K)
        assert (r["output_ids"] == o["output_ids"]).mean() > 0.6, what
        return
    r = gh.run_op(op, ids, lens, out, V, top_k=1)
    o = m.generate(ids, lens, out, return_logits=True)
    for b in range(B):
        for t in range(out):
            ref = o["logits"][t, b]
            scale = np.abs(ref).max()
            assert np.abs(r["logits"][t, b] - ref).max() <= 5e-3 * scale, (what, b, t, np.abs(r["logits"][t, b] - ref).max() / scale)
            if o["output_ids"][b, lens[b] + t] == cfg["end_id"]:
                break  # the row finished: later logits are not consumed
            if r["output_ids"][b, lens[b] + t] != o["output_ids"][b, lens[b] + t]:
                top2 = np.sort(ref)[-2:]
                assert top2[1] - top2[0] <= 1e-2 * scale, ("token flip without a near tie", what, b, t)
                break


def test_own_group_layout_on_random_shapes_and_grids(gh):
    """A fixed-seed slice of tools/fuzz_own_layout.py: the one- / two-row persistent kernel with the own-group layout of its out-proj /
    FFN2 stage forced wherever the shape divides (FTCF_PERSIST_OWN=1) on random hidden sizes (64..1024), grids of 4..64 workgroups, one
    and two rows, fp16 / int8 -- every case against the oracle (2 % of the logit range, arg max equal outside a near tie), and the
    layout must really have run in a good part of them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_own_layout", os.path.join(ROOT, "tools", "fuzz_own_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        spec.loader.exec_module(mod)
        bad, own = mod.run(1, 16, verbose=False)
    finally:
        os.chdir(cwd)
    assert bad == 0
    assert own >= 4
