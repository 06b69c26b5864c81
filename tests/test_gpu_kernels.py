"""-m gpu: kernel-level parity of the HIP path (through the C ABI) against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

capi = None


@pytest.fixture(scope="module", autouse=True)
def _lib():
    global capi
    from fastertransformer4codefuse_amd import capi as _c
    capi = _c
    capi.require_gpu()
    yield


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_KEEP = []


def D(t):
    """device copy that stays alive (a temporary `.cuda()` would be freed -- and its block reused -- as soon as the
    ctypes pointer has been taken)"""
    d = t.cuda() if isinstance(t, torch.Tensor) else torch.from_numpy(t).cuda()
    _KEEP.append(d)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return d


def quant(w_np):
    from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as qf
    q, s = qf(torch.from_numpy(w_np).half().contiguous())
    return q, s


_GRID_CACHE = {}


def _grid_weights(n, k):
    """one quantised N(0, 0.002) matrix per (n, k) of the reference's grid, shared by all m"""
    if (n, k) not in _GRID_CACHE:
        _GRID_CACHE.clear()  # (a 16384 x 4096 fp32 matrix is 268 MB: keep one)
        g = torch.Generator().manual_seed(734876213 + n * 7 + k)
        w = (torch.randn(k, n, generator=g) * 0.002).half().float().numpy()
        q, s = quant(w)
        q_rm, s_o = orc.symmetric_quantize_int8(w, True)
        _GRID_CACHE[(n, k)] = (q.cuda(), s.cuda(), q_rm, s_o)
    return _GRID_CACHE[(n, k)]


# th_gemm_dequantize.py:111-115: the reference's whole grid (compute_m x compute_n x compute_k); k = 16384 is the long-K /
# split-K accumulation case, m covers the GEMV (1..4), burst (<= 16) and tiled MFMA (> 16) forms incl. ragged row counts
@pytest.mark.parametrize("k", [4096, 8192, 16384])
@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_fpA_intB_gemm_matches_reference_formula(n, k):
    # th_gemm_dequantize.py:65-115: weights N(0, 0.002), rtol 1e-3 / atol 2e-3 vs matmul(act, q.to(fp16) * scale)
    qd, sd, q_rm, s_o = _grid_weights(n, k)
    for m in (256, 177, 195, 125, 66, 33, 8, 2, 1, 3, 4, 16):
        g = torch.Generator().manual_seed(m * 1000003 + n + k)
        act = torch.randn(m, k, generator=g).half()
        ref = orc.gemm(act.float().numpy(), q=q_rm, scale=s_o, fp16=True)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        A = act.cuda()
        capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(A), capi.vp(qd), capi.vp(sd), None, 0, capi.vp(out),
                                                 m, n, k, sp()))
        torch.cuda.synchronize()
        torch.testing.assert_close(out.cpu().float(), torch.from_numpy(ref), rtol=1e-3, atol=2e-3, msg=lambda t: f"m={m}: {t}")


@pytest.mark.parametrize("k", [1088, 2304])
def test_fpA_intB_gemm_split_k_forms(k):
    """17..256 rows cut the K extent of their tiles into slices: k = 1088 (17 k-steps: not a whole number of rounds of the
    four-deep ring -> the plain loop) and 2304 (36 k-steps: the deep loop, ragged last slice); repeated launches re-arm the
    tickets; rtol / atol as the reference's grid test (th_gemm_dequantize.py:65-115)."""
    n = 1280
    qd, sd, q_rm, s_o = _grid_weights(n, k)
    for m in (17, 33, 48, 64, 65, 200, 256, 300):  # tile heights 32 / 48 / 64, one to five row blocks
        g = torch.Generator().manual_seed(m * 7 + k)
        act = torch.randn(m, k, generator=g).half()
        bias = torch.randn(n, generator=g).half()
        ref = orc.gemm(act.float().numpy(), q=q_rm, scale=s_o, bias=bias.float().numpy(), act=1, fp16=True)
        A, bd = act.cuda(), bias.cuda()
        outs = []
        for _ in range(3):
            out = torch.empty((m, n), dtype=torch.float16, device="cuda")
            capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(A), capi.vp(qd), capi.vp(sd), capi.vp(bd), 1, capi.vp(out), m, n, k, sp()))
            torch.cuda.synchronize()
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])  # slices are added in slice order
        torch.testing.assert_close(outs[0].float(), torch.from_numpy(ref), rtol=1e-3, atol=2e-3, msg=lambda t: f"m={m}: {t}")


@pytest.mark.parametrize("m", [4, 128])
def test_identity_activation_dequant_is_bit_exact(m):
    # th_gemm_dequantize.py:22-40
    torch.manual_seed(0)
    k, n = 128, 256
    w = (torch.randn(k, n) * 0.05).half()
    q, s = quant(w.float().numpy())
    q_rm, s_o = orc.symmetric_quantize_int8(w.float().numpy(), True)
    ref = (torch.from_numpy(q_rm).half() * torch.from_numpy(s_o).half())
    A = torch.eye(k, dtype=torch.float16)[:m].contiguous().cuda()
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(A), capi.vp(D(q)), capi.vp(D(s)), None, 0, capi.vp(out),
                                             m, n, k, sp()))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref[:m])


@pytest.mark.parametrize("m", [1, 4, 16, 130])
def test_fpA_intB_gemm_bias_gelu_epilogue(m):
    torch.manual_seed(1)
    n, k = 1024, 512
    w = (torch.randn(k, n) * 0.05).half().float().numpy()
    q, s = quant(w)
    q_rm, s_o = orc.symmetric_quantize_int8(w, True)
    act = torch.randn(m, k).half()
    bias = torch.randn(n).half()
    ref = orc.gemm(act.float().numpy(), q=q_rm, scale=s_o, bias=bias.float().numpy(), act=1, fp16=True)
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(D(act)), capi.vp(D(q)), capi.vp(D(s)),
                                             capi.vp(D(bias)), 1, capi.vp(out), m, n, k, sp()))
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().float(), torch.from_numpy(ref), rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("m", [1, 3, 7, 200])
def test_fp16_gemm(m):
    torch.manual_seed(2)
    n, k = 512, 1024
    w = (torch.randn(k, n) * 0.03).half()
    act = torch.randn(m, k).half()
    bias = torch.randn(n).half()
    wt = torch.empty((k, n), dtype=torch.float16, device="cuda")
    capi.check(capi.lib().ftcf_fp16_rowmajor_to_tiled(capi.vp(D(w)), C.c_size_t(k), C.c_size_t(n), capi.vp(wt), sp()))
    for act_kind, b in ((0, None), (1, bias)):
        ref = orc.gemm(act.float().numpy(), W=w.float().numpy(), bias=None if b is None else b.float().numpy(),
                       act=act_kind, fp16=True)
        out = torch.empty((m, n), dtype=torch.float16, device="cuda")
        capi.check(capi.lib().ftcf_fp16_gemm(capi.vp(D(act)), capi.vp(wt), capi.vp(None if b is None else D(b)),
                                             act_kind, capi.vp(out), m, n, k, sp()))
        torch.cuda.synchronize()
        torch.testing.assert_close(out.cpu().float(), torch.from_numpy(ref), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("m", [1, 4, 9])
def test_lm_head(m):
    torch.manual_seed(3)
    V, H = 2000, 512
    W = (torch.randn(V, H) * 0.1).half()
    x = torch.randn(m, H).half()
    ref = orc.lm_head(x.float().numpy(), W.float().numpy())
    out = torch.empty((m, V), dtype=torch.float32, device="cuda")
    capi.check(capi.lib().ftcf_lm_head(capi.vp(D(x)), capi.vp(D(W)), capi.vp(out), m, V, H, V, sp()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-3)


def test_layernorm_and_residual_match_oracle_rounding_points():
    rng = np.random.RandomState(5)
    m, n = 7, 5120
    x = orc.round_half(rng.randn(m, n).astype(np.float32))
    g = orc.round_half(1 + 0.1 * rng.randn(n).astype(np.float32))
    b = orc.round_half(0.1 * rng.randn(n).astype(np.float32))
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    X, G, Bt = (torch.from_numpy(a).half().cuda() for a in (x, g, b))
    capi.check(capi.lib().ftcf_layernorm(capi.vp(X), capi.vp(G), capi.vp(Bt), capi.vp(out), m, n, C.c_float(1e-5), 1, sp()))
    torch.cuda.synchronize()
    ref = orc.layernorm(x, g, b, fp16=True)
    got = out.cpu().float().numpy()
    # identical rounding points; fp32 reduction order may flip half(mean)/half(rstd) by one ulp on rare rows
    assert np.mean(got != ref) < 0.02
    np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-3)
    ffn, att = orc.round_half(x * 0.5), orc.round_half(x * 0.25)
    for inplace in (0, 1):
        for tp in (1, 2):
            ref = orc.add_bias_attn_ffn_residual(ffn, att, x, b, tp=tp, inplace_variant=bool(inplace), fp16=True)
            o2 = torch.empty((m, n), dtype=torch.float16, device="cuda")
            capi.check(capi.lib().ftcf_add_bias_attn_ffn_residual(
                capi.vp(o2), capi.vp(D(torch.from_numpy(ffn).half())), capi.vp(D(torch.from_numpy(att).half())),
                capi.vp(X), capi.vp(Bt), m, n, tp, inplace, 1, sp()))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(o2.cpu().float().numpy(), ref)  # elementwise: bit exact


# every head size of the reference (DecoderSelfAttentionLayer.cc:280-282): 64 / 128 on the whole (tl, s_max) grid, the others
# -- lane groups with idle lanes (48, 80, 96, 144 ... 224) or other group widths (32, 256) -- on a short and a looped case
_MMHA_CASES = [(dh, nh, rot, tl, s_max) for (dh, nh, rot) in [(128, 5, 32), (64, 4, 16)]
               for (tl, s_max) in [(0, 1024), (1, 1024), (31, 1024), (100, 1024), (700, 1024), (5000, 6000), (4099, 4100)]]
_MMHA_CASES += [(dh, 3, rot, tl, s_max) for (dh, rot) in [(32, 32), (48, 16), (80, 80), (96, 24), (144, 64), (160, 32), (192, 192),
                                                          (224, 56), (256, 128)]
                for (tl, s_max) in [(100, 1024), (2500, 2600)]]


@pytest.mark.parametrize("dh,nh,rot,tl,s_max", _MMHA_CASES)
def test_masked_multihead_attention_matches_oracle(dh, nh, rot, tl, s_max):
    """s_max 1024: every KV split fits the all-in-registers form; 6000 / 4100 / 2600: the looped form."""
    rng = np.random.RandomState(tl + dh)
    B = 3
    hl = nh * dh
    kc = orc.round_half(rng.randn(B, nh, s_max, dh).astype(np.float32))
    vc = orc.round_half(rng.randn(B, nh, s_max, dh).astype(np.float32))
    qkv = orc.round_half(rng.randn(B, 3 * hl).astype(np.float32))
    bias = orc.round_half(0.1 * rng.randn(3 * hl).astype(np.float32))
    seq_len = np.array([tl, max(tl - 1, 0), tl], dtype=np.int32)
    pad = np.array([0, 2, 0], dtype=np.int32)
    masked = np.zeros((B, s_max), dtype=np.uint8)
    if tl > 8:
        masked[1, 3:5] = 1
    finished = np.array([0, 0, 1], dtype=np.uint8)
    step = tl + 1
    kc_o, vc_o = kc.copy(), vc.copy()
    ref = orc.mmha_step(qkv, bias, kc_o, vc_o, seq_len, pad, masked, finished, nh, dh, rot, step, fp16=True)
    t = lambda a, dt=torch.float16: D(torch.from_numpy(a).to(dt))
    Kc, Vc = t(kc), t(vc)
    ctx = torch.zeros((B, hl), dtype=torch.float16, device="cuda")
    wsb = capi.lib().ftcf_masked_multihead_attention_workspace(B, nh, dh, s_max)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    capi.check(capi.lib().ftcf_masked_multihead_attention(
        capi.vp(t(qkv)), capi.vp(t(bias)), capi.vp(Kc), capi.vp(Vc), capi.vp(t(seq_len, torch.int32)),
        capi.vp(t(pad, torch.int32)), capi.vp(t(masked, torch.uint8)), capi.vp(t(finished, torch.uint8)), B, nh, dh, rot,
        s_max, step, capi.vp(ctx), capi.vp(ws), C.c_size_t(wsb), sp()))
    torch.cuda.synchronize()
    got = ctx.cpu().float().numpy()
    np.testing.assert_allclose(got[:2], ref[:2], rtol=1e-2, atol=2e-3)
    # the new key / value rows were appended to the cache exactly (bias + rotary, half rounding points)
    for b in range(2):
        np.testing.assert_allclose(Kc.cpu().float().numpy()[b, :, seq_len[b]], kc_o[b, :, seq_len[b]], rtol=2e-3, atol=2e-3)
        np.testing.assert_array_equal(Vc.cpu().float().numpy()[b, :, seq_len[b]], vc_o[b, :, seq_len[b]])
    assert np.all(got[2] == 0)  # finished row untouched


@pytest.mark.parametrize("dh,nh,rot", [(128, 3, 32), (64, 4, 16), (32, 2, 32), (48, 3, 16), (80, 2, 80), (96, 2, 24), (144, 2, 64),
                                       (160, 2, 32), (192, 2, 192), (224, 1, 56), (256, 2, 128)])
def test_context_attention_matches_oracle(dh, nh, rot):
    rng = np.random.RandomState(11)
    B, S, s_max = 2, 70, 96
    hl = nh * dh
    lens = np.array([70, 37], dtype=np.int32)
    qkv = orc.round_half(rng.randn(B * S, 3 * hl).astype(np.float32))
    bias = orc.round_half(0.1 * rng.randn(3 * hl).astype(np.float32))
    kc_o = np.zeros((B, nh, s_max, dh), dtype=np.float32)
    vc_o = np.zeros_like(kc_o)
    ref = orc.context_attention(qkv, bias, lens, kc_o, vc_o, B, S, nh, dh, rot, fp16=True)
    t = lambda a, dt=torch.float16: D(torch.from_numpy(a).to(dt))
    Kc = torch.zeros((B, nh, s_max, dh), dtype=torch.float16, device="cuda")
    Vc = torch.zeros_like(Kc)
    ctx = torch.zeros((B * S, hl), dtype=torch.float16, device="cuda")
    capi.check(capi.lib().ftcf_context_attention(capi.vp(t(qkv)), capi.vp(t(bias)), capi.vp(t(lens, torch.int32)),
                                                 capi.vp(Kc), capi.vp(Vc), B, S, nh, dh, rot, s_max, capi.vp(ctx), sp()))
    torch.cuda.synchronize()
    got = ctx.cpu().float().numpy().reshape(B, S, hl)
    refr = ref.reshape(B, S, hl)
    for b in range(B):
        np.testing.assert_allclose(got[b, :lens[b]], refr[b, :lens[b]], rtol=1e-2, atol=2e-3)
    np.testing.assert_allclose(Kc.cpu().float().numpy()[:, :, :S], kc_o[:, :, :S], rtol=2e-3, atol=2e-3)
    np.testing.assert_array_equal(Vc.cpu().float().numpy()[:, :, :S], vc_o[:, :, :S])


@pytest.mark.parametrize("dh,nh", [(128, 3), (64, 2)])
def test_context_attention_key_split_matches_oracle_and_the_unsplit_form(dh, nh, monkeypatch):
    """Prompts of >= 8 key tiles: the heaviest query blocks are cut in two along their keys and merged inside the launch
    (FTCF_CTX_SPLIT, default on).  Ragged rows (one shorter than the split blocks' start, one ending inside a split block, one full)
    against the oracle, the split and the unsplit form against each other, and the split form bit-identical on repeats (whichever
    half arrives second merges: the merge is symmetric)."""
    rng = np.random.RandomState(5)
    B, S, s_max, rot = 3, 600, 640, 32
    hl = nh * dh
    lens = np.array([600, 130, 421], dtype=np.int32)
    qkv = orc.round_half(rng.randn(B * S, 3 * hl).astype(np.float32))
    bias = orc.round_half(0.1 * rng.randn(3 * hl).astype(np.float32))
    kc_o = np.zeros((B, nh, s_max, dh), dtype=np.float32)
    vc_o = np.zeros_like(kc_o)
    ref = orc.context_attention(qkv, bias, lens, kc_o, vc_o, B, S, nh, dh, rot, fp16=True).reshape(B, S, hl)
    t = lambda a, dt=torch.float16: D(torch.from_numpy(a).to(dt))

    def run(split):
        monkeypatch.setenv("FTCF_CTX_SPLIT", split)
        Kc = torch.zeros((B, nh, s_max, dh), dtype=torch.float16, device="cuda")
        Vc = torch.zeros_like(Kc)
        ctx = torch.zeros((B * S, hl), dtype=torch.float16, device="cuda")
        capi.check(capi.lib().ftcf_context_attention(capi.vp(t(qkv)), capi.vp(t(bias)), capi.vp(t(lens, torch.int32)),
                                                     capi.vp(Kc), capi.vp(Vc), B, S, nh, dh, rot, s_max, capi.vp(ctx), sp()))
        torch.cuda.synchronize()
        return ctx.cpu().float().numpy().reshape(B, S, hl)

    on, on2, off = run("1"), run("1"), run("0")
    for b in range(B):
        np.testing.assert_allclose(on[b, :lens[b]], ref[b, :lens[b]], rtol=1e-2, atol=2e-3)
        np.testing.assert_allclose(on[b, :lens[b]], off[b, :lens[b]], rtol=2e-3, atol=1e-3)
        np.testing.assert_array_equal(on[b, :lens[b]], on2[b, :lens[b]])
    assert not np.array_equal(on[0], off[0])  # (the split really ran: a different association of the same sums)


def test_gemv_and_mfma_paths_agree_at_codefuse_13b_shapes():
    """Full BASELINE sizes (H=5120, I=20480): the m<=4 VALU GEMV and the m>4 MFMA GEMM read the same tiled image;
    they must agree with each other and with an fp32 torch reference of the dequantised weights on the device."""
    torch.manual_seed(7)
    for (k, n) in ((5120, 15360), (20480, 5120)):
        w = (torch.randn(k, n, device="cuda") * 0.02).half()
        q, s = quant(w.cpu().float().numpy())
        Q, S = q.cuda(), s.cuda()
        x = torch.randn(1, k, device="cuda").half()
        o1 = torch.empty((1, n), dtype=torch.float16, device="cuda")
        capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(x), capi.vp(Q), capi.vp(S), None, 0, capi.vp(o1), 1, n, k, sp()))
        x8 = x.repeat(8, 1).contiguous()
        o8 = torch.empty((8, n), dtype=torch.float16, device="cuda")
        capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(x8), capi.vp(Q), capi.vp(S), None, 0, capi.vp(o8), 8, n, k, sp()))
        torch.cuda.synchronize()
        q_rm = torch.empty((k, n), dtype=torch.int8)
        capi.check(capi.lib().ftcf_int8_tiled_to_rowmajor(capi.vp(q), C.c_size_t(k), C.c_size_t(n), capi.vp(q_rm)))
        deq = (q_rm.cuda().half() * S).float()
        ref = (x.float() @ deq)
        torch.testing.assert_close(o1.float(), ref, rtol=2e-3, atol=3e-3)
        torch.testing.assert_close(o8[3:4].float(), ref, rtol=2e-3, atol=3e-3)
        assert torch.equal(o8[0], o8[7])
