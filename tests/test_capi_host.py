"""CPU (-m "not gpu"): the C-ABI library loads, exports every symbol include/ftcf.h declares, fails loudly without a
GPU, and its host-side quantiser (libth_common counterpart) matches the oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from fastertransformer4codefuse_amd import capi
from fastertransformer4codefuse_amd.gptneox_op import GptNeoXOp, symmetric_quantize_last_axis_of_batched_matrix_int8
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ftcf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ftcf_[a-z0-9_A-Z]+)\s*\(", hdr)) - {"ftcf_token_callback"})
    assert declared, "no declarations parsed"
    lib = capi.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTED) == declared
    assert lib.ftcf_version() == 100


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_ops_fail_loudly_without_a_gpu():
    assert capi.device_count() == 0
    with pytest.raises(capi.FtcfError):
        capi.require_gpu()
    rc = capi.lib().ftcf_layernorm(None, None, None, None, 1, 8, C.c_float(1e-5), 1, None)
    assert rc == -5 and b"no HIP device" in capi.lib().ftcf_last_error()
    with pytest.raises(capi.FtcfError):
        GptNeoXOp(None, 0, 4, 64, 1024, 2, 512, 16, 0, 2, 1, 1, 0, 1024, True, [torch.zeros(1)], [], [])


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_host_quantizer_matches_oracle_bit_exact(dtype):
    torch.manual_seed(0)
    K, N = 256, 96
    w = (torch.randn(K, N) * 0.02).to(dtype).contiguous()
    w[:, 5] = 0  # all-zero column: NaN path of the reference
    q, s = symmetric_quantize_last_axis_of_batched_matrix_int8(w)
    assert q.shape == w.shape and q.dtype == torch.int8 and s.dtype == dtype and s.shape == (N,)
    q_rm = torch.empty((K, N), dtype=torch.int8)
    capi.check(capi.lib().ftcf_int8_tiled_to_rowmajor(capi.vp(q), C.c_size_t(K), C.c_size_t(N), capi.vp(q_rm)))
    oq, os_ = orc.symmetric_quantize_int8(w.float().numpy(), weight_is_half=(dtype == torch.float16))
    np.testing.assert_array_equal(q_rm.numpy(), oq)
    np.testing.assert_array_equal(s.float().numpy(), os_)
    # round trip of the tile layout
    q2 = torch.empty_like(q)
    capi.check(capi.lib().ftcf_int8_rowmajor_to_tiled(capi.vp(q_rm), C.c_size_t(K), C.c_size_t(N), capi.vp(q2)))
    assert torch.equal(q, q2)


def test_tile_layout_known_answer():
    """DESIGN.md layout: byte ((nt*KT+kt)*64 + lane)*16 + j = u8(q[kt*64 + (lane>>4)*16 + j][nt*16 + (lane&15)] + 128)."""
    K, N = 128, 32
    q = (np.arange(K * N, dtype=np.int64).reshape(K, N) % 251 - 125).astype(np.int8)
    t = np.empty_like(q)
    capi.check(capi.lib().ftcf_int8_rowmajor_to_tiled(capi.vp(q), C.c_size_t(K), C.c_size_t(N), capi.vp(t)))
    flat = t.reshape(-1).view(np.uint8)
    for (nt, kt, lane, j) in [(0, 0, 0, 0), (1, 1, 37, 9), (0, 1, 63, 15), (1, 0, 16, 3)]:
        k = kt * 64 + (lane >> 4) * 16 + j
        n = nt * 16 + (lane & 15)
        assert flat[((nt * 2 + kt) * 64 + lane) * 16 + j] == (int(q[k, n]) + 128)


def test_batched_3d_quantize_and_argument_checks():
    w = (torch.randn(2, 64, 16) * 0.1).half()
    q, s = symmetric_quantize_last_axis_of_batched_matrix_int8(w)
    assert q.shape == (2, 64, 16) and s.shape == (2, 16)
    with pytest.raises(RuntimeError):
        symmetric_quantize_last_axis_of_batched_matrix_int8(torch.zeros(4))
    with pytest.raises(capi.FtcfError):
        symmetric_quantize_last_axis_of_batched_matrix_int8(torch.zeros(60, 16))  # K % 64 != 0


def test_full_size_quantize_and_layout_round_trips():
    """CodeFuse-13B FFN matrix (5120 x 20480), where only size-independent properties are checked: tiled <-> row-major and
    CUDA-SM80 <-> row-major are exact inverses, every column reaches 127 / -128 (the scale is max|w| / 128 as in
    cutlass_preprocessors.cc:603-643), dequantised error <= scale / 2 except where +128 was clamped to 127."""
    torch.manual_seed(3)
    K, N = 5120, 20480
    w = (torch.randn(K, N) * 0.02).half().contiguous()
    q, s = symmetric_quantize_last_axis_of_batched_matrix_int8(w)
    L = capi.lib()
    q_rm = torch.empty((K, N), dtype=torch.int8)
    capi.check(L.ftcf_int8_tiled_to_rowmajor(capi.vp(q), C.c_size_t(K), C.c_size_t(N), capi.vp(q_rm)))
    q2 = torch.empty_like(q)
    capi.check(L.ftcf_int8_rowmajor_to_tiled(capi.vp(q_rm), C.c_size_t(K), C.c_size_t(N), capi.vp(q2)))
    assert torch.equal(q, q2)
    cu = torch.empty((K * N,), dtype=torch.int8)
    capi.check(L.ftcf_int8_rowmajor_to_cuda_sm80(capi.vp(q_rm), C.c_size_t(K), C.c_size_t(N), capi.vp(cu)))
    back = torch.empty((K, N), dtype=torch.int8)
    capi.check(L.ftcf_int8_cuda_sm80_to_rowmajor(capi.vp(cu), C.c_size_t(K), C.c_size_t(N), capi.vp(back)))
    assert torch.equal(back, q_rm)
    qa = q_rm.to(torch.int16).abs()
    assert bool((qa.max(dim=0).values >= 127).all())
    err = (q_rm.float() * s.float()[None, :] - w.float()).abs()
    lim = torch.where(q_rm == 127, s.float()[None, :] * 1.0, s.float()[None, :] * 0.5) + 1e-6
    assert bool((err <= lim).all())


def test_quantiser_accepts_bf16_like_the_reference():
    """WeightOnlyQuantOps.cc:149,205: bf16 weights go through symmetric_quantize<__nv_bfloat16, __nv_bfloat16> -- float(x)
    of every element, fp32 column maxima, int8 = round(w / (max / 128)) with the UNROUNDED scale, scales stored as bf16."""
    import torch
    from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as qf
    from fastertransformer4codefuse_amd import capi
    import ctypes as C
    torch.manual_seed(3)
    K, N = 128, 48
    w = (torch.randn(K, N) * 0.05).to(torch.bfloat16).contiguous()
    q, s = qf(w)
    assert q.dtype == torch.int8 and s.dtype == torch.bfloat16 and tuple(s.shape) == (N,)
    wf = w.float()
    col = wf.abs().max(dim=0).values / 128.0
    assert torch.equal(s, col.to(torch.bfloat16))  # round-to-nearest-even of the fp32 scale
    quo = wf / col
    ref = torch.clamp(torch.sign(quo) * torch.floor(quo.abs() + 0.5), -128, 127).to(torch.int8)  # std::round: half away from 0
    q_rm = torch.empty((K, N), dtype=torch.int8)
    capi.check(capi.lib().ftcf_int8_tiled_to_rowmajor(capi.vp(q), C.c_size_t(K), C.c_size_t(N), capi.vp(q_rm)))
    assert torch.equal(q_rm, ref)
    # the same values as fp32 input: identical integers, fp32 scales
    q32, s32 = qf(wf.contiguous())
    assert torch.equal(q32, q) and torch.equal(s32, col)
