"""Pins the oracle's weight-only quantiser / dequant GEMM to the reference's own test expectations.

Reference tests (need a CUDA build to run, but their expected values are formulas / literals):
  tests/gemm_dequantize/th_gemm_dequantize.py:22-40   identity-activation dequant must be bit exact
  tests/gemm_dequantize/th_gemm_dequantize.py:65-115  fp16 x int8 GEMM vs torch.matmul(act, q.to(fp16)*scale),
                                                      rtol 1e-3 / atol 2e-3, weights N(0, 0.002), seed 734876213
  tests/gemm_dequantize/th_gemm_dequantize.py:139-169 bias / gelu epilogues
  kernels/cutlass_kernels/cutlass_preprocessors.cc:603-643  scale = max|w|/128, q = clamp(round(w/scale))
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc


def test_quantizer_formula_known_answers():
    # column max 1.28 -> scale 0.01 ; +max maps to 128 -> clamped to 127 ; -max maps to -128
    w = np.array([[1.28, -0.64], [-1.28, 0.005], [0.014999, 0.64], [0.0, -0.0025]], dtype=np.float32)
    q, s = orc.symmetric_quantize_int8(w, weight_is_half=False)
    np.testing.assert_allclose(s, [0.01, 0.005], rtol=1e-6)
    assert q[:, 0].tolist() == [127, -128, 1, 0]
    # round() is half-away-from-zero: 0.005/0.005 = 1 ; -0.0025/0.005 = -0.5 -> -1
    assert q[:, 1].tolist() == [-128, 1, 127, -1]


def test_quantizer_scale_is_stored_in_weight_dtype_but_division_uses_fp32_scale():
    rng = np.random.RandomState(0)
    w = orc.round_half(rng.randn(64, 8).astype(np.float32) * 0.02)
    q, s = orc.symmetric_quantize_int8(w, weight_is_half=True)
    s32 = np.abs(w).max(0) / 128.0
    np.testing.assert_array_equal(s, orc.round_half(s32))
    exp = np.clip(np.sign(w / s32) * np.floor(np.abs(w / s32) + 0.5), -128, 127).astype(np.int8)
    np.testing.assert_array_equal(q, exp)


def test_zero_column_quantizes_like_reference_nan_path():
    w = np.zeros((4, 2), dtype=np.float32)
    w[:, 1] = [1, -1, 0.5, 0]
    q, s = orc.symmetric_quantize_int8(w, weight_is_half=False)
    assert s[0] == 0.0 and q[:, 0].tolist() == [127] * 4  # 0/0 -> NaN -> min/max chain yields 127; dequant = 0


def test_identity_activation_dequant_is_bit_exact():
    # th_gemm_dequantize.py:22-40: act = I, result must equal q.to(fp16) * scale exactly
    torch.manual_seed(734876213)
    k = n = 128
    q = torch.randint(-128, 128, (k, n), dtype=torch.int8)
    scale = (torch.rand(n) * 0.01 + 0.001).half()
    ref = (q.half() * scale).float().numpy()
    out = orc.gemm(np.eye(k, dtype=np.float32), q=q.numpy(), scale=scale.float().numpy(), fp16=True)
    np.testing.assert_array_equal(out, ref)


@pytest.mark.parametrize("m,n,k", [(1, 1024, 4096), (8, 2048, 4096), (33, 1024, 8192), (177, 1024, 4096)])
def test_gemm_matches_reference_formula_within_reference_tolerance(m, n, k):
    torch.manual_seed(734876213)
    w = torch.randn(k, n) * 0.002
    q_np, s_np = orc.symmetric_quantize_int8(w.half().float().numpy(), weight_is_half=True)
    act = torch.randn(m, k).half()
    ref = torch.matmul(act.float(), (torch.from_numpy(q_np).half() * torch.from_numpy(s_np).half()).float()).half()
    out = orc.gemm(act.float().numpy(), q=q_np, scale=s_np, fp16=True)
    torch.testing.assert_close(torch.from_numpy(out).half(), ref, rtol=1e-3, atol=2e-3)


def test_gemm_bias_gelu_epilogue():
    torch.manual_seed(1)
    m, n, k = 16, 256, 512
    w = torch.randn(k, n) * 0.05
    q_np, s_np = orc.symmetric_quantize_int8(w.half().float().numpy(), weight_is_half=True)
    act = torch.randn(m, k).half()
    bias = torch.randn(n).half()
    pre = torch.matmul(act.float(), (torch.from_numpy(q_np).half() * torch.from_numpy(s_np).half()).float()) + bias.float()
    ref = torch.nn.functional.gelu(pre, approximate="tanh").half()
    out = orc.gemm(act.float().numpy(), q=q_np, scale=s_np, bias=bias.float().numpy(), act=1, fp16=True)
    torch.testing.assert_close(torch.from_numpy(out).half(), ref, rtol=1e-3, atol=2e-3)


def test_layernorm_and_residual_against_torch():
    rng = np.random.RandomState(3)
    x = orc.round_half(rng.randn(5, 256).astype(np.float32))
    g = orc.round_half(1 + 0.1 * rng.randn(256).astype(np.float32))
    b = orc.round_half(0.1 * rng.randn(256).astype(np.float32))
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x), (256,), torch.from_numpy(g), torch.from_numpy(b), 1e-5)
    np.testing.assert_allclose(orc.layernorm(x, g, b, fp16=False), ref.numpy(), atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(orc.layernorm(x, g, b, fp16=True), ref.numpy(), atol=1e-2, rtol=1e-2)
    out = orc.add_bias_attn_ffn_residual(x, x * 0.5, x * 0.25, b, tp=1, inplace_variant=True, fp16=False)
    np.testing.assert_allclose(out, x + x * 0.5 + x * 0.25 + b, rtol=1e-6, atol=1e-6)


def test_uniform_is_in_half_open_interval_and_deterministic():
    u = [orc.lib().orc_uniform(7, 3, i) for i in range(1000)]
    assert min(u) > 0.0 and max(u) <= 1.0
    assert u[:5] == [orc.lib().orc_uniform(7, 3, i) for i in range(5)]
    assert abs(np.mean(u) - 0.5) < 0.05


def test_hardware_half_rounding_is_the_software_definition():
    """oracle/ftcf_oracle.c uses the F16C conversion where the host has it; the software round-to-nearest-even
    conversion is the definition.  Every class of value: normals, half subnormals, ties, overflow, inf, nan, signed 0."""
    import ctypes as C
    lib = orc.lib()
    rng = np.random.RandomState(7)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.randint(-9, 6, 20000).astype(np.float32),
        np.float32([0.0, -0.0, 65504.0, 65519.99, 65520.0, 65536.0, 1e9, -1e9, np.inf, -np.inf, 2.0 ** -14, 2.0 ** -24,
                    2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -26, 5.9604645e-08, 6.1035156e-05, 1.0 + 2.0 ** -11,
                    1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20]),
        (np.arange(0, 2049, dtype=np.float32) + 0.5) * np.float32(2.0 ** -24),  # ties between half subnormals
    ])
    for v in vals:
        a, b = lib.orc_round_half(C.c_float(v)), lib.orc_round_half_soft(C.c_float(v))
        assert a == b and np.signbit(a) == np.signbit(b), (v, a, b)
    nan = lib.orc_round_half(C.c_float(np.nan))
    assert np.isnan(nan) and np.isnan(lib.orc_round_half_soft(C.c_float(np.nan)))
