"""The CUDA build's SM75..SM89 int8 weight layout: oracle restatement pinned by the reference's own
known-answer tests (tests/weight_only_quant_ops/th_weight_quant_ops_unit_tests.py), product importer pinned by the oracle.
Host only: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest
import torch  # noqa: F401  -- before libftcf.so: one HIP / OpenMP runtime per process (capi.lib docstring)

from oracle import oracle as orc
from fastertransformer4codefuse_amd import capi

I8P = C.POINTER(C.c_int8)


def test_oracle_row_permutation_is_the_reference_map():
    # th_weight_quant_ops_unit_tests.py:31-47 (reference_interleave): groups of 16 rows, map 0 1 8 9 2 3 10 11 ...
    rng = np.random.RandomState(0)
    for K, N in ((128, 128), (256, 512), (1024, 1024)):
        t = rng.randint(-128, 128, size=(K, N)).astype(np.int8)
        pm = [0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15]
        ref = t.reshape(-1, 16, N)[:, pm, :].reshape(K, N)
        np.testing.assert_array_equal(orc.sm80_preprocess_int8(t, "permute"), ref)


def test_oracle_add_bias_interleave_int8_known_answer():
    # th_weight_quant_ops_unit_tests.py:110-116
    x = np.array([[-104, -70, -36, 127, 16, 50, 84, 118]], dtype=np.int8)
    exp = (np.array([[-104, -36, -70, 127, 16, 84, 50, 118]], dtype=np.int32) + 128).astype(np.uint8).view(np.int8)
    np.testing.assert_array_equal(orc.sm80_preprocess_int8(x, "bias"), exp)


def test_oracle_transpose_known_answer():
    # th_weight_quant_ops_unit_tests.py:118-143: _transpose == permute([.., 1, 0])
    rng = np.random.RandomState(1)
    t = rng.randint(-128, 128, size=(128, 4096)).astype(np.int8)
    np.testing.assert_array_equal(orc.sm80_preprocess_int8(t, "transpose"), t.T)


def test_oracle_column_interleave_structure():
    # cutlass_preprocessors.cc:437-498 with rows_per_column_tile = 64, columns_interleaved = 2
    # (mixed_gemm_B_layout.h:59-72): a 128-byte line = 64 k of column 2j followed by the same 64 k of column 2j+1
    K, N = 256, 8
    col = (np.arange(N)[:, None] * 1000 + np.arange(K)[None, :]).astype(np.int32)  # value identifies (n, k)
    tagged = (col % 251).astype(np.int8)  # any int8 image; positions are checked through a second, exact pass below
    out = orc.sm80_preprocess_int8(tagged, "interleave").reshape(N // 2, K // 64, 2, 64)
    for j in range(N // 2):
        for tile in range(K // 64):
            np.testing.assert_array_equal(out[j, tile, 0], tagged[2 * j, tile * 64:(tile + 1) * 64])
            np.testing.assert_array_equal(out[j, tile, 1], tagged[2 * j + 1, tile * 64:(tile + 1) * 64])


@pytest.mark.parametrize("K,N", [(64, 16), (128, 256), (512, 1536), (1024, 256)])
def test_product_importer_inverts_the_oracle_layout(K, N):
    rng = np.random.RandomState(K + N)
    q = rng.randint(-128, 128, size=(K, N)).astype(np.int8)
    cuda = orc.sm80_preprocess_int8(q)  # what a CUDA build's .q.bin holds
    L = capi.lib()
    back = np.empty_like(q)
    capi.check(L.ftcf_int8_cuda_sm80_to_rowmajor(cuda.ctypes.data_as(I8P), K, N, back.ctypes.data_as(I8P)))
    np.testing.assert_array_equal(back, q)
    fwd = np.empty(K * N, dtype=np.int8)
    capi.check(L.ftcf_int8_rowmajor_to_cuda_sm80(q.ctypes.data_as(I8P), K, N, fwd.ctypes.data_as(I8P)))
    np.testing.assert_array_equal(fwd, cuda)


def test_import_command_relayouts_a_cuda_checkpoint(tmp_path):
    """A tiny FT checkpoint whose .q.bin files are in the CUDA layout -> `import-cuda-qbin` -> identical to what our own
    quantiser writes for the same weights."""
    import configparser
    from fastertransformer4codefuse_amd import convert
    H, nh, I, L_ = 128, 2, 256, 1
    src, ours, imported = tmp_path / "cuda", tmp_path / "ours", tmp_path / "imported"
    src.mkdir()
    cfg = configparser.ConfigParser()
    cfg["gptneox"] = dict(model_name="t", head_num=str(nh), size_per_head=str(H // nh), inter_size=str(I), num_layer=str(L_),
                          vocab_size="64", rotary_embedding=str(16), start_id="0", end_id="2", use_gptj_residual="1",
                          weight_data_type="fp16")
    with open(src / "config.ini", "w") as f:
        cfg.write(f)
    rng = np.random.RandomState(3)
    shapes = {"attention.query_key_value.weight": (H, 3 * H), "attention.dense.weight": (H, H),
              "mlp.dense_h_to_4h.weight": (H, I), "mlp.dense_4h_to_h.weight": (I, H)}
    for fn, (K, N) in shapes.items():
        (rng.standard_normal((K, N)) * 0.05).astype(np.float16).tofile(src / f"model.layers.0.{fn}.0.bin")
    convert.quant_and_save(str(src), str(ours), 1)  # our layout, from the fp16 weights
    # emulate the CUDA build: same int8 values, CUDA layout
    L = capi.lib()
    for fn, (K, N) in shapes.items():
        tiled = np.fromfile(ours / f"model.layers.0.{fn}.0.q.bin", dtype=np.int8)
        rm = np.empty(K * N, np.int8)
        capi.check(L.ftcf_int8_tiled_to_rowmajor(tiled.ctypes.data_as(I8P), K, N, rm.ctypes.data_as(I8P)))
        orc.sm80_preprocess_int8(rm.reshape(K, N)).tofile(src / f"model.layers.0.{fn}.0.q.bin")
        np.fromfile(ours / f"model.layers.0.{fn}.0.s.bin", dtype=np.float16).tofile(src / f"model.layers.0.{fn}.0.s.bin")
    convert.import_cuda_qbin(str(src), str(imported), 1)
    for fn in shapes:
        a = np.fromfile(imported / f"model.layers.0.{fn}.0.q.bin", dtype=np.int8)
        b = np.fromfile(ours / f"model.layers.0.{fn}.0.q.bin", dtype=np.int8)
        np.testing.assert_array_equal(a, b)
