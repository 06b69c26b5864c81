"""Random-shape check of the GEMM entry points (ftcf_fpA_intB_gemm / ftcf_fp16_gemm) against the oracle's restatement
(orc.gemm: q.half() * scale in fp16, fp32 accumulation, the reference's epilogue): m in 1..400 (GEMV, burst, split-K tiled with 32 / 48 /
64-row tiles, ring and plain forms), n a multiple of 16, k a multiple of 64, bias / gelu on and off, every launch repeated (the
in-launch reductions must be order independent).  Usage: python tools/fuzz_gemm.py [--seconds 300] [--seed 0]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastertransformer4codefuse_amd import capi  # noqa: E402
from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as quantize  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def run(seconds=300.0, seed=0, max_cases=None):
    """random cases until `seconds` have passed or `max_cases` have run (tests/test_gpu_headline_shapes.py runs a fixed-seed
    slice inside the -m gpu suite); returns (cases, worst error as a fraction of the tolerance)"""
    a = argparse.Namespace(seconds=seconds, seed=seed)
    capi.require_gpu()
    L = capi.lib()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.RandomState(a.seed)
    t0, cases, worst = time.time(), 0, 0.0
    while time.time() - t0 < a.seconds and (max_cases is None or cases < max_cases):
        n = 16 * int(rng.randint(1, 129))
        k = 64 * int(rng.choice([1, 2, 3, 4, 5, 8, 9, 10, 16, 17, 20, 33, 36, 40, 64, 80]))
        int8 = bool(rng.randint(0, 2))
        w = (rng.standard_normal((k, n)) * 0.02).astype(np.float32)
        w16 = torch.from_numpy(w).half()
        if int8:
            q, s = quantize(w16.contiguous())
            qd, sd = q.cuda(), s.cuda()
            q_rm, s_o = orc.symmetric_quantize_int8(w16.float().numpy(), True)
        else:
            wt = torch.empty((k, n), dtype=torch.float16, device="cuda")
            wd = w16.cuda()
            capi.check(L.ftcf_fp16_rowmajor_to_tiled(capi.vp(wd), C.c_size_t(k), C.c_size_t(n), capi.vp(wt), sp))
        for m in sorted(set(int(x) for x in rng.choice([1, 2, 4, 5, 9, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 96, 100, 128, 129, 177,
                                                        200, 255, 256, 257, 300, 320, 321, 400], size=5))):
            act_kind = int(rng.randint(0, 2))
            use_bias = bool(rng.randint(0, 2)) or act_kind == 1
            x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)).half()
            b = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).half()
            xd, bd = x.cuda(), b.cuda()
            outs = []
            for _ in range(2):
                out = torch.empty((m, n), dtype=torch.float16, device="cuda")
                if int8:
                    capi.check(L.ftcf_fpA_intB_gemm(capi.vp(xd), capi.vp(qd), capi.vp(sd), capi.vp(bd) if use_bias else None, act_kind,
                                                    capi.vp(out), m, n, k, sp))
                else:
                    capi.check(L.ftcf_fp16_gemm(capi.vp(xd), capi.vp(wt), capi.vp(bd) if use_bias else None, act_kind, capi.vp(out), m,
                                                n, k, sp))
                torch.cuda.synchronize()
                outs.append(out.cpu())
            assert torch.equal(outs[0], outs[1]), ("not repeatable", m, n, k, int8)
            bias_np = b.float().numpy() if use_bias else None
            if int8:
                ref = orc.gemm(x.float().numpy(), q=q_rm, scale=s_o, bias=bias_np, act=act_kind, fp16=True)
            else:
                ref = orc.gemm(x.float().numpy(), W=w16.float().numpy(), bias=bias_np, act=act_kind, fp16=True)
            got = outs[0].float().numpy()
            err = np.abs(got - ref) / (2e-3 + 2e-3 * np.abs(ref))  # rtol / atol of tests/test_gpu_kernels.py::test_fp16_gemm
            worst = max(worst, float(err.max()))
            assert err.max() <= 1.0, ("mismatch", m, n, k, int8, act_kind, use_bias, float(err.max()))
            cases += 1
    return cases, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    t0 = time.time()
    cases, worst = run(a.seconds, a.seed)
    print(f"fuzz_gemm: {cases} cases in {time.time() - t0:.0f} s, worst error {worst:.3f} of the tolerance (rtol 2e-3, atol 2e-3)")


if __name__ == "__main__":
    main()
