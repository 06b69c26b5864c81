"""Per-shape timing of the prefill GEMMs through the C ABI (ftcf_fpA_intB_gemm / ftcf_fp16_gemm): the four GEMMs of a
CodeFuse-13B layer at m prompt tokens, HIP events around `reps` launches each.  Prints one JSON line per shape with
TFLOP/s and the fraction of the 2.5 PFLOP/s dense fp16 MFMA peak.  Usage: python tools/bench_gemm.py [--m 1024] [--fp16]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastertransformer4codefuse_amd import capi  # noqa: E402

PEAK = 2.5e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tp", type=int, default=1)
    a = ap.parse_args()
    lib = capi.lib()
    dev = torch.device("cuda:0")
    H, I = 5120, 20480
    shapes = [("qkv", 3 * H // a.tp, H), ("out", H, H // a.tp), ("ffn1", I // a.tp, H), ("ffn2", H, I // a.tp)]
    s = torch.cuda.current_stream().cuda_stream
    tot_t, tot_f = 0.0, 0.0
    for name, n, k in shapes:
        A = (torch.randn(a.m, k, device=dev) * 0.5).half()
        out = torch.empty(a.m, n, device=dev, dtype=torch.half)
        bias = torch.zeros(n, device=dev, dtype=torch.half)
        if a.fp16:
            W = torch.randint(-3, 4, (k * n,), device=dev, dtype=torch.int16).view(torch.half)  # layout irrelevant for timing
            W = (torch.randn(k * n, device=dev) * 0.02).half()
            call = lambda: capi.check(lib.ftcf_fp16_gemm(capi.vp(A), capi.vp(W), capi.vp(bias), 0, capi.vp(out), a.m, n, k,
                                                         C.c_void_p(s)))
        else:
            W = torch.randint(-127, 128, (k * n,), device=dev, dtype=torch.int8)
            sc = torch.full((n,), 0.001, device=dev, dtype=torch.half)
            call = lambda: capi.check(lib.ftcf_fpA_intB_gemm(capi.vp(A), capi.vp(W), capi.vp(sc), capi.vp(bias), 0,
                                                             capi.vp(out), a.m, n, k, C.c_void_p(s)))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        fl = 2.0 * a.m * n * k
        tot_t += us
        tot_f += fl
        print(json.dumps({"gemm": name, "m": a.m, "n": n, "k": k, "weights": "fp16" if a.fp16 else "int8", "us": round(us, 1),
                          "tflops": round(fl / us / 1e6, 1), "mfma_frac": round(fl / us / 1e6 / (PEAK / 1e12), 3)}))
    print(json.dumps({"gemm": "layer", "us": round(tot_t, 1), "tflops": round(tot_f / tot_t / 1e6, 1),
                      "mfma_frac": round(tot_f / tot_t / 1e6 / (PEAK / 1e12), 3)}))


if __name__ == "__main__":
    main()
