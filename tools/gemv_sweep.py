"""Micro-benchmark: m=1 weight-only int8 GEMV through the C ABI for several (k, n): fixed cost vs streaming slope."""
import sys; sys.path.insert(0, '.')
import ctypes as C
import torch
from fastertransformer4codefuse_amd import capi
L = capi.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(k, n, iters=200, m=1):
    q = torch.randint(0, 255, (k * n,), dtype=torch.uint8, device='cuda').view(torch.int8)
    s = (torch.rand(n, device='cuda') * 2e-4 + 1e-4).half()
    x = torch.randn(m, k, device='cuda').half()
    o = torch.empty(m, n, device='cuda', dtype=torch.float16)
    # rotate over several weight copies so that the 256 MB infinity cache cannot hold the matrix
    copies = max(1, int(600e6 // (k * n)) )
    qs = [q] + [q.clone() for _ in range(min(copies, 8) - 1)]
    for w in qs[:2]:
        capi.check(L.ftcf_fpA_intB_gemm(capi.vp(x), capi.vp(w), capi.vp(s), None, 0, capi.vp(o), m, n, k, sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        capi.check(L.ftcf_fpA_intB_gemm(capi.vp(x), capi.vp(qs[i % len(qs)]), capi.vp(s), None, 0, capi.vp(o), m, n, k, sp))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"k={k:6d} n={n:6d} MB={k*n/1e6:7.1f} us={us:7.2f} TB/s={k*n/us/1e6:5.2f}")
import os
shapes = [(5120, 15360), (5120, 20480), (5120, 35840)] if os.environ.get('SWEEP') == 'ka' else [(25600, 5120), (20480, 5120), (5120, 5120), (5120, 20480)] if os.environ.get('SWEEP') == 'k3' else [(5120, 1024), (5120, 5120), (5120, 15360), (5120, 35840), (5120, 71680), (5120, 143360), (20480, 5120), (20480, 10240), (20480, 20480), (1024, 5120)]
for (k, n) in shapes:
    run(k, n)
