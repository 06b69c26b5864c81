import sys; sys.path.insert(0, '.')
import ctypes as C, os
import torch
from fastertransformer4codefuse_amd import capi
L = capi.lib()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, nh, dh, rot = int(os.environ.get('B', 1)), 40, 128, 32
s_max, tl = int(os.environ.get('SMAX', 1536)), int(os.environ.get('TL', 1100))
hl = nh * dh
nlayers = 8   # rotate over several caches so that the 256 MB infinity cache cannot hold them
Ks = [torch.randn(B, nh, s_max, dh, device='cuda').half() for _ in range(nlayers)]
Vs = [torch.randn(B, nh, s_max, dh, device='cuda').half() for _ in range(nlayers)]
qkv = torch.randn(B, 3 * hl, device='cuda').half(); bias = torch.randn(3 * hl, device='cuda').half()
seq = torch.full((B,), tl, dtype=torch.int32, device='cuda'); pad = torch.zeros(B, dtype=torch.int32, device='cuda')
msk = torch.zeros(B, s_max, dtype=torch.uint8, device='cuda'); fin = torch.zeros(B, dtype=torch.uint8, device='cuda')
ctx = torch.zeros(B, hl, device='cuda', dtype=torch.float16)
wsb = L.ftcf_masked_multihead_attention_workspace(B, nh, dh, s_max) * 4
ws = torch.zeros(wsb, dtype=torch.uint8, device='cuda')
def call(i):
    capi.check(L.ftcf_masked_multihead_attention(capi.vp(qkv), capi.vp(bias), capi.vp(Ks[i % nlayers]), capi.vp(Vs[i % nlayers]), capi.vp(seq), capi.vp(pad), capi.vp(msk), capi.vp(fin), B, nh, dh, rot, s_max, tl + 1 + i % 3, capi.vp(ctx), capi.vp(ws), C.c_size_t(wsb), sp))
for i in range(5): call(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for i in range(n): call(i)
e1.record(); torch.cuda.synchronize()
print(f"B={B} s_max={s_max} tl={tl} wgs={os.environ.get('FTCF_MMHA_WGS','640')}: {e0.elapsed_time(e1)*1e3/n:.2f} us per call (incl. memset of the slab)")
