import sys; sys.path.insert(0,'.')
import torch, numpy as np, ctypes as C
from fastertransformer4codefuse_amd import capi
from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as qf
from oracle import oracle as orc
sp=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
m,k,n=4,128,256
w=(torch.randn(k,n)*0.05).half()
q,s=qf(w.contiguous())
q_rm,s_o=orc.symmetric_quantize_int8(w.float().numpy(),True)
ref=(torch.from_numpy(q_rm).half()*torch.from_numpy(s_o).half())
A=torch.eye(k,dtype=torch.float16)[:m].contiguous().cuda(); Q=q.cuda(); S=s.cuda()
out=torch.empty((m,n),dtype=torch.float16,device='cuda')
capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(A),capi.vp(Q),capi.vp(S),None,0,capi.vp(out),m,n,k,sp()))
torch.cuda.synchronize()
bad=(out.cpu()!=ref[:m]).nonzero()
print('identity nbad',len(bad), sorted(set(bad[:,1].tolist()))[:70])
# LN
rng=np.random.RandomState(5); mm,nn=7,5120
x=orc.round_half(rng.randn(mm,nn).astype(np.float32)); g=orc.round_half(1+0.1*rng.randn(nn).astype(np.float32)); b=orc.round_half(0.1*rng.randn(nn).astype(np.float32))
X,G,Bt=(torch.from_numpy(a).half().cuda() for a in (x,g,b))
o=torch.empty((mm,nn),dtype=torch.float16,device='cuda')
capi.check(capi.lib().ftcf_layernorm(capi.vp(X),capi.vp(G),capi.vp(Bt),capi.vp(o),mm,nn,C.c_float(1e-5),1,sp()))
torch.cuda.synchronize()
ref=torch.nn.functional.layer_norm(X.float(),(nn,),G.float(),Bt.float(),1e-5)
print('LN maxerr vs torch', (o.float()-ref).abs().max().item(), 'oracle', np.abs(o.cpu().float().numpy()-orc.layernorm(x,g,b,fp16=True)).max())
print(o[0,:4].tolist(), ref[0,:4].tolist())
