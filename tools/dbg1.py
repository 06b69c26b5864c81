import sys; sys.path.insert(0,'.')
import torch, numpy as np, ctypes as C
from fastertransformer4codefuse_amd import capi
from fastertransformer4codefuse_amd.gptneox_op import symmetric_quantize_last_axis_of_batched_matrix_int8 as qf
from oracle import oracle as orc
sp=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
for (m,k,n) in [(4,128,256),(1,128,256),(4,512,256),(4,576,256),(2,640,64)]:
    w=(torch.randn(k,n)*0.05).half()
    q,s=qf(w.contiguous())
    q_rm,s_o=orc.symmetric_quantize_int8(w.float().numpy(),True)
    ref=(torch.from_numpy(q_rm).half()*torch.from_numpy(s_o).half())
    A=torch.eye(k,dtype=torch.float16)[:m].contiguous().cuda()
    out=torch.empty((m,n),dtype=torch.float16,device='cuda')
    capi.check(capi.lib().ftcf_fpA_intB_gemm(capi.vp(A),capi.vp(q.cuda()),capi.vp(s.cuda()),None,0,capi.vp(out),m,n,k,sp()))
    torch.cuda.synchronize()
    bad=(out.cpu()!=ref[:m]).nonzero()
    print(m,k,n,'nbad',len(bad), bad[:12].tolist())
