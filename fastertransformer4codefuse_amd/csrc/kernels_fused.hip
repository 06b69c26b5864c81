// Horizontally fused decode-step launch: decoder attention (split-KV MMHA) and the LN2 -> FFN1 (+bias, gelu) weight
// stream share ONE grid.  The attention workgroups are latency bound (QK^T -> softmax -> PV -> merge) and use a small
// part of the chip; running them beside ~105 MB of FFN weight streaming hides their whole duration under HBM traffic
// that has to happen anyway (parallel residual: FFN1 depends on the layer input only, GptNeoXDecoder.cc:301-340).
//   blocks [0, n_gemv)             : ln_gemv_block, segment 1 only (layernorm_kernels.cu:157-286 + FfnLayer.cc:203-217)
//   blocks [n_gemv, n_gemv+n_attn) : mmha_block  (decoder_masked_multihead_attention_template.hpp:1099-1919)
// The streaming blocks come first in the grid: they are few (1.25 per CU) and long running, so every CU starts pulling
// weights at once and the many short attention workgroups fill the remaining wave slots around them (measured:
// attention-first leaves one slot per CU for the weight stream until the attention blocks retire, 30.7 us vs ...).
#include "attn_device.hip.h"
#include "gemv_device.hip.h"

namespace ftcf {

template<bool INT8, int M, int DH>
__global__ __launch_bounds__(256) void k_mmha_ln_gemv(const MmhaParams ap, const LnGemvParams gp, const int n_attn)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_last;
    const int      n_gemv = gp.blocks0 + gp.blocks1;
    const int      bid    = (int)blockIdx.x;
    if (bid < n_gemv) {
        ln_gemv_group_block<INT8, M>(gp, smem, bid, 2);  // 2 column groups x 2 K-halves per 4-wave workgroup
    }
    else {
        const int a  = bid - n_gemv;
        const int sp = a % ap.nsplit;
        const int hb = a / ap.nsplit;
        mmha_block<DH>(ap, smem, s_last, hb % ap.nh, hb / ap.nh, sp);
    }
    (void)n_attn;
}

template<bool INT8, int M>
static void launch_m(const MmhaParams& ap, const LnGemvParams& gp, hipStream_t s)
{
    const int    n_attn = ap.nh * ap.B * ap.nsplit;
    const size_t smem_a = mmha_smem_bytes(ap.dh, ap.s_max, ap.nsplit);
    const size_t smem_g = (size_t)M * (gp.K + XPAD) * 2 + (2 * 4 + 4 * M * 16) * sizeof(float);
    const size_t smem   = std::max(smem_a, smem_g);
    const int grid = n_attn + gp.blocks0 + gp.blocks1;
    if (ap.dh == 128) {
        hipLaunchKernelGGL((k_mmha_ln_gemv<INT8, M, 128>), dim3(grid), dim3(256), smem, s, ap, gp, n_attn);
    }
    else {
        hipLaunchKernelGGL((k_mmha_ln_gemv<INT8, M, 64>), dim3(grid), dim3(256), smem, s, ap, gp, n_attn);
    }
}

void launch_mmha_ln_gemv(const MmhaParams& ap, const LnGemvParams& gp, bool int8, int M, hipStream_t s)
{
    FTCF_CHECK_ARG(M >= 1 && M <= 4 && M == ap.B, "fused attention + GEMV supports 1..4 rows");
    FTCF_CHECK_ARG(ap.dh == 64 || ap.dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(ap.nsplit >= 1 && ap.nsplit <= 16 && ap.gran != nullptr, "bad split-KV configuration");
    FTCF_CHECK_ARG(gp.K % 64 == 0, "K must be a multiple of 64");
    if (int8) {
        switch (M) {
            case 1: launch_m<true, 1>(ap, gp, s); break;
            case 2: launch_m<true, 2>(ap, gp, s); break;
            case 3: launch_m<true, 3>(ap, gp, s); break;
            default: launch_m<true, 4>(ap, gp, s); break;
        }
    }
    else {
        switch (M) {
            case 1: launch_m<false, 1>(ap, gp, s); break;
            case 2: launch_m<false, 2>(ap, gp, s); break;
            case 3: launch_m<false, 3>(ap, gp, s); break;
            default: launch_m<false, 4>(ap, gp, s); break;
        }
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
