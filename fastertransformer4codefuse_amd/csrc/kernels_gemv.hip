// Weight-streaming GEMV kernels for the per-token decode step (m <= 4 rows), gfx950.
//
// These replace, on the decode path, the reference's CutlassFpAIntBGemmRunner<half,uint8_t>::gemm / gemm_bias_act
// (kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:511-581) and cublasMMWrapper::Gemm
// (utils/cublasMMWrapper.cc:94-386) call sites of DecoderSelfAttentionLayer.cc:532-577,635-678 and
// FfnLayer.cc:203-231,330-343, fused with invokeGeneralLayerNorm (kernels/layernorm_kernels.cu:157-286) and
// invokeAddBiasAttentionFfnResidual (kernels/add_residual_kernels.cu:116-178).
//
// Roofline: HBM.  Every weight byte is read exactly once per token; algorithmic bytes per launch = K*N (int8)
// or 2*K*N (fp16) (+ 2N scales).  One wave-level load instruction fetches one 1 KiB tile (16 B per lane,
// non-temporal), 8 tiles are kept in flight per wave while the previous 8 are consumed from registers.
#include <cmath>

#include "gemv_device.hip.h"
#include "lm_head_device.hip.h"

namespace ftcf {

template<bool INT8, int M>
__global__ __launch_bounds__(256) void k_ln_gemv(const LnGemvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ln_gemv_block<INT8, M>(p, smem, (int)blockIdx.x);
}

template<bool INT8, int M>
__global__ __launch_bounds__(256) void k_ln_gemv_group(const LnGemvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ln_gemv_group_block<INT8, M>(p, smem, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Generic split-K GEMV: one block per 16-column group, its waves split up to two K segments
// (segment A = x_a * W_a, segment B = x_b * W_b).  Epilogues:
//   EPI_PLAIN    : out = half(acc_a [+ bias, gelu])                                   (single segment)
//   EPI_RESIDUAL : attn = half(acc_a), ffn = half(acc_b), out = ffn + attn + bias + x_in/TP
//                  (add_residual_kernels.cu:116-152; `inplace_variant` selects the fp32-sum form)
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int M, int EPI>
__global__ __launch_bounds__(GEMV_SPLITK_MAX_WAVES * 64) void k_gemv_splitk(const SplitKParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TK   = TileK<INT8>::value;
    const int     lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int     nw   = blockDim.x >> 6;
    const int     grp  = blockIdx.x;
    const int     c = lane & 15, g = lane >> 4;

    const int seg   = p.wave_seg[wid];
    const int t0    = p.wave_t0[wid];
    const int nt    = p.wave_nt[wid];
    const int KTseg = seg ? p.KT_b : p.KT_a;
    const int Kseg  = KTseg * TK;

    // per-wave private x slice in LDS: [M][slice_halves]
    f16*       xs     = reinterpret_cast<f16*>(smem) + (size_t)wid * M * p.slice_halves;
    const f16* xsrc   = seg ? p.x_b : p.x_a;
    const int  nhalf  = nt * TK;
    const char*  wbase = reinterpret_cast<const char*>(seg ? p.W_b : p.W_a);
    const u32x4* wp    = reinterpret_cast<const u32x4*>(wbase + (((size_t)grp * KTseg + t0) * 64 + lane) * 16);
    WaveStream<INT8, M> ws;
    constexpr int SV = (M == 1) ? 8 : ((M == 2) ? 4 : 2);  // slice vectors per lane held in registers
    if (nhalf <= 512 * SV) {
        // x slice loads first (L2 hits), then the first weight batch: the LDS staging overlaps the HBM latency
        f16x8 xv[M][SV];
#pragma unroll
        for (int j = 0; j < SV; j++) {
            const int i = lane * 8 + j * 512;
#pragma unroll
            for (int m = 0; m < M; m++) {
                xv[m][j] = (i < nhalf) ? *reinterpret_cast<const f16x8*>(xsrc + (size_t)m * Kseg + (size_t)t0 * TK + i) :
                                         f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        ws.prime(wp, nt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < SV; j++) {
            const int i = lane * 8 + j * 512;
            if (i < nhalf) {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    *reinterpret_cast<f16x8*>(xs + (size_t)m * p.slice_halves + i) = xv[m][j];
                }
            }
        }
    }
    else {
#pragma unroll
        for (int m = 0; m < M; m++) {
            for (int i = lane * 8; i < nhalf; i += 64 * 8) {
                *reinterpret_cast<f16x8*>(xs + (size_t)m * p.slice_halves + i) =
                    *reinterpret_cast<const f16x8*>(xsrc + (size_t)m * Kseg + (size_t)t0 * TK + i);
            }
        }
        ws.prime(wp, nt);
    }
    // no barrier needed: the slice is read by the wave that wrote it (LDS ops of one wave are ordered)

    const int    n     = grp * 16 + c;
    f16x2        scale2 = {(f16)1.0f, (f16)1.0f};
    if constexpr (INT8) {
        const f16 sc = (seg ? p.scale_b : p.scale_a)[n];
        scale2       = f16x2{sc, sc};
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    ws.run(wp, nt, a_frag_ptr<INT8, M>(xs, p.slice_halves, lane), scale2, acc);

    // cross-wave reduction in a fixed order (deterministic)
    float* red = reinterpret_cast<float*>(smem + (size_t)nw * M * p.slice_halves * 2);  // [nw][M][16]
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            red[(wid * M + m) * 16 + c] = acc_row(acc, m);
        }
    }
    __syncthreads();
    if (threadIdx.x < M * 16) {
        const int m  = threadIdx.x >> 4;
        const int cc = threadIdx.x & 15;
        const int nn = grp * 16 + cc;
        float     sa = 0.f, sb = 0.f;
        for (int w = 0; w < nw; w++) {
            const float v = red[(w * M + m) * 16 + cc];
            if (p.wave_seg[w]) {
                sb += v;
            }
            else {
                sa += v;
            }
        }
        const size_t oidx = (size_t)m * p.N + nn;
        if constexpr (EPI == EPI_PLAIN) {
            if constexpr (INT8) {
                float v = sa;
                if (p.bias) {
                    v += (float)p.bias[nn];
                }
                if (p.act == 1) {
                    v = gelu_f32(v);
                }
                p.out[oidx] = (f16)v;
            }
            else {
                f16 h = (f16)sa;
                if (p.act == 1) {
                    h = gelu_f16(p.bias ? (f16)(h + p.bias[nn]) : h);
                }
                else if (p.bias) {
                    h = h + p.bias[nn];
                }
                p.out[oidx] = h;
            }
        }
        else {
            const f16 attn = (f16)sa, ffn = p.ffn_in ? p.ffn_in[oidx] : (f16)sb;
            const f16 xin  = (f16)((float)p.x_in[oidx] / (float)p.tp);
            const f16 b    = p.bias[nn];
            f16       r;
            if (p.inplace_variant) {
                r = (f16)((float)xin + (float)ffn + (float)attn + (float)b);
            }
            else {
                r = ((ffn + attn) + b) + xin;
            }
            p.out[oidx] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LM head: logits_f32[m, n] = x[m, :] . W[n, :]  with W the replicated fp16 [V, H] tensor, read in place
// (models/gptneox/GptNeoX.cc:866-912).  One wave per 4 vocabulary rows, lanes along k.
// ---------------------------------------------------------------------------------------------------------------
template<int M>
__global__ __launch_bounds__(256) void k_lm_head(const f16* __restrict__ x, const f16* __restrict__ W,
                                                 float* __restrict__ logits, int n_rows, int K, int ldc,
                                                 const f16* __restrict__ gamma, const f16* __restrict__ beta, float eps,
                                                 const int* d_stop)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (d_stop && *d_stop) {
        return;  // every row has finished: a token of a multi-token graph behind the request's last one
    }
    f16*   xs  = reinterpret_cast<f16*>(smem);  // [M][K]
    float* red = reinterpret_cast<float*>(smem + (size_t)M * K * 2);
    lm_head_stage_x<M>(x, K, gamma, beta, eps, xs, red);
    lm_head_rows<M>(W, logits, n_rows, K, ldc, xs, [](int, int, float) {});
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int M>
static void launch_ln_gemv_m(const LnGemvParams& p, hipStream_t s)
{
    const size_t smem = (size_t)M * (p.K + XPAD) * 2 + 64;
    const int    grid = p.blocks0 + p.blocks1;
    hipLaunchKernelGGL((k_ln_gemv<INT8, M>), dim3(grid), dim3(256), smem, s, p);
}

void launch_ln_gemv(const LnGemvParams& p, bool int8, int M, hipStream_t s)
{
    FTCF_CHECK_ARG(M >= 1 && M <= 4, "ln_gemv supports 1..4 rows");
    FTCF_CHECK_ARG(p.K % 64 == 0 && p.K % 8 == 0, "K must be a multiple of 64");
    if (int8) {
        switch (M) {
            case 1: launch_ln_gemv_m<true, 1>(p, s); break;
            case 2: launch_ln_gemv_m<true, 2>(p, s); break;
            case 3: launch_ln_gemv_m<true, 3>(p, s); break;
            default: launch_ln_gemv_m<true, 4>(p, s); break;
        }
    }
    else {
        switch (M) {
            case 1: launch_ln_gemv_m<false, 1>(p, s); break;
            case 2: launch_ln_gemv_m<false, 2>(p, s); break;
            case 3: launch_ln_gemv_m<false, 3>(p, s); break;
            default: launch_ln_gemv_m<false, 4>(p, s); break;
        }
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

template<bool INT8, int M>
static void launch_ln_gemv_group_m(const LnGemvParams& p, int wpg, hipStream_t s)
{
    const size_t smem = (size_t)M * (p.K + XPAD) * 2 + (2 * wpg + wpg * M * 16) * 4 + 64;
    hipLaunchKernelGGL((k_ln_gemv_group<INT8, M>), dim3(p.NT0 + p.NT1), dim3(wpg * 64), smem, s, p);
}

void launch_ln_gemv_group(const LnGemvParams& p, bool int8, int M, int wpg, hipStream_t s)
{
    FTCF_CHECK_ARG(M >= 1 && M <= 4, "ln_gemv supports 1..4 rows");
    FTCF_CHECK_ARG(p.K % 64 == 0, "K must be a multiple of 64");
    FTCF_CHECK_ARG(wpg >= 1 && wpg <= 4, "1..4 waves per column group");
    if (int8) {
        switch (M) {
            case 1: launch_ln_gemv_group_m<true, 1>(p, wpg, s); break;
            case 2: launch_ln_gemv_group_m<true, 2>(p, wpg, s); break;
            case 3: launch_ln_gemv_group_m<true, 3>(p, wpg, s); break;
            default: launch_ln_gemv_group_m<true, 4>(p, wpg, s); break;
        }
    }
    else {
        switch (M) {
            case 1: launch_ln_gemv_group_m<false, 1>(p, wpg, s); break;
            case 2: launch_ln_gemv_group_m<false, 2>(p, wpg, s); break;
            case 3: launch_ln_gemv_group_m<false, 3>(p, wpg, s); break;
            default: launch_ln_gemv_group_m<false, 4>(p, wpg, s); break;
        }
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// Splits the tiles of up to two segments over `nw` waves proportionally (every non-empty segment gets >= 1 wave).
void plan_splitk(SplitKParams& p, bool int8, int M, int max_waves)
{
    const int TK    = int8 ? TILE_K_I8 : TILE_K_F16;
    const int total = p.KT_a + p.KT_b;
    int       nw    = max_waves;
    // keep >= 8 tiles per wave when possible
    while (nw > 1 && total / nw < GEMV_U) {
        nw--;
    }
    int wa = p.KT_b == 0 ? nw : (int)((double)p.KT_a / total * nw + 0.5);
    if (p.KT_b > 0) {
        if (wa < 1) {
            wa = 1;
        }
        if (wa > nw - 1) {
            wa = nw - 1;
        }
        if (nw == 1) {  // two segments need two waves
            nw = 2;
            wa = 1;
        }
    }
    const int wb  = nw - wa;
    int       idx = 0, maxnt = 0;
    for (int s = 0; s < 2; s++) {
        const int KT = s ? p.KT_b : p.KT_a;
        const int W  = s ? wb : wa;
        for (int w = 0; w < W; w++) {
            const int t0 = (int)((long)KT * w / W), t1 = (int)((long)KT * (w + 1) / W);
            p.wave_seg[idx] = s;
            p.wave_t0[idx]  = t0;
            p.wave_nt[idx]  = t1 - t0;
            maxnt           = std::max(maxnt, t1 - t0);
            idx++;
        }
    }
    p.nwaves       = idx;
    p.slice_halves = maxnt * TK + XPAD;  // + pad: rows of a slice in different banks
    (void)M;
}

template<bool INT8, int M>
static void launch_splitk_m(const SplitKParams& p, int epi, hipStream_t s)
{
    const size_t smem = (size_t)p.nwaves * M * p.slice_halves * 2 + (size_t)p.nwaves * M * 16 * 4;
    const dim3   grid(p.N / 16), block(p.nwaves * 64);
    if (epi == EPI_PLAIN) {
        hipLaunchKernelGGL((k_gemv_splitk<INT8, M, EPI_PLAIN>), grid, block, smem, s, p);
    }
    else {
        hipLaunchKernelGGL((k_gemv_splitk<INT8, M, EPI_RESIDUAL>), grid, block, smem, s, p);
    }
}

void launch_gemv_splitk(const SplitKParams& p, bool int8, int M, int epi, hipStream_t s)
{
    FTCF_CHECK_ARG(M >= 1 && M <= 4, "gemv supports 1..4 rows");
    FTCF_CHECK_ARG(p.N % 16 == 0, "N must be a multiple of 16");
    if (int8) {
        switch (M) {
            case 1: launch_splitk_m<true, 1>(p, epi, s); break;
            case 2: launch_splitk_m<true, 2>(p, epi, s); break;
            case 3: launch_splitk_m<true, 3>(p, epi, s); break;
            default: launch_splitk_m<true, 4>(p, epi, s); break;
        }
    }
    else {
        switch (M) {
            case 1: launch_splitk_m<false, 1>(p, epi, s); break;
            case 2: launch_splitk_m<false, 2>(p, epi, s); break;
            case 3: launch_splitk_m<false, 3>(p, epi, s); break;
            default: launch_splitk_m<false, 4>(p, epi, s); break;
        }
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

void launch_lm_head(const f16* x, const f16* W, float* logits, int M, int n_rows, int K, int ldc, hipStream_t s,
                    const f16* gamma, const f16* beta, float eps, const int* d_stop)
{
    FTCF_CHECK_ARG(K % 8 == 0, "K must be a multiple of 8");
    const size_t smem = (size_t)M * (K + XPAD) * 2 + 64;
    int          grid = (n_rows + 15) / 16;
    if (grid > 2048) {
        grid = 2048;
    }
#define FTCF_LM(MM)                                                                                                    \
    hipLaunchKernelGGL((k_lm_head<MM>), dim3(grid), dim3(256), smem, s, x, W, logits, n_rows, K, ldc, gamma, beta, eps, d_stop)
    switch (M) {
        case 1: FTCF_LM(1); break;
        case 2: FTCF_LM(2); break;
        case 3: FTCF_LM(3); break;
        case 4: FTCF_LM(4); break;
        default: throw Error(-1, "lm_head GEMV supports 1..4 rows");
    }
#undef FTCF_LM
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// K3, balanced: out = residual(half(x_a * W_a), half(x_b * W_b))  (out-proj U FFN2 -> add_residual_kernels.cu:116-152)
// The one-workgroup-per-column-group form has N/16 = 320 workgroups for 256 CUs: the 64 CUs that get two of them set
// the kernel time (+35 %).  Here each group's concatenated K range is cut into Q chunks -> NT*Q = 1280 two-wave
// workgroups = exactly 5 per CU.  Partials travel as granules to the chunk-0 workgroup (only it waits; bounded spin).
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned long long u64g;
__device__ __forceinline__ void st_gran(u64g* g, unsigned tag, float v)
{
    __hip_atomic_store(g, ((u64g)tag << 32) | (u64g)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template<bool INT8, int M>
__global__ __launch_bounds__(128) void k_gemv_chunked(const ChunkParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TK   = TileK<INT8>::value;
    const int     lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int     grp = blockIdx.x / p.Q, q = blockIdx.x % p.Q;
    const int     c = lane & 15, g = lane >> 4;
    const int     KTt = p.KT_a + p.KT_b;
    // flat tile range of this wave
    const int lo = min(q * p.T, KTt), hi = min(lo + p.T, KTt);
    const int half = (hi - lo + 1) / 2;
    const int w_lo = min(lo + wid * half, hi), w_hi = min(w_lo + half, hi);
    // pieces: A = [w_lo, w_hi) ^ [0, KT_a) ; B = [w_lo, w_hi) ^ [KT_a, KTt) shifted by KT_a
    const int a0 = min(w_lo, p.KT_a), a1 = min(w_hi, p.KT_a);
    const int b0 = max(w_lo, p.KT_a) - p.KT_a, b1 = max(w_hi, p.KT_a) - p.KT_a;
    const int ntA = a1 - a0, ntB = b1 - b0;
    const int slice = (p.T / 2 + 2) * TK + XPAD;  // halves per wave per row (+ pad: rows in different banks)
    f16*      xs    = reinterpret_cast<f16*>(smem) + (size_t)wid * M * slice;
    const u32x4* wpA = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.W_a)
                                                      + (((size_t)grp * p.KT_a + a0) * 64 + lane) * 16);
    const u32x4* wpB = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.W_b)
                                                      + (((size_t)grp * p.KT_b + b0) * 64 + lane) * 16);
    WaveStream<INT8, M> ws;
    // x slices (L2 hits) first, then the first weight batch
    {
        const int nhA = ntA * TK, nhB = ntB * TK;
        constexpr int SV = (M == 1) ? 8 : ((M == 2) ? 4 : 2);
        if (nhA + nhB <= 512 * SV) {
            f16x8 xv[M][SV];
#pragma unroll
            for (int j = 0; j < SV; j++) {
                const int i = lane * 8 + j * 512;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (i < nhA) {
                        v = *reinterpret_cast<const f16x8*>(p.x_a + (size_t)m * p.KT_a * TK + (size_t)a0 * TK + i);
                    }
                    else if (i < nhA + nhB) {
                        v = *reinterpret_cast<const f16x8*>(p.x_b + (size_t)m * p.KT_b * TK + (size_t)b0 * TK + (i - nhA));
                    }
                    xv[m][j] = v;
                }
            }
            if (ntA > 0) {
                ws.prime(wpA, ntA);
            }
            else {
                ws.prime(wpB, ntB);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SV; j++) {
                const int i = lane * 8 + j * 512;
                if (i < nhA + nhB) {
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        *reinterpret_cast<f16x8*>(xs + (size_t)m * slice + i) = xv[m][j];
                    }
                }
            }
        }
        else {
#pragma unroll
            for (int m = 0; m < M; m++) {
                for (int i = lane * 8; i < nhA + nhB; i += 512) {
                    *reinterpret_cast<f16x8*>(xs + (size_t)m * slice + i) =
                        (i < nhA) ? *reinterpret_cast<const f16x8*>(p.x_a + (size_t)m * p.KT_a * TK + (size_t)a0 * TK + i) :
                                    *reinterpret_cast<const f16x8*>(p.x_b + (size_t)m * p.KT_b * TK + (size_t)b0 * TK + (i - nhA));
                }
            }
            if (ntA > 0) {
                ws.prime(wpA, ntA);
            }
            else {
                ws.prime(wpB, ntB);
            }
        }
    }
    const int n = grp * 16 + c;
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
    if (ntA > 0) {
        f16x2 sc2 = {(f16)1.0f, (f16)1.0f};
        if constexpr (INT8) {
            const f16 sc = p.scale_a[n];
            sc2          = f16x2{sc, sc};
        }
        ws.run(wpA, ntA, a_frag_ptr<INT8, M>(xs, slice, lane), sc2, accA);
        if (ntB > 0) {
            ws.prime(wpB, ntB);
        }
    }
    if (ntB > 0) {
        f16x2 sc2 = {(f16)1.0f, (f16)1.0f};
        if constexpr (INT8) {
            const f16 sc = p.scale_b[n];
            sc2          = f16x2{sc, sc};
        }
        ws.run(wpB, ntB, a_frag_ptr<INT8, M>(xs + (size_t)ntA * TK, slice, lane), sc2, accB);
    }
    // two-wave reduce in LDS, fixed order
    float* part = reinterpret_cast<float*>(smem + (size_t)2 * M * slice * 2);  // [2 waves][2][M][16]
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            part[((wid * 2 + 0) * M + m) * 16 + c] = acc_row(accA, m);
            part[((wid * 2 + 1) * M + m) * 16 + c] = acc_row(accB, m);
        }
    }
    __syncthreads();
    const int      step = p.d_step ? *p.d_step : p.step;
    const unsigned tag  = (unsigned)(step * 1024 + p.salt) + 1u;
    u64g*          gg   = p.gran + (size_t)grp * p.Q * 2 * M * 16;
    if (threadIdx.x < 2 * M * 16) {  // index = (s*M + m)*16 + c
        const float v = part[threadIdx.x] + part[2 * M * 16 + threadIdx.x];
        if (p.Q > 1) {
            st_gran(&gg[(size_t)q * 2 * M * 16 + threadIdx.x], tag, v);
        }
        else {
            part[threadIdx.x] = v;
        }
    }
    if (q != 0) {
        return;
    }
    __syncthreads();
    if (threadIdx.x < M * 16) {
        const int m = threadIdx.x >> 4, cc = threadIdx.x & 15;
        float     sa = 0.f, sb = 0.f;
        if (p.Q > 1) {
            u64g va[8], vb[8];
            int  spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k < p.Q) {
                        va[k] = __hip_atomic_load(&gg[(size_t)k * 2 * M * 16 + (0 * M + m) * 16 + cc], __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
                        vb[k] = __hip_atomic_load(&gg[(size_t)k * 2 * M * 16 + (1 * M + m) * 16 + cc], __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (k < p.Q) {
                        ok &= ((unsigned)(va[k] >> 32) == tag) && ((unsigned)(vb[k] >> 32) == tag);
                    }
                }
                if (ok || ++spins > (1 << 22)) {
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {  // chunk order: deterministic
                if (k < p.Q) {
                    sa += __uint_as_float((unsigned)va[k]);
                    sb += __uint_as_float((unsigned)vb[k]);
                }
            }
        }
        else {
            sa = part[(0 * M + m) * 16 + cc];
            sb = part[(1 * M + m) * 16 + cc];
        }
        const int    nn   = grp * 16 + cc;
        const size_t oidx = (size_t)m * p.N + nn;
        const f16    attn = (f16)sa, ffn = (f16)sb;
        const f16    xin  = (f16)((float)p.x_in[oidx] / (float)p.tp);
        const f16    b    = p.bias[nn];
        f16          r;
        if (p.inplace_variant) {
            r = (f16)((float)xin + (float)ffn + (float)attn + (float)b);
        }
        else {
            r = ((ffn + attn) + b) + xin;
        }
        p.out[oidx] = r;
    }
}

int chunk_pick_q(int NT, int KT_total)
{
    int    best = 1;
    double best_cost = 1e30;
    for (int Q = 1; Q <= 8; Q *= 2) {
        const int T = (KT_total + Q - 1) / Q;
        if (Q > 1 && T / 2 < 16) {
            break;  // keep >= 16 tiles per wave
        }
        const double per_cu = (double)NT * Q / 256.0;
        const double cost   = std::ceil(per_cu) / per_cu + 0.01 * Q;  // imbalance factor, slight preference for small Q
        if (cost < best_cost) {
            best_cost = cost;
            best      = Q;
        }
    }
    return best;
}

size_t chunk_workspace_bytes(int N, int M, int Q)
{
    return (size_t)(N / 16) * Q * 2 * M * 16 * sizeof(unsigned long long);
}

template<bool INT8, int M>
static void launch_chunked_m(const ChunkParams& p, hipStream_t s)
{
    const int    TK    = INT8 ? TILE_K_I8 : TILE_K_F16;
    const int    slice = (p.T / 2 + 2) * TK + XPAD;
    const size_t smem  = (size_t)2 * M * slice * 2 + (size_t)2 * 2 * M * 16 * 4;
    hipLaunchKernelGGL((k_gemv_chunked<INT8, M>), dim3((p.N / 16) * p.Q), dim3(128), smem, s, p);
}

void launch_gemv_chunked(const ChunkParams& p, bool int8, int M, hipStream_t s)
{
    FTCF_CHECK_ARG(M >= 1 && M <= 4, "gemv supports 1..4 rows");
    FTCF_CHECK_ARG(p.N % 16 == 0 && p.Q >= 1 && p.Q <= 8 && (p.Q == 1 || p.gran != nullptr), "bad chunked GEMV config");
    if (int8) {
        switch (M) {
            case 1: launch_chunked_m<true, 1>(p, s); break;
            case 2: launch_chunked_m<true, 2>(p, s); break;
            case 3: launch_chunked_m<true, 3>(p, s); break;
            default: launch_chunked_m<true, 4>(p, s); break;
        }
    }
    else {
        switch (M) {
            case 1: launch_chunked_m<false, 1>(p, s); break;
            case 2: launch_chunked_m<false, 2>(p, s); break;
            case 3: launch_chunked_m<false, 3>(p, s); break;
            default: launch_chunked_m<false, 4>(p, s); break;
        }
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
