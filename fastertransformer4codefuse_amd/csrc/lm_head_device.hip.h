// Device side of the LM head GEMV: logits_f32[m, n] = h[m, :] . W[n, :] with W the replicated fp16 [V, H] tensor read in
// place (models/gptneox/GptNeoX.cc:866-912), the final LayerNorm (GptNeoX.cc:854-863) fused in front of it.  Shared by
// k_lm_head (kernels_gemv.hip) and k_lm_head_greedy (kernels_sampling.hip: the same logits, and the all-greedy dynamic decode
// of the token from inside the launch).  256 threads per workgroup; one wave per 4 vocabulary rows, lanes along k.
#pragma once
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

// stages the M rows of x in LDS (xs: [M][K] halves), normalised when gamma != NULL (invokeGeneralLayerNorm, half2-path
// numerics): every workgroup normalises the tiny [M, K] hidden state itself instead of paying a kernel boundary for it.
// red: 2 * (blockDim / 64) floats.  Ends with a workgroup barrier.
template<int M>
__device__ __forceinline__ void lm_head_stage_x(const f16* __restrict__ x, const int K, const f16* __restrict__ gamma,
                                                const f16* __restrict__ beta, const float eps, f16* xs, float* red)
{
    if (gamma) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const f16* xr   = x + (size_t)m * K;
            float      s[2] = {0.f, 0.f};
            for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(xr + i);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float f = (float)v[j];
                    s[0] += f;
                    s[1] += f * f;
                }
            }
            block_sum<2>(s, red);
            const float mean = s[0] / (float)K;
            const float rstd = rsqrtf(s[1] / (float)K - mean * mean + eps);
            const f16   mh = (f16)mean, rh = (f16)rstd;
            for (int i = threadIdx.x * 8; i < K; i += 256 * 8) {
                const f16x8 v  = *reinterpret_cast<const f16x8*>(xr + i);
                const f16x8 gg = *reinterpret_cast<const f16x8*>(gamma + i);
                const f16x8 bb = *reinterpret_cast<const f16x8*>(beta + i);
                f16x8       o;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    o[j] = (((v[j] - mh) * rh) * gg[j]) + bb[j];
                }
                *reinterpret_cast<f16x8*>(xs + (size_t)m * K + i) = o;
            }
        }
    }
    else {
        for (int i = threadIdx.x * 8; i < M * K; i += 256 * 8) {
            *reinterpret_cast<f16x8*>(xs + i) = *reinterpret_cast<const f16x8*>(x + i);
        }
    }
    __syncthreads();
}

// the vocabulary rows of this workgroup's waves; on_logit(m, row, value) runs in every lane of the wave that produced the
// logit (the value is wave-uniform), after it has been stored
template<int M, typename F>
__device__ __forceinline__ void lm_head_rows(const f16* __restrict__ W, float* __restrict__ logits, const int n_rows,
                                             const int K, const int ldc, const f16* xs, F&& on_logit)
{
    const int     lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int     nwaves = gridDim.x * 4;
    constexpr int R = 4;
    for (int r0 = (blockIdx.x * 4 + wid) * R; r0 < n_rows; r0 += nwaves * R) {
        float acc[R][M];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                acc[r][m] = 0.f;
            }
        }
        for (int k = lane * 8; k < K; k += 64 * 8) {
            u32x4 w[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int row = (r0 + r < n_rows) ? (r0 + r) : (n_rows - 1);
                w[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + (size_t)row * K + k));
            }
#pragma unroll
            for (int m = 0; m < M; m++) {
                const f16x8 xv = *reinterpret_cast<const f16x8*>(xs + (size_t)m * K + k);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const f16x8 b = __builtin_bit_cast(f16x8, w[r]);
                    float       a = acc[r][m];
                    a             = dot2(f16x2{b[0], b[1]}, f16x2{xv[0], xv[1]}, a);
                    a             = dot2(f16x2{b[2], b[3]}, f16x2{xv[2], xv[3]}, a);
                    a             = dot2(f16x2{b[4], b[5]}, f16x2{xv[4], xv[5]}, a);
                    a             = dot2(f16x2{b[6], b[7]}, f16x2{xv[6], xv[7]}, a);
                    acc[r][m]     = a;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                const float v = wave_sum(acc[r][m]);
                if (r0 + r < n_rows) {
                    if (lane == 0) {
                        logits[(size_t)m * ldc + r0 + r] = v;
                    }
                    on_logit(m, r0 + r, v);
                }
            }
        }
    }
}

}  // namespace ftcf
