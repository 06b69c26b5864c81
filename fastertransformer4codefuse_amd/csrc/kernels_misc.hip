// Small HBM-bound helper kernels of the GPT-NeoX path (gfx950): LayerNorm, residual, embedding lookups, re-tiling.
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

// invokeGeneralLayerNorm, fp16 half2-path numerics (kernels/layernorm_kernels.cu:157-286, dispatch :1652-1735):
// stats in fp32 with var = E[x^2] - mean^2, normalise/scale/shift in half arithmetic.  One block per row.
__global__ __launch_bounds__(256) void k_layernorm_f16(const f16* __restrict__ x, const f16* __restrict__ gamma,
                                                       const f16* __restrict__ beta, f16* __restrict__ out, int n,
                                                       float eps)
{
    __shared__ float red[8];
    const f16*       xr = x + (size_t)blockIdx.x * n;
    f16*             o  = out + (size_t)blockIdx.x * n;
    float            s[2] = {0.f, 0.f};
    const bool       vec = (n % 8) == 0;
    if (vec) {
        for (int i = threadIdx.x * 8; i < n; i += 256 * 8) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xr + i);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float f = (float)v[j];
                s[0] += f;
                s[1] += f * f;
            }
        }
    }
    else {
        for (int i = threadIdx.x; i < n; i += 256) {
            const float f = (float)xr[i];
            s[0] += f;
            s[1] += f * f;
        }
    }
    block_sum<2>(s, red);
    const float mean = s[0] / (float)n;
    const float rstd = rsqrtf(s[1] / (float)n - mean * mean + eps);
    const f16   mh = (f16)mean, rh = (f16)rstd;
    if (vec) {
        for (int i = threadIdx.x * 8; i < n; i += 256 * 8) {
            const f16x8 xv = *reinterpret_cast<const f16x8*>(xr + i);
            const f16x8 gv = *reinterpret_cast<const f16x8*>(gamma + i);
            f16x8       ov;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                ov[j] = ((xv[j] - mh) * rh) * gv[j];
            }
            if (beta) {
                const f16x8 bv = *reinterpret_cast<const f16x8*>(beta + i);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    ov[j] = ov[j] + bv[j];
                }
            }
            *reinterpret_cast<f16x8*>(o + i) = ov;
        }
        return;
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        f16 v = ((xr[i] - mh) * rh) * gamma[i];
        if (beta) {
            v = v + beta[i];
        }
        o[i] = v;
    }
}

// Batched decode layers (general path): the two LayerNorms of a parallel-residual layer read the same x, so one pass
// computes the statistics once and writes both normalised copies; with RESID the layer-closing
// invokeAddBiasAttentionFfnResidual of the PREVIOUS layer (add_residual_kernels.cu:116-178) runs first in the same
// pass (x is updated in place).  Same arithmetic as k_layernorm_f16 / k_add_bias_attn_ffn_residual (same per-thread
// element sets and summation order); the row stays in registers.  n % 8 == 0, n <= 8192.
constexpr int DLN_NV = 4;
template<bool RESID>
__global__ __launch_bounds__(256) void k_residual_dual_ln(f16* __restrict__ x, const f16* __restrict__ ffn,
                                                          const f16* __restrict__ attn, const f16* __restrict__ bias,
                                                          int tp, int inplace_variant, const f16* __restrict__ g1,
                                                          const f16* __restrict__ b1, const f16* __restrict__ g2,
                                                          const f16* __restrict__ b2, f16* __restrict__ out1,
                                                          f16* __restrict__ out2, int n, float eps, int bias_mul)
{
    // (bias_mul: tensor-parallel layers whose attn / ffn arrive already all-reduced add the rank's bias / TP that many times)
    __shared__ float red[8];
    const size_t     row = (size_t)blockIdx.x * n;
    f16x8            v[DLN_NV];
    float            s[2] = {0.f, 0.f};
#pragma unroll
    for (int it = 0; it < DLN_NV; it++) {
        const int i = threadIdx.x * 8 + it * 2048;
        if (i < n) {
            f16x8 xv = *reinterpret_cast<const f16x8*>(x + row + i);
            if constexpr (RESID) {
                const f16x8 fv = *reinterpret_cast<const f16x8*>(ffn + row + i);
                const f16x8 av = *reinterpret_cast<const f16x8*>(attn + row + i);
                f16x8       bv = *reinterpret_cast<const f16x8*>(bias + i);
                if (bias_mul != 1) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        bv[j] = (f16)((float)bv[j] * (float)bias_mul);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const f16 xin = (f16)((float)xv[j] / (float)tp);
                    if (inplace_variant) {
                        xv[j] = (f16)((float)xin + (float)fv[j] + (float)av[j] + (float)bv[j]);
                    }
                    else {
                        xv[j] = ((fv[j] + av[j]) + bv[j]) + xin;
                    }
                }
                *reinterpret_cast<f16x8*>(x + row + i) = xv;
            }
            v[it] = xv;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float f = (float)xv[j];
                s[0] += f;
                s[1] += f * f;
            }
        }
    }
    if (g1 == nullptr) {  // last layer: residual only
        return;
    }
    block_sum<2>(s, red);
    const float mean = s[0] / (float)n;
    const float rstd = rsqrtf(s[1] / (float)n - mean * mean + eps);
    const f16   mh = (f16)mean, rh = (f16)rstd;
#pragma unroll
    for (int it = 0; it < DLN_NV; it++) {
        const int i = threadIdx.x * 8 + it * 2048;
        if (i < n) {
            const f16x8 ga = *reinterpret_cast<const f16x8*>(g1 + i), gb = *reinterpret_cast<const f16x8*>(g2 + i);
            const f16x8 ba = *reinterpret_cast<const f16x8*>(b1 + i), bb = *reinterpret_cast<const f16x8*>(b2 + i);
            f16x8       oa, ob;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const f16 c = (v[it][j] - mh) * rh;
                f16       a = c * ga[j];
                a           = a + ba[j];
                f16 b       = c * gb[j];
                b           = b + bb[j];
                oa[j]       = a;
                ob[j]       = b;
            }
            *reinterpret_cast<f16x8*>(out1 + row + i) = oa;
            *reinterpret_cast<f16x8*>(out2 + row + i) = ob;
        }
    }
}

bool residual_dual_ln_supported(int n)
{
    return n % 8 == 0 && n <= 2048 * DLN_NV;
}

void launch_residual_dual_ln(f16* x, const f16* ffn, const f16* attn, const f16* bias, int tp, int inplace_variant,
                             const f16* g1, const f16* b1, const f16* g2, const f16* b2, f16* out1, f16* out2, int m,
                             int n, float eps, hipStream_t s, int bias_mul)
{
    FTCF_CHECK_ARG(residual_dual_ln_supported(n), "fused residual + LayerNorm needs n % 8 == 0 and n <= 8192");
    if (m == 0) {
        return;
    }
    if (ffn) {
        hipLaunchKernelGGL((k_residual_dual_ln<true>), dim3(m), dim3(256), 0, s, x, ffn, attn, bias, tp, inplace_variant,
                           g1, b1, g2, b2, out1, out2, n, eps, bias_mul);
    }
    else {
        hipLaunchKernelGGL((k_residual_dual_ln<false>), dim3(m), dim3(256), 0, s, x, ffn, attn, bias, tp,
                           inplace_variant, g1, b1, g2, b2, out1, out2, n, eps, bias_mul);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// fp32 engine variant: two-pass generalLayerNorm (kernels/layernorm_kernels.cu:1565-1650)
__global__ __launch_bounds__(256) void k_layernorm_f32(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out, int n,
                                                       float eps)
{
    __shared__ float red[8];
    const float*     xr = x + (size_t)blockIdx.x * n;
    float*           o  = out + (size_t)blockIdx.x * n;
    float            s[1] = {0.f};
    for (int i = threadIdx.x; i < n; i += 256) {
        s[0] += xr[i];
    }
    block_sum<1>(s, red);
    const float mean = s[0] / (float)n;
    float       v[1] = {0.f};
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = xr[i] - mean;
        v[0] += d * d;
    }
    block_sum<1>(v, red);
    const float rstd = rsqrtf(v[0] / (float)n + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        o[i] = ((xr[i] - mean) * rstd) * gamma[i] + (beta ? beta[i] : 0.f);
    }
}

void launch_layernorm(const void* x, const void* gamma, const void* beta, void* out, int m, int n, float eps,
                      bool fp16, hipStream_t s)
{
    if (m == 0) {
        return;
    }
    if (fp16) {
        hipLaunchKernelGGL(k_layernorm_f16, dim3(m), dim3(256), 0, s, (const f16*)x, (const f16*)gamma,
                           (const f16*)beta, (f16*)out, n, eps);
    }
    else {
        hipLaunchKernelGGL(k_layernorm_f32, dim3(m), dim3(256), 0, s, (const float*)x, (const float*)gamma,
                           (const float*)beta, (float*)out, n, eps);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// invokeAddBiasAttentionFfnResidual (kernels/add_residual_kernels.cu:116-178)
template<typename T>
__global__ void k_add_bias_attn_ffn_residual(T* out, const T* ffn, const T* attn, const T* in, const T* bias, size_t total,
                                             int n, int tp, int inplace_variant)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % n);
        const T   xin = (T)((float)in[i] / (float)tp);
        T         r;
        if (inplace_variant) {
            r = (T)((float)xin + (float)ffn[i] + (float)attn[i] + (float)bias[col]);
        }
        else {
            r = ((ffn[i] + attn[i]) + bias[col]) + xin;
        }
        out[i] = r;
    }
}

void launch_add_bias_attn_ffn_residual(void* out, const void* ffn, const void* attn, const void* in, const void* bias,
                                       int m, int n, int tp, int inplace_variant, bool fp16, hipStream_t s)
{
    const size_t total = (size_t)m * n;
    if (total == 0) {
        return;
    }
    const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    if (fp16) {
        hipLaunchKernelGGL(k_add_bias_attn_ffn_residual<f16>, dim3(grid), dim3(256), 0, s, (f16*)out, (const f16*)ffn,
                           (const f16*)attn, (const f16*)in, (const f16*)bias, total, n, tp, inplace_variant);
    }
    else {
        hipLaunchKernelGGL(k_add_bias_attn_ffn_residual<float>, dim3(grid), dim3(256), 0, s, (float*)out,
                           (const float*)ffn, (const float*)attn, (const float*)in, (const float*)bias, total, n, tp,
                           inplace_variant);
    }
    FTCF_HIP_CHECK(hipGetLastError());
}

// sequential-residual layers (use_gptj_residual == 0): out = a + b + bias in fp32, rounded once -- the element-wise part of
// invokeGeneralAddBiasResidualPreLayerNorm (layernorm_kernels.cu:158-260, the LayerNorm follows as its own launch) and
// invokeAddBiasResidual (add_residual_kernels.cu:22-60) as GptNeoXDecoder.cc:313-331,362-367 use them
__global__ void k_add_bias_residual(f16* out, const f16* a, const f16* b, const f16* bias, size_t total, int n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float bv = bias ? (float)bias[i % n] : 0.f;
        out[i]         = (f16)((bv + (float)a[i]) + (float)b[i]);
    }
}

void launch_add_bias_residual(f16* out, const f16* a, const f16* b, const f16* bias, int m, int n, hipStream_t s)
{
    const size_t total = (size_t)m * n;
    if (total == 0) {
        return;
    }
    const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_add_bias_residual, dim3(grid), dim3(256), 0, s, out, a, b, bias, total, n);
    FTCF_HIP_CHECK(hipGetLastError());
}

// kernels/gpt_kernels.cu:31-104 start_id_embedding_position_lookups_kernel (no position table for NeoX)
__global__ void k_prompt_embedding(f16* out, int* output_ids, const f16* table, const int* ids, int B, int S, int H)
{
    const int row = blockIdx.x;  // b*S + s
    const int b = row / S, s = row % S;
    const int id = ids[row];
    if (threadIdx.x == 0 && output_ids) {
        output_ids[(size_t)s * B + b] = id;
    }
    const f16* src = table + (size_t)id * H;
    f16*       dst = out + (size_t)row * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}

void launch_prompt_embedding(f16* out, int* output_ids, const f16* table, const int* ids, int B, int S, int H,
                             hipStream_t s)
{
    FTCF_CHECK_ARG(H % 8 == 0, "hidden size must be a multiple of 8");
    hipLaunchKernelGGL(k_prompt_embedding, dim3(B * S), dim3(256), 0, s, out, output_ids, table, ids, B, S, H);
    FTCF_HIP_CHECK(hipGetLastError());
}

// kernels/decoding_kernels.cu:145-191 embeddingLookupPosEncoding: token of the previous step -> [B,H]
__global__ void k_step_embedding(f16* out, const f16* table, const int* output_ids, const int* d_step, int B, int H)
{
    const int  b    = blockIdx.x;
    const int  step = *d_step;
    const int  id   = output_ids[(size_t)(step - 1) * B + b];
    const f16* src  = table + (size_t)id * H;
    f16*       dst  = out + (size_t)b * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}

void launch_step_embedding(f16* out, const f16* table, const int* output_ids, const int* d_step, int B, int H,
                           hipStream_t s)
{
    hipLaunchKernelGGL(k_step_embedding, dim3(B), dim3(256), 0, s, out, table, output_ids, d_step, B, H);
    FTCF_HIP_CHECK(hipGetLastError());
}

__global__ void k_embedding(f16* out, const f16* table, const int* ids, int H)
{
    const int  id  = ids[blockIdx.x];
    const f16* src = table + (size_t)id * H;
    f16*       dst = out + (size_t)blockIdx.x * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}
void launch_embedding(f16* out, const f16* table, const int* ids, int n_ids, int H, hipStream_t s)
{
    hipLaunchKernelGGL(k_embedding, dim3(n_ids), dim3(256), 0, s, out, table, ids, H);
    FTCF_HIP_CHECK(hipGetLastError());
}

// kernels/gpt_kernels.cu:438-470 lookupHiddenStateOfLastToken
__global__ void k_gather_last_token(f16* out, const f16* hidden, const int* input_lengths, int S, int H, int tile)
{
    const int  b   = blockIdx.x / tile;
    const f16* src = hidden + ((size_t)b * S + (input_lengths[b] - 1)) * H;
    f16*       dst = out + (size_t)blockIdx.x * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}
void launch_gather_last_token(f16* out, const f16* hidden, const int* input_lengths, int B, int S, int H,
                              hipStream_t s, int tile)
{
    hipLaunchKernelGGL(k_gather_last_token, dim3(B * tile), dim3(256), 0, s, out, hidden, input_lengths, S, H, tile);
    FTCF_HIP_CHECK(hipGetLastError());
}

__global__ void k_tile_prompt_ids(int* output_ids, const int* ids, int B, int K, int S)
{
    const size_t total = (size_t)S * B * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int    bb = (int)(i % ((size_t)B * K));
        const size_t s2 = i / ((size_t)B * K);
        output_ids[i]   = ids[(size_t)(bb / K) * S + s2];
    }
}
void launch_tile_prompt_ids(int* output_ids, const int* ids, int B, int K, int S, hipStream_t s)
{
    const size_t total = (size_t)S * B * K;
    hipLaunchKernelGGL(k_tile_prompt_ids, dim3((int)std::min<size_t>((total + 255) / 256, 1024)), dim3(256), 0, s,
                       output_ids, ids, B, K, S);
    FTCF_HIP_CHECK(hipGetLastError());
}

// fp16 [K,N] row major -> engine tile layout (ftcf_common.h).  One block per tile; reads 16 columns x 32 rows.
__global__ __launch_bounds__(64) void k_fp16_to_tiled(const f16* __restrict__ w, size_t K, size_t N, f16* __restrict__ out)
{
    const size_t KT = K / TILE_K_F16;
    const size_t nt = blockIdx.x / KT, kt = blockIdx.x % KT;
    const int    lane = threadIdx.x;
    const int    c = lane & 15, g = lane >> 4;
    f16x8        v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v[j] = w[(kt * TILE_K_F16 + g * 8 + j) * N + nt * 16 + c];
    }
    *reinterpret_cast<f16x8*>(out + ((nt * KT + kt) * 64 + lane) * 8) = v;
}

void launch_fp16_rowmajor_to_tiled(const f16* w, size_t K, size_t N, f16* out, hipStream_t s)
{
    FTCF_CHECK_ARG(K % TILE_K_F16 == 0 && N % TILE_N == 0, "fp16 tiling needs K % 32 == 0 and N % 16 == 0");
    const size_t tiles = (K / TILE_K_F16) * (N / TILE_N);
    FTCF_CHECK_ARG(tiles < (size_t)1 << 31, "matrix too large");
    hipLaunchKernelGGL(k_fp16_to_tiled, dim3((unsigned)tiles), dim3(64), 0, s, w, K, N, out);
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
