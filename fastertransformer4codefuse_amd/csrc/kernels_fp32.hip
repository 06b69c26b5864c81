// fp32 instantiation of the GPT-NeoX path: FTGptNeoX<float> (th_op/gptneox/GptNeoXOp.cc:56-70), i.e. GptNeoX<float>,
// GptNeoXContextDecoder<float>, GptNeoXDecoder<float> with cuBLAS sgemm, the float masked-multihead-attention kernel and the
// float LayerNorm / residual kernels.
//
// This is the VALIDATION instantiation (the reference's own unit tests and the tiny BASELINE config run it; nobody serves a
// 13B model in fp32): everything is computed and accumulated in fp32 at the reference's rounding points -- none -- and
// the kernels are written for clarity, not for the roofline.  The GEMM uses v_mfma_f32_16x16x4_f32 (true fp32 inputs, no
// xf32 / tf32 truncation); attention and the LM head are plain VALU kernels.  No weight re-layout: the caller's row-major
// [K, N] matrices and the [V, H] head are read in place.  int8_mode is a half-only feature of the reference
// (CutlassFpAIntBGemmRunner<half, uint8_t>) and is refused for fp32.
#include "attn_device.hip.h"
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------
// embedding / bookkeeping (kernels/gpt_kernels.cu:31-104, decoding_kernels.cu:145-191, gpt_kernels.cu:438-470)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k32_prompt_embedding(float* out, int* output_ids, const float* table, const int* ids, int B, int S, int H)
{
    const int row = blockIdx.x;  // b*S + s
    const int b = row / S, s = row % S;
    const int id = ids[row];
    if (threadIdx.x == 0 && output_ids) {
        output_ids[(size_t)s * B + b] = id;
    }
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        out[(size_t)row * H + i] = table[(size_t)id * H + i];
    }
}
void launch32_prompt_embedding(float* out, int* output_ids, const float* table, const int* ids, int B, int S, int H,
                               hipStream_t s)
{
    hipLaunchKernelGGL(k32_prompt_embedding, dim3(B * S), dim3(256), 0, s, out, output_ids, table, ids, B, S, H);
    FTCF_HIP_CHECK(hipGetLastError());
}

__global__ void k32_step_prologue(float* out, const float* table, const int* output_ids, const int* d_step, float* rot_table,
                                  const int* pad_count, int B, int H, int rot)
{
    const int b    = blockIdx.x;
    const int step = *d_step;
    if ((int)threadIdx.x < rot / 2) {
        const int pos = (step - 1) - (pad_count ? pad_count[b] : 0);
        float     cs, sn;
        rotary_coef(threadIdx.x, rot, pos, cs, sn);
        rot_table[((size_t)b * (rot / 2) + threadIdx.x) * 2]     = cs;
        rot_table[((size_t)b * (rot / 2) + threadIdx.x) * 2 + 1] = sn;
    }
    const int id = output_ids[(size_t)(step - 1) * B + b];
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        out[(size_t)b * H + i] = table[(size_t)id * H + i];
    }
}
void launch32_step_prologue(float* out, const float* table, const int* output_ids, const int* d_step, float* rot_table,
                            const int* pad_count, int B, int H, int rot, hipStream_t s)
{
    FTCF_CHECK_ARG(rot / 2 <= 256, "rotary_embedding_dim must be <= 512");
    hipLaunchKernelGGL(k32_step_prologue, dim3(B), dim3(256), 0, s, out, table, output_ids, d_step, rot_table, pad_count, B,
                       H, rot);
    FTCF_HIP_CHECK(hipGetLastError());
}

__global__ void k32_gather_last_token(float* out, const float* hidden, const int* input_lengths, int S, int H, int tile)
{
    const int    b   = blockIdx.x / tile;
    const float* src = hidden + ((size_t)b * S + (input_lengths[b] - 1)) * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        out[(size_t)blockIdx.x * H + i] = src[i];
    }
}
void launch32_gather_last_token(float* out, const float* hidden, const int* input_lengths, int B, int S, int H,
                                hipStream_t s, int tile)
{
    hipLaunchKernelGGL(k32_gather_last_token, dim3(B * tile), dim3(256), 0, s, out, hidden, input_lengths, S, H, tile);
    FTCF_HIP_CHECK(hipGetLastError());
}

// invokeGeneralAddBiasResidualPreLayerNorm's element-wise part / invokeAddBiasResidual for the sequential-residual layers
__global__ void k32_add_bias_residual(float* out, const float* a, const float* b, const float* bias, size_t total, int n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        out[i] = ((bias ? bias[i % n] : 0.f) + a[i]) + b[i];
    }
}
void launch32_add_bias_residual(float* out, const float* a, const float* b, const float* bias, int m, int n, hipStream_t s)
{
    const size_t total = (size_t)m * n;
    if (total == 0) {
        return;
    }
    hipLaunchKernelGGL(k32_add_bias_residual, dim3((int)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, out,
                       a, b, bias, total, n);
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM: C[m, n] = A[m, k] x W[k, n] (+ bias) (+ gelu), all fp32 (cublasMMWrapper::Gemm with CUDA_R_32F,
// utils/cublasMMWrapper.cc:94-386; bias / gelu as invokeAddBiasGeluV2 computes them in float, activation_kernels.cu:401-426)
// Workgroup: 4 waves, tile 16 rows x 64 columns; A chunk 16 x 64 through LDS, W fragments straight from global (lanes of a
// k row read 16 consecutive columns).  v_mfma_f32_16x16x4_f32: A lane (row = l & 15, k = l >> 4), B lane (col = l & 15,
// k = l >> 4), C rows (l >> 4) * 4 + q, col l & 15.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k32_gemm(const float* __restrict__ A, const float* __restrict__ W,
                                                const float* __restrict__ bias, int act, float* __restrict__ C, int m, int n,
                                                int k)
{
    constexpr int KC = 64;
    __shared__ float As[16][KC + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64 + wid * 16;
    const int col = n0 + c;
    const int colc = col < n ? col : n - 1;
    f32x4_t   acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < k; k0 += KC) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * KC; i += 256) {
            const int r = i / KC, kk = i % KC;
            int       row = m0 + r;
            row           = row < m ? row : m - 1;
            As[r][kk]     = (k0 + kk < k) ? A[(size_t)row * k + k0 + kk] : 0.f;
        }
        __syncthreads();
        const int kend = (k - k0 < KC) ? k - k0 : KC;
        for (int kk = 0; kk < kend; kk += 4) {
            const int   kr = k0 + kk + g;
            const float b  = (kr < k) ? W[(size_t)kr * n + colc] : 0.f;
            const float a  = As[c][kk + g];
            acc            = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
    }
    if (col < n) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = m0 + g * 4 + q;
            if (row < m) {
                float v = acc[q] + bv;
                if (act == 1) {
                    v = gelu_f32(v);
                }
                C[(size_t)row * n + col] = v;
            }
        }
    }
}
void launch32_gemm(const float* A, const float* W, const float* bias, int act, float* C, int m, int n, int k, hipStream_t s)
{
    if (m == 0) {
        return;
    }
    dim3 grid((n + 63) / 64, (m + 15) / 16);
    hipLaunchKernelGGL(k32_gemm, grid, dim3(256), 0, s, A, W, bias, act, C, m, n, k);
    FTCF_HIP_CHECK(hipGetLastError());
}

// logits[m, n] = A[m, k] x W[n, k]^T (GptNeoX.cc:866-912, fp32): a wave per vocabulary row, all m rows of A per pass
__global__ __launch_bounds__(256) void k32_lm_head(const float* __restrict__ A, const float* __restrict__ W,
                                                   float* __restrict__ C, int m, int n, int k, int ldc)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int v = blockIdx.x * 4 + wid;
    if (v >= n) {
        return;
    }
    const float* w = W + (size_t)v * k;
    for (int r0 = 0; r0 < m; r0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < k; i += 64) {
            const float wv = w[i];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (r0 + r < m) {
                    acc[r] = fmaf(A[(size_t)(r0 + r) * k + i], wv, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float sum = wave_sum(acc[r]);
            if (lane == 0 && r0 + r < m) {
                C[(size_t)(r0 + r) * ldc + v] = sum;
            }
        }
    }
}
void launch32_lm_head(const float* A, const float* W_nk, float* logits, int m, int n, int k, int ldc, hipStream_t s)
{
    hipLaunchKernelGGL(k32_lm_head, dim3((n + 3) / 4), dim3(256), 0, s, A, W_nk, logits, m, n, k, ldc);
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// prefill attention (GptContextAttentionLayer.cc:142-345 in float: add_fusedQKV_bias_transpose with NeoX rotary, the
// attention mask of gpt_kernels.cu:359-402, softmax(qk_scale * QK^T + mask) with the +1e-6 of
// unfused_attention_kernels.cu:322, PV)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k32_qkv_bias_rotary_cache(float* qkv, const float* __restrict__ qkv_bias,
                                                                 const int* __restrict__ input_lengths, float* k_cache,
                                                                 float* v_cache, int S, int nh, int dh, int rot, int s_max,
                                                                 int crm)
{
    __shared__ float s_cs[256], s_sn[256];
    const int  row = blockIdx.x;
    const int  b = row / S, s = row % S;
    const int  hl = nh * dh, half = rot / 2;
    const bool valid = s < input_lengths[b];
    if ((int)threadIdx.x < half) {
        rotary_coef(threadIdx.x, rot, s, s_cs[threadIdx.x], s_sn[threadIdx.x]);  // position = index in the (right padded) row
    }
    __syncthreads();
    float* base = qkv + (size_t)row * 3 * hl;
    for (int i = threadIdx.x; i < hl; i += blockDim.x) {
        const int h = i / dh, d = i % dh;
        if (d >= half && d < rot) {
            continue;  // written by the thread that owns d - rot/2
        }
        const size_t cidx = (((size_t)b * crm * nh + h) * s_max + s) * dh + d;  // cache row b * crm (beam search: beam 0)
        float        q = 0.f, k = 0.f, v = 0.f;
        if (valid) {
            q = base[i] + qkv_bias[i];
            k = base[hl + i] + qkv_bias[hl + i];
            v = base[2 * hl + i] + qkv_bias[2 * hl + i];
        }
        if (d < half) {
            float q2 = 0.f, k2 = 0.f, v2 = 0.f;
            if (valid) {
                q2 = base[i + half] + qkv_bias[i + half];
                k2 = base[hl + i + half] + qkv_bias[hl + i + half];
                v2 = base[2 * hl + i + half] + qkv_bias[2 * hl + i + half];
                const float cs = s_cs[d], sn = s_sn[d];
                const float qa = q, ka = k;
                q  = cs * qa - sn * q2;
                q2 = cs * q2 + sn * qa;
                k  = cs * ka - sn * k2;
                k2 = cs * k2 + sn * ka;
            }
            base[i + half]       = q2;
            k_cache[cidx + half] = k2;
            v_cache[cidx + half] = v2;
        }
        base[i]       = q;
        k_cache[cidx] = k;
        v_cache[cidx] = v;
    }
}

// one workgroup (128 threads) per (query, head, row); dynamic LDS: scores of the keys 0..query
__global__ __launch_bounds__(128) void k32_context_attention(const float* __restrict__ qkv,
                                                             const int* __restrict__ input_lengths,
                                                             const float* __restrict__ k_cache,
                                                             const float* __restrict__ v_cache, int S, int nh, int dh,
                                                             int s_max, float* __restrict__ ctx, float qk_scale, int crm)
{
    extern __shared__ float sm32[];
    float* sc = sm32;            // [S]
    float* sq = sm32 + S;        // [dh]
    float* red = sq + dh;        // [4]
    const int qi = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = input_lengths[b];
    if (qi >= len) {
        return;  // padded query rows are discarded by the reference
    }
    const int hl = nh * dh;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < dh; d += blockDim.x) {
        sq[d] = qkv[((size_t)b * S + qi) * 3 * hl + h * dh + d];
    }
    __syncthreads();
    const float* kc = k_cache + ((size_t)b * crm * nh + h) * s_max * dh;
    const float* vc = v_cache + ((size_t)b * crm * nh + h) * s_max * dh;
    float        mx = -INFINITY;
    for (int t = threadIdx.x; t <= qi; t += blockDim.x) {
        float a = 0.f;
        for (int d = 0; d < dh; d++) {
            a = fmaf(sq[d], kc[(size_t)t * dh + d], a);
        }
        a     = a * qk_scale;
        sc[t] = a;
        mx    = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    if (lane == 0) {
        red[wid] = mx;
    }
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    __syncthreads();
    float sum = 0.f;
    for (int t = threadIdx.x; t <= qi; t += blockDim.x) {
        const float e = __expf(sc[t] - mx);
        sc[t]         = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) {
        red[wid] = sum;
    }
    __syncthreads();
    const float inv = 1.f / ((red[0] + red[1]) + 1e-6f);
    for (int d = threadIdx.x; d < dh; d += blockDim.x) {
        float o = 0.f;
        for (int t = 0; t <= qi; t++) {
            o = fmaf(sc[t] * inv, vc[(size_t)t * dh + d], o);
        }
        ctx[((size_t)b * S + qi) * hl + h * dh + d] = o;
    }
}

void launch32_context_attention(float* qkv, const float* qkv_bias, const int* input_lengths, float* k_cache, float* v_cache,
                                int B, int S, int nh, int dh, int rot, int s_max, float* ctx, hipStream_t s, int crm)
{
    FTCF_CHECK_ARG(rot % 2 == 0 && rot <= dh && rot / 2 <= 256, "rotary_embedding_dim must be even and <= size_per_head");
    hipLaunchKernelGGL(k32_qkv_bias_rotary_cache, dim3(B * S), dim3(256), 0, s, qkv, qkv_bias, input_lengths, k_cache,
                       v_cache, S, nh, dh, rot, s_max, crm);
    const size_t smem = (size_t)(S + dh + 4) * sizeof(float);
    FTCF_CHECK_ARG(smem <= 64 * 1024, "fp32 prefill attention keeps a query's scores in LDS: max_input_len <= ~16000");
    hipLaunchKernelGGL(k32_context_attention, dim3(S, nh, B), dim3(128), smem, s, qkv, input_lengths, k_cache, v_cache, S, nh,
                       dh, s_max, ctx, 1.0f / sqrtf((float)dh), crm);
    FTCF_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// decode attention: masked_multihead_attention_kernel<float, Dh, ...> (decoder_masked_multihead_attention_template.hpp:
// 1099-1919) -- bias, NeoX rotary at position step - 1 - pad_count, append to the cache at tlength, scores over 0..tlength
// with the padding mask, softmax with +1e-6, PV; HAS_BEAMS reads through the cache indirection (:1290-1296,:1509-1512,
// :1733-1736).  One workgroup per (head, row).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k32_mmha(const Mmha32Params p)
{
    extern __shared__ float sm32[];
    const int dh = p.dh, hl = p.nh * dh;
    float*    sq = sm32;           // [dh]
    float*    sk = sq + dh;        // [dh]
    float*    sv = sk + dh;        // [dh]
    float*    sc = sv + dh;        // [s_max + 1]
    float*    red = sc + p.s_max + 1;  // [8]
    const int h = blockIdx.x, b = blockIdx.y;
    if (p.finished && p.finished[b]) {
        return;  // :1176
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tl   = p.seq_len[b];
    const int step = *p.d_step;
    const int pos  = (step - 1) - (p.pad_count ? p.pad_count[b] : 0);
    for (int d = threadIdx.x; d < dh; d += blockDim.x) {
        const size_t base = (size_t)b * 3 * hl + h * dh + d;
        sq[d] = p.qkv[base] + (p.qkv_bias ? p.qkv_bias[h * dh + d] : 0.f);
        sk[d] = p.qkv[base + hl] + (p.qkv_bias ? p.qkv_bias[hl + h * dh + d] : 0.f);
        sv[d] = p.qkv[base + 2 * hl] + (p.qkv_bias ? p.qkv_bias[2 * hl + h * dh + d] : 0.f);
    }
    __syncthreads();
    if ((int)threadIdx.x < p.rot / 2) {
        float cs, sn;
        rotary_coef(threadIdx.x, p.rot, pos, cs, sn);
        const int   j = threadIdx.x, j2 = j + p.rot / 2;
        const float qa = sq[j], qb = sq[j2], ka = sk[j], kb = sk[j2];
        sq[j]  = cs * qa - sn * qb;
        sq[j2] = cs * qb + sn * qa;
        sk[j]  = cs * ka - sn * kb;
        sk[j2] = cs * kb + sn * ka;
    }
    __syncthreads();
    float* kc = p.k_cache + ((size_t)b * p.nh + h) * p.s_max * dh;
    float* vc = p.v_cache + ((size_t)b * p.nh + h) * p.s_max * dh;
    for (int d = threadIdx.x; d < dh; d += blockDim.x) {
        kc[(size_t)tl * dh + d] = sk[d];
        vc[(size_t)tl * dh + d] = sv[d];
    }
    const int* indir = nullptr;
    int        b_first = 0;
    if (p.cache_indir) {
        indir   = p.cache_indir + (size_t)((step - p.max_input_len) & 1) * p.indir_plane + (size_t)b * p.s_max;
        b_first = b / p.beam_width * p.beam_width;
    }
    const ptrdiff_t row_kv = (ptrdiff_t)p.nh * p.s_max * dh;
    const uint8_t*  mask   = p.masked_tokens ? p.masked_tokens + (size_t)b * p.s_max : nullptr;
    const float     inv_sqrt_dh = 1.f / sqrtf((float)dh);
    float           mx = -FLT_MAX;
    for (int t = threadIdx.x; t <= tl; t += blockDim.x) {
        float a = 0.f;
        if (t == tl) {
            for (int d = 0; d < dh; d++) {
                a = fmaf(sq[d], sk[d], a);
            }
        }
        else {
            const float* kt = kc + (indir ? (ptrdiff_t)(b_first + indir[t] - b) * row_kv : 0) + (size_t)t * dh;
            for (int d = 0; d < dh; d++) {
                a = fmaf(sq[d], kt[d], a);
            }
        }
        a = a * inv_sqrt_dh;
        const bool m = (t < tl) && mask && mask[t];
        sc[t]        = m ? -INFINITY : a;  // masked keys get probability 0 (:1570,:1610-1622)
        if (!m) {
            mx = fmaxf(mx, a);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) {
        red[wid] = mx;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int t = threadIdx.x; t <= tl; t += blockDim.x) {
        const float e = (sc[t] == -INFINITY) ? 0.f : __expf(sc[t] - mx);
        sc[t]         = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) {
        red[wid] = sum;
    }
    __syncthreads();
    const float inv = 1.f / (((red[0] + red[1]) + (red[2] + red[3])) + 1.e-6f);  // :1632
    for (int d = threadIdx.x; d < dh; d += blockDim.x) {
        float o = 0.f;
        for (int t = 0; t < tl; t++) {
            const float* vt = vc + (indir ? (ptrdiff_t)(b_first + indir[t] - b) * row_kv : 0) + (size_t)t * dh;
            o               = fmaf(sc[t] * inv, vt[d], o);
        }
        o = fmaf(sc[tl] * inv, sv[d], o);
        p.ctx[(size_t)b * hl + h * dh + d] = o;
    }
}

void launch32_mmha(const Mmha32Params& p, hipStream_t s)
{
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh && p.rot / 2 <= 256, "rotary_embedding_dim must be even and <= size_per_head");
    const size_t smem = (size_t)(3 * p.dh + p.s_max + 1 + 8) * sizeof(float);
    FTCF_CHECK_ARG(smem <= 64 * 1024, "fp32 decode attention keeps a row's scores in LDS: max_input_len + output_len <= ~16000");
    hipLaunchKernelGGL(k32_mmha, dim3(p.nh, p.B), dim3(256), smem, s, p);
    FTCF_HIP_CHECK(hipGetLastError());
}

}  // namespace ftcf
