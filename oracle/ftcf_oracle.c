/*
 * ftcf_oracle.c -- CPU restatement (ORACLE) of the reference's GPT-NeoX / CodeFuse decode path.
 * TEST INFRASTRUCTURE ONLY (see ftcf_oracle.h).  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/fastertransformer unless noted).
 */
#include "ftcf_oracle.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* ------------------------------------------------------------------------------------------------ */
/* binary16 emulation                                                                                */
/* ------------------------------------------------------------------------------------------------ */
uint16_t orc_float_to_half_bits(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t abs  = x & 0x7fffffffu;
    if (abs >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((abs > 0x7f800000u) ? 0x200u : 0));
    }
    if (abs >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (abs < 0x38800000u) { /* subnormal half or zero: < 2^-14 */
        if (abs < 0x33000000u) { /* < 2^-25 -> 0 */
            return (uint16_t)sign;
        }
        int      e    = (int)(abs >> 23);           /* biased float exponent */
        uint32_t mant = (abs & 0x7fffffu) | 0x800000u;
        int      shift = 126 - e;                   /* 14..24 : value = mant * 2^(e-150); target unit 2^-24 */
        /* result = round(mant * 2^(e-150) / 2^-24) = round(mant >> (126 - e)) */
        uint32_t r    = mant >> shift;
        uint32_t rem  = mant & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) {
            r++;
        }
        return (uint16_t)(sign | r);
    }
    /* normal */
    uint32_t e    = (abs >> 23) - 112u; /* re-bias 127 -> 15 */
    uint32_t mant = abs & 0x7fffffu;
    uint32_t r    = (e << 10) | (mant >> 13);
    uint32_t rem  = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) {
        r++; /* may carry into exponent: correct */
    }
    return (uint16_t)(sign | r);
}

float orc_half_bits_to_float(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e    = (h >> 10) & 0x1fu;
    uint32_t m    = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        }
        else {
            int s = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                s++;
            }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
        }
    }
    else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    }
    else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

int orc_num_threads(void)
{
    return omp_get_max_threads();
}

/* the software conversion above is the definition; where the host has F16C (every AVX2 machine) the hardware
 * conversion -- round to nearest even, subnormals, overflow to inf: the same function, checked value by value in
 * tests/test_oracle_quant.py -- replaces it, which makes the m = 1 GEMVs of the CPU baseline ~10x faster */
float orc_round_half_soft(float x)
{
    return orc_half_bits_to_float(orc_float_to_half_bits(x));
}
#if defined(__F16C__)
#include <immintrin.h>
static inline float orc_round_half_inl(float x)
{
    return _cvtsh_ss(_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
}
float orc_round_half(float x)
{
    return orc_round_half_inl(x);
}
#define orc_round_half(x) orc_round_half_inl(x)
#else
float orc_round_half(float x)
{
    return orc_round_half_soft(x);
}
#endif

#define RT(x) (fp16 ? orc_round_half(x) : (x))
#define HALF_FLT_MAX 65504.f

/* counter based RNG shared bit-for-bit with the HIP engine (csrc/sampling.hip: ftcf_uniform).
 * The reference uses curand XORWOW (sampling_topk_kernels.cu:32-65,283); its stream is not reproducible
 * without curand, so only the distribution is mirrored -- greedy (top_k=1) is unaffected. */
static uint64_t splitmix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
float orc_uniform(uint64_t seed, uint64_t row, uint64_t draw)
{
    uint64_t z = splitmix64(seed ^ splitmix64(row * 0x632be59bd9b4e019ULL + draw));
    uint32_t r = (uint32_t)(z >> 40); /* 24 bits */
    return (float)(r + 1u) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------------------------------------ */
/* quantiser: kernels/cutlass_kernels/cutlass_preprocessors.cc:576-673                              */
/*   scale[n] = max_k|W[k,n]| / 128 (stored rounded to the weight dtype), q = clamp(round(w / scale_f32)) */
/*   q is returned in the plain row-major [K,N] ("unprocessed") layout.                             */
/* ------------------------------------------------------------------------------------------------ */
void orc_symmetric_quantize_int8(const float* w, int K, int N, int weight_is_half, int8_t* q, float* scale)
{
    float* col_max = (float*)calloc((size_t)N, sizeof(float));
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < N; j++) {
            float a = fabsf(w[(size_t)i * N + j]);
            if (a > col_max[j]) {
                col_max[j] = a;
            }
        }
    }
    for (int j = 0; j < N; j++) {
        col_max[j] *= (1.f / 128.f);
        scale[j] = weight_is_half ? orc_round_half(col_max[j]) : col_max[j];
    }
    for (int i = 0; i < K; i++) {
        for (int j = 0; j < N; j++) {
            float s = roundf(w[(size_t)i * N + j] / col_max[j]); /* divide by the UNROUNDED fp32 scale (:626-643) */
            /* std::max(-128.f, std::min(127.f, s)) with NaN (0/0) -> 127 */
            float c = (s < 127.f) ? s : 127.f;
            if (!(c > -128.f)) {
                c = (c != c) ? 127.f : -128.f;
            }
            q[(size_t)i * N + j] = (int8_t)c;
        }
    }
    free(col_max);
}

/* ------------------------------------------------------------------------------------------------ */
/* activation helpers                                                                                */
/* ------------------------------------------------------------------------------------------------ */
/* cutlass_extensions/.../ft_fused_activations.h:72-90 (GELU_taylor<float>): 0.5 z (1 + tanh(k0 z (1 + k1 z^2))) */
static float gelu_f32(float z)
{
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    return 0.5f * z * (1.0f + tanhf(k0 * z * (1.0f + k1 * z * z)));
}

/* kernels/activation_kernels.cu:54-85 GeluActivation<half2> and :401-426 addBiasGeluV2 */
void orc_add_bias_gelu(float* x, const float* bias, int m, int n, int fp16)
{
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < n; j++) {
            float v = x[(size_t)i * n + j];
            if (fp16) {
                if (bias) {
                    v = orc_round_half(v + bias[j]); /* hadd2 */
                }
                float p3  = orc_round_half(v * orc_round_half(v * v)); /* __hmul2(val, __hmul2(val, val)) */
                float cdf = 0.5f * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * p3)));
                v         = orc_round_half(v * orc_round_half(cdf));
            }
            else {
                if (bias) {
                    v = v + bias[j];
                }
                float cdf = 0.5f * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
                v         = v * cdf;
            }
            x[(size_t)i * n + j] = v;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* GEMM                                                                                              */
/*  fp path: utils/cublasMMWrapper.cc:94-386 (fp16 in, fp32 accumulate, T out)                       */
/*  int8 path: kernels/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:45-197,               */
/*     cutlass_extensions/.../interleaved_numeric_conversion.h:50-83 (u8 -> fp16 exactly),           */
/*     gemm/warp/mma_tensorop_dequantizer.h (B_f16 = half(q) * scale_f16, rounded to half BEFORE MMA),*/
/*     epilogue_helpers.h:52-62 (bias + activation in fp32, then convert to half).                   */
/* ------------------------------------------------------------------------------------------------ */
void orc_gemm(const float* A, int m, int k, int n, const float* W, const int8_t* q, const float* scale,
              const float* bias, int act, float* C, int fp16, int out_fp32)
{
    /* decode GEMVs over int8 weights (m <= 4: what bench.py's cpu_baseline times, 13.6 GB per token at 13B): parallel over
     * (CHUNK OF 64 K-ROWS) x (QUARTER OF THE COLUMNS) work items -- 320 for a 5120-row matrix -- each streaming kilobytes of
     * contiguous bytes per row (the column-block walk below touches 128 bytes per 15 KB stride: a page per access) in tiles of
     * 512 columns whose accumulators and dequantised row stay in the L1; per element the 64 products of a chunk are added in k
     * order to a double, then the chunks in chunk order -- a fixed order whatever the thread count.  5.8 -> 1.2 s per 13B token
     * on the 128 cores of the GPU box */
    if (q && m >= 1 && m <= 4) {
        enum { KC = 64, JT = 512, NQ = 4 };
        const int nch = (k + KC - 1) / KC, nq = ((n + NQ - 1) / NQ + JT - 1) / JT * JT; /* columns per quarter: whole tiles */
        double*   part = (double*)malloc(sizeof(double) * (size_t)nch * (size_t)m * (size_t)n);
#pragma omp parallel
        {
            float* brow = (float*)malloc(sizeof(float) * JT);
#pragma omp for schedule(static)
            for (int item = 0; item < nch * NQ; item++) {
                const int c = item / NQ, q0 = (item % NQ) * nq, q1 = q0 + nq < n ? q0 + nq : n;
                double*   pc = part + (size_t)c * m * n;
                const int k1 = (c + 1) * KC < k ? (c + 1) * KC : k;
                for (int j0 = q0; j0 < q1; j0 += JT) {
                    const int nb = q1 - j0 < JT ? q1 - j0 : JT;
                    for (int i = 0; i < m; i++) {
                        for (int j = 0; j < nb; j++) {
                            pc[(size_t)i * n + j0 + j] = 0.0;
                        }
                    }
                    for (int kk = c * KC; kk < k1; kk++) {
                        const int8_t* qr = q + (size_t)kk * n + j0;
                        const float*  sc = scale + j0;
                        int           j  = 0;
#if defined(__AVX2__) && defined(__F16C__)
                        for (; j + 8 <= nb; j += 8) {
                            const __m256 qf = _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_loadl_epi64((const __m128i*)(qr + j))));
                            const __m256 pr = _mm256_mul_ps(qf, _mm256_loadu_ps(sc + j));
                            _mm256_storeu_ps(brow + j, _mm256_cvtph_ps(_mm256_cvtps_ph(pr, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)));
                        }
#endif
                        for (; j < nb; j++) {
                            brow[j] = orc_round_half((float)qr[j] * sc[j]);
                        }
                        for (int i = 0; i < m; i++) {
                            const double a  = (double)A[(size_t)i * k + kk];
                            double*      ar = pc + (size_t)i * n + j0;
                            j               = 0;
#if defined(__AVX2__) && defined(__FMA__)
                            const __m256d av = _mm256_set1_pd(a);
                            for (; j + 4 <= nb; j += 4) {
                                _mm256_storeu_pd(ar + j, _mm256_fmadd_pd(av, _mm256_cvtps_pd(_mm_loadu_ps(brow + j)), _mm256_loadu_pd(ar + j)));
                            }
#endif
                            for (; j < nb; j++) {
                                ar[j] += a * (double)brow[j];
                            }
                        }
                    }
                }
            }
            free(brow);
#pragma omp for schedule(static)
            for (int j = 0; j < n; j++) {
                for (int i = 0; i < m; i++) {
                    double acc = 0.0;
                    for (int c = 0; c < nch; c++) {
                        acc += part[((size_t)c * m + i) * n + j];
                    }
                    float v = (float)acc;
                    if (bias) { /* fused epilogue in fp32 */
                        v += bias[j];
                    }
                    if (act == 1) {
                        v = gelu_f32(v);
                    }
                    if (!out_fp32) {
                        v = RT(v);
                    }
                    C[(size_t)i * n + j] = v;
                }
            }
        }
        free(part);
        return;
    }
    /* otherwise: parallel over column blocks; per element the k loop runs in order with a double accumulator */
    const int NB = 128;
#pragma omp parallel
    {
        double* acc  = (double*)malloc(sizeof(double) * (size_t)NB * (size_t)(m > 0 ? m : 1));
        float*  brow = (float*)malloc(sizeof(float) * (size_t)NB);
#pragma omp for schedule(static)
        for (int j0 = 0; j0 < n; j0 += NB) {
            const int nb = (n - j0 < NB) ? (n - j0) : NB;
            for (int i = 0; i < m * NB; i++) {
                acc[i] = 0.0;
            }
            for (int kk = 0; kk < k; kk++) {
                if (q) {
                    const int8_t* qr = q + (size_t)kk * n + j0;
                    int           j  = 0;
#if defined(__AVX2__) && defined(__F16C__)
                    /* eight columns at a time, the same operations per element: fp32 product, round to half (nearest even), back */
                    for (; j + 8 <= nb; j += 8) {
                        const __m256 qf = _mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_loadl_epi64((const __m128i*)(qr + j))));
                        const __m256 pr = _mm256_mul_ps(qf, _mm256_loadu_ps(scale + j0 + j));
                        _mm256_storeu_ps(brow + j, _mm256_cvtph_ps(_mm256_cvtps_ph(pr, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)));
                    }
#endif
                    for (; j < nb; j++) {
                        brow[j] = orc_round_half((float)qr[j] * scale[j0 + j]);
                    }
                }
                else {
                    const float* wr = W + (size_t)kk * n + j0;
                    for (int j = 0; j < nb; j++) {
                        brow[j] = wr[j];
                    }
                }
                for (int i = 0; i < m; i++) {
                    const double a = (double)A[(size_t)i * k + kk];
                    double*      ar = acc + (size_t)i * NB;
                    int          j  = 0;
#if defined(__AVX2__) && defined(__FMA__)
                    /* (the scalar statement below contracts to a fused multiply-add under gcc's default -ffp-contract=fast: the
                     * same single rounding here) */
                    const __m256d av = _mm256_set1_pd(a);
                    for (; j + 4 <= nb; j += 4) {
                        const __m256d bv = _mm256_cvtps_pd(_mm_loadu_ps(brow + j));
                        _mm256_storeu_pd(ar + j, _mm256_fmadd_pd(av, bv, _mm256_loadu_pd(ar + j)));
                    }
#endif
                    for (; j < nb; j++) {
                        ar[j] += a * (double)brow[j];
                    }
                }
            }
            for (int i = 0; i < m; i++) {
                for (int j = 0; j < nb; j++) {
                    float v = (float)acc[(size_t)i * NB + j];
                    if (q) { /* fused epilogue in fp32 */
                        if (bias) {
                            v += bias[j0 + j];
                        }
                        if (act == 1) {
                            v = gelu_f32(v);
                        }
                    }
                    if (!out_fp32) {
                        v = RT(v);
                    }
                    C[(size_t)i * n + j0 + j] = v;
                }
            }
        }
        free(acc);
        free(brow);
    }
    if (!q && (bias || act)) {
        /* non-fused path: FfnLayer.cc:264-308 -> invokeAddBiasGeluV2 ; plain bias add is done by callers */
        if (act == 1) {
            orc_add_bias_gelu(C, bias, m, n, fp16);
        }
        else if (bias) {
            for (int i = 0; i < m; i++) {
                for (int j = 0; j < n; j++) {
                    C[(size_t)i * n + j] = RT(C[(size_t)i * n + j] + bias[j]);
                }
            }
        }
    }
}

/* models/gptneox/GptNeoX.cc:866-912: logits_f32 = normed_hidden (T) x lm_head^T (T), fp32 out */
void orc_lm_head(const float* A, int m, int k, int n, const float* Wt, float* C)
{
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; j++) {
        const float* wr = Wt + (size_t)j * k;
        for (int i = 0; i < m; i++) {
            double       acc = 0.0;
            const float* ar  = A + (size_t)i * k;
            for (int kk = 0; kk < k; kk++) {
                acc += (double)ar[kk] * (double)wr[kk];
            }
            C[(size_t)i * n + j] = (float)acc;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* LayerNorm: kernels/layernorm_kernels.cu:157-286 (half2 path: var = E[x^2]-mean^2, normalise in half2)  */
/*            and :1565-1650 (fp32 two-pass), dispatch :1652-1735 ; eps 1e-5 (GptNeoX.h:43)         */
/* ------------------------------------------------------------------------------------------------ */
void orc_layernorm(const float* x, const float* gamma, const float* beta, int m, int n, float eps, float* out, int fp16)
{
    for (int i = 0; i < m; i++) {
        const float* xr = x + (size_t)i * n;
        float*       o  = out + (size_t)i * n;
        if (fp16 && (n % 2 == 0)) {
            double s = 0, s2 = 0;
            for (int j = 0; j < n; j++) {
                s += xr[j];
                s2 += (double)xr[j] * xr[j];
            }
            float mean   = (float)(s / n);
            float rstd   = 1.0f / sqrtf((float)(s2 / n) - mean * mean + eps);
            float mean_h = orc_round_half(mean);
            float rstd_h = orc_round_half(rstd);
            for (int j = 0; j < n; j++) {
                /* hmul2(hsub2(x, mean), var, gamma) = ((x - mean) * var) * gamma, each op rounded to half */
                float v = orc_round_half(xr[j] - mean_h);
                v       = orc_round_half(v * rstd_h);
                v       = orc_round_half(v * gamma[j]);
                if (beta) {
                    v = orc_round_half(v + beta[j]);
                }
                o[j] = v;
            }
        }
        else {
            double s = 0;
            for (int j = 0; j < n; j++) {
                s += xr[j];
            }
            float  mean = (float)(s / n);
            double v2   = 0;
            for (int j = 0; j < n; j++) {
                float d = xr[j] - mean;
                v2 += (double)d * d;
            }
            float rstd = 1.0f / sqrtf((float)(v2 / n) + eps);
            for (int j = 0; j < n; j++) {
                float b = beta ? beta[j] : 0.f;
                o[j]    = RT(((xr[j] - mean) * rstd) * gamma[j] + b);
            }
        }
    }
}

/* kernels/add_residual_kernels.cu:116-178: out = ffn + attn + bias + in/TP.
 * Two variants exist: the out-of-place one chains T additions left to right (:116-133); the in-place one
 * (block_output == block_input, :135-152) sums the four terms in fp32 and rounds once (cuda_type_utils.cuh:132). */
void orc_add_bias_attn_ffn_residual(float* out, const float* ffn, const float* attn, const float* in, const float* bias,
                                    int m, int n, int tp, int inplace_variant, int fp16)
{
    for (int i = 0; i < m; i++) {
        for (int j = 0; j < n; j++) {
            size_t idx = (size_t)i * n + j;
            float  x   = RT(in[idx] / (float)tp);
            float  r;
            if (inplace_variant) {
                r = RT(x + ffn[idx] + attn[idx] + bias[j]);
            }
            else {
                r = RT(ffn[idx] + attn[idx]);
                r = RT(r + bias[j]);
                r = RT(r + x);
            }
            out[idx] = r;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* rotary: kernels/decoder_masked_multihead_attention_utils.h:1325-1345 (NeoX pairing (j, j+rot/2))  */
/* ------------------------------------------------------------------------------------------------ */
static void rotary_neox(float* v, int rot, int pos, int fp16)
{
    int half = rot / 2;
    for (int j = 0; j < half; j++) {
        float inv_freq = (float)pos / powf(10000.0f, (float)(2 * j) / (float)rot);
        float c = cosf(inv_freq), s = sinf(inv_freq);
        float a = v[j], b = v[j + half];
        v[j]        = RT(c * a - s * b);
        v[j + half] = RT(c * b + s * a);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* decode-step attention: decoder_masked_multihead_attention_template.hpp:1099-1919,                 */
/* launcher decoder_masked_multihead_attention/decoder_masked_multihead_attention_128.cu:28-77       */
/* ------------------------------------------------------------------------------------------------ */
void orc_mmha_step(const float* qkv, const float* qkv_bias, float* k_cache, float* v_cache, const int* seq_len,
                   const int* pad_count, const uint8_t* masked_tokens, const uint8_t* finished, int B, int nh, int dh,
                   int rot, int s_max, int step, float* ctx, int fp16)
{
    orc_mmha_step_beam(qkv, qkv_bias, k_cache, v_cache, seq_len, pad_count, masked_tokens, finished, B, nh, dh, rot, s_max,
                       step, ctx, fp16, NULL, 1);
}

/* The same with beam search (HAS_BEAMS, decoder_masked_multihead_attention_template.hpp:1465-1475, 1730-1745): row b of the
 * B = batch * beam_width rows reads the cached key / value of time t from the row of beam cache_indir[b][t] of its batch
 * (cache_indir: [batch][beam][s_max]); the current token goes to the row's own cache. */
void orc_mmha_step_beam(const float* qkv, const float* qkv_bias, float* k_cache, float* v_cache, const int* seq_len,
                        const int* pad_count, const uint8_t* masked_tokens, const uint8_t* finished, int B, int nh, int dh,
                        int rot, int s_max, int step, float* ctx, int fp16, const int* cache_indir, int beam_width)
{
    const int   hl          = nh * dh;
    const float inv_sqrt_dh = 1.f / sqrtf((float)dh); /* DecoderSelfAttentionLayer.cc:118, q_scaling = 1 */
    const int   timestep    = step - 1;               /* :112 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++) {
        for (int h = 0; h < nh; h++) {
            if (finished && finished[b]) {
                continue; /* :1176 */
            }
            const int tl = seq_len[b]; /* tlength (:1204-1207) */
            float     q[256], k[256], v[256];
            for (int d = 0; d < dh; d++) {
                float bq = qkv_bias ? qkv_bias[h * dh + d] : 0.f;
                float bk = qkv_bias ? qkv_bias[hl + h * dh + d] : 0.f;
                float bv = qkv_bias ? qkv_bias[2 * hl + h * dh + d] : 0.f;
                q[d]     = RT(qkv[(size_t)b * 3 * hl + h * dh + d] + bq);
                k[d]     = RT(qkv[(size_t)b * 3 * hl + hl + h * dh + d] + bk);
                v[d]     = RT(qkv[(size_t)b * 3 * hl + 2 * hl + h * dh + d] + bv);
            }
            /* the reference's float kernel with Dh == 32 silently skips NeoX rotary
             * (Qk_vec_m_<float,32> scalar no-op overloads); use Dh in {64,128}. */
            if (rot > 0) {
                int pos = timestep - (pad_count ? pad_count[b] : 0); /* :1303,:1343-1344 */
                rotary_neox(q, rot, pos, fp16);
                rotary_neox(k, rot, pos, fp16);
            }
            float* kc = k_cache + ((size_t)b * nh + h) * s_max * dh;
            float* vc = v_cache + ((size_t)b * nh + h) * s_max * dh;
            for (int d = 0; d < dh; d++) {
                kc[(size_t)tl * dh + d] = k[d];
                vc[(size_t)tl * dh + d] = v[d];
            }
            float* p     = (float*)malloc(sizeof(float) * (size_t)(tl + 1));
            float  qkmax = -FLT_MAX;
            for (int t = 0; t <= tl; t++) {
                double acc = 0;
                const float* kt = kc;
                if (cache_indir && t < tl) {
                    const int src = (b / beam_width) * beam_width + cache_indir[(size_t)b * s_max + t];
                    kt            = k_cache + ((size_t)src * nh + h) * s_max * dh;
                }
                for (int d = 0; d < dh; d++) {
                    acc += (double)q[d] * (double)kt[(size_t)t * dh + d];
                }
                float qk  = (float)acc * inv_sqrt_dh;
                p[t]      = qk;
                int mask  = (t < tl) && masked_tokens && masked_tokens[(size_t)b * s_max + t];
                if (!mask && qk > qkmax) {
                    qkmax = qk; /* :1570 ; the current step is never masked (:1434) */
                }
            }
            double sum = 0;
            for (int t = 0; t <= tl; t++) {
                int   mask = masked_tokens && masked_tokens[(size_t)b * s_max + t];
                float e    = mask ? 0.f : expf(p[t] - qkmax); /* :1610-1622 */
                p[t]       = e;
                sum += e;
            }
            float inv_sum = 1.f / ((float)sum + 1.e-6f); /* :1632 */
            for (int t = 0; t <= tl; t++) {
                p[t] = RT(p[t] * inv_sum); /* logits stored as T before P.V (:1643) */
            }
            /* P.V with fp32 accumulation in V_PER_ITER thread groups, then a tree reduction whose upper half
             * passes through T shared memory (:1692-1696, :1866-1890). */
            int threads = (tl < 32) ? 64 : ((tl < 2048) ? 128 : 256);
            /* THREADS_PER_VALUE comes from Dh_MAX, the head size rounded up to a power of two: the dispatch of
             * decoder_masked_multihead_attention.cu:29-59 launches <T, Dh, Dh_MAX> = <48, 64>, <80, 128>, <144, 256> ... and
             * the template sizes its thread groups by Dh_MAX (…template.hpp:1112), so V_PER_ITER is a power of two for
             * every head size and the tree below never drops a group */
            int dh_max = 32;
            while (dh_max < dh) {
                dh_max *= 2;
            }
            int tpv = fp16 ? (dh_max * 2 / 16) : (dh_max * 4 / 16); /* THREADS_PER_VALUE */
            if (tpv < 1) {
                tpv = 1;
            }
            int G = threads / tpv;
            if (G < 1) {
                G = 1;
            }
            double* part = (double*)calloc((size_t)G * dh, sizeof(double));
            for (int t = 0; t < tl; t++) {
                int          g  = t % G;
                const float* vt = vc;
                if (cache_indir) {
                    const int src = (b / beam_width) * beam_width + cache_indir[(size_t)b * s_max + t];
                    vt            = v_cache + ((size_t)src * nh + h) * s_max * dh;
                }
                for (int d = 0; d < dh; d++) {
                    part[(size_t)g * dh + d] += (double)p[t] * (double)vt[(size_t)t * dh + d];
                }
            }
            {
                int g = tl % G; /* :1799 */
                for (int d = 0; d < dh; d++) {
                    part[(size_t)g * dh + d] += (double)p[tl] * (double)v[d];
                }
            }
            for (int active = G; active >= 2; active /= 2) {
                int mid = active / 2;
                for (int g = 0; g < mid; g++) {
                    for (int d = 0; d < dh; d++) {
                        float up = (float)part[(size_t)(g + mid) * dh + d];
                        part[(size_t)g * dh + d] = (double)RT(up) + part[(size_t)g * dh + d];
                    }
                }
            }
            for (int d = 0; d < dh; d++) {
                ctx[(size_t)b * hl + h * dh + d] = RT((float)part[d]);
            }
            free(part);
            free(p);
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* prefill attention: layers/attention_layers/GptContextAttentionLayer.cc:101-393 (UNFUSED_MHA with   */
/* padding removal), kernels/unfused_attention_kernels.cu:1326-1484 (bias+rotary+transpose),         */
/* :255-332 (masked softmax), kernels/gpt_kernels.cu:359-402 (mask).                                 */
/* ------------------------------------------------------------------------------------------------ */
void orc_context_attention(const float* qkv, const float* qkv_bias, const int* input_lengths, float* k_cache,
                           float* v_cache, int B, int S, int nh, int dh, int rot, int s_max, float* ctx, int fp16)
{
    const int hl = nh * dh;
    /* qk_scale is computed in T (GptContextAttentionLayer.cc: `const T qk_scale = static_cast<T>(1/sqrtf(dh))`) */
    const float qk_scale = RT(1.0f / sqrtf((float)dh));
    float*      Q        = (float*)calloc((size_t)B * nh * S * dh, sizeof(float));
    float*      K        = (float*)calloc((size_t)B * nh * S * dh, sizeof(float));
    float*      V        = (float*)calloc((size_t)B * nh * S * dh, sizeof(float));
    /* padded rows stay zero: q_buf_2_/k/v are memset before the un-padded scatter (:156 ff.) */
    for (int b = 0; b < B; b++) {
        for (int s = 0; s < input_lengths[b] && s < S; s++) {
            const float* row = qkv + ((size_t)b * S + s) * 3 * hl;
            for (int h = 0; h < nh; h++) {
                float q[256], k[256];
                for (int d = 0; d < dh; d++) {
                    float bq = qkv_bias ? qkv_bias[h * dh + d] : 0.f;
                    float bk = qkv_bias ? qkv_bias[hl + h * dh + d] : 0.f;
                    float bv = qkv_bias ? qkv_bias[2 * hl + h * dh + d] : 0.f;
                    q[d]     = RT(row[h * dh + d] + bq);
                    k[d]     = RT(row[hl + h * dh + d] + bk);
                    V[(((size_t)b * nh + h) * S + s) * dh + d] = RT(row[2 * hl + h * dh + d] + bv);
                }
                if (rot > 0) {
                    rotary_neox(q, rot, s, fp16); /* position = index in the padded row */
                    rotary_neox(k, rot, s, fp16);
                }
                memcpy(Q + (((size_t)b * nh + h) * S + s) * dh, q, sizeof(float) * dh);
                memcpy(K + (((size_t)b * nh + h) * S + s) * dh, k, sizeof(float) * dh);
            }
        }
    }
    /* caches: kernels/unfused_attention_kernels.cu:1673-1749 (layout is engine private) */
    for (int b = 0; b < B; b++) {
        for (int h = 0; h < nh; h++) {
            for (int s = 0; s < S; s++) {
                memcpy(k_cache + (((size_t)b * nh + h) * s_max + s) * dh, K + (((size_t)b * nh + h) * S + s) * dh,
                       sizeof(float) * dh);
                memcpy(v_cache + (((size_t)b * nh + h) * s_max + s) * dh, V + (((size_t)b * nh + h) * S + s) * dh,
                       sizeof(float) * dh);
            }
        }
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; b++) {
        for (int h = 0; h < nh; h++) {
            float*       p  = (float*)malloc(sizeof(float) * (size_t)S);
            const float* Qh = Q + ((size_t)b * nh + h) * S * dh;
            const float* Kh = K + ((size_t)b * nh + h) * S * dh;
            const float* Vh = V + ((size_t)b * nh + h) * S * dh;
            const int    len = input_lengths[b];
            for (int qi = 0; qi < len && qi < S; qi++) { /* padded query rows are computed and discarded */
                float mx = -1e20f;
                for (int ki = 0; ki < S; ki++) {
                    double acc = 0;
                    for (int d = 0; d < dh; d++) {
                        acc += (double)Qh[(size_t)qi * dh + d] * (double)Kh[(size_t)ki * dh + d];
                    }
                    float mask = (qi < len && ki <= qi) ? 1.f : 0.f;
                    float val  = qk_scale * (float)acc + (1.0f - mask) * -10000.0f;
                    p[ki]      = val;
                    if (val > mx) {
                        mx = val;
                    }
                }
                double sum = 0;
                for (int ki = 0; ki < S; ki++) {
                    p[ki] = expf(p[ki] - mx);
                    sum += p[ki];
                }
                float inv = 1.0f / ((float)sum + 1e-6f);
                for (int ki = 0; ki < S; ki++) {
                    p[ki] = RT(p[ki] * inv);
                }
                for (int d = 0; d < dh; d++) {
                    double acc = 0;
                    for (int ki = 0; ki < S; ki++) {
                        acc += (double)p[ki] * (double)Vh[(size_t)ki * dh + d];
                    }
                    ctx[((size_t)b * S + qi) * hl + h * dh + d] = RT((float)acc);
                }
            }
            free(p);
        }
    }
    free(Q);
    free(K);
    free(V);
}

/* ------------------------------------------------------------------------------------------------ */
/* dynamic decode (beam_width == 1)                                                                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    float v;
    int   i;
} orc_vi;

/* k largest, value descending, ties -> lower index first (the reference's cub reduction leaves ties unspecified,
 * kernels/reduce_kernel_utils.cuh:344-348) */
static void topk_desc(const float* x, int n, int k, orc_vi* out)
{
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        float v = x[i];
        if (cnt == k && !(v > out[cnt - 1].v)) {
            continue;
        }
        int pos = (cnt < k) ? cnt : k - 1;
        while (pos > 0 && v > out[pos - 1].v) {
            out[pos] = out[pos - 1];
            pos--;
        }
        out[pos].v = v;
        out[pos].i = i;
        if (cnt < k) {
            cnt++;
        }
    }
    for (int j = cnt; j < k; j++) {
        out[j].v = -FLT_MAX;
        out[j].i = -1;
    }
}

static int cmp_desc(const void* a, const void* b)
{
    const orc_vi* x = (const orc_vi*)a;
    const orc_vi* y = (const orc_vi*)b;
    if (x->v > y->v) {
        return -1;
    }
    if (x->v < y->v) {
        return 1;
    }
    return (x->i > y->i) - (x->i < y->i);
}

/* kernels/sampling_topp_kernels.cu:1296-1345 addBiasSoftMax (bias == nullptr) */
static void softmax_endmask(float* l, int V, int finish, int end_id)
{
    float mx = -FLT_MAX;
    for (int j = 0; j < V; j++) {
        if (finish) {
            l[j] = (j == end_id) ? FLT_MAX : -FLT_MAX;
        }
        if (l[j] > mx) {
            mx = l[j];
        }
    }
    double sum = 0;
    for (int j = 0; j < V; j++) {
        l[j] = expf(l[j] - mx);
        sum += l[j];
    }
    float s = (float)sum + 1e-6f;
    for (int j = 0; j < V; j++) {
        l[j] = l[j] / s;
    }
}

void orc_dynamic_decode(float* logits, int B, int V, int step, int max_input_len, const int* input_lengths,
                        const orc_sampling* sp, int end_id, int* output_ids, uint8_t* finished, int* seq_len,
                        float* cum_log_probs, uint64_t* draw_counter, int total_len)
{
    (void)total_len;
    /* K15: kernels/select_optional_last_tokens.cu:22-85, first generated step only (DynamicDecodeLayer.cc:250-267) */
    if (step == max_input_len && sp->optional_last_tokens) {
        for (int b = 0; b < B; b++) {
            uint8_t* allow = (uint8_t*)calloc((size_t)V, 1);
            for (int j = 0; j < sp->optional_count; j++) {
                int t = sp->optional_last_tokens[(size_t)b * sp->optional_count + j];
                if (t >= 0 && t < V) {
                    allow[t] = 1;
                }
            }
            for (int j = 0; j < V; j++) {
                if (!allow[j]) {
                    logits[(size_t)b * V + j] = -INFINITY;
                }
            }
            free(allow);
        }
    }
    /* runtime-arg routing: TopKSamplingLayer.cu:27-77 / TopPSamplingLayer.cu:30-110 */
    int*   k_eff   = (int*)malloc(sizeof(int) * B);
    float* p_topk  = (float*)malloc(sizeof(float) * B);
    float* p_topp  = (float*)malloc(sizeof(float) * B);
    int    any_topk = 0, any_topp = 0;
    for (int b = 0; b < B; b++) {
        int   k = sp->top_k ? sp->top_k[b] : 0;
        float p = sp->top_p ? sp->top_p[b] : 0.f;
        if (k == 0 && p == 0.0f) {
            k = 1;
        }
        float pk = p;
        if (k > 0 && pk == 0.0f) {
            pk = 1.0f;
        }
        k_eff[b]  = k > 1024 ? 1024 : k;
        p_topk[b] = pk < 0.f ? 0.f : (pk > 1.f ? 1.f : pk);
        p_topp[b] = p < 0.f ? 0.f : (p > 1.f ? 1.f : p);
        if (k > 0) {
            any_topk = 1;
        }
        else {
            any_topp = 1;
        }
    }
    /* BaseSamplingLayer.cc:255-359.  Both layers run the same penalty pipeline on the rows they own (when a layer
     * skips some rows it works on a private copy, so every row is penalised exactly once). */
    for (int layer = 0; layer < 2; layer++) { /* 0: top-k layer, 1: top-p layer */
        if ((layer == 0 && !any_topk) || (layer == 1 && !any_topp)) {
            continue;
        }
        /* temperature is applied to EVERY row of the layer as soon as one row differs from 1.0 (:283-292) */
        int temp_all_one = 1, rep_all_default = 1;
        for (int b = 0; b < B; b++) {
            if (sp->temperature && sp->temperature[b] != 1.0f) {
                temp_all_one = 0;
            }
            if (sp->repetition_penalty && sp->repetition_penalty[b] != 1.0f) {
                rep_all_default = 0;
            }
        }
        for (int b = 0; b < B; b++) {
            int owns = (layer == 0) ? (k_eff[b] > 0) : (k_eff[b] == 0);
            if (!owns) {
                continue;
            }
            float* l = logits + (size_t)b * V;
            if (!temp_all_one) { /* kernels/sampling_penalty_kernels.cu:117-147 */
                float inv = 1.0f / (sp->temperature[b] + 1e-6f);
                for (int j = 0; j < V; j++) {
                    l[j] *= inv;
                }
            }
            if (step > 1 && sp->repetition_penalty && !rep_all_default) { /* :367-425 */
                float    pen     = sp->repetition_penalty[b];
                int      in_len  = input_lengths ? input_lengths[b] : max_input_len;
                float*   newv    = (float*)malloc(sizeof(float) * (size_t)step);
                int*     idx     = (int*)malloc(sizeof(int) * (size_t)step);
                int      cnt     = 0;
                for (int t = 0; t < step; t++) {
                    if (t >= in_len && t < max_input_len) {
                        continue;
                    }
                    int   id = output_ids[(size_t)t * B + b];
                    float lg = l[id];
                    idx[cnt]   = id;
                    newv[cnt++] = lg < 0.0f ? lg * pen : lg / pen;
                }
                for (int c = 0; c < cnt; c++) {
                    l[idx[c]] = newv[c];
                }
                free(newv);
                free(idx);
            }
            if (sp->min_length) { /* :485-520 */
                if (seq_len[b] + 1 - max_input_len < sp->min_length[b]) {
                    l[end_id] = -FLT_MAX;
                }
            }
            const int fin = finished[b];
            if (layer == 0) {
                /* kernels/sampling_topk_kernels.cu:67-110 end mask */
                if (fin) {
                    for (int j = 0; j < V; j++) {
                        l[j] = (j == end_id) ? FLT_MAX : -FLT_MAX;
                    }
                }
                if (sp->return_cum_log_probs) {
                    softmax_endmask(l, V, fin, end_id); /* TopKSamplingLayer.cu:235-246 */
                }
                if (fin) {
                    output_ids[(size_t)step * B + b] = end_id; /* :239-242 stage2 */
                    continue;
                }
                int     k    = k_eff[b];
                orc_vi* top  = (orc_vi*)malloc(sizeof(orc_vi) * (size_t)k);
                topk_desc(l, V, k, top);
                float  smax = top[0].v;
                float  ssum = 0.f;
                float* val2 = (float*)malloc(sizeof(float) * (size_t)k);
                for (int i = 0; i < k; i++) {
                    float u = top[i].v;
                    if (!sp->return_cum_log_probs) {
                        u = expf(u - smax); /* :271-275 */
                    }
                    val2[i] = u;
                    ssum += u;
                }
                float u01  = orc_uniform(sp->random_seed ? sp->random_seed[b] : 0, 0, draw_counter[b]++);
                float rnd  = u01 * p_topk[b] * ssum; /* :283 */
                int   pick = k - 1;
                for (int i = 0; i < k; i++) {
                    rnd -= val2[i];
                    if (rnd <= 0.0f || i == k - 1) {
                        pick = i;
                        break;
                    }
                }
                int id                           = top[pick].i;
                output_ids[(size_t)step * B + b] = id;
                if (sp->return_cum_log_probs && cum_log_probs) {
                    cum_log_probs[b] += logf(val2[pick]);
                }
                seq_len[b] += 1; /* :305-308 */
                finished[b] = (id == end_id);
                free(top);
                free(val2);
            }
            else {
                /* TopPSamplingLayer.cu runSampling: softmax always, then kernels/sampling_topp_kernels.cu:802-1000 */
                softmax_endmask(l, V, fin, end_id);
                float u01 = orc_uniform(sp->random_seed ? sp->random_seed[b] : 0, 0, draw_counter[b]++);
                float thr = p_topp[b];
                int   best = 0;
                for (int j = 1; j < V; j++) {
                    if (l[j] > l[best]) {
                        best = j;
                    }
                }
                int   id;
                float pr;
                if (l[best] >= thr) { /* topp_beam_topk_kernel<T,1> shortcut */
                    id = best;
                    pr = l[best];
                }
                else {
                    orc_vi* srt = (orc_vi*)malloc(sizeof(orc_vi) * (size_t)V);
                    for (int j = 0; j < V; j++) {
                        srt[j].v = l[j];
                        srt[j].i = j;
                    }
                    qsort(srt, (size_t)V, sizeof(orc_vi), cmp_desc);
                    float rnd = u01 * thr;
                    float cum = 0.f;
                    int   sel = V - 1;
                    for (int j = 0; j < V; j++) {
                        cum += srt[j].v;
                        if (rnd <= cum) {
                            sel = j;
                            break;
                        }
                    }
                    id = srt[sel].i;
                    pr = srt[sel].v;
                    free(srt);
                }
                output_ids[(size_t)step * B + b] = id;
                if (sp->return_cum_log_probs && cum_log_probs) {
                    cum_log_probs[b] += logf(pr);
                }
                seq_len[b]  = fin ? seq_len[b] : seq_len[b] + 1;
                finished[b] = (id == end_id);
            }
        }
    }
    /* kernels/stop_criteria_kernels.cu:24-83 */
    if (sp->stop_words) {
        for (int b = 0; b < B; b++) {
            const int* words = sp->stop_words + (size_t)b * 2 * sp->stop_len;
            const int* offs  = words + sp->stop_len;
            for (int id = 0; id < sp->stop_len; id++) {
                if (offs[id] < 0) {
                    continue;
                }
                int item_end   = offs[id];
                int item_start = id > 0 ? offs[id - 1] : 0;
                int item_size  = item_end - item_start;
                int stop       = 0;
                if (step + 1 >= item_size) {
                    stop = 1;
                    for (int t = item_size - 1; t >= 0; t--) {
                        int prev = output_ids[(size_t)(step - (item_size - 1) + t) * B + b];
                        if (prev != words[item_start + t]) {
                            stop = 0;
                            break;
                        }
                    }
                }
                if (stop) {
                    finished[b] = 1;
                }
            }
        }
    }
    free(k_eff);
    free(p_topk);
    free(p_topp);
}

/* ------------------------------------------------------------------------------------------------ */
/* decoder stack helpers                                                                             */
/* ------------------------------------------------------------------------------------------------ */
static void layer_gemm(const orc_config* c, const float* A, int m, int k, int n, const float* W, const int8_t* q,
                       const float* s, const float* bias, int act, float* C)
{
    if (c->int8_mode == 1) {
        orc_gemm(A, m, k, n, NULL, q, s, bias, act, C, c->fp16, 0);
    }
    else {
        orc_gemm(A, m, k, n, W, NULL, NULL, bias, act, C, c->fp16, 0);
    }
}

/* models/gptneox/GptNeoXDecoder.cc:245-384 */
void orc_decoder_step(const orc_config* c, const orc_weights* w, const float* x_in, float* k_cache, float* v_cache,
                      const int* seq_len, const int* pad_count, const uint8_t* masked_tokens, const uint8_t* finished,
                      int B, int s_max, int step, float* y)
{
    orc_decoder_step_beam(c, w, x_in, k_cache, v_cache, seq_len, pad_count, masked_tokens, finished, B, s_max, step, y, NULL,
                          1);
}

void orc_decoder_step_beam(const orc_config* c, const orc_weights* w, const float* x_in, float* k_cache, float* v_cache,
                           const int* seq_len, const int* pad_count, const uint8_t* masked_tokens,
                           const uint8_t* finished, int B, int s_max, int step, float* y, const int* cache_indir,
                           int beam_width)
{
    const int H = c->head_num * c->size_per_head, nhl = c->head_num / c->tp_size, hl = nhl * c->size_per_head;
    const int il   = c->inter_size / c->tp_size;
    const int L    = c->num_layer;
    const int fp16 = c->fp16;
    float*    x    = (float*)malloc(sizeof(float) * (size_t)B * H);
    float*    nrm  = (float*)malloc(sizeof(float) * (size_t)B * H);
    float*    qkv  = (float*)malloc(sizeof(float) * (size_t)B * 3 * hl);
    float*    ctx  = (float*)calloc((size_t)B * hl, sizeof(float));
    float*    att  = (float*)malloc(sizeof(float) * (size_t)B * H);
    float*    mid  = (float*)malloc(sizeof(float) * (size_t)B * il);
    float*    ffn  = (float*)malloc(sizeof(float) * (size_t)B * H);
    memcpy(x, x_in, sizeof(float) * (size_t)B * H);
    const size_t cache_l = (size_t)B * nhl * s_max * c->size_per_head;
    for (int l = 0; l < L; l++) {
        orc_layernorm(x, w->ln1_g[l], w->ln1_b[l], B, H, 1e-5f, nrm, fp16);
        /* DecoderSelfAttentionLayer.cc:532-577 QKV GEMM (no bias: MMHA adds it), :581-614 MMHA, :635-678 out proj */
        layer_gemm(c, nrm, B, H, 3 * hl, w->qkv_w ? w->qkv_w[l] : NULL, w->qkv_q ? w->qkv_q[l] : NULL,
                   w->qkv_s ? w->qkv_s[l] : NULL, NULL, 0, qkv);
        orc_mmha_step_beam(qkv, w->qkv_b[l], k_cache + l * cache_l, v_cache + l * cache_l, seq_len, pad_count,
                           masked_tokens, finished, B, nhl, c->size_per_head, c->rotary_dim, s_max, step, ctx, fp16,
                           cache_indir, beam_width);
        layer_gemm(c, ctx, B, hl, H, w->out_w ? w->out_w[l] : NULL, w->out_q ? w->out_q[l] : NULL,
                   w->out_s ? w->out_s[l] : NULL, NULL, 0, att);
        if (c->use_gptj_residual) {
            orc_layernorm(x, w->ln2_g[l], w->ln2_b[l], B, H, 1e-5f, nrm, fp16);
        }
        else {
            /* invokeGeneralAddBiasResidualPreLayerNorm: att = att + bias + x ; nrm = LN(att) (GptNeoXDecoder.cc:313-331) */
            for (int i = 0; i < B * H; i++) {
                float bsum = w->out_b && w->out_b[l] ? w->out_b[l][i % H] : 0.f;
                att[i]     = RT(bsum + x[i] + att[i]);
            }
            orc_layernorm(att, w->ln2_g[l], w->ln2_b[l], B, H, 1e-5f, nrm, fp16);
        }
        /* FfnLayer.cc:172-372 */
        layer_gemm(c, nrm, B, H, il, w->ffn1_w ? w->ffn1_w[l] : NULL, w->ffn1_q ? w->ffn1_q[l] : NULL,
                   w->ffn1_s ? w->ffn1_s[l] : NULL, w->ffn1_b[l], 1, mid);
        layer_gemm(c, mid, B, il, H, w->ffn2_w ? w->ffn2_w[l] : NULL, w->ffn2_q ? w->ffn2_q[l] : NULL,
                   w->ffn2_s ? w->ffn2_s[l] : NULL, NULL, 0, ffn);
        if (c->use_gptj_residual) {
            /* layer_input/output alias for 0 < l < L-1 -> in-place variant (GptNeoXDecoder.cc:249-250,342-356) */
            int inplace = (l > 0 && l < L - 1);
            orc_add_bias_attn_ffn_residual(x, ffn, att, x, w->ffn2_b[l], B, H, c->tp_size, inplace, fp16);
            if (c->tp_size > 1 && c->allreduce) {
                c->allreduce(x, (long)B * H, c->comm_ctx);
                for (int i = 0; i < B * H; i++) {
                    x[i] = RT(x[i]);
                }
            }
        }
        else {
            /* invokeAddBiasResidual: out = ffn + att(residual) + bias (:362-367); all-reduces happen inside the
             * TensorParallel* wrappers (2 per layer) -- not modelled for tp>1 here. */
            for (int i = 0; i < B * H; i++) {
                x[i] = RT(ffn[i] + att[i] + w->ffn2_b[l][i % H]);
            }
        }
    }
    memcpy(y, x, sizeof(float) * (size_t)B * H);
    free(x);
    free(nrm);
    free(qkv);
    free(ctx);
    free(att);
    free(mid);
    free(ffn);
}

/* models/gptneox/GptNeoXContextDecoder.cc:283-507 ; returns hidden state of every (padded) position [B,S,H] */
static void context_decoder(const orc_config* c, const orc_weights* w, float* x /*[B*S,H] in/out*/, float* k_cache,
                            float* v_cache, const int* input_lengths, int B, int S, int s_max)
{
    const int H = c->head_num * c->size_per_head, nhl = c->head_num / c->tp_size, hl = nhl * c->size_per_head;
    const int il   = c->inter_size / c->tp_size;
    const int L    = c->num_layer;
    const int fp16 = c->fp16;
    const int M    = B * S;
    float*    nrm  = (float*)malloc(sizeof(float) * (size_t)M * H);
    float*    qkv  = (float*)malloc(sizeof(float) * (size_t)M * 3 * hl);
    float*    ctx  = (float*)calloc((size_t)M * hl, sizeof(float));
    float*    att  = (float*)malloc(sizeof(float) * (size_t)M * H);
    float*    mid  = (float*)malloc(sizeof(float) * (size_t)M * il);
    float*    ffn  = (float*)malloc(sizeof(float) * (size_t)M * H);
    const size_t cache_l = (size_t)B * nhl * s_max * c->size_per_head;
    for (int l = 0; l < L; l++) {
        orc_layernorm(x, w->ln1_g[l], w->ln1_b[l], M, H, 1e-5f, nrm, fp16);
        layer_gemm(c, nrm, M, H, 3 * hl, w->qkv_w ? w->qkv_w[l] : NULL, w->qkv_q ? w->qkv_q[l] : NULL,
                   w->qkv_s ? w->qkv_s[l] : NULL, NULL, 0, qkv);
        memset(ctx, 0, sizeof(float) * (size_t)M * hl);
        orc_context_attention(qkv, w->qkv_b[l], input_lengths, k_cache + l * cache_l, v_cache + l * cache_l, B, S, nhl,
                              c->size_per_head, c->rotary_dim, s_max, ctx, fp16);
        layer_gemm(c, ctx, M, hl, H, w->out_w ? w->out_w[l] : NULL, w->out_q ? w->out_q[l] : NULL,
                   w->out_s ? w->out_s[l] : NULL, NULL, 0, att);
        if (c->use_gptj_residual) {
            orc_layernorm(x, w->ln2_g[l], w->ln2_b[l], M, H, 1e-5f, nrm, fp16);
        }
        else {
            for (int i = 0; i < M * H; i++) {
                float bsum = w->out_b && w->out_b[l] ? w->out_b[l][i % H] : 0.f;
                att[i]     = RT(bsum + x[i] + att[i]);
            }
            orc_layernorm(att, w->ln2_g[l], w->ln2_b[l], M, H, 1e-5f, nrm, fp16);
        }
        layer_gemm(c, nrm, M, H, il, w->ffn1_w ? w->ffn1_w[l] : NULL, w->ffn1_q ? w->ffn1_q[l] : NULL,
                   w->ffn1_s ? w->ffn1_s[l] : NULL, w->ffn1_b[l], 1, mid);
        layer_gemm(c, mid, M, il, H, w->ffn2_w ? w->ffn2_w[l] : NULL, w->ffn2_q ? w->ffn2_q[l] : NULL,
                   w->ffn2_s ? w->ffn2_s[l] : NULL, NULL, 0, ffn);
        if (c->use_gptj_residual) {
            /* with padding removal layer_input == layer_output == decoder_layer_output_ for every layer
             * (GptNeoXContextDecoder.cc:311-322) -> always the in-place (fp32 sum, one rounding) variant */
            orc_add_bias_attn_ffn_residual(x, ffn, att, x, w->ffn2_b[l], M, H, c->tp_size, 1, fp16);
            if (c->tp_size > 1 && c->allreduce) {
                c->allreduce(x, (long)M * H, c->comm_ctx);
                for (int i = 0; i < M * H; i++) {
                    x[i] = RT(x[i]);
                }
            }
        }
        else {
            for (int i = 0; i < M * H; i++) {
                x[i] = RT(ffn[i] + att[i] + w->ffn2_b[l][i % H]);
            }
        }
    }
    free(nrm);
    free(qkv);
    free(ctx);
    free(att);
    free(mid);
    free(ffn);
}

/* ------------------------------------------------------------------------------------------------ */
/* whole path: models/gptneox/GptNeoX.cc:386-1052 + setOutputTensors :1090-1181 (gatherTree           */
/* kernels/decoding_kernels.cu:452-583 with beam_width 1)                                            */
/* ------------------------------------------------------------------------------------------------ */
int orc_generate(const orc_config* c, const orc_weights* w, const int* input_ids, const int* input_lengths, int B,
                 int S, int out_len, const orc_sampling* sp, int* output_ids, int* sequence_lengths,
                 float* cum_log_probs_out, float* dbg_logits, float* dbg_hidden)
{
    const int H = c->head_num * c->size_per_head, nhl = c->head_num / c->tp_size;
    const int V = c->vocab_size, L = c->num_layer;
    const int fp16  = c->fp16;
    const int total = S + out_len; /* max_output_seq_len == max_seq_len == max_cache_seq_len */
    const int s_max = total;
    const int vl    = V / c->tp_size;

    const size_t cache_sz = (size_t)L * B * nhl * s_max * c->size_per_head;
    float*       k_cache  = (float*)calloc(cache_sz, sizeof(float));
    float*       v_cache  = (float*)calloc(cache_sz, sizeof(float));
    int*         ids      = (int*)calloc((size_t)total * B, sizeof(int)); /* output_ids_buf_ [time, batch] */
    uint8_t*     finished = (uint8_t*)calloc((size_t)B, 1);
    int*         seq_len  = (int*)calloc((size_t)B, sizeof(int));
    int*         pad_cnt  = (int*)calloc((size_t)B, sizeof(int));
    uint8_t*     masked   = (uint8_t*)calloc((size_t)B * s_max, 1);
    float*       cum      = (float*)calloc((size_t)B, sizeof(float));
    uint64_t*    draws    = (uint64_t*)calloc((size_t)B, sizeof(uint64_t));
    float*       hid      = (float*)malloc(sizeof(float) * (size_t)B * H);
    float*       hid_out  = (float*)malloc(sizeof(float) * (size_t)B * H);
    float*       nrm      = (float*)malloc(sizeof(float) * (size_t)B * H);
    float*       logits   = (float*)malloc(sizeof(float) * (size_t)B * V);
    float*       gather   = (float*)malloc(sizeof(float) * (size_t)B * V);

    int max_input_length = S;
    if (S > 1) {
        /* kernels/gpt_kernels.cu:31-104: ids -> time-major output_ids + embedding lookup */
        float* x = (float*)malloc(sizeof(float) * (size_t)B * S * H);
        for (int b = 0; b < B; b++) {
            for (int s = 0; s < S; s++) {
                int id               = input_ids[(size_t)b * S + s];
                ids[(size_t)s * B + b] = id;
                memcpy(x + ((size_t)b * S + s) * H, w->wte + (size_t)id * H, sizeof(float) * H);
            }
        }
        context_decoder(c, w, x, k_cache, v_cache, input_lengths, B, S, s_max);
        for (int b = 0; b < B; b++) { /* lookupHiddenStateOfLastToken (:438-470) */
            memcpy(hid_out + (size_t)b * H, x + ((size_t)b * S + (input_lengths[b] - 1)) * H, sizeof(float) * H);
        }
        free(x);
        for (int b = 0; b < B; b++) { /* invokeDecodingInitialize (decoding_kernels.cu:26-65) */
            finished[b] = 0;
            seq_len[b]  = max_input_length - 1;
            cum[b]      = 0.f;
        }
    }
    else {
        /* S == 1: no prefill (GptNeoX.cc:719-747) */
        for (int b = 0; b < B; b++) {
            finished[b]  = 0;
            seq_len[b]   = max_input_length - 1;
            cum[b]       = 0.f;
            ids[b]       = input_ids[b];
        }
    }
    /* invokeMaskPaddingTokens (gpt_kernels.cu:1035-1082) */
    for (int b = 0; b < B; b++) {
        for (int s = input_lengths[b]; s < max_input_length; s++) {
            masked[(size_t)b * s_max + s] = 1;
        }
    }

    int steps_run = 0;
    for (int step = max_input_length; step < total; step++) {
        if (!(max_input_length > 1 && step == max_input_length)) {
            /* embedding of the previous token (decoding_kernels.cu:145-191), no position table for NeoX */
            for (int b = 0; b < B; b++) {
                int id = ids[(size_t)(step - 1) * B + b];
                memcpy(hid + (size_t)b * H, w->wte + (size_t)id * H, sizeof(float) * H);
            }
            orc_decoder_step(c, w, hid, k_cache, v_cache, seq_len, pad_cnt, masked, finished, B, s_max, step, hid_out);
        }
        orc_layernorm(hid_out, w->final_ln_g, w->final_ln_b, B, H, 1e-5f, nrm, fp16);
        if (c->tp_size == 1) {
            orc_lm_head(nrm, B, H, V, w->lm_head, logits);
        }
        else {
            /* GptNeoX.cc:888-925: rank r computes rows [r*vl, (r+1)*vl) of the replicated lm_head, all-gather
             * [tp][B][vl] then transposeAxis01 -> [B][V] */
            orc_lm_head(nrm, B, H, vl, w->lm_head + (size_t)c->tp_rank * vl * H,
                        gather + (size_t)c->tp_rank * B * vl);
            if (c->allgather) {
                c->allgather(gather, (long)B * vl, c->comm_ctx);
            }
            for (int r = 0; r < c->tp_size; r++) {
                for (int b = 0; b < B; b++) {
                    memcpy(logits + (size_t)b * V + (size_t)r * vl, gather + ((size_t)r * B + b) * vl,
                           sizeof(float) * vl);
                }
            }
        }
        if (dbg_logits) {
            memcpy(dbg_logits + (size_t)(step - max_input_length) * B * V, logits, sizeof(float) * (size_t)B * V);
        }
        orc_dynamic_decode(logits, B, V, step, max_input_length, input_lengths, sp, c->end_id, ids, finished, seq_len,
                           cum, draws, total);
        /* length criterion (stop_criteria_kernels.cu:106-158): limit == total -> never hit inside the loop */
        int all = 1;
        for (int b = 0; b < B; b++) {
            if (step >= total) {
                finished[b] = 1;
            }
            all &= finished[b];
        }
        steps_run++;
        if (all) {
            break;
        }
        if (step == max_input_length) { /* invokeUpdatePaddingCount (gpt_kernels.cu:981-1033) */
            for (int b = 0; b < B; b++) {
                pad_cnt[b] += max_input_length - input_lengths[b];
            }
        }
    }
    if (dbg_hidden) {
        memcpy(dbg_hidden, hid_out, sizeof(float) * (size_t)B * H);
    }

    /* setOutputTensors -> gatherTree (beam 1, no prompts) */
    for (int b = 0; b < B; b++) {
        int  tmp_len   = seq_len[b] + 1;
        sequence_lengths[b] = tmp_len;
        int  max_len   = tmp_len;
        int  msl       = max_len < total ? max_len : total;
        int* beams     = (int*)calloc((size_t)total, sizeof(int));
        int  in_len    = input_lengths[b];
        int  pad_off   = max_input_length - in_len;
        if (msl > 0) {
            beams[msl - 1 - pad_off] = ids[(size_t)(msl - 1) * B + b];
            for (int level = msl - 2; level >= 0; level--) {
                if (level >= in_len && level < max_input_length) {
                    continue;
                }
                int tgt    = level >= max_input_length ? level - pad_off : level;
                beams[tgt] = ids[(size_t)level * B + b];
            }
            for (int index = max_len - pad_off; index < total; index++) {
                beams[index] = c->end_id;
            }
            int fin = 0;
            for (int t = max_input_length; t < msl; t++) {
                if (fin) {
                    beams[t] = c->end_id;
                }
                else if (beams[t] == c->end_id) {
                    fin = 1;
                }
            }
        }
        memcpy(output_ids + (size_t)b * total, beams, sizeof(int) * (size_t)total);
        free(beams);
        if (cum_log_probs_out) {
            cum_log_probs_out[b] = cum[b];
        }
    }
    free(k_cache);
    free(v_cache);
    free(ids);
    free(finished);
    free(seq_len);
    free(pad_cnt);
    free(masked);
    free(cum);
    free(draws);
    free(hid);
    free(hid_out);
    free(nrm);
    free(logits);
    free(gather);
    return steps_run;
}

/* ------------------------------------------------------------------------------------------------------------------
 * CUDA (SM75..SM89) int8 weight layout: restatement of cutlass_preprocessors.cc:139-201, 207-348, 350-370, 437-498,
 * 500-539 for QuantType::INT8_WEIGHT_ONLY (LayoutDetailsB<uint8_t, Sm75+>: ColumnMajorTileInterleave<64, 2>,
 * mixed_gemm_B_layout.h:59-72).  Used to pin the product's importer of .q.bin files written by the CUDA build.
 * ------------------------------------------------------------------------------------------------------------------ */
void orc_sm80_permute_rows(const int8_t* in, int K, int N, int8_t* out)
{
    /* write row r of every 16-row group reads row 8*((r%4)/2) + r%2 + 2*(r/4)  (:185-187 with ELTS_PER_REG = 4) */
    for (int base = 0; base < K; base += 16) {
        for (int r = 0; r < 16; r++) {
            const int rd = 8 * ((r % 4) / 2) + r % 2 + 2 * (r / 4);
            memcpy(out + (size_t)(base + r) * N, in + (size_t)(base + rd) * N, (size_t)N);
        }
    }
}
void orc_sm80_transpose(const int8_t* in, int K, int N, int8_t* out)
{
    for (int k = 0; k < K; k++) {
        for (int n = 0; n < N; n++) {
            out[(size_t)n * K + k] = in[(size_t)k * N + n];
        }
    }
}
void orc_sm80_interleave_columns(const int8_t* in, int K, int N, int8_t* out)
{
    /* column-major input: column n = K bytes = K/4 32-bit words; tiles of 64 rows (16 words), 2 columns interleaved */
    const int       vr = K / 4, per_tile = 16, il = 2;
    const uint32_t* src = (const uint32_t*)in;
    uint32_t*       dst = (uint32_t*)out;
    for (int n = 0; n < N; n++) {
        const size_t wc = (size_t)(n / il);
        for (int base = 0; base < vr; base += per_tile) {
            for (int v = base; v < vr && v < base + per_tile; v++) {
                const size_t vw = (size_t)il * base + (size_t)per_tile * (n % il) + (size_t)(v % per_tile);
                dst[wc * vr * il + vw] = src[(size_t)n * vr + v];
            }
        }
    }
}
void orc_sm80_add_bias_interleave_int8(int8_t* t, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        t[i] = (int8_t)((int)t[i] + 128);
    }
    for (size_t b = 0; b + 3 < n; b += 4) { /* [e3 e2 e1 e0] -> [e3 e1 e2 e0] */
        const int8_t x = t[b + 1];
        t[b + 1]       = t[b + 2];
        t[b + 2]       = x;
    }
}
void orc_sm80_preprocess_int8(const int8_t* row_major, int K, int N, int8_t* out)
{
    int8_t* a = (int8_t*)malloc((size_t)K * N);
    int8_t* b = (int8_t*)malloc((size_t)K * N);
    orc_sm80_permute_rows(row_major, K, N, a);
    orc_sm80_transpose(a, K, N, b);
    orc_sm80_interleave_columns(b, K, N, a);
    orc_sm80_add_bias_interleave_int8(a, (size_t)K * N);
    memcpy(out, a, (size_t)K * N);
    free(a);
    free(b);
}

/* ==================================================================================================================
 * Beam search (beam_width > 1): layers/DynamicDecodeLayer.cc:309-393 -> OnlineBeamSearchLayer / BaseBeamSearchLayer
 * without BeamHypotheses (GptNeoX.cc:948-985 passes none), kernels/online_softmax_beamsearch_kernels.cu,
 * kernels/beam_search_penalty_kernels.cu, decoding_kernels.cu:452-583 (gatherTree with parents).
 * Rows are bb = batch * K + beam.  output_ids / parent_ids are time-major [total][B*K].
 * ================================================================================================================== */
typedef struct {
    float v;
    int   i;
} beam_cand;

static int cmp_cand(const void* a, const void* b)
{
    const beam_cand* x = (const beam_cand*)a;
    const beam_cand* y = (const beam_cand*)b;
    if (x->v > y->v) {
        return -1;
    }
    if (x->v < y->v) {
        return 1;
    }
    return x->i - y->i; /* ties: lower index first (the reference's block reduce leaves this unspecified) */
}

void orc_beam_search_step(float* logits, int B, int K, int V, int step, int max_input_len, const int* input_lengths,
                          const orc_beam_params* bp, int end_id, int* output_ids, int* parent_ids, uint8_t* finished,
                          int* seq_len, float* cum_log_probs, const int* src_indir, int* tgt_indir, int s_max)
{
    const int BK = B * K;
    /* K15: select_optional_last_tokens.cu:22-85 at the first generated step (DynamicDecodeLayer.cc:250-267) */
    if (step == max_input_len && bp->optional_last_tokens) {
        for (int bb = 0; bb < BK; bb++) {
            const int b     = bb / K;
            uint8_t*  allow = (uint8_t*)calloc((size_t)V, 1);
            for (int j = 0; j < bp->optional_count; j++) {
                int t = bp->optional_last_tokens[(size_t)b * bp->optional_count + j];
                if (t >= 0 && t < V) {
                    allow[t] = 1;
                }
            }
            for (int j = 0; j < V; j++) {
                if (!allow[j]) {
                    logits[(size_t)bb * V + j] = -INFINITY;
                }
            }
            free(allow);
        }
    }
    beam_cand* cand = (beam_cand*)malloc(sizeof(beam_cand) * (size_t)K * K);
    float*     cy   = (float*)malloc(sizeof(float) * (size_t)K * K); /* un-penalised scores */
    int*       cx   = (int*)malloc(sizeof(int) * (size_t)K * K);     /* absolute ids: token + row * V */
    int*       new_parent = (int*)malloc(sizeof(int) * (size_t)BK);
    int*       new_word   = (int*)malloc(sizeof(int) * (size_t)BK);
    float*     new_cum    = (float*)malloc(sizeof(float) * (size_t)BK);
    for (int b = 0; b < B; b++) {
        const float temperature = bp->temperature ? bp->temperature[b] : 1.0f;
        const float rep         = bp->repetition_penalty ? bp->repetition_penalty[b] : 1.0f;
        const float diversity   = bp->diversity_rate ? bp->diversity_rate[b] : 0.0f;
        const float len_pen     = bp->len_penalty ? bp->len_penalty[b] : 0.0f;
        const int   min_length  = bp->min_length ? bp->min_length[b] : 0;
        for (int k = 0; k < K; k++) {
            const int bb = b * K + k;
            float*    l  = logits + (size_t)bb * V;
            /* invokeAddBiasApplyPenalties (beam_search_penalty_kernels.cu:171-262) */
            if (temperature != 1.0f) {
                const float inv = 1.0f / (temperature + 1e-6f);
                for (int j = 0; j < V; j++) {
                    l[j] *= inv;
                }
            }
            if (bp->repetition_penalty && step > 0 && rep != 1.0f) { /* :89-153: history along the parent chain */
                const int in_len = input_lengths ? input_lengths[bb] : max_input_len;
                float*    nv     = (float*)malloc(sizeof(float) * (size_t)step);
                int*      ni     = (int*)malloc(sizeof(int) * (size_t)step);
                int       cnt    = 0;
                int       prev   = output_ids[(size_t)(step - 1) * BK + bb];
                ni[cnt]          = prev;
                nv[cnt++]        = l[prev] > 0.f ? l[prev] / rep : l[prev] * rep;
                int parent_beam  = k;
                for (int i = step - 2; i >= 0; i--) {
                    if (i >= in_len && i < max_input_len) {
                        continue;
                    }
                    parent_beam = parent_ids[(size_t)i * BK + b * K + parent_beam];
                    prev        = output_ids[(size_t)i * BK + b * K + parent_beam];
                    ni[cnt]     = prev;
                    nv[cnt++]   = l[prev] > 0.f ? l[prev] / rep : l[prev] * rep;
                }
                for (int c = 0; c < cnt; c++) {
                    l[ni[c]] = nv[c];
                }
                free(nv);
                free(ni);
            }
            if (step - max_input_len < min_length) { /* :155-169, 252-261 */
                if (seq_len[bb] + 1 - max_input_len < min_length) {
                    l[end_id] = -FLT_MAX;
                }
            }
            /* beam_online_softmax_topk_kernel (online_softmax_beamsearch_kernels.cu:296-365): per row, top K of
             * log_softmax + cum_log_prob; a finished row offers its end token at cum + 0 and nothing else */
            beam_cand* row = (beam_cand*)malloc(sizeof(beam_cand) * (size_t)V);
            if (finished[bb]) {
                for (int j = 0; j < V; j++) {
                    row[j].v = (j == end_id) ? 0.0f : -INFINITY;
                    row[j].i = j;
                }
            }
            else {
                float mx = -FLT_MAX;
                for (int j = 0; j < V; j++) {
                    if (l[j] > mx) {
                        mx = l[j];
                    }
                }
                float d = 0.f;
                for (int j = 0; j < V; j++) {
                    d += expf(l[j] - mx);
                }
                const float logd = logf(d);
                for (int j = 0; j < V; j++) {
                    row[j].v = l[j] - mx - logd;
                    row[j].i = j;
                }
            }
            qsort(row, (size_t)V, sizeof(beam_cand), cmp_cand);
            for (int i = 0; i < K; i++) {
                cx[k * K + i] = row[i].i + bb * V;
                cy[k * K + i] = row[i].v + cum_log_probs[bb];
            }
            free(row);
        }
        /* batch_topk_kernel (:100-262, beam_hyps.num_beams == nullptr): K best of the K*K candidates of the batch.
         * NB the kernel indexes finished / sequence_lengths by the BATCH id (its blockIdx), not by batch*K+beam: restated
         * as written (only matters with len_penalty != 0, which the harness never sets). */
        for (int e = 0; e < K * K; e++) {
            float v = cy[e];
            if (len_pen != 0.0f) {
                const int length = finished[b] ? seq_len[b] : seq_len[b] + 1;
                if (length != 1) {
                    v = v / powf((float)length, len_pen);
                }
            }
            v += diversity * (float)(e % K);
            cand[e].v = v;
            cand[e].i = e;
        }
        qsort(cand, (size_t)K * K, sizeof(beam_cand), cmp_cand);
        for (int k = 0; k < K; k++) {
            const int e             = cand[k].i;
            const int z             = cx[e];
            new_parent[b * K + k]   = (z / V) % K;
            new_word[b * K + k]     = z % V;
            new_cum[b * K + k]      = cy[e];
        }
    }
    /* update_kernel (OnlineBeamSearchLayer.cu:25-58): lengths follow the parent beam */
    int*     old_seq = (int*)malloc(sizeof(int) * (size_t)BK);
    uint8_t* old_fin = (uint8_t*)malloc((size_t)BK);
    memcpy(old_seq, seq_len, sizeof(int) * (size_t)BK);
    memcpy(old_fin, finished, (size_t)BK);
    for (int bb = 0; bb < BK; bb++) {
        const int b = bb / K, pb = b * K + new_parent[bb];
        seq_len[bb]                          = old_fin[pb] ? old_seq[pb] : old_seq[pb] + 1;
        finished[bb]                         = new_word[bb] == end_id;
        parent_ids[(size_t)step * BK + bb]   = new_parent[bb];
        output_ids[(size_t)step * BK + bb]   = new_word[bb];
        cum_log_probs[bb]                    = new_cum[bb];
    }
    /* update_indir_cache_kernel (BaseBeamSearchLayer.cu:30-62): rows that just finished keep their stale entries */
    for (int bb = 0; bb < BK; bb++) {
        if (finished[bb]) {
            continue;
        }
        const int b = bb / K, k = bb % K, src_beam = new_parent[bb];
        for (int t = 0; t <= step && t < s_max; t++) {
            tgt_indir[(size_t)bb * s_max + t] = (t == step) ? k : src_indir[((size_t)b * K + src_beam) * s_max + t];
        }
    }
    /* stop words along the parent chain (stop_criteria_kernels.cu:24-83 with parent_ids) */
    if (bp->stop_words) {
        for (int bb = 0; bb < BK; bb++) {
            const int  b     = bb / K;
            const int* words = bp->stop_words + (size_t)b * 2 * bp->stop_len;
            const int* offs  = words + bp->stop_len;
            for (int id = 0; id < bp->stop_len; id++) {
                if (offs[id] < 0) {
                    continue;
                }
                const int item_end = offs[id], item_start = id > 0 ? offs[id - 1] : 0, item_size = item_end - item_start;
                int       stop = 0;
                if (step + 1 >= item_size) {
                    stop       = 1;
                    int parent = bb % K;
                    for (int t = item_size - 1; t >= 0; t--) {
                        const int ts  = step - (item_size - 1) + t;
                        const int tok = output_ids[(size_t)ts * BK + b * K + parent];
                        if (tok != words[item_start + t]) {
                            stop = 0;
                            break;
                        }
                        parent = parent_ids[(size_t)ts * BK + b * K + parent];
                    }
                }
                if (stop) {
                    finished[bb] = 1;
                }
            }
        }
    }
    free(old_seq);
    free(old_fin);
    free(cand);
    free(cy);
    free(cx);
    free(new_parent);
    free(new_word);
    free(new_cum);
}

/* gatherTree (decoding_kernels.cu:452-583) + the transpose to [B][K][total]; ids / parents time-major [total][B*K] */
void orc_gather_tree_beam(const int* ids, const int* parents, const int* seq_len, const int* t_len, int B, int K,
                          int max_input_len, int total, int end_id, int* output_ids, int* sequence_lengths)
{
    const int BK = B * K;
    for (int b = 0; b < B; b++) {
        int max_len = -1;
        for (int j = 0; j < K; j++) {
            const int tmp_len             = seq_len[b * K + j] + 1; /* max_sequence_length_final_step = 1 */
            sequence_lengths[b * K + j]   = tmp_len;
            if (tmp_len > max_len) {
                max_len = tmp_len;
            }
        }
        const int msl = max_len < total ? max_len : total;
        for (int k = 0; k < K; k++) {
            const int bb      = b * K + k;
            int*      beams   = output_ids + (size_t)bb * total;
            const int in_len  = t_len[bb];
            const int pad_off = max_input_len - in_len;
            for (int t = 0; t < total; t++) {
                beams[t] = 0;
            }
            if (msl <= 0) {
                continue;
            }
            beams[msl - 1 - pad_off] = ids[(size_t)(msl - 1) * BK + bb];
            int parent               = parents[(size_t)(msl - 1) * BK + bb] % K;
            for (int level = msl - 2; level >= 0; level--) {
                if (level >= in_len && level < max_input_len) {
                    continue;
                }
                const int tl2 = level >= max_input_len ? level - pad_off : level;
                beams[tl2]    = ids[(size_t)level * BK + b * K + parent];
                parent        = parents[(size_t)level * BK + b * K + parent] % K;
            }
            for (int index = max_len - pad_off; index < total; index++) {
                beams[index] = end_id;
            }
            int fin = 0;
            for (int t = max_input_len; t < msl; t++) {
                if (fin) {
                    beams[t] = end_id;
                }
                else if (beams[t] == end_id) {
                    fin = 1;
                }
            }
        }
    }
}

/* GptNeoX<T>::forward with beam_width = K > 1 (GptNeoX.cc:386-1052): inputs tiled K times, the context phase runs on all
 * B*K rows, cum_log_probs of beams > 0 start at -1e20 so that the first step expands beam 0 only. */
int orc_generate_beam(const orc_config* c, const orc_weights* w, const int* input_ids, const int* input_lengths, int B,
                      int S, int out_len, int K, const orc_beam_params* bp, int* output_ids, int* sequence_lengths,
                      float* cum_log_probs_out)
{
    const int H = c->head_num * c->size_per_head, nhl = c->head_num / c->tp_size;
    const int V = c->vocab_size, L = c->num_layer, fp16 = c->fp16;
    const int total = S + out_len, s_max = total, BK = B * K;
    const size_t cache_sz = (size_t)L * BK * nhl * s_max * c->size_per_head;
    float*   k_cache = (float*)calloc(cache_sz, sizeof(float));
    float*   v_cache = (float*)calloc(cache_sz, sizeof(float));
    int*     ids     = (int*)calloc((size_t)total * BK, sizeof(int));
    int*     parents = (int*)calloc((size_t)total * BK, sizeof(int));
    uint8_t* finished = (uint8_t*)calloc((size_t)BK, 1);
    int*     seq_len = (int*)calloc((size_t)BK, sizeof(int));
    int*     pad_cnt = (int*)calloc((size_t)BK, sizeof(int));
    int*     t_len   = (int*)calloc((size_t)BK, sizeof(int));
    uint8_t* masked  = (uint8_t*)calloc((size_t)BK * s_max, 1);
    float*   cum     = (float*)calloc((size_t)BK, sizeof(float));
    int*     indir[2];
    indir[0] = (int*)calloc((size_t)BK * s_max, sizeof(int));
    indir[1] = (int*)calloc((size_t)BK * s_max, sizeof(int));
    float* hid     = (float*)malloc(sizeof(float) * (size_t)BK * H);
    float* hid_out = (float*)malloc(sizeof(float) * (size_t)BK * H);
    float* nrm     = (float*)malloc(sizeof(float) * (size_t)BK * H);
    float* logits  = (float*)malloc(sizeof(float) * (size_t)BK * V);
    const int max_input_length = S;
    for (int bb = 0; bb < BK; bb++) {
        t_len[bb]    = input_lengths[bb / K]; /* invokeTileGptInputs */
        finished[bb] = 0;
        seq_len[bb]  = max_input_length - 1;
        cum[bb]      = (bb % K == 0) ? 0.f : -1e20f; /* decodingInitialize (decoding_kernels.cu:26-47) */
    }
    if (S > 1) {
        float* x = (float*)malloc(sizeof(float) * (size_t)BK * S * H);
        for (int bb = 0; bb < BK; bb++) {
            for (int s2 = 0; s2 < S; s2++) {
                const int id             = input_ids[(size_t)(bb / K) * S + s2];
                ids[(size_t)s2 * BK + bb] = id;
                memcpy(x + ((size_t)bb * S + s2) * H, w->wte + (size_t)id * H, sizeof(float) * H);
            }
        }
        context_decoder(c, w, x, k_cache, v_cache, t_len, BK, S, s_max);
        for (int bb = 0; bb < BK; bb++) {
            memcpy(hid_out + (size_t)bb * H, x + ((size_t)bb * S + (t_len[bb] - 1)) * H, sizeof(float) * H);
        }
        free(x);
    }
    else {
        for (int bb = 0; bb < BK; bb++) {
            ids[bb] = input_ids[bb / K];
        }
    }
    for (int bb = 0; bb < BK; bb++) {
        for (int s2 = t_len[bb]; s2 < max_input_length; s2++) {
            masked[(size_t)bb * s_max + s2] = 1;
        }
    }
    int steps_run = 0;
    for (int step = max_input_length; step < total; step++) {
        const int src = (step - max_input_length) % 2, tgt = 1 - src; /* GptNeoX.cc:778-780 */
        if (!(max_input_length > 1 && step == max_input_length)) {
            for (int bb = 0; bb < BK; bb++) {
                const int id = ids[(size_t)(step - 1) * BK + bb];
                memcpy(hid + (size_t)bb * H, w->wte + (size_t)id * H, sizeof(float) * H);
            }
            orc_decoder_step_beam(c, w, hid, k_cache, v_cache, seq_len, pad_cnt, masked, finished, BK, s_max, step, hid_out,
                                  indir[src], K);
        }
        orc_layernorm(hid_out, w->final_ln_g, w->final_ln_b, BK, H, 1e-5f, nrm, fp16);
        orc_lm_head(nrm, BK, H, V, w->lm_head, logits);
        orc_beam_search_step(logits, B, K, V, step, max_input_length, t_len, bp, c->end_id, ids, parents, finished, seq_len,
                             cum, indir[src], indir[tgt], s_max);
        int all = 1;
        for (int bb = 0; bb < BK; bb++) {
            all &= finished[bb];
        }
        steps_run++;
        if (all) {
            break;
        }
        if (step == max_input_length) {
            for (int bb = 0; bb < BK; bb++) {
                pad_cnt[bb] += max_input_length - t_len[bb];
            }
        }
    }
    orc_gather_tree_beam(ids, parents, seq_len, t_len, B, K, max_input_length, total, c->end_id, output_ids,
                         sequence_lengths);
    if (cum_log_probs_out) {
        memcpy(cum_log_probs_out, cum, sizeof(float) * (size_t)BK);
    }
    free(k_cache);
    free(v_cache);
    free(ids);
    free(parents);
    free(finished);
    free(seq_len);
    free(pad_cnt);
    free(t_len);
    free(masked);
    free(cum);
    free(indir[0]);
    free(indir[1]);
    free(hid);
    free(hid_out);
    free(nrm);
    free(logits);
    return steps_run;
}
