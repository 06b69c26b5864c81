// Device-side building blocks of the weight-streaming GEMV kernels (shared by kernels_gemv.hip and kernels_fused.hip).
#pragma once
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

constexpr int GEMV_U = 8;  // tiles per batch (x2 batches in flight)
// The rows of x staged in LDS are XPAD halves (16 B) further apart than their length: K is a multiple of 64 halves, so
// unpadded rows start in the same bank and the A-fragment reads of m rows conflict m ways (measured: the m = 4 GEMV
// step took 1.7x the m = 1 step although the MFMA work is identical)
constexpr int XPAD = 8;

template<bool INT8>
struct TileK {
    static constexpr int value = INT8 ? TILE_K_I8 : TILE_K_F16;
};

// Consumes one 16-byte weight fragment against the M (<= 16) rows of x held in LDS -- on the MATRIX pipe.
// The lane's 16 bytes are exactly its B fragment(s) of v_mfma_f32_16x16x32_f16 (column lane&15, k group lane>>4); the
// A fragment is x[row][same k order] with row = min(lane&15, M-1) (rows >= M are duplicates whose results are never
// read).  D[row][col] accumulates over every tile in one f32x4 per lane: lane (g, c) register j = row 4g+j, column c.
// Compared with the VALU dot2c form this removes 8 of 32 VALU instructions per tile-lane (and the slow dot2c ones),
// leaves only the exact u8->f16 dequant on the VALU, needs no cross-lane fold, and costs the same for 1..16 rows.
template<bool INT8, int M>
__device__ __forceinline__ void consume_tile(const u32x4 w, const f16* xr, const f16x2 scale2, f32x4& acc)
{
    if constexpr (INT8) {
        f16x2 d[8];
        dequant4(w.x, scale2, d[0], d[1]);
        dequant4(w.y, scale2, d[2], d[3]);
        dequant4(w.z, scale2, d[4], d[5]);
        dequant4(w.w, scale2, d[6], d[7]);
        const f16x8 b0 = {d[0][0], d[0][1], d[1][0], d[1][1], d[2][0], d[2][1], d[3][0], d[3][1]};
        const f16x8 b1 = {d[4][0], d[4][1], d[5][0], d[5][1], d[6][0], d[6][1], d[7][0], d[7][1]};
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(xr);
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(xr + 8);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc, 0, 0, 0);
    }
    else {
        const f16x8 b0 = __builtin_bit_cast(f16x8, w);
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(xr);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
    }
}

// LDS pointer of this lane's A fragments: row min(lane&15, M-1), k group lane>>4 of the first tile
template<bool INT8, int M>
__device__ __forceinline__ const f16* a_frag_ptr(const f16* xs_tile0, const int xstride, const int lane)
{
    const int r = (M == 1) ? 0 : (((lane & 15) < M) ? (lane & 15) : (M - 1));
    return xs_tile0 + (size_t)r * xstride + (lane >> 4) * (INT8 ? 16 : 8);
}

// value of output row m, column lane&15 (valid on lanes 0..15 for m < 4)
__device__ __forceinline__ float acc_row(const f32x4& acc, const int m)
{
    return m == 0 ? acc[0] : (m == 1 ? acc[1] : (m == 2 ? acc[2] : acc[3]));
}

// Streams `ntiles` consecutive tiles of one column group.  wp: this lane's 16 B of the first tile.
// xl: LDS pointer to this lane's k offset of the first tile (row 0); rows are xstride halves apart.
// Two register batches (A/B) of GEMV_U tiles ping-pong WITHOUT register copies, so that the batch being consumed only
// waits for its own loads (counted vmcnt) while the other batch stays in flight.
template<bool INT8, int M>
struct WaveStream {
    u32x4 A[GEMV_U], B[GEMV_U];

    __device__ __forceinline__ void load(u32x4 (&r)[GEMV_U], const u32x4* __restrict__ wp, int batch)
    {
#pragma unroll
        for (int u = 0; u < GEMV_U; u++) {
            r[u] = __builtin_nontemporal_load(wp + (size_t)(batch * GEMV_U + u) * 64);
        }
    }
    __device__ __forceinline__ void consume(const u32x4 (&r)[GEMV_U], int batch, const f16* xr, const f16x2 scale2,
                                            f32x4& acc)
    {
        constexpr int TK = TileK<INT8>::value;
#pragma unroll
        for (int u = 0; u < GEMV_U; u++) {
            consume_tile<INT8, M>(r[u], xr + (batch * GEMV_U + u) * TK, scale2, acc);
        }
    }
    // issue the first batch early (before a prologue that does not depend on the weights)
    __device__ __forceinline__ void prime(const u32x4* __restrict__ wp, int ntiles)
    {
        if (ntiles >= GEMV_U) {
            load(A, wp, 0);
        }
    }
    // xr: this lane's A-fragment pointer for the first tile (a_frag_ptr)
    __device__ __forceinline__ void run(const u32x4* __restrict__ wp, int ntiles, const f16* xr, const f16x2 scale2,
                                        f32x4& acc)
    {
        constexpr int TK = TileK<INT8>::value;
        const int     nb = ntiles / GEMV_U;
        int           b  = 0;
        while (b + 2 <= nb) {
            // sched_barrier: hipcc's scheduler otherwise sinks each load batch below the preceding consume (to save
            // registers), which serialises load and compute -- the pipeline must keep one batch in flight
            load(B, wp, b + 1);
            __builtin_amdgcn_sched_barrier(0);
            consume(A, b, xr, scale2, acc);
            __builtin_amdgcn_sched_barrier(0);
            // unconditional (clamped) reload keeps the loop body branch free; the last one is a harmless re-read
            load(A, wp, (b + 2 < nb) ? b + 2 : nb - 1);
            __builtin_amdgcn_sched_barrier(0);
            consume(B, b + 1, xr, scale2, acc);
            __builtin_amdgcn_sched_barrier(0);
            b += 2;
        }
        if (b < nb) {
            consume(A, b, xr, scale2, acc);
            b++;
        }
        for (int t = nb * GEMV_U; t < ntiles; t++) {
            const u32x4 w = __builtin_nontemporal_load(wp + (size_t)t * 64);
            consume_tile<INT8, M>(w, xr + t * TK, scale2, acc);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// K_A: y0 = LN1(x) * W0        (QKV, no bias: the attention kernel adds it like the reference's MMHA)
//      y1 = gelu(LN2(x) * W1 + b1)   (FFN first projection, fused epilogue of gemm_bias_act)
// One launch, one wave per 16-column group over the full K extent; the block recomputes the LayerNorm of the
// (tiny, L2 resident) layer input instead of paying a kernel boundary for it.
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int M>
__device__ __forceinline__ void ln_gemv_block(const LnGemvParams& p, char* smem, const int block_id)
{
    f16*   xs  = reinterpret_cast<f16*>(smem);                      // [M][K]
    float* red = reinterpret_cast<float*>(smem + (size_t)M * (p.K + XPAD) * 2);  // 2*4 floats

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int seg  = (block_id >= p.blocks0) ? 1 : 0;
    const int grp  = (seg ? (block_id - p.blocks0) : block_id) * 4 + wid;  // column group in segment
    const int NT   = seg ? p.NT1 : p.NT0;
    const int K    = p.K;
    constexpr int TK = TileK<INT8>::value;
    const int     KT = K / TK;
    const int     c = lane & 15, g = lane >> 4;
    const bool    active = grp < NT;
    const char*   wbase  = reinterpret_cast<const char*>(seg ? p.W1 : p.W0);
    const u32x4*  wp     = reinterpret_cast<const u32x4*>(wbase + ((size_t)(active ? grp : 0) * KT * 64 + lane) * 16);

    const f16* gamma = seg ? p.gamma1 : p.gamma0;
    const f16* beta  = seg ? p.beta1 : p.beta0;
    WaveStream<INT8, M> ws;
    constexpr int XV = 4;  // register-resident LayerNorm for K <= 8192 (single-row decode: the bs=1 hot path)
    if (M == 1 && K <= 2048 * XV) {
        // ---- all global loads first: x, gamma, beta (L2 hits), THEN the first weight batch; the LayerNorm math
        //      below runs while the weights are in flight (counted vmcnt: the x loads are the oldest) ----
        f16x8 xv[M][XV], gv[XV], bv[XV];
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int  i  = threadIdx.x * 8 + j * 2048;
            const bool ok = i < K;
#pragma unroll
            for (int m = 0; m < M; m++) {
                xv[m][j] = ok ? *reinterpret_cast<const f16x8*>(p.x + (size_t)m * K + i) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            gv[j] = ok ? *reinterpret_cast<const f16x8*>(gamma + i) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            bv[j] = ok ? *reinterpret_cast<const f16x8*>(beta + i) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (active) {
            ws.prime(wp, KT);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- LayerNorm: fp16 half2-path numerics of layernorm_kernels.cu:157-286 ----
#pragma unroll
        for (int m = 0; m < M; m++) {
            float s[2] = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < XV; j++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float f = (float)xv[m][j][e];
                    s[0] += f;
                    s[1] += f * f;
                }
            }
            block_sum<2>(s, red);
            const float mean = s[0] / (float)K;
            const float rstd = rsqrtf(s[1] / (float)K - mean * mean + p.eps);
            const f16   mh = (f16)mean, rh = (f16)rstd;
#pragma unroll
            for (int j = 0; j < XV; j++) {
                const int i = threadIdx.x * 8 + j * 2048;
                if (i < K) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        o[e] = (((xv[m][j][e] - mh) * rh) * gv[j][e]) + bv[j][e];
                    }
                    *reinterpret_cast<f16x8*>(xs + (size_t)m * (K + XPAD) + i) = o;
                }
            }
        }
    }
    else {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const f16* xr = p.x + (size_t)m * K;
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
            const float rstd = rsqrtf(s[1] / (float)K - mean * mean + p.eps);
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
                *reinterpret_cast<f16x8*>(xs + (size_t)m * (K + XPAD) + i) = o;
            }
        }
        if (active) {
            ws.prime(wp, KT);
        }
    }
    __syncthreads();
    if (!active) {
        return;
    }
    const int n      = grp * 16 + c;
    f16x2     scale2 = {(f16)1.0f, (f16)1.0f};
    if constexpr (INT8) {
        const f16 sc = (seg ? p.scale1 : p.scale0)[n];
        scale2       = f16x2{sc, sc};
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    ws.run(wp, KT, a_frag_ptr<INT8, M>(xs, K + XPAD, lane), scale2, acc);
    if (g == 0) {
        f16*      out = seg ? p.out1 : p.out0;
        const int N   = NT * 16;
#pragma unroll
        for (int m = 0; m < M; m++) {
            float v = acc_row(acc, m);
            if (seg == 1) {
                if constexpr (INT8) {
                    // fused epilogue in fp32 (epilogue_helpers.h:52-62): bias + gelu, one rounding
                    v      = gelu_f32(v + (float)p.bias1[n]);
                    out[(size_t)m * N + n] = (f16)v;
                }
                else {
                    // fp16 engine: GEMM rounds to half, then invokeAddBiasGeluV2 in half (activation_kernels.cu:401-426)
                    f16 h                  = (f16)v + p.bias1[n];
                    out[(size_t)m * N + n] = gelu_f16(h);
                }
            }
            else {
                out[(size_t)m * N + n] = (f16)v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Balanced variant of K_A: ONE 16-column group per workgroup, its waves split K (LDS reduce in wave order).
// Workgroup counts: QKV 960, FFN1 1280 at CodeFuse-13B -> 3.75 / 5 per CU instead of the 0.94 / 1.25 of the
// 4-groups-per-workgroup form, whose CUs with one extra workgroup set the kernel time (measured +30 %).
// Segment 0 blocks are [0, NT0), segment 1 blocks [NT0, NT0 + NT1).
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int M>
__device__ __forceinline__ void ln_gemv_group_block(const LnGemvParams& p, char* smem, const int block_id,
                                                    const int gpb = 1)
{
    // gpb column groups per workgroup (the waves are dealt to the groups evenly; NT0 must be a multiple of gpb)
    const int nw   = blockDim.x >> 6;
    const int wpg  = nw / gpb;
    f16*      xs   = reinterpret_cast<f16*>(smem);                          // [M][K]
    float*    red  = reinterpret_cast<float*>(smem + (size_t)M * (p.K + XPAD) * 2);  // 2*nw floats (LN), then [nw][M][16]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gsel = wid / wpg, kw = wid - gsel * wpg;
    const int gi   = block_id * gpb + gsel;
    const int seg  = (gi >= p.NT0) ? 1 : 0;
    const int grp  = seg ? gi - p.NT0 : gi;
    const int K    = p.K;
    constexpr int TK = TileK<INT8>::value;
    const int     KT = K / TK;
    const int     c = lane & 15, g = lane >> 4;
    const int     t0 = (int)((long)KT * kw / wpg), t1 = (int)((long)KT * (kw + 1) / wpg);
    const int     nt = t1 - t0;
    const char*   wbase = reinterpret_cast<const char*>(seg ? p.W1 : p.W0);
    const u32x4*  wp    = reinterpret_cast<const u32x4*>(wbase + (((size_t)grp * KT + t0) * 64 + lane) * 16);
    WaveStream<INT8, M> ws;
    const f16* gamma = seg ? p.gamma1 : p.gamma0;
    const f16* beta  = seg ? p.beta1 : p.beta0;
    const int  nthr  = blockDim.x;
    constexpr int XV = 5;  // register-resident LayerNorm: K <= 40 * nthr (5120 at 128 threads); larger K takes the loop path
    if (M == 1 && K <= nthr * 8 * XV) {
        // x, gamma, beta (L2 hits) are requested first, then the first weight batch; the LayerNorm math runs while
        // the weights are in flight (counted vmcnt: the x loads are the oldest)
        f16x8 xv[XV], gv[XV], bv[XV];
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int  i  = (threadIdx.x + j * nthr) * 8;
            const bool ok = i < K;
            xv[j] = ok ? *reinterpret_cast<const f16x8*>(p.x + i) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            gv[j] = ok ? *reinterpret_cast<const f16x8*>(gamma + i) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            bv[j] = ok ? *reinterpret_cast<const f16x8*>(beta + i) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        ws.prime(wp, nt);
        __builtin_amdgcn_sched_barrier(0);
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < XV; j++) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float f = (float)xv[j][e];
                s[0] += f;
                s[1] += f * f;
            }
        }
        block_sum<2>(s, red);
        const float mean = s[0] / (float)K;
        const float rstd = rsqrtf(s[1] / (float)K - mean * mean + p.eps);
        const f16   mh = (f16)mean, rh = (f16)rstd;
#pragma unroll
        for (int j = 0; j < XV; j++) {
            const int i = (threadIdx.x + j * nthr) * 8;
            if (i < K) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    o[e] = (((xv[j][e] - mh) * rh) * gv[j][e]) + bv[j][e];
                }
                *reinterpret_cast<f16x8*>(xs + i) = o;
            }
        }
    }
    else {
        ws.prime(wp, nt);  // weights first: the LayerNorm below runs under their HBM latency
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < M; m++) {
            const f16* xr   = p.x + (size_t)m * K;
            float      s[2] = {0.f, 0.f};
            for (int i = threadIdx.x * 8; i < K; i += nthr * 8) {
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
            const float rstd = rsqrtf(s[1] / (float)K - mean * mean + p.eps);
            const f16   mh = (f16)mean, rh = (f16)rstd;
            for (int i = threadIdx.x * 8; i < K; i += nthr * 8) {
                const f16x8 v  = *reinterpret_cast<const f16x8*>(xr + i);
                const f16x8 gg = *reinterpret_cast<const f16x8*>(gamma + i);
                const f16x8 bb = *reinterpret_cast<const f16x8*>(beta + i);
                f16x8       o;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    o[j] = (((v[j] - mh) * rh) * gg[j]) + bb[j];
                }
                *reinterpret_cast<f16x8*>(xs + (size_t)m * (K + XPAD) + i) = o;
            }
        }
    }
    __syncthreads();
    const int n      = grp * 16 + c;
    f16x2     scale2 = {(f16)1.0f, (f16)1.0f};
    if constexpr (INT8) {
        const f16 sc = (seg ? p.scale1 : p.scale0)[n];
        scale2       = f16x2{sc, sc};
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    ws.run(wp, nt, a_frag_ptr<INT8, M>(xs + (size_t)t0 * TK, K + XPAD, lane), scale2, acc);
    float* part = red + 2 * nw;  // [nw][M][16]
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            part[(wid * M + m) * 16 + c] = acc_row(acc, m);
        }
    }
    __syncthreads();
    if (threadIdx.x < gpb * M * 16) {
        const int gs = threadIdx.x / (M * 16), r = threadIdx.x - gs * (M * 16);
        const int m = r >> 4, cc = r & 15;
        const int go   = block_id * gpb + gs;
        const int oseg = (go >= p.NT0) ? 1 : 0;
        const int nn   = (oseg ? go - p.NT0 : go) * 16 + cc;
        float     v    = 0.f;
        for (int w = gs * wpg; w < (gs + 1) * wpg; w++) {
            v += part[(w * M + m) * 16 + cc];
        }
        f16*      out = oseg ? p.out1 : p.out0;
        const int N   = (oseg ? p.NT1 : p.NT0) * 16;
        if (oseg == 1) {
            if constexpr (INT8) {
                out[(size_t)m * N + nn] = (f16)gelu_f32(v + (float)p.bias1[nn]);  // epilogue_helpers.h:52-62
            }
            else {
                out[(size_t)m * N + nn] = gelu_f16((f16)v + p.bias1[nn]);  // activation_kernels.cu:401-426
            }
        }
        else {
            out[(size_t)m * N + nn] = (f16)v;
        }
    }
}

}  // namespace ftcf
