// Device-side decoder attention (split-KV + in-launch merge), shared by kernels_attn.hip and kernels_fused.hip.
#pragma once
#include "ftcf_common.h"
#include "kernels.h"

namespace ftcf {

__device__ __forceinline__ float group_sum(float v, int lanes_per_key)
{
    for (int o = lanes_per_key >> 1; o >= 1; o >>= 1) {
        v += __shfl_xor(v, o, 64);
    }
    return v;
}

// the same sum inside groups of 8 or 16 lanes on the VALU's data-parallel-primitive path: no LDS crossbar round trips
// (__shfl_xor is a ds_bpermute: ~64 cycles each, four in a row per key block).  Every lane of the group gets the sum.
template<int CTRL>
__device__ __forceinline__ float dpp_read(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template<int LPK>
__device__ __forceinline__ float group_sum_dpp(float v)
{
    static_assert(LPK == 8 || LPK == 16, "8 or 16 lanes per key");
    v += dpp_read<0xB1>(v);  // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_read<0x4E>(v);  // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_read<0x141>(v);  // row_half_mirror: the other quad of the eight
    if constexpr (LPK == 16) {
        v += dpp_read<0x140>(v);  // row_mirror: the other eight of the row
    }
    return v;
}

// wave-wide max / sum without the LDS crossbar: four DPP steps inside every row of 16 lanes, then the four row results
// through readlane (uniform result)
__device__ __forceinline__ float wave_max_dpp(float v)
{
    v = fmaxf(v, dpp_read<0xB1>(v));
    v = fmaxf(v, dpp_read<0x4E>(v));
    v = fmaxf(v, dpp_read<0x141>(v));
    v = fmaxf(v, dpp_read<0x140>(v));
    const int   b  = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v = group_sum_dpp<16>(v);
    const int   b  = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// the value of lane + 1 (same row of 16 lanes; the last lane of a row reads 0): packs pairs of halves without ds_bpermute
__device__ __forceinline__ unsigned next_lane_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);  // row_shl:1
}
// v + (the same lane of the row 16 / 32 lanes away), on gfx950's lane-swap instructions (VALU, no LDS crossbar).  The
// instruction swaps its two operands in place (odd rows of the first with even rows of the second / upper half of the
// first with lower half of the second); with ONE value in both, hipcc's builtin adds the first result to itself (measured,
// ROCm 7.2), so the statement is written out: two copies in, both out, two wait states after the last VALU write.
__device__ __forceinline__ float xor16_sum(float v)
{
    float a = v, c = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(c));  // {rows 0 0 2 2}, {rows 1 1 3 3}
    return a + c;
}
__device__ __forceinline__ float xor32_sum(float v)
{
    float a = v, c = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(c));  // {lower half twice}, {upper half twice}
    return a + c;
}
// sum over the key groups of a wave: lanes l, l + LPK, l + 2 LPK, ... (every lane gets it)
template<int LPK>
__device__ __forceinline__ float across_groups_sum(float v)
{
    if constexpr (LPK == 8) {
        v += dpp_read<0x128>(v);  // row_ror:8
    }
    return xor32_sum(xor16_sum(v));
}

// rotary coefficient exactly as the reference computes it (decoder_masked_multihead_attention_utils.h:1325-1329):
// inv_freq = t / 10000^(2j/rot) ; {cos, sin}(inv_freq) in fp32
__device__ __forceinline__ void rotary_coef(int j, int rot, int pos, float& cs, float& sn)
{
    const float inv_freq = (float)pos / powf(10000.0f, (float)(2 * j) / (float)rot);
    cs = cosf(inv_freq);
    sn = sinf(inv_freq);
}
__device__ __forceinline__ void rotary_apply(f16& a, f16& b, const float cs, const float sn)
{
    const float fa = (float)a, fb = (float)b;
    a = (f16)(cs * fa - sn * fb);
    b = (f16)(cs * fb + sn * fa);
}
__device__ __forceinline__ void rotary_pair(f16& a, f16& b, int j, int rot, int pos)
{
    float cs, sn;
    rotary_coef(j, rot, pos, cs, sn);
    rotary_apply(a, b, cs, sn);
}

// In-launch hand-off of the split-KV partials uses 8-byte {tag, value} GRANULES written with ONE relaxed agent-scope
// (sc1, write-through) store each and polled with relaxed agent-scope loads: the data is its own flag, so there is no
// drain / ticket / fence round trip on the latency chain (cdna_hip_programming.md G16 recipe R2, MI355X_MICROARCH.md
// row "handoff-1to1").  tag = f(step, layer) + 1 is unique per launch; the slab is zeroed when a request begins.
typedef unsigned long long u64;
// (granule slabs always live in global memory: the explicit address space keeps these from becoming FLAT accesses, which
// count on lgkmcnt as well and serialise with LDS waits, when the pointer comes out of a struct)
typedef __attribute__((address_space(1))) u64 gu64;
__device__ __forceinline__ void st_granule(u64* g, unsigned tag, float v)
{
    __hip_atomic_store((gu64*)g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_granule(const u64* g)
{
    return __hip_atomic_load((const gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int MMHA_MAX_SPLIT = 16;
constexpr int MMHA_SPIN_LIMIT = 1 << 22;  // bounded: a protocol bug must not hang the GPU

// Split `sp` of (row b, head h): computes the un-normalised partial (max, sum, out[DH]) over its key range and
// publishes it as granules.  Latency chain: ONE round trip for {finished, seq_len, step, q, bias, K rows, V rows}
// (the K/V rows of the whole fixed chunk are requested before tlength is known and masked afterwards), then math.
// BEAMS (beam search, beam_width > 1): the key / value of time t < tlength is read from the row the cache indirection
// names, (b / beam_width) * beam_width + cache_indir[b][t] (decoder_masked_multihead_attention_template.hpp:1290-1296,
// :1509-1512, :1733-1736); the new key / value always goes to the row's own cache line.
template<int DH, bool BEAMS = false>
__device__ __forceinline__ bool mmha_partial(const MmhaParams& p, char* smem, u64* gout, const unsigned tag, int h,
                                             int b, int sp)
{
    // lanes per key/value row (16 B each): DH / 8 of them carry data, rounded up to a power of two so that the groups tile a
    // wave -- the reference's head sizes 48, 80, 96, 144 ... 224 (DecoderSelfAttentionLayer.cc:280-282) leave lanes of a
    // group idle (they re-read the group's first piece and contribute zeros)
    static_assert(DH % 8 == 0 && DH >= 32 && DH <= 256, "size_per_head: a multiple of 8 in [32, 256]");
    constexpr int NSUB = DH / 8;
    constexpr int LPK  = NSUB <= 4 ? 4 : NSUB <= 8 ? 8 : NSUB <= 16 ? 16 : 32;
    constexpr int KPI  = 64 / LPK;  // rows per wave-load
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int  grp = lane / LPK;
    const bool act = (lane % LPK) < NSUB;
    const int  sub = act ? lane % LPK : 0;
    const int chunk = (((p.s_max + p.nsplit - 1) / p.nsplit) + 15) & ~15;
    const int t_beg = sp * chunk;
    const f16* kc = p.k_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
    const f16* vc = p.v_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
    const int*      indir   = nullptr;
    const ptrdiff_t row_kv  = (ptrdiff_t)p.nh * p.s_max * DH;
    int             b_first = 0;
    if constexpr (BEAMS) {
        const int step0 = p.d_step ? *p.d_step : p.step;
        // source plane of this step: (step - max_input_len) % 2 (GptNeoX.cc:778-780)
        indir   = p.cache_indir + (size_t)((step0 - p.max_input_len) & 1) * p.indir_plane + (size_t)b * p.s_max;
        b_first = b / p.beam_width * p.beam_width;
    }
    auto row_of = [&](const f16* own, const int t) -> const f16* {
        if constexpr (BEAMS) {
            return own + (ptrdiff_t)(b_first + indir[t] - b) * row_kv + (size_t)t * DH;
        }
        else {
            return own + (size_t)t * DH;
        }
    };
    constexpr int UK   = 8;
    const bool    fast = (chunk <= 4 * KPI * UK);
    u32x4         kreg[UK], vreg[UK];
    if (fast) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t   = t_beg + u * 4 * KPI + wid * KPI + grp;
            t       = t < p.s_max ? t : p.s_max - 1;  // always inside the cache; rows >= tlength are masked below
            kreg[u] = *reinterpret_cast<const u32x4*>(row_of(kc, t) + sub * 8);
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t   = t_beg + u * 4 * KPI + wid * KPI + grp;
            t       = t < p.s_max ? t : p.s_max - 1;
            vreg[u] = *reinterpret_cast<const u32x4*>(row_of(vc, t) + sub * 8);
        }
    }
    const int  hl = p.nh * DH;
    f16        q_in = (f16)0.f, k_in = (f16)0.f, v_in = (f16)0.f;
    if (threadIdx.x < DH) {
        const int    d    = threadIdx.x;
        const size_t base = (size_t)b * 3 * hl + h * DH + d;
        q_in = p.qkv[base] + (p.qkv_bias ? p.qkv_bias[h * DH + d] : (f16)0.f);
        k_in = p.qkv[base + hl] + (p.qkv_bias ? p.qkv_bias[hl + h * DH + d] : (f16)0.f);
        v_in = p.qkv[base + 2 * hl] + (p.qkv_bias ? p.qkv_bias[2 * hl + h * DH + d] : (f16)0.f);
    }
    float rot_cs = 1.f, rot_sn = 0.f;
    if (p.rot_table && threadIdx.x < p.rot / 2) {
        rot_cs = p.rot_table[((size_t)b * (p.rot / 2) + threadIdx.x) * 2];
        rot_sn = p.rot_table[((size_t)b * (p.rot / 2) + threadIdx.x) * 2 + 1];
    }
    // padding mask of the rows this lane scores, fetched with the same round trip (not inside the qk loop)
    unsigned mask_bits = 0u;
    if (fast && p.masked_tokens && (lane % LPK) == 0) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t = t_beg + u * 4 * KPI + wid * KPI + grp;
            t     = t < p.s_max ? t : p.s_max - 1;
            mask_bits |= (p.masked_tokens[(size_t)b * p.s_max + t] ? 1u : 0u) << u;
        }
    }
    const bool fin  = p.finished && p.finished[b];
    const int  tl   = p.seq_len[b];  // tlength: number of cached keys; the new token goes to index tl
    const int  step = p.d_step ? *p.d_step : p.step;
    const int  pos  = (step - 1) - (p.pad_count ? p.pad_count[b] : 0);  // :1303,:1343-1344
    __builtin_amdgcn_sched_barrier(0);
    if (fin) {
        return false;  // :1176 (ctx of a finished row is never consumed); uniform for all splits of the row
    }
    int t_end = t_beg + chunk;  // exclusive, over positions 0..tl (tl = current token)
    if (t_end > tl + 1) {
        t_end = tl + 1;
    }
    if (t_beg > tl) {  // empty split
        if (threadIdx.x < DH) {
            st_granule(&gout[threadIdx.x], tag, 0.f);
        }
        if (threadIdx.x == 0) {
            st_granule(&gout[DH], tag, -INFINITY);
            st_granule(&gout[DH + 1], tag, 0.f);
        }
        return true;
    }
    const bool owns_cur     = (tl >= t_beg && tl < t_end);
    const int  t_cached_end = owns_cur ? tl : t_end;  // cached keys of this split: [t_beg, t_cached_end)

    f16*   s_q    = reinterpret_cast<f16*>(smem);       // [DH]
    f16*   s_k    = s_q + DH;                            // [DH] new key
    f16*   s_v    = s_k + DH;                            // [DH] new value
    float* s_red  = reinterpret_cast<float*>(s_v + DH);  // [8 + 4*DH]
    float* s_p    = s_red + 8 + 4 * DH;                  // [chunk]
    if (threadIdx.x < DH) {
        s_q[threadIdx.x] = q_in;
        s_k[threadIdx.x] = k_in;
        s_v[threadIdx.x] = v_in;
    }
    __syncthreads();
    if (p.rot > 0 && threadIdx.x < p.rot / 2) {
        const int j = threadIdx.x;
        float     cs, sn;
        if (p.rot_table) {  // {cos, sin} of this step's position, computed once per token (k_rotary_table)
            cs = rot_cs;
            sn = rot_sn;
        }
        else {
            rotary_coef(j, p.rot, pos, cs, sn);
        }
        f16 a = s_q[j], c = s_q[j + p.rot / 2];
        rotary_apply(a, c, cs, sn);
        s_q[j]             = a;
        s_q[j + p.rot / 2] = c;
        if (owns_cur) {
            f16 ka = s_k[j], kc2 = s_k[j + p.rot / 2];
            rotary_apply(ka, kc2, cs, sn);
            s_k[j]             = ka;
            s_k[j + p.rot / 2] = kc2;
        }
    }
    __syncthreads();
    if (owns_cur && threadIdx.x < DH) {  // append to the cache (:1397, :1837)
        p.k_cache[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + threadIdx.x] = s_k[threadIdx.x];
        p.v_cache[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + threadIdx.x] = s_v[threadIdx.x];
    }

    const float inv_sqrt_dh = rsqrtf((float)DH);  // DecoderSelfAttentionLayer.cc:118 with q_scaling 1
    f16x8       qv  = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
    if (!act) {
        qv = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const bool     sub0 = (lane % LPK) == 0;
    const uint8_t* mask = p.masked_tokens ? p.masked_tokens + (size_t)b * p.s_max : nullptr;

    // ---- phase 1: qk for the cached keys (fp32 accumulate, MMHA_USE_FP32_ACUM_FOR_FMA) ----
    float lmax = -INFINITY;
    // looped form (key ranges beyond the all-in-registers one): rounds of U rows per lane, two rounds in flight (the loads
    // of round i + 1 are requested before round i is consumed; clamped, never conditional)
    constexpr int U = 8;
    auto stream_rows = [&](const f16* base, auto&& consume) {
        constexpr int STEP = 4 * KPI * U;
        int           t0   = t_beg + wid * KPI;
        if (t0 >= t_cached_end) {
            return;
        }
        u32x4 ra[U], rb[U];
        auto  ld = [&](u32x4 (&r)[U], const int tt) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                int t = tt + u * 4 * KPI + grp;
                t     = t < t_cached_end ? t : t_cached_end - 1;
                r[u]  = *reinterpret_cast<const u32x4*>(row_of(base, t) + sub * 8);
            }
        };
        ld(ra, t0);
        for (;;) {
            ld(rb, t0 + STEP);
#pragma unroll
            for (int u = 0; u < U; u++) {
                consume(ra[u], t0 + u * 4 * KPI + grp);
            }
            t0 += STEP;
            if (t0 >= t_cached_end) {
                break;
            }
            ld(ra, t0 + STEP);
#pragma unroll
            for (int u = 0; u < U; u++) {
                consume(rb[u], t0 + u * 4 * KPI + grp);
            }
            t0 += STEP;
            if (t0 >= t_cached_end) {
                break;
            }
        }
    };
    auto qk_one = [&](const u32x4 raw, const int t, const int u_fast) {
        const f16x8 kv = __builtin_bit_cast(f16x8, raw);
        float       a  = 0.f;
        a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
        a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
        a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
        a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
        a              = group_sum(a, LPK) * inv_sqrt_dh;
        if (t < t_cached_end && sub0) {
            const bool m = (u_fast >= 0) ? ((mask_bits >> u_fast) & 1u) != 0u : (mask && mask[t]);
            s_p[t - t_beg] = m ? -INFINITY : a;  // masked keys get probability 0 (:1570,:1610-1622)
            if (!m) {
                lmax = fmaxf(lmax, a);
            }
        }
    };
    if (fast) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            qk_one(kreg[u], t_beg + u * 4 * KPI + wid * KPI + grp, u);
        }
    }
    else {
        stream_rows(kc, [&](const u32x4 raw, const int t) { qk_one(raw, t, -1); });
    }
    if (owns_cur && wid == 0) {  // current token from LDS (:1407-1437)
        float a = 0.f;
        if (lane < NSUB) {
            const f16x8 kv = *reinterpret_cast<const f16x8*>(s_k + lane * 8);
            const f16x8 q8 = *reinterpret_cast<const f16x8*>(s_q + lane * 8);
            a              = dot2(f16x2{q8[0], q8[1]}, f16x2{kv[0], kv[1]}, a);
            a              = dot2(f16x2{q8[2], q8[3]}, f16x2{kv[2], kv[3]}, a);
            a              = dot2(f16x2{q8[4], q8[5]}, f16x2{kv[4], kv[5]}, a);
            a              = dot2(f16x2{q8[6], q8[7]}, f16x2{kv[6], kv[7]}, a);
        }
        a = wave_sum(a) * inv_sqrt_dh;
        if (lane == 0) {
            s_p[tl - t_beg] = a;
            lmax            = fmaxf(lmax, a);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) {
        s_red[wid] = lmax;
    }
    __syncthreads();
    const float m_loc = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    // ---- phase 2: exp, local sum ----
    float lsum = 0.f;
    for (int i = threadIdx.x; i < t_end - t_beg; i += 256) {
        const float e = (s_p[i] == -INFINITY) ? 0.f : __expf(s_p[i] - m_loc);
        s_p[i]        = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();  // s_red reuse + s_p visible
    if (lane == 0) {
        s_red[4 + wid] = lsum;
    }
    // ---- phase 3: P.V (fp32 accumulate) ----
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        acc[j] = 0.f;
    }
    auto pv_one = [&](const u32x4 raw, const int t) {
        if (t < t_cached_end) {  // rows beyond tlength were fetched speculatively and may hold anything (NaN bits)
            const float pt = s_p[t - t_beg];
            const f16x8 vv = __builtin_bit_cast(f16x8, raw);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                acc[j] = fmaf(pt, (float)vv[j], acc[j]);
            }
        }
    };
    if (fast) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            pv_one(vreg[u], t_beg + u * 4 * KPI + wid * KPI + grp);
        }
    }
    else {
        stream_rows(vc, [&](const u32x4 raw, const int t) { pv_one(raw, t); });
    }
    if (owns_cur && wid == 0 && grp == 0 && act) {
        const float pt = s_p[tl - t_beg];
        const f16x8 vv = *reinterpret_cast<const f16x8*>(s_v + sub * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = fmaf(pt, (float)vv[j], acc[j]);
        }
    }
    // fold the KPI row groups of the wave, then the 4 waves
#pragma unroll
    for (int j = 0; j < 8; j++) {
        for (int o = LPK; o < 64; o <<= 1) {
            acc[j] += __shfl_xor(acc[j], o, 64);
        }
    }
    float* s_o = s_red + 8;  // [4][DH]
    if (grp == 0 && act) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s_o[wid * DH + sub * 8 + j] = acc[j];
        }
    }
    __syncthreads();
    if (p.nsplit == 1) {
        // a single split: nothing to merge -- normalise and write ctx here (the same arithmetic as the merger with one
        // partial of weight exp(0) = 1) instead of publishing granules and polling them back (a memory round trip)
        if (threadIdx.x < DH) {
            const int   d   = threadIdx.x;
            const float o   = (s_o[d] + s_o[DH + d]) + (s_o[2 * DH + d] + s_o[3 * DH + d]);
            const float L   = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
            const float w   = (m_loc == -INFINITY) ? 0.f : 1.f;
            const float inv = 1.f / (w * L + 1.e-6f);  // :1632
            p.ctx[(size_t)b * p.nh * DH + h * DH + d] = (f16)((w * o) * inv);
        }
        return false;
    }
    if (threadIdx.x < DH) {
        const int d = threadIdx.x;
        st_granule(&gout[d], tag, (s_o[d] + s_o[DH + d]) + (s_o[2 * DH + d] + s_o[3 * DH + d]));
    }
    if (threadIdx.x == 0) {
        st_granule(&gout[DH], tag, m_loc);
        st_granule(&gout[DH + 1], tag, (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    }
    return true;
}

// One launch: every split workgroup publishes its partial as granules; the LAST split's workgroup of each (row, head) then
// polls the nsplit partials (all loads of one pass in flight together), merges them in split order (deterministic)
// and writes ctx.  Only the merger ever waits, producers never do, and every launcher dispatches a (row, head)'s last split
// after its other splits: a merger never holds a slot that a producer it waits for still needs (with split 0 as the merger and
// more workgroups than slots -- 16 rows x 40 heads x 2 splits -- the first wave of workgroups spun until their limit:
// 10..80 ms per step).  Every spin is bounded.
template<int DH, bool BEAMS = false>
__device__ __forceinline__ void mmha_block(const MmhaParams& p, char* smem, int& s_last, const int h, const int b, const int sp)
{
    (void)s_last;
    const int      step = p.d_step ? *p.d_step : p.step;
    const unsigned tag  = (unsigned)(step * 1024 + p.layer) + 1u;  // salt < 1024: layer + row group * num_layer
    u64*           gall = p.gran + ((size_t)b * p.nh + h) * p.nsplit * (DH + 2);
    const bool     live = mmha_partial<DH, BEAMS>(p, smem, gall + (size_t)sp * (DH + 2), tag, h, b, sp);
    if (!live || sp != p.nsplit - 1) {
        return;
    }
    // ---- merger (the last split) ----
    __syncthreads();
    float* sval = reinterpret_cast<float*>(smem);  // [nsplit][DH+2], then [nsplit] weights + [1] denominator
    const int ne = DH + 2;
    for (int i = threadIdx.x; i < ne; i += blockDim.x) {
        u64 gv[MMHA_MAX_SPLIT];
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int s2 = 0; s2 < MMHA_MAX_SPLIT; s2++) {
                if (s2 < p.nsplit) {
                    gv[s2] = ld_granule(&gall[(size_t)s2 * ne + i]);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < MMHA_MAX_SPLIT; s2++) {
                if (s2 < p.nsplit) {
                    ok &= ((unsigned)(gv[s2] >> 32) == tag);
                }
            }
            if (ok || ++spins > MMHA_SPIN_LIMIT) {
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int s2 = 0; s2 < MMHA_MAX_SPLIT; s2++) {
            if (s2 < p.nsplit) {
                sval[s2 * ne + i] = __uint_as_float((unsigned)gv[s2]);
            }
        }
    }
    __syncthreads();
    float* sw = sval + p.nsplit * ne;
    if (threadIdx.x < 64) {
        float ms = -INFINITY, ls = 0.f;
        if ((int)threadIdx.x < p.nsplit) {
            ms = sval[threadIdx.x * ne + DH];
            ls = sval[threadIdx.x * ne + DH + 1];
        }
        const float m = wave_max(ms);
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - m);
        if ((int)threadIdx.x < p.nsplit) {
            sw[threadIdx.x] = w;
        }
        float L = 0.f;  // fixed-order sum over splits
        for (int s2 = 0; s2 < p.nsplit; s2++) {
            L += __shfl(w * ls, s2, 64);
        }
        if (threadIdx.x == 0) {
            sw[p.nsplit] = L;
        }
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const int d = threadIdx.x;
        float     o = 0.f;
        for (int s2 = 0; s2 < p.nsplit; s2++) {
            o += sw[s2] * sval[s2 * ne + d];
        }
        const float inv = 1.f / (sw[p.nsplit] + 1.e-6f);  // :1632
        p.ctx[(size_t)b * p.nh * DH + h * DH + d] = (f16)(o * inv);
    }
}

}  // namespace ftcf
