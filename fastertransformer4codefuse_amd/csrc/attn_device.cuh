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

// rotary coefficient exactly as the reference computes it (decoder_masked_multihead_attention_utils.h:1325-1329):
// inv_freq = t / 10000^(2j/rot) ; {cos, sin}(inv_freq) in fp32
__device__ __forceinline__ void rotary_pair(f16& a, f16& b, int j, int rot, int pos)
{
    const float inv_freq = (float)pos / powf(10000.0f, (float)(2 * j) / (float)rot);
    const float cs = cosf(inv_freq), sn = sinf(inv_freq);
    const float fa = (float)a, fb = (float)b;
    a = (f16)(cs * fa - sn * fb);
    b = (f16)(cs * fb + sn * fa);
}

// write-through (sc1) store / L1-bypassing load at agent scope: the partials are handed to another workgroup inside
// the launch, so they must not linger in this CU's L1 / this XCD's L2 write-back state (MI355X_MICROARCH.md,
// "Workgroup dispatch, XCD placement & inter-workgroup visibility": `sc1` stores + `sc1` loads on both sides).
__device__ __forceinline__ void st_agent(float* p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template<int DH>
__device__ __forceinline__ void mmha_partial(const MmhaParams& p, char* smem, float* wsout, int h, int b, int sp)
{
    constexpr int LPK = DH / 8;    // lanes per key/value row (16 B each)
    constexpr int KPI = 64 / LPK;  // rows per wave-load
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tl    = p.seq_len[b];  // tlength: number of cached keys; the new token goes to index tl
    const int chunk = (((p.s_max + p.nsplit - 1) / p.nsplit) + 15) & ~15;
    const int t_beg = sp * chunk;
    int       t_end = t_beg + chunk;  // exclusive, over positions 0..tl (tl = current token)
    if (t_end > tl + 1) {
        t_end = tl + 1;
    }
    if (t_beg > tl) {  // empty split
        if (threadIdx.x == 0) {
            st_agent(&wsout[DH], -INFINITY);
            st_agent(&wsout[DH + 1], 0.f);
        }
        if (threadIdx.x < DH) {
            st_agent(&wsout[threadIdx.x], 0.f);
        }
        return;
    }
    const bool owns_cur = (tl >= t_beg && tl < t_end);
    const int  sub = lane % LPK, grp = lane / LPK;
    const int  t_cached_end = owns_cur ? tl : t_end;  // cached keys of this split: [t_beg, t_cached_end)
    const f16* kc = p.k_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
    const f16* vc = p.v_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
    // Latency chain: when the split's keys fit in registers (<= 4 waves x KPI x UK rows) the K AND V rows are requested
    // up front, before the q/bias/rotary prologue -- one HBM round trip instead of two dependent ones.
    constexpr int UK   = 8;
    const bool    fast = (chunk <= 4 * KPI * UK) && (t_cached_end > t_beg);
    u32x4         kreg[UK], vreg[UK];
    if (fast) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t   = t_beg + u * 4 * KPI + wid * KPI + grp;
            t       = t < t_cached_end ? t : t_cached_end - 1;
            kreg[u] = *reinterpret_cast<const u32x4*>(kc + (size_t)t * DH + sub * 8);
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t   = t_beg + u * 4 * KPI + wid * KPI + grp;
            t       = t < t_cached_end ? t : t_cached_end - 1;
            vreg[u] = *reinterpret_cast<const u32x4*>(vc + (size_t)t * DH + sub * 8);
        }
    }

    f16*   s_q    = reinterpret_cast<f16*>(smem);       // [DH]
    f16*   s_k    = s_q + DH;                            // [DH] new key
    f16*   s_v    = s_k + DH;                            // [DH] new value
    float* s_red  = reinterpret_cast<float*>(s_v + DH);  // [8 + 4*DH]
    float* s_p    = s_red + 8 + 4 * DH;                  // [chunk]

    const int hl   = p.nh * DH;
    const int step = p.d_step ? *p.d_step : p.step;
    const int pos  = (step - 1) - (p.pad_count ? p.pad_count[b] : 0);  // :1303,:1343-1344
    // ---- q (+bias, rotary); new k/v for the split that owns the current position ----
    if (threadIdx.x < DH) {
        const int    d    = threadIdx.x;
        const size_t base = (size_t)b * 3 * hl + h * DH + d;
        const f16    bq   = p.qkv_bias ? p.qkv_bias[h * DH + d] : (f16)0.f;
        s_q[d]            = p.qkv[base] + bq;
        if (owns_cur) {
            const f16 bk = p.qkv_bias ? p.qkv_bias[hl + h * DH + d] : (f16)0.f;
            const f16 bv = p.qkv_bias ? p.qkv_bias[2 * hl + h * DH + d] : (f16)0.f;
            s_k[d]       = p.qkv[base + hl] + bk;
            s_v[d]       = p.qkv[base + 2 * hl] + bv;
        }
    }
    __syncthreads();
    if (p.rot > 0 && threadIdx.x < p.rot / 2) {
        const int j = threadIdx.x;
        f16       a = s_q[j], c = s_q[j + p.rot / 2];
        rotary_pair(a, c, j, p.rot, pos);
        s_q[j]             = a;
        s_q[j + p.rot / 2] = c;
        if (owns_cur) {
            f16 ka = s_k[j], kc = s_k[j + p.rot / 2];
            rotary_pair(ka, kc, j, p.rot, pos);
            s_k[j]             = ka;
            s_k[j + p.rot / 2] = kc;
        }
    }
    __syncthreads();
    if (owns_cur && threadIdx.x < DH) {  // append to the cache (:1397, :1837)
        p.k_cache[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + threadIdx.x] = s_k[threadIdx.x];
        p.v_cache[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + threadIdx.x] = s_v[threadIdx.x];
    }

    const float inv_sqrt_dh = rsqrtf((float)DH);  // DecoderSelfAttentionLayer.cc:118 with q_scaling 1
    const f16x8 qv  = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
    const uint8_t* mask = p.masked_tokens ? p.masked_tokens + (size_t)b * p.s_max : nullptr;

    // ---- phase 1: qk for the cached keys (fp32 accumulate, MMHA_USE_FP32_ACUM_FOR_FMA) ----
    float lmax = -INFINITY;
    constexpr int U = 4;
    auto qk_one = [&](const u32x4 raw, const int t) {
        const f16x8 kv = __builtin_bit_cast(f16x8, raw);
        float       a  = 0.f;
        a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
        a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
        a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
        a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
        a              = group_sum(a, LPK) * inv_sqrt_dh;
        if (t < t_cached_end && sub == 0) {
            const bool m = mask && mask[t];
            s_p[t - t_beg] = m ? -INFINITY : a;  // masked keys get probability 0 (:1570,:1610-1622)
            if (!m) {
                lmax = fmaxf(lmax, a);
            }
        }
    };
    if (fast) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            qk_one(kreg[u], t_beg + u * 4 * KPI + wid * KPI + grp);
        }
    }
    else {
        for (int t0 = t_beg + wid * KPI; t0 < t_cached_end; t0 += 4 * KPI * U) {
            u32x4 kr[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                int t = t0 + u * 4 * KPI + grp;
                t     = t < t_cached_end ? t : t_cached_end - 1;
                kr[u] = *reinterpret_cast<const u32x4*>(kc + (size_t)t * DH + sub * 8);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                qk_one(kr[u], t0 + u * 4 * KPI + grp);
            }
        }
    }
    if (owns_cur && wid == 0) {  // current token from LDS (:1407-1437)
        float a = 0.f;
        if (lane < LPK) {
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
        const float pt = (t < t_cached_end) ? s_p[t - t_beg] : 0.f;
        const f16x8 vv = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = fmaf(pt, (float)vv[j], acc[j]);
        }
    };
    if (fast) {
#pragma unroll
        for (int u = 0; u < UK; u++) {
            pv_one(vreg[u], t_beg + u * 4 * KPI + wid * KPI + grp);
        }
    }
    else {
        for (int t0 = t_beg + wid * KPI; t0 < t_cached_end; t0 += 4 * KPI * U) {
            u32x4 vr[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                int t = t0 + u * 4 * KPI + grp;
                t     = t < t_cached_end ? t : t_cached_end - 1;
                vr[u] = *reinterpret_cast<const u32x4*>(vc + (size_t)t * DH + sub * 8);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                pv_one(vr[u], t0 + u * 4 * KPI + grp);
            }
        }
    }
    if (owns_cur && wid == 0 && grp == 0) {
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
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            s_o[wid * DH + sub * 8 + j] = acc[j];
        }
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const int d = threadIdx.x;
        st_agent(&wsout[d], (s_o[d] + s_o[DH + d]) + (s_o[2 * DH + d] + s_o[3 * DH + d]));
    }
    if (threadIdx.x == 0) {
        st_agent(&wsout[DH], m_loc);
        st_agent(&wsout[DH + 1], (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
    }
}

// One launch: every split workgroup publishes its (max, sum, out[DH]) partial, takes a ticket, and the LAST arriver
// of each (row, head) merges the partials in split order (deterministic) -- the in-launch hand-off recipe of
// cdna_hip_programming.md G16: plain stores -> per-wave vmcnt(0) -> barrier -> one-lane agent release (+ asm vmcnt(0))
// -> relaxed agent ticket ; consumer: one-lane agent acquire -> barrier -> plain loads.  Placement independent.
template<int DH>
__device__ __forceinline__ void mmha_block(const MmhaParams& p, char* smem, int& s_last, const int h, const int b, const int sp)
{
    if (p.finished && p.finished[b]) {
        return;  // :1176 (ctx of a finished row is never consumed); uniform for all splits of the row
    }
    float* wsout = p.ws + (((size_t)b * p.nh + h) * p.nsplit + sp) * (DH + 2);
    mmha_partial<DH>(p, smem, wsout, h, b, sp);
    if (p.nsplit == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x < DH) {
            const float inv = 1.f / (ld_agent(&wsout[DH + 1]) + 1.e-6f);  // :1632
            p.ctx[(size_t)b * p.nh * DH + h * DH + threadIdx.x] = (f16)(ld_agent(&wsout[threadIdx.x]) * inv);
        }
        return;
    }
    // publish: sc1 payload (st_agent) -> every wave drains its stores -> barrier -> ONE relaxed agent ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* counter = p.counters + (size_t)b * p.nh + h;
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last      = (t == p.nsplit - 1) ? 1 : 0;
        if (s_last) {
            __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
    }
    __syncthreads();
    if (!s_last) {
        return;
    }
    const float* ws = p.ws + ((size_t)b * p.nh + h) * p.nsplit * (DH + 2);
    float*       sw = reinterpret_cast<float*>(smem);  // [nsplit] weights, then [1] denominator
    if (threadIdx.x < 64) {
        float ms = -INFINITY, ls = 0.f;
        if ((int)threadIdx.x < p.nsplit) {  // nsplit <= 64
            ms = ld_agent(&ws[threadIdx.x * (DH + 2) + DH]);
            ls = ld_agent(&ws[threadIdx.x * (DH + 2) + DH + 1]);
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
#pragma unroll 4
        for (int s2 = 0; s2 < p.nsplit; s2++) {
            o += sw[s2] * ld_agent(&ws[s2 * (DH + 2) + d]);
        }
        const float inv = 1.f / (sw[p.nsplit] + 1.e-6f);  // :1632
        p.ctx[(size_t)b * p.nh * DH + h * DH + d] = (f16)(o * inv);
    }
}


}  // namespace ftcf
