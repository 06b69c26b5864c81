// Persistent decode layers for 3..16 rows (the batched decode step, BASELINE config 5's regime): ONE launch runs layers
// [l_begin, l_end) of GptNeoXDecoder<T>::forward (models/gptneox/GptNeoXDecoder.cc:245-384) on one resident 8-wave workgroup per CU.
//
// The one- and two-row kernel (persist_device.hip.h) stages a row's LayerNorm outputs, its whole FFN intermediate and its context
// in LDS: 51 KB per row.  Sixteen rows do not fit, so this kernel keeps NO activations in LDS:
//   * a weight pass walks its tiles K-MAJOR over up to RW_G 16-column groups: the eight waves of a workgroup cut the K extent, a
//     wave loads per k-step one tile of each group plus the rows' MFMA A fragment of that k-step (16 rows x 64 k = 2 KB against
//     up to 5 KB of weights) straight from L2 into registers -- a ring of RW_R such steps is in flight -- and the waves' partial
//     sums meet in LDS;
//   * the layer input travels as raw halves + per-row {sum, sum of squares} partials; every wave normalises its own k-steps on the
//     fly (gamma / beta of the layer sit in LDS), with the half2 arithmetic of layernorm_kernels.cu:157-286;
//   * a layer is FIVE chip-wide streams in this order: QKV -> FFN1 -> attention K/V rows -> FFN2 -> out-proj.  q/k/v travel under
//     the FFN1 stream, mid under the attention, the merged context under FFN2: only the layer boundary (pieces -> x' -> statistics)
//     is exposed;
//   * the attention is one wave per (row, head, KV split): a flash-decoding stream over the split's K and V rows with four
//     independent online soft-max states (one per 16-lane key group) and no LDS at all;
//   * hand-offs are flags behind drained write-through (sc1) stores, consumed with sc1 loads (cdna_hip_programming.md G16
//     recipe R1; MI355X_MICROARCH.md "publish-large": payloads here are KBs per workgroup, not the 8-byte granules of the
//     one-row kernel).  Flags are monotone tags (step * 256 + layer + 1), compared with >=.
// Wave 0 of a workgroup is its CONTROL wave: it polls the flags a stream depends on while the other seven are still inside the
// previous stream, reduces the waves' partial sums, applies the epilogue, publishes, merges KV splits and the K pieces.  The
// streamer waves only ever wait on LDS words.  Every spin is bounded and reports through RowsParams::err.
#pragma once
#include "attn_device.hip.h"
#include "gemv_device.hip.h"

namespace ftcf {

constexpr int RW_NW       = 8;
constexpr int RW_NT       = RW_NW * 64;
constexpr int RW_NS       = RW_NW - 1;  // streamer waves
constexpr int RW_G        = 5;          // 16-column groups per k-step (compile-time maximum; the QKV pass may use 4)
constexpr int RW_R        = 4;          // ring slots of a weight pass (three k-steps in flight while one is consumed)
constexpr int RW_KVB      = 4;          // K (and V) wave-loads per ring slot of the attention stream
constexpr int RW_KVR      = 4;          // its ring slots
constexpr int RW_SPIN     = 1 << 18;
constexpr int RW_MAXSPLIT = 32;
constexpr int RW_PHASES   = 5;  // QKV, FFN1, AT, FFN2, OUT
constexpr int RW_PA_PAD   = 4;  // floats behind the DH outputs of an attention partial: {max, sum} + padding to 16 bytes

typedef __attribute__((address_space(1))) unsigned rw_gu32;
#define RW_GP(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))
#define RW_RLX __ATOMIC_RELAXED
#define RW_AGT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ int rw_rfl(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}
// a buffer descriptor over [p, p + bytes) built from provably wave-uniform words (cdna_hip_programming.md T20)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rw_rsrc(const void* p, size_t bytes)
{
    const unsigned long long a  = (unsigned long long)p;
    const unsigned           lo = (unsigned)rw_rfl((int)(unsigned)a), hi = (unsigned)rw_rfl((int)(unsigned)(a >> 32));
    void*                    q  = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, rw_rfl((int)(bytes > 0xfffffff0u ? 0xfffffff0u : bytes)), 0x00020000);
}
// 16-byte loads / stores that bypass L1 and write through (sc1): the payload side of every hand-off
__device__ __forceinline__ u32x4 rw_ld16(const __amdgpu_buffer_rsrc_t r, const int voff, const int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16);
}
__device__ __forceinline__ void rw_st16(const u32x4 v, const __amdgpu_buffer_rsrc_t r, const int voff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 16);
}
__device__ __forceinline__ void rw_drain()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void rw_st_flag(unsigned* f, const unsigned v)
{
    __hip_atomic_store((rw_gu32*)f, v, RW_RLX, RW_AGT);
}
__device__ __forceinline__ bool rw_give_up(int& spins, int* err, const int code)
{
    if (++spins > RW_SPIN) {
        __hip_atomic_store((__attribute__((address_space(1))) int*)err, code, RW_RLX, RW_AGT);
        return true;
    }
    return (spins & 255) == 0 && __hip_atomic_load((__attribute__((address_space(1))) int*)err, RW_RLX, RW_AGT) != 0;
}
// one wave re-reads flags f[0 .. n) (four per lane and pass) until every one has reached `tag`
__device__ __forceinline__ void rw_poll(const unsigned* f, const int n, const unsigned tag, const int lane, int* err, const int code)
{
    for (int base = 0; base < n; base += 256) {
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int      i = base + k * 64 + lane;
                const unsigned v = __hip_atomic_load((const rw_gu32*)(f + (i < n ? i : n - 1)), RW_RLX, RW_AGT);
                ok &= (int)(v - tag) >= 0;
            }
            if (__all(ok)) {
                break;
            }
            if (rw_give_up(spins, err, code)) {
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}
// a wave waits until the LDS word has reached `target`
__device__ __forceinline__ void rw_lds_wait(const int* w, const int target, int* err, const int code)
{
    int spins = 0;
    while (rw_rfl(*(const volatile __attribute__((address_space(3))) int*)w) < target) {
        if (rw_give_up(spins, err, code)) {
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void rw_lds_set(int* w, const int v, const int lane)
{
    if (lane == 0) {
        *(volatile __attribute__((address_space(3))) int*)w = v;
    }
}
__device__ __forceinline__ void rw_lds_bump(int* w, const int lane)
{
    if (lane == 0) {
        atomicAdd(w, 1);
    }
}

// owner of column group g when NG groups are dealt to NB workgroups in contiguous ranges [w NG / NB, (w + 1) NG / NB)
__host__ __device__ inline int rw_group_begin(const int w, const int NG, const int NB)
{
    return (int)((long)w * NG / NB);
}
// k-steps [kb, ke) of wave `wid` when a workgroup streams k-steps [K0, K1): the control wave takes the first nc, the streamer
// waves cut the rest into contiguous shares
__host__ __device__ inline void rw_wave_ksteps(const int K0, const int K1, const int nc, const int wid, int& kb, int& ke)
{
    const int n = K1 - K0;
    const int c = nc < n ? nc : n;
    if (wid == 0) {
        kb = K0;
        ke = K0 + c;
        return;
    }
    const int r = n - c, s = wid - 1;
    kb = K0 + c + (int)((long)r * s / RW_NS);
    ke = K0 + c + (int)((long)r * (s + 1) / RW_NS);
}

template<bool INT8>
struct RwK {
    static constexpr int KS = INT8 ? TILE_K_I8 : TILE_K_F16;  // k per k-step (one tile along K)
    static constexpr int AV = INT8 ? 2 : 1;                   // 16-byte pieces of a lane's A fragment
};

// one weight tile against the rows' A fragment held in registers (gemv_device.hip.h consume_tile with the fragment from L2)
template<bool INT8>
__device__ __forceinline__ void rw_tile(const u32x4 w, const f16x8 a0, const f16x8 a1, const f16x2 sc2, f32x4& acc)
{
    if constexpr (INT8) {
        f16x2 d[8];
        dequant4(w.x, sc2, d[0], d[1]);
        dequant4(w.y, sc2, d[2], d[3]);
        dequant4(w.z, sc2, d[4], d[5]);
        dequant4(w.w, sc2, d[6], d[7]);
        const f16x8 b0 = {d[0][0], d[0][1], d[1][0], d[1][1], d[2][0], d[2][1], d[3][0], d[3][1]};
        const f16x8 b1 = {d[4][0], d[4][1], d[5][0], d[5][1], d[6][0], d[6][1], d[7][0], d[7][1]};
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc, 0, 0, 0);
        acc            = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc, 0, 0, 0);
    }
    else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, __builtin_bit_cast(f16x8, w), acc, 0, 0, 0);
    }
}

// LayerNorm of eight halves of one row: (((x - mean) * rstd) * gamma) + beta, every operation rounded to half
// (layernorm_kernels.cu:243-259; the same expression as kernels_misc.hip k_residual_dual_ln)
__device__ __forceinline__ f16x8 rw_ln8(const f16x8 x, const f16 mh, const f16 rh, const f16x8 g, const f16x8 b)
{
#pragma clang fp contract(off)
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const f16 c = (x[e] - mh) * rh;
        f16       a = c * g[e];
        a           = a + b[e];
        o[e]        = a;
    }
    return o;
}

struct RwSmem {
    f16*   gb;    // [4][H]: ln1 gamma, ln1 beta, ln2 gamma, ln2 beta of the current layer
    float* part;  // [2][NW][RW_G][4][64] partial sums of the waves (two buffers: consecutive passes alternate)
    float* stat;  // [16][2] mean, rstd of the layer input's rows
    float* scr;   // [512] scratch of the control wave
    f16*   att;   // [NW][3 * DH] q | k | v of a wave's attention unit
    int*   sync;  // [0] go (highest phase the streamers may enter), [1..2] partial sums written, [3..4] ... reduced
};

// ---------------------------------------------------------------------------------------------------------------
// One pass of one wave: k-steps [kb, ke) of column groups [g0, g0 + ng) (ng <= G; the groups beyond ng re-read the last one and
// are never stored).  A fragments: row min(lane & 15, M - 1), k = kstep * KS + (lane >> 4) * (KS / 4) .. + KS / 4 of the
// [M][lda] halves behind `ar`.  LN: normalised on the fly with the row's statistics and gamma / beta from LDS.
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int G, bool LN>
struct RwPass {
    static constexpr int KS = RwK<INT8>::KS;
    static constexpr int AV = RwK<INT8>::AV;
    struct Slot {
        u32x4 w[G];
        u32x4 a[AV];
    };
    Slot                    S[RW_R];
    f32x4                   acc[G];
    f16x2                   sc2[G];
    const char*             wb;       // weight image (uniform)
    unsigned                woff[G];  // byte offset of this lane's 16 bytes of group g's tile at k-step 0
    __amdgpu_buffer_rsrc_t  ar;
    int                     aoff;  // byte offset of this lane's fragment at k-step 0
    f16                     mh, rh;
    const f16 *             lg, *lb;  // LDS gamma / beta (+ the lane's k offset inside a k-step)

    __device__ __forceinline__ void bind(const void* w, const int KT, const int g0, const int ng, const f16* scale,
                                         const __amdgpu_buffer_rsrc_t ar_, const int lda_bytes, const int M, const int lane,
                                         const float* stat = nullptr, const f16* gamma = nullptr, const f16* beta = nullptr)
    {
        wb = reinterpret_cast<const char*>(w);
        ar = ar_;
        const int row = (lane & 15) < M ? (lane & 15) : M - 1;
        aoff          = row * lda_bytes + (lane >> 4) * (KS / 4) * 2;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int gg = g0 + (g < ng ? g : (ng > 0 ? ng - 1 : 0));
            woff[g]      = (unsigned)(((size_t)gg * KT * 64 + lane) * 16);
            acc[g]       = f32x4{0.f, 0.f, 0.f, 0.f};
            sc2[g]       = f16x2{(f16)1.f, (f16)1.f};
            if constexpr (INT8) {
                const f16 sc = *RW_GP(f16, scale + gg * 16 + (lane & 15));
                sc2[g]       = f16x2{sc, sc};
            }
        }
        if constexpr (LN) {
            mh = (f16)stat[row * 2];
            rh = (f16)stat[row * 2 + 1];
            lg = gamma + (lane >> 4) * (KS / 4);
            lb = beta + (lane >> 4) * (KS / 4);
        }
    }
    __device__ __forceinline__ void load(Slot& s, const int k)
    {
        const char* wk = wb + (size_t)k * TILE_BYTES;  // uniform
#pragma unroll
        for (int g = 0; g < G; g++) {
            s.w[g] = __builtin_nontemporal_load(RW_GP(u32x4, wk + woff[g]));
        }
#pragma unroll
        for (int v = 0; v < AV; v++) {
            s.a[v] = rw_ld16(ar, aoff + v * 16, k * KS * 2);
        }
    }
    __device__ __forceinline__ void consume(const Slot& s, const int k)
    {
        f16x8 a0 = __builtin_bit_cast(f16x8, s.a[0]);
        f16x8 a1 = __builtin_bit_cast(f16x8, s.a[AV - 1]);
        if constexpr (LN) {
            const f16x8* gp = reinterpret_cast<const f16x8*>(lg + k * KS);
            const f16x8* bp = reinterpret_cast<const f16x8*>(lb + k * KS);
            a0              = rw_ln8(a0, mh, rh, gp[0], bp[0]);
            if constexpr (AV == 2) {
                a1 = rw_ln8(a1, mh, rh, gp[1], bp[1]);
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            rw_tile<INT8>(s.w[g], a0, a1, sc2[g], acc[g]);
        }
    }
    // streams k-steps [kb, ke): the whole ring is requested first; a slot is re-requested as soon as it has been consumed
    __device__ __forceinline__ void run(const int kb, const int ke)
    {
        const int n = ke - kb;
        if (n <= 0) {
            return;
        }
        const int nrot = (n + RW_R - 1) / RW_R;
#pragma unroll
        for (int r = 0; r < RW_R; r++) {
            load(S[r], kb + (r < n ? r : n - 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int it = 0; it < nrot - 1; it++) {
#pragma unroll
            for (int r = 0; r < RW_R; r++) {
                const int i = it * RW_R + r;
                consume(S[r], kb + i);
                __builtin_amdgcn_sched_barrier(0);
                const int nx = i + RW_R;
                load(S[r], kb + (nx < n ? nx : n - 1));  // clamped, never conditional (the compiler counts vmcnt exactly)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int last = (nrot - 1) * RW_R;
#pragma unroll
        for (int r = 0; r < RW_R; r++) {
            if (last + r < n) {
                consume(S[r], kb + last + r);
            }
        }
    }
    // the wave's partial sums -> part[wid][g][j][lane] (j = accumulator register: row 4 (lane >> 4) + j, column lane & 15)
    __device__ __forceinline__ void dump(float* part, const int wid, const int lane) const
    {
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                part[((wid * RW_G + g) * 4 + j) * 64 + lane] = acc[g][j];
            }
        }
    }
};

// The control wave adds the waves' partial sums of one pass in wave order and hands epi(row, group, half, v[8]) eight
// consecutive columns of one row at a time (columns group * 16 + half * 8 ..).
template<typename EPI>
__device__ __forceinline__ void rw_reduce(const float* part, const int ng, const int M, const int lane, EPI&& epi)
{
    const int items = M * ng * 2;
    for (int it = lane; it < items; it += 64) {
        const int r = it / (ng * 2), q = it - r * (ng * 2), g = q >> 1, h8 = q & 1;
        float     v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            v[e] = 0.f;
        }
#pragma unroll
        for (int w = 0; w < RW_NW; w++) {
            const float* pp = part + ((w * RW_G + g) * 4 + (r & 3)) * 64 + (r >> 2) * 16 + h8 * 8;
            const f32x4  x0 = *reinterpret_cast<const f32x4*>(pp), x1 = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[e] += x0[e];
                v[4 + e] += x1[e];
            }
        }
        epi(r, g, h8, v);
    }
}

__device__ __forceinline__ u32x4 rw_pack8(const f16 (&h)[8])
{
    const f16x8 v = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    return __builtin_bit_cast(u32x4, v);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention of one (row b, head h, KV split sp) on ONE wave
// (decoder_masked_multihead_attention_template.hpp:1099-1919; the arithmetic of attn_device.hip.h mmha_partial).
// Lane (grp, sub) = key group lane / LPK, 16-byte piece lane % LPK of a K / V row; a wave-load covers KPI = 64 / LPK keys.
// Each key group keeps its own running {max, sum, out[8 per lane]}; the groups are merged at the end.
// ---------------------------------------------------------------------------------------------------------------
template<int DH, bool PAGED>
struct RwAttn {
    static constexpr int LPK = DH / 8;
    static constexpr int KPI = 64 / LPK;
    static_assert(DH == 64 || DH == 128, "size_per_head 64 or 128");
    struct Slot {
        u32x4 k[RW_KVB], v[RW_KVB];
    };

    __device__ __forceinline__ static void run(const RowsParams& p, const PersistLayer& lw, const __amdgpu_buffer_rsrc_t rq,
                                               const int b, const int h, const int sp, const unsigned tag, const int lane,
                                               f16* scr, const __amdgpu_buffer_rsrc_t rctx, const __amdgpu_buffer_rsrc_t rpa)
    {
        const int  sub = lane % LPK, grp = lane / LPK;
        const int  ns  = p.plan.nsplit;
        const int  hl  = p.nh * DH;
        const bool fin = p.finished && p.finished[b];
        const int  tl  = p.seq_len[b];  // cached keys; the new token goes to index tl
        // the request's splits cut the row's CURRENT length (not the cache's capacity): equal shares at every step
        const int chunk = (((tl + 1 + ns - 1) / ns) + KPI - 1) / KPI * KPI;
        const int t_beg = sp * chunk;
        int       t_end = t_beg + chunk;
        t_end           = t_end > tl + 1 ? tl + 1 : t_end;
        const bool owns_cur = !fin && tl >= t_beg && tl < t_end;
        const int  t_cend   = fin ? t_beg : (owns_cur ? tl : t_end);  // cached keys of the split: [t_beg, t_cend)
        const int  in_len   = p.input_lengths ? p.input_lengths[b] : 0x7fffffff;

        auto row_off = [&](const int t) -> size_t {  // element offset of key t's row inside the layer's cache / pool
            if constexpr (PAGED) {
                const int pg = p.page_table[(size_t)b * p.max_pages + t / p.page_tokens];
                return (((size_t)pg * p.nh + h) * p.page_tokens + (t % p.page_tokens)) * DH;
            }
            else {
                return (((size_t)b * p.nh + h) * p.s_max + t) * DH;
            }
        };
        // ---- q, k, v of the new token: bias (added here like the reference's MMHA), rotary ----
        const int   qo = (b * 3 * hl + h * DH + sub * 8) * 2;
        const u32x4 qr = rw_ld16(rq, qo, 0), kr = rw_ld16(rq, qo + hl * 2, 0), vr = rw_ld16(rq, qo + 2 * hl * 2, 0);
        f16x8       q8 = __builtin_bit_cast(f16x8, qr), k8 = __builtin_bit_cast(f16x8, kr), v8 = __builtin_bit_cast(f16x8, vr);
        if (lw.b_qkv) {
            const f16x8 bq = *RW_GP(f16x8, lw.b_qkv + h * DH + sub * 8), bk = *RW_GP(f16x8, lw.b_qkv + hl + h * DH + sub * 8),
                        bv = *RW_GP(f16x8, lw.b_qkv + 2 * hl + h * DH + sub * 8);
            q8 = q8 + bq;
            k8 = k8 + bk;
            v8 = v8 + bv;
        }
        if (p.rot > 0) {
            // NeoX pairing (x[j], x[j + rot / 2]) (decoder_masked_multihead_attention_utils.h:1325-1345): through the wave's LDS
            // scratch, the partner element may sit in another lane
            if (grp == 0) {
                *reinterpret_cast<f16x8*>(scr + sub * 8)      = q8;
                *reinterpret_cast<f16x8*>(scr + DH + sub * 8) = k8;
            }
            const int hr = p.rot / 2;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int d = sub * 8 + e;
                if (d < p.rot) {
                    const int   j  = d < hr ? d : d - hr;
                    const float cs = p.rot_table[((size_t)b * hr + j) * 2], sn = p.rot_table[((size_t)b * hr + j) * 2 + 1];
                    const int   pd = d < hr ? d + hr : d - hr;
                    const float qa = (float)q8[e], qb = (float)scr[pd], ka = (float)k8[e], kb = (float)scr[DH + pd];
                    // first of the pair: cs * a - sn * b ; second: cs * b' + sn * a' (b' = itself, a' = its partner)
                    q8[e] = d < hr ? (f16)(cs * qa - sn * qb) : (f16)(cs * qa + sn * qb);
                    k8[e] = d < hr ? (f16)(cs * ka - sn * kb) : (f16)(cs * ka + sn * kb);
                }
            }
        }
        if (owns_cur && grp == 0) {  // append to the cache (:1397, :1837)
            const size_t o = row_off(tl) + sub * 8;
            *reinterpret_cast<f16x8*>(lw.k_cache + o) = k8;
            *reinterpret_cast<f16x8*>(lw.v_cache + o) = v8;
        }
        const float inv_sqrt_dh = rsqrtf((float)DH);
        float       m = -INFINITY, l = 0.f, o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = 0.f;
        }
        auto qk = [&](const f16x8 kv) {
            float a = 0.f;
            a       = dot2(f16x2{q8[0], q8[1]}, f16x2{kv[0], kv[1]}, a);
            a       = dot2(f16x2{q8[2], q8[3]}, f16x2{kv[2], kv[3]}, a);
            a       = dot2(f16x2{q8[4], q8[5]}, f16x2{kv[4], kv[5]}, a);
            a       = dot2(f16x2{q8[6], q8[7]}, f16x2{kv[6], kv[7]}, a);
            return group_sum_dpp<LPK>(a) * inv_sqrt_dh;
        };
        // ---- the cached keys: a ring of RW_KVR slots of RW_KVB K rows + RW_KVB V rows per lane ----
        const int nkeys = t_cend - t_beg;
        if (nkeys > 0) {
            constexpr int STEP = RW_KVB * KPI;  // keys per slot
            const int     nst  = (nkeys + STEP - 1) / STEP;
            const int     nrot = (nst + RW_KVR - 1) / RW_KVR;
            const f16*    kc   = lw.k_cache + sub * 8;
            const f16*    vc   = lw.v_cache + sub * 8;
            Slot          S[RW_KVR];
            auto          ld = [&](Slot& s, const int st) {
#pragma unroll
                for (int u = 0; u < RW_KVB; u++) {
                    int t = t_beg + st * STEP + u * KPI + grp;
                    t     = t < t_cend ? t : t_cend - 1;  // clamped, never conditional; masked below
                    const size_t ro = row_off(t);
                    s.k[u]          = __builtin_nontemporal_load(RW_GP(u32x4, kc + ro));
                    s.v[u]          = __builtin_nontemporal_load(RW_GP(u32x4, vc + ro));
                }
            };
            auto use = [&](const Slot& s, const int st) {
                float sc[RW_KVB];
                float mx = m;
#pragma unroll
                for (int u = 0; u < RW_KVB; u++) {
                    const int  t  = t_beg + st * STEP + u * KPI + grp;
                    const bool ok = t < t_cend && !(t >= in_len && t < p.max_input_len);  // padding keys: probability 0 (:1570)
                    const float a = qk(__builtin_bit_cast(f16x8, s.k[u]));
                    sc[u]         = ok ? a : -INFINITY;
                    mx            = fmaxf(mx, sc[u]);
                }
                if (mx == -INFINITY) {
                    return;  // (uniform inside the key group: nothing to add yet)
                }
                const float f = __expf(m - mx);  // m = -inf: 0
                l *= f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    o[e] *= f;
                }
                m = mx;
#pragma unroll
                for (int u = 0; u < RW_KVB; u++) {
                    const float pt = (sc[u] == -INFINITY) ? 0.f : __expf(sc[u] - mx);
                    const f16x8 vv = __builtin_bit_cast(f16x8, s.v[u]);
                    l += pt;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        // (rows beyond the split were fetched as copies of its last row: pt is exactly 0 for them)
                        o[e] = fmaf(pt, (float)vv[e], o[e]);
                    }
                }
            };
#pragma unroll
            for (int r = 0; r < RW_KVR; r++) {
                ld(S[r], r < nst ? r : nst - 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int it = 0; it < nrot - 1; it++) {
#pragma unroll
                for (int r = 0; r < RW_KVR; r++) {
                    const int i = it * RW_KVR + r;
                    use(S[r], i);
                    __builtin_amdgcn_sched_barrier(0);
                    const int nx = i + RW_KVR;
                    ld(S[r], nx < nst ? nx : nst - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int last = (nrot - 1) * RW_KVR;
#pragma unroll
            for (int r = 0; r < RW_KVR; r++) {
                if (last + r < nst) {
                    use(S[r], last + r);
                }
            }
        }
        if (owns_cur) {  // the current token, from registers (:1407-1437): key group 0 takes it
            const float a = qk(k8);
            if (grp == 0) {
                const float mx = fmaxf(m, a);
                const float f  = __expf(m - mx);
                const float pt = __expf(a - mx);
                l              = l * f + pt;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    o[e] = fmaf(pt, (float)v8[e], o[e] * f);
                }
                m = mx;
            }
        }
        // ---- merge the key groups of the wave ----
        float mw = m;
        for (int off = LPK; off < 64; off <<= 1) {
            mw = fmaxf(mw, __shfl_xor(mw, off, 64));
        }
        const float wg = (m == -INFINITY) ? 0.f : __expf(m - mw);
        l              = across_groups_sum<LPK>(l * wg);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = across_groups_sum<LPK>(o[e] * wg);
        }
        const int pair = b * p.nh + h;
        if (ns == 1) {
            // a single split: normalise and write ctx here (the merger's arithmetic with one partial of weight 1)
            if (grp == 0) {
                const float w   = (mw == -INFINITY) ? 0.f : 1.f;
                const float inv = 1.f / (w * l + 1.e-6f);  // :1632
                f16         hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    hv[e] = (f16)((w * o[e]) * inv);
                }
                rw_st16(rw_pack8(hv), rctx, (b * hl + h * DH + sub * 8) * 2);
            }
            rw_drain();
            if (lane == 0) {
                rw_st_flag(p.fc + pair, tag);
            }
            return;
        }
        // partial {out[DH], max, sum} of the split (fp32)
        const int po = ((pair * ns + sp) * (DH + RW_PA_PAD)) * 4;
        if (grp == 0) {
            rw_st16(__builtin_bit_cast(u32x4, f32x4{o[0], o[1], o[2], o[3]}), rpa, po + sub * 32);
            rw_st16(__builtin_bit_cast(u32x4, f32x4{o[4], o[5], o[6], o[7]}), rpa, po + sub * 32 + 16);
        }
        if (lane == 0) {
            __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(p.pa + (size_t)(pair * ns + sp) * (DH + RW_PA_PAD) + DH),
                               ((unsigned long long)__float_as_uint(l) << 32) | (unsigned long long)__float_as_uint(mw), RW_RLX,
                               RW_AGT);
        }
        rw_drain();
        if (lane == 0) {
            rw_st_flag(p.fa + pair * ns + sp, tag);
        }
    }
};

// the control wave merges the KV splits of (row, head) pair `pair` in split order (attn_device.hip.h mmha_block's merger)
template<int DH>
__device__ __forceinline__ void rw_merge_splits(const RowsParams& p, const int pair, const int lane, const __amdgpu_buffer_rsrc_t rpa,
                                                const __amdgpu_buffer_rsrc_t rctx)
{
    const int ns = p.plan.nsplit;
    const int ne = DH + RW_PA_PAD;
    float     ms = -INFINITY, ls = 0.f;
    if (lane < ns) {
        const unsigned long long ml = __hip_atomic_load(
            (const __attribute__((address_space(1))) unsigned long long*)(p.pa + (size_t)(pair * ns + lane) * ne + DH), RW_RLX, RW_AGT);
        ms = __uint_as_float((unsigned)ml);
        ls = __uint_as_float((unsigned)(ml >> 32));
    }
    const float mx = wave_max(ms);
    const float w  = (ms == -INFINITY) ? 0.f : __expf(ms - mx);
    float       L  = 0.f;
    for (int s2 = 0; s2 < ns; s2++) {  // fixed order
        L += __shfl(w * ls, s2, 64);
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        o[e] = 0.f;
    }
    const int sub = lane < DH / 8 ? lane : DH / 8 - 1;
    for (int s0 = 0; s0 < ns; s0 += 4) {
        u32x4 x[4][2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int s2 = s0 + j < ns ? s0 + j : ns - 1;
            x[j][0]      = rw_ld16(rpa, ((pair * ns + s2) * ne) * 4 + sub * 32, 0);
            x[j][1]      = rw_ld16(rpa, ((pair * ns + s2) * ne) * 4 + sub * 32 + 16, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (s0 + j < ns) {
                const float ws = __shfl(w, s0 + j, 64);
                const f32x4 a = __builtin_bit_cast(f32x4, x[j][0]), c = __builtin_bit_cast(f32x4, x[j][1]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    o[e] += ws * a[e];
                    o[4 + e] += ws * c[e];
                }
            }
        }
    }
    if (lane < DH / 8) {
        const float inv = 1.f / (L + 1.e-6f);  // :1632
        f16         hv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            hv[e] = (f16)(o[e] * inv);
        }
        const int b = pair / p.nh, h = pair - b * p.nh;
        rw_st16(rw_pack8(hv), rctx, (b * p.nh * DH + h * DH + lane * 8) * 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int DH, int G1, bool PAGED>
__global__ __launch_bounds__(RW_NT) void k_decode_rows(const RowsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = RwK<INT8>::KS;
    if (p.d_stop && *p.d_stop) {
        return;  // every row has finished: uniform over the grid
    }
    int       tx = threadIdx.x;
    const int lane = tx & 63, wid = rw_rfl(tx >> 6);
    const int wg = blockIdx.x, NB = p.plan.NB;
    const int M = p.M, H = p.H, Hl = p.Hl, Il = p.Il;
    const int CB = p.plan.CB, KP2 = p.plan.KP2, KP3 = p.plan.KP3, ns = p.plan.nsplit;
    const int      step     = *p.d_step;
    const unsigned tag_base = (unsigned)step * 256u + 1u;

    RwSmem s;
    {
        char* q = smem;
        s.gb    = reinterpret_cast<f16*>(q);
        q += (size_t)4 * H * 2;
        s.part = reinterpret_cast<float*>(q);
        q += (size_t)2 * RW_NW * RW_G * 256 * 4;
        s.stat = reinterpret_cast<float*>(q);
        q += 32 * 4;
        s.scr = reinterpret_cast<float*>(q);
        q += 512 * 4;
        s.att = reinterpret_cast<f16*>(q);
        q += (size_t)RW_NW * 3 * DH * 2;
        s.sync = reinterpret_cast<int*>(q);
    }
    auto stamp = [&](const int l, const int slot) {
        if (p.ts && lane == 0) {
            p.ts[(((size_t)wg * p.L + l) * RW_NW + wid) * 16 + slot] = wall_clock64();
        }
    };
    // ---- static shares ----
    const int KT1 = H / KS, KT2 = Il / KS, KT3 = Hl / KS;     // k-steps of QKV / FFN1, FFN2, out-proj
    const int NGq = 3 * Hl / 16, NGf = Il / 16, NGo = H / 16;  // column groups
    const int q0 = rw_group_begin(wg, NGq, NB), nq = rw_group_begin(wg + 1, NGq, NB) - q0;
    const int f0 = rw_group_begin(wg, NGf, NB), nf = rw_group_begin(wg + 1, NGf, NB) - f0;
    // row-parallel GEMMs: workgroup -> (column block, K piece); the same column blocks for FFN2 and out-proj
    const bool has2 = wg < CB * KP2, has3 = wg < CB * KP3;
    const int  cb2 = has2 ? wg / KP2 : 0, kp2 = has2 ? wg % KP2 : 0;
    const int  cb3 = has3 ? wg / KP3 : 0, kp3 = has3 ? wg % KP3 : 0;
    const int  ng2 = has2 ? ((cb2 + 1) * RW_G < NGo ? RW_G : NGo - cb2 * RW_G) : 0;
    const int  ng3 = has3 ? ((cb3 + 1) * RW_G < NGo ? RW_G : NGo - cb3 * RW_G) : 0;
    const int  k2a = (int)((long)KT2 * kp2 / KP2), k2b = has2 ? (int)((long)KT2 * (kp2 + 1) / KP2) : k2a;
    const int  k3a = (int)((long)KT3 * kp3 / KP3), k3b = has3 ? (int)((long)KT3 * (kp3 + 1) / KP3) : k3a;
    int        kb1, ke1, kb2, ke2, kb3, ke3;
    rw_wave_ksteps(0, KT1, p.plan.nc1, wid, kb1, ke1);
    rw_wave_ksteps(k2a, k2b, p.plan.nc2, wid, kb2, ke2);
    rw_wave_ksteps(k3a, k3b, p.plan.nc3, wid, kb3, ke3);
    // attention unit of this wave (streamer waves): slot (wid - 1) * NB + wg of M * nh * ns units, pair major
    const int  unit     = (wid - 1) * NB + wg;
    const bool has_unit = wid > 0 && unit < M * p.nh * ns;
    const int  upair = has_unit ? unit / ns : 0, usp = has_unit ? unit - upair * ns : 0;
    const int  ub = upair / p.nh, uh = upair - ub * p.nh;
    const int  npairs = M * p.nh;

    // buffers behind descriptors (sc1 accesses)
    const __amdgpu_buffer_rsrc_t r_qkv = rw_rsrc(p.qkv, (size_t)M * 3 * Hl * 2), r_mid = rw_rsrc(p.mid, (size_t)M * Il * 2),
                                 r_ctx = rw_rsrc(p.ctx, (size_t)M * Hl * 2), r_p2 = rw_rsrc(p.p2, (size_t)KP2 * M * H * 4),
                                 r_p3 = rw_rsrc(p.p3, (size_t)KP3 * M * H * 4),
                                 r_pa = rw_rsrc(p.pa, (size_t)M * p.nh * ns * (DH + RW_PA_PAD) * 4);

    // ---- kernel start: statistics of x_in's rows and the first layer's LayerNorm parameters, by everybody ----
    {
        const PersistLayer& lw = p.layers[p.l_begin];
        for (int i = tx; i < H / 8; i += RW_NT) {
            reinterpret_cast<u32x4*>(s.gb)[i]             = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln1_g) + i);
            reinterpret_cast<u32x4*>(s.gb + H)[i]         = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln1_b) + i);
            reinterpret_cast<u32x4*>(s.gb + 2 * H)[i]     = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln2_g) + i);
            reinterpret_cast<u32x4*>(s.gb + 3 * (size_t)H)[i] = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln2_b) + i);
        }
        for (int r = wid; r < M; r += RW_NW) {
            float s0 = 0.f, s1 = 0.f;
            for (int i = lane; i < H / 8; i += 64) {
                const f16x8 v = *RW_GP(f16x8, reinterpret_cast<const f16x8*>(p.x_in + (size_t)r * H) + i);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float f = (float)v[e];
                    s0 += f;
                    s1 += f * f;
                }
            }
            s0 = wave_sum(s0);
            s1 = wave_sum(s1);
            if (lane == 0) {
                const float mean  = s0 / (float)H;
                s.stat[r * 2]     = mean;
                s.stat[r * 2 + 1] = rsqrtf(s1 / (float)H - mean * mean + p.eps);
            }
        }
        if (tx < 8) {
            s.sync[tx] = 0;
        }
        __syncthreads();
    }

    for (int l = p.l_begin; l < p.l_end; l++) {
        asm volatile("" : "+v"(tx));
        const PersistLayer& lw  = p.layers[l];
        const unsigned      tag = tag_base + (unsigned)l;
        const int           ph0 = (l - p.l_begin) * RW_PHASES + 1;  // go value that admits QKV of this layer
        const int           rp0 = (l - p.l_begin) * 4;              // index of this layer's first reducing pass
        const bool          first = l == p.l_begin, last = l == p.l_end - 1;
        const f16*          xin = first ? p.x_in : p.xb[l & 1];
        f16*                xout = last ? p.x_out : p.xb[(l + 1) & 1];
        const __amdgpu_buffer_rsrc_t r_x = rw_rsrc(xin, (size_t)M * H * 2);
        // partial-sum buffer of reducing pass rp: before rewriting it the waves wait for the reduction two passes back
        auto part_of = [&](const int rp) { return s.part + (size_t)(rp & 1) * RW_NW * RW_G * 256; };
        auto wait_part_free = [&](const int rp) { rw_lds_wait(&s.sync[3 + (rp & 1)], rp >> 1, p.err, 20); };
        auto part_done      = [&](const int rp) { rw_lds_bump(&s.sync[1 + (rp & 1)], lane); };

        if (wid != 0) {
            // =========================================== streamer waves ===========================================
            stamp(l, 0);
            rw_lds_wait(&s.sync[0], ph0, p.err, 10);
            stamp(l, 1);
            {
                RwPass<INT8, G1, true> ps;
                ps.bind(lw.w_qkv, KT1, q0, nq, lw.s_qkv, r_x, H * 2, M, lane, s.stat, s.gb, s.gb + H);
                ps.run(kb1, nq > 0 ? ke1 : kb1);
                wait_part_free(rp0);
                ps.dump(part_of(rp0), wid, lane);
                part_done(rp0);
            }
            stamp(l, 2);
            {
                RwPass<INT8, RW_G, true> ps;
                ps.bind(lw.w_ffn1, KT1, f0, nf, lw.s_ffn1, r_x, H * 2, M, lane, s.stat, s.gb + 2 * H, s.gb + 3 * (size_t)H);
                ps.run(kb1, nf > 0 ? ke1 : kb1);
                wait_part_free(rp0 + 1);
                ps.dump(part_of(rp0 + 1), wid, lane);
                part_done(rp0 + 1);
            }
            stamp(l, 3);
            rw_lds_wait(&s.sync[0], ph0 + 2, p.err, 11);
            stamp(l, 4);
            if (has_unit) {
                RwAttn<DH, PAGED>::run(p, lw, r_qkv, ub, uh, usp, tag, lane, s.att + (size_t)wid * 3 * DH, r_ctx, r_pa);
            }
            stamp(l, 5);
            rw_lds_wait(&s.sync[0], ph0 + 3, p.err, 12);
            stamp(l, 6);
            {
                RwPass<INT8, RW_G, false> ps;
                ps.bind(lw.w_ffn2, KT2, cb2 * RW_G, ng2, lw.s_ffn2, r_mid, Il * 2, M, lane);
                ps.run(kb2, ng2 > 0 ? ke2 : kb2);
                wait_part_free(rp0 + 2);
                ps.dump(part_of(rp0 + 2), wid, lane);
                part_done(rp0 + 2);
            }
            stamp(l, 7);
            rw_lds_wait(&s.sync[0], ph0 + 4, p.err, 13);
            stamp(l, 8);
            {
                RwPass<INT8, RW_G, false> ps;
                ps.bind(lw.w_out, KT3, cb3 * RW_G, ng3, lw.s_out, r_ctx, Hl * 2, M, lane);
                ps.run(kb3, ng3 > 0 ? ke3 : kb3);
                wait_part_free(rp0 + 3);
                ps.dump(part_of(rp0 + 3), wid, lane);
                part_done(rp0 + 3);
            }
            stamp(l, 9);
            continue;
        }
        // =============================================== control wave ===============================================
        stamp(l, 0);
        if (!first) {
            // the layer input is complete when every merger has published its piece of x'; then the rows' statistics: the
            // mergers' partials summed in column-block order
            rw_poll(p.fx, CB * KP3, tag - 1u, lane, p.err, 1);
            const unsigned long long* st = p.stats + (size_t)(l & 1) * M * CB;
            const int                 r = lane & 15, qd = lane >> 4;
            float                     s0 = 0.f, s1 = 0.f;
            for (int c = qd; c < CB; c += 4) {
                const unsigned long long v = __hip_atomic_load(
                    (const __attribute__((address_space(1))) unsigned long long*)(st + (size_t)(r < M ? r : M - 1) * CB + c), RW_RLX, RW_AGT);
                s0 += __uint_as_float((unsigned)v);
                s1 += __uint_as_float((unsigned)(v >> 32));
            }
            const float a0 = __shfl(s0, r, 64), a1 = __shfl(s0, r + 16, 64), a2 = __shfl(s0, r + 32, 64), a3 = __shfl(s0, r + 48, 64);
            const float b0 = __shfl(s1, r, 64), b1 = __shfl(s1, r + 16, 64), b2 = __shfl(s1, r + 32, 64), b3 = __shfl(s1, r + 48, 64);
            const float t0 = (a0 + a1) + (a2 + a3), t1 = (b0 + b1) + (b2 + b3);
            if (lane < M) {
                const float mean     = t0 / (float)H;
                s.stat[lane * 2]     = mean;
                s.stat[lane * 2 + 1] = rsqrtf(t1 / (float)H - mean * mean + p.eps);
            }
        }
        rw_lds_set(&s.sync[0], ph0 + 1, lane);  // QKV and FFN1 may run (FFN1 needs nothing new)
        stamp(l, 1);
        // ---- QKV: own share, then the waves' sums -> q | k | v (no bias: the attention adds it, like the reference's MMHA) ----
        {
            RwPass<INT8, G1, true> ps;
            ps.bind(lw.w_qkv, KT1, q0, nq, lw.s_qkv, r_x, H * 2, M, lane, s.stat, s.gb, s.gb + H);
            ps.run(kb1, nq > 0 ? ke1 : kb1);
            wait_part_free(rp0);
            ps.dump(part_of(rp0), wid, lane);
            part_done(rp0);
        }
        rw_lds_wait(&s.sync[1 + (rp0 & 1)], ((rp0 >> 1) + 1) * RW_NW, p.err, 21);
        stamp(l, 2);
        if (nq > 0) {
            rw_reduce(part_of(rp0), nq, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                f16 hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    hv[e] = (f16)v[e];
                }
                rw_st16(rw_pack8(hv), r_qkv, (r * 3 * Hl + (q0 + g) * 16 + h8 * 8) * 2);
            });
        }
        rw_drain();
        if (lane == 0) {
            rw_st_flag(p.fq + wg, tag);
        }
        rw_lds_bump(&s.sync[3 + (rp0 & 1)], lane);
        // the next layer's ln1 parameters (every wave of this workgroup is past its QKV pass)
        if (!last) {
            const PersistLayer& nx = p.layers[l + 1];
            for (int i = lane; i < H / 8; i += 64) {
                reinterpret_cast<u32x4*>(s.gb)[i]     = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(nx.ln1_g) + i);
                reinterpret_cast<u32x4*>(s.gb + H)[i] = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(nx.ln1_b) + i);
            }
        }
        stamp(l, 3);
        // ---- FFN1: own share, then mid = gelu(. + bias) ----
        {
            RwPass<INT8, RW_G, true> ps;
            ps.bind(lw.w_ffn1, KT1, f0, nf, lw.s_ffn1, r_x, H * 2, M, lane, s.stat, s.gb + 2 * H, s.gb + 3 * (size_t)H);
            ps.run(kb1, nf > 0 ? ke1 : kb1);
            wait_part_free(rp0 + 1);
            ps.dump(part_of(rp0 + 1), wid, lane);
            part_done(rp0 + 1);
        }
        // q | k | v of every producer (they travelled under the FFN1 stream): the attention may start
        rw_poll(p.fq, NB, tag, lane, p.err, 2);
        rw_lds_set(&s.sync[0], ph0 + 2, lane);
        stamp(l, 4);
        rw_lds_wait(&s.sync[1 + ((rp0 + 1) & 1)], (((rp0 + 1) >> 1) + 1) * RW_NW, p.err, 22);
        if (nf > 0) {
            rw_reduce(part_of(rp0 + 1), nf, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                const int   col = (f0 + g) * 16 + h8 * 8;
                const f16x8 bv  = *RW_GP(f16x8, lw.b_ffn1 + col);
                f16         hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    if constexpr (INT8) {
                        hv[e] = (f16)gelu_f32(v[e] + (float)bv[e]);  // fused fp32 epilogue (epilogue_helpers.h:52-62)
                    }
                    else {
                        hv[e] = gelu_f16((f16)v[e] + bv[e]);  // cuBLAS rounds to half, invokeAddBiasGeluV2 in half
                    }
                }
                rw_st16(rw_pack8(hv), r_mid, (r * Il + col) * 2);
            });
        }
        rw_drain();
        if (lane == 0) {
            rw_st_flag(p.fm + wg, tag);
        }
        rw_lds_bump(&s.sync[3 + ((rp0 + 1) & 1)], lane);
        if (!last) {
            const PersistLayer& nx = p.layers[l + 1];
            for (int i = lane; i < H / 8; i += 64) {
                reinterpret_cast<u32x4*>(s.gb + 2 * H)[i]         = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(nx.ln2_g) + i);
                reinterpret_cast<u32x4*>(s.gb + 3 * (size_t)H)[i] = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(nx.ln2_b) + i);
            }
        }
        stamp(l, 5);
        // ---- mid of every producer (it travels under the attention): FFN2 may start ----
        rw_poll(p.fm, NB, tag, lane, p.err, 3);
        rw_lds_set(&s.sync[0], ph0 + 3, lane);
        stamp(l, 6);
        // ---- KV splits -> ctx for the (row, head) pairs of this workgroup ----
        if (ns > 1) {
            for (int pr = wg; pr < npairs; pr += NB) {
                rw_poll(p.fa + (size_t)pr * ns, ns, tag, lane, p.err, 4);
                rw_merge_splits<DH>(p, pr, lane, r_pa, r_ctx);
            }
            rw_drain();
            for (int pr = wg; pr < npairs; pr += NB) {
                if (lane == 0) {
                    rw_st_flag(p.fc + pr, tag);
                }
            }
        }
        stamp(l, 7);
        // ---- FFN2: own share, the waves' sums -> fp32 partial of this K piece ----
        {
            RwPass<INT8, RW_G, false> ps;
            ps.bind(lw.w_ffn2, KT2, cb2 * RW_G, ng2, lw.s_ffn2, r_mid, Il * 2, M, lane);
            ps.run(kb2, ng2 > 0 ? ke2 : kb2);
            wait_part_free(rp0 + 2);
            ps.dump(part_of(rp0 + 2), wid, lane);
            part_done(rp0 + 2);
        }
        // the merged context of every pair (it travelled under the FFN2 stream): out-proj may start
        rw_poll(p.fc, npairs, tag, lane, p.err, 5);
        rw_lds_set(&s.sync[0], ph0 + 4, lane);
        stamp(l, 8);
        rw_lds_wait(&s.sync[1 + ((rp0 + 2) & 1)], (((rp0 + 2) >> 1) + 1) * RW_NW, p.err, 23);
        if (ng2 > 0) {
            rw_reduce(part_of(rp0 + 2), ng2, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                const int o = ((kp2 * M + r) * H + (cb2 * RW_G + g) * 16 + h8 * 8) * 4;
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r_p2, o);
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r_p2, o + 16);
            });
        }
        rw_drain();
        if (lane == 0) {
            rw_st_flag(p.f2 + wg, tag);
        }
        rw_lds_bump(&s.sync[3 + ((rp0 + 2) & 1)], lane);
        stamp(l, 9);
        // ---- out-proj: own share, partial of this K piece ----
        {
            RwPass<INT8, RW_G, false> ps;
            ps.bind(lw.w_out, KT3, cb3 * RW_G, ng3, lw.s_out, r_ctx, Hl * 2, M, lane);
            ps.run(kb3, ng3 > 0 ? ke3 : kb3);
            wait_part_free(rp0 + 3);
            ps.dump(part_of(rp0 + 3), wid, lane);
            part_done(rp0 + 3);
        }
        rw_lds_wait(&s.sync[1 + ((rp0 + 3) & 1)], (((rp0 + 3) >> 1) + 1) * RW_NW, p.err, 24);
        stamp(l, 10);
        if (ng3 > 0) {
            rw_reduce(part_of(rp0 + 3), ng3, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                const int o = ((kp3 * M + r) * H + (cb3 * RW_G + g) * 16 + h8 * 8) * 4;
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r_p3, o);
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r_p3, o + 16);
            });
        }
        rw_drain();
        if (lane == 0) {
            rw_st_flag(p.f3 + wg, tag);
        }
        rw_lds_bump(&s.sync[3 + ((rp0 + 3) & 1)], lane);
        stamp(l, 11);
        // ---- x' = residual(x, attention pieces, FFN pieces, bias) for rows [r0, r1) of this workgroup's column block
        //      (invokeAddBiasAttentionFfnResidual, add_residual_kernels.cu:116-178) + the rows' partial statistics ----
        if (has3) {
            rw_poll(p.f2 + (size_t)cb3 * KP2, KP2, tag, lane, p.err, 6);
            rw_poll(p.f3 + (size_t)cb3 * KP3, KP3, tag, lane, p.err, 7);
            const int r0 = (int)((long)M * kp3 / KP3), r1 = (int)((long)M * (kp3 + 1) / KP3);
            // layer_input/output alias for 0 < l < L-1 in the reference (GptNeoXDecoder.cc:249-250) -> residual form
            const int inplace = (l > 0 && l < p.L - 1) ? 1 : 0;
            const __amdgpu_buffer_rsrc_t r_xo = rw_rsrc(xout, (size_t)M * H * 2);
            const int                    per  = ng3 * 2, items = (r1 - r0) * per;
            for (int base = 0; base < items; base += 64) {
                const int  it = base + lane;
                const bool on = it < items;
                const int  r = r0 + (on ? it / per : 0), q = on ? it % per : 0;
                const int  col = (cb3 * RW_G + (q >> 1)) * 16 + (q & 1) * 8;
                float      sa[8], sb[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    sa[e] = sb[e] = 0.f;
                }
                for (int j = 0; j < KP3; j++) {  // piece order: deterministic
                    const f32x4 a = __builtin_bit_cast(f32x4, rw_ld16(r_p3, ((j * M + r) * H + col) * 4, 0));
                    const f32x4 c = __builtin_bit_cast(f32x4, rw_ld16(r_p3, ((j * M + r) * H + col) * 4 + 16, 0));
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sa[e] += a[e];
                        sa[4 + e] += c[e];
                    }
                }
                for (int j = 0; j < KP2; j++) {
                    const f32x4 a = __builtin_bit_cast(f32x4, rw_ld16(r_p2, ((j * M + r) * H + col) * 4, 0));
                    const f32x4 c = __builtin_bit_cast(f32x4, rw_ld16(r_p2, ((j * M + r) * H + col) * 4 + 16, 0));
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sb[e] += a[e];
                        sb[4 + e] += c[e];
                    }
                }
                const f16x8 xv = __builtin_bit_cast(f16x8, rw_ld16(r_x, (r * H + col) * 2, 0));
                const f16x8 bv = *RW_GP(f16x8, lw.b_res + col);
                f16         hv[8];
                float       q0s = 0.f, q1s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const f16 attn = (f16)sa[e], ffn = (f16)sb[e];
                    const f16 xi   = (f16)((float)xv[e] / (float)p.tp);
                    f16       o;
                    if (inplace) {
                        o = (f16)((float)xi + (float)ffn + (float)attn + (float)bv[e]);
                    }
                    else {
                        o = ((ffn + attn) + bv[e]) + xi;
                    }
                    hv[e]         = o;
                    const float f = (float)o;
                    q0s += f;
                    q1s += f * f;
                }
                if (on) {
                    if (last) {
                        *reinterpret_cast<u32x4*>(xout + (size_t)r * H + col) = rw_pack8(hv);
                    }
                    else {
                        rw_st16(rw_pack8(hv), r_xo, (r * H + col) * 2);
                    }
                    s.scr[it * 2]     = q0s;
                    s.scr[it * 2 + 1] = q1s;
                }
            }
            // per-row partial statistics over this column block, items in fixed order (items <= 128: 16 rows x 5 groups x 2 / KP3
            // at the shapes planned for; rows_plan() checks it)
            if (!last && lane < r1 - r0) {
                float t0 = 0.f, t1 = 0.f;
                for (int q = 0; q < per; q++) {
                    t0 += s.scr[(lane * per + q) * 2];
                    t1 += s.scr[(lane * per + q) * 2 + 1];
                }
                __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(p.stats + (size_t)((l + 1) & 1) * M * CB
                                                                                          + (size_t)(r0 + lane) * CB + cb3),
                                   ((unsigned long long)__float_as_uint(t1) << 32) | (unsigned long long)__float_as_uint(t0), RW_RLX,
                                   RW_AGT);
            }
            rw_drain();
            if (lane == 0) {
                rw_st_flag(p.fx + wg, tag);
            }
        }
        stamp(l, 12);
    }
}

}  // namespace ftcf
