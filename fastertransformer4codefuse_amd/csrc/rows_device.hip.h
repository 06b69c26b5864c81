// Persistent decode layers for 3..16 rows (the batched decode step, BASELINE config 5's regime): ONE launch runs layers
// [l_begin, l_end) of GptNeoXDecoder<T>::forward (models/gptneox/GptNeoXDecoder.cc:245-384) on one resident 8-wave workgroup per CU.
//
// The one- and two-row kernel (persist_device.hip.h) stages a row's LayerNorm outputs, its whole FFN intermediate and its context
// in LDS: 51 KB per row.  Sixteen rows do not fit, so this kernel keeps NO activations in LDS:
//   * a weight pass walks its tiles K-MAJOR over up to RW_G 16-column groups: the seven streamer waves of a workgroup cut the K
//     extent, a wave loads per k-step one tile of each group plus the rows' MFMA A fragment of that k-step (16 rows x 64 k = 2 KB
//     against up to 5 KB of weights) straight from L2 into registers -- a ring of RW_R such steps is in flight -- and the waves'
//     partial sums meet in LDS;
//   * the layer input travels as raw halves + per-row {sum, sum of squares} partials; every wave normalises its own k-steps on the
//     fly (gamma / beta of the layer sit in LDS), with the half2 arithmetic of layernorm_kernels.cu:157-286;
//   * a layer is FIVE chip-wide streams in this order: QKV -> FFN1 -> attention K/V rows -> FFN2 -> out-proj.  q/k/v travel under
//     the FFN1 stream, mid under the attention, the context under FFN2: only the layer boundary (pieces -> x' + statistics) is
//     exposed, and every wait of a streamer wave is entered with the NEXT stream's first ring of weight tiles (or K/V rows)
//     already requested: they do not depend on what the wave waits for, only the A fragments / q do;
//   * the attention of a (row, head) pair stays inside ONE workgroup (pair = workgroup + u * NB): its seven streamer waves take the
//     pair's 16-key blocks round robin (a flash-decoding stream per wave with one online soft-max state per 16-lane key group, the
//     K/V ring running on across the workgroup's pairs), leave {out, max, sum} partials in LDS, and the control wave merges them in
//     wave order and publishes the context: no KV-split hop between workgroups;
//   * wave 0 of a workgroup is its CONTROL wave and streams nothing: it polls the flags a stream depends on while the streamers are
//     still inside the previous stream, reduces the waves' partial sums, applies the epilogues, and is the ONLY wave that stores to
//     the hand-off region (the streamers write LDS -- and the K/V cache row of the new token) -- so a streamer never drains its
//     loads.  Hand-offs are flags behind drained write-through (sc1) stores, consumed with sc1 loads (cdna_hip_programming.md G16
//     recipe R1; MI355X_MICROARCH.md "publish-large").  Flags are monotone tags (step * 256 + layer + 1), compared with >=;
//   * the layer boundary is ONE hop of 16-byte granules: merger (row, column range) sums the K pieces of FFN2 and out-proj, adds the
//     residual, stores its piece of x' and then ONE granule {tag, sum, sum of squares}; every control wave sweeps the M * CR granules
//     (tags and statistics in the same loads).
// Every spin is bounded and reports through RowsParams::err.
#pragma once
#include "attn_device.hip.h"
#include "gemv_device.hip.h"

namespace ftcf {

constexpr int RW_NW     = 8;
constexpr int RW_NT     = RW_NW * 64;
constexpr int RW_NS     = RW_NW - 1;  // streamer waves
constexpr int RW_G      = 5;          // 16-column groups per k-step (compile-time maximum; the QKV pass may use 4)
constexpr int RW_R      = 4;          // ring slots of a weight pass (three k-steps in flight while one is consumed)
// ring slots of the QKV pass: its weights are requested at the layer boundary, while the pieces of x' travel and the HBM has nothing
// else to do -- deep enough to take most of the wave's share (G1 = 4: 8 x 24 registers; G1 = 5: 6 x 28)
// (deeper rings -- 8 / 6 slots -- and A fragments requested fewer steps ahead were measured no better: profiles/r05_notes.md)
template<int G1>
struct RwQ {
    static constexpr int R  = 4;
    static constexpr int RA = R;
};
constexpr int RW_KVB    = 4;          // K (and V) wave-loads per ring slot of the attention stream
constexpr int RW_KVR    = 4;          // its ring slots
constexpr int RW_UMAX   = 4;          // (row, head) pairs of a workgroup at most
constexpr int RW_SPIN   = 1 << 18;
constexpr int RW_PHASES = 4;  // go values of a layer: QKV (+ FFN1), AT, FFN2, OUT
constexpr int RW_PA     = 4;  // floats behind the DH outputs of an attention partial: {max, sum} + padding to 16 bytes

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
// loads / stores that bypass L1 and write through (sc1): both sides of every hand-off (voff: per lane, soff: uniform)
__device__ __forceinline__ u32x4 rw_ld16(const __amdgpu_buffer_rsrc_t r, const int voff, const int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16);
}
__device__ __forceinline__ void rw_st16(const u32x4 v, const __amdgpu_buffer_rsrc_t r, const int voff, const int soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 16);
}
__device__ __forceinline__ unsigned rw_ld4(const __amdgpu_buffer_rsrc_t r, const int voff, const int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 16);
}
__device__ __forceinline__ void rw_st4(const unsigned v, const __amdgpu_buffer_rsrc_t r, const int voff, const int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, 16);
}
__device__ __forceinline__ void rw_drain()  // every vector-memory operation of this wave has completed (stores written through)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void rw_lds_fence()  // this wave's LDS operations have completed; nothing moves across
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// one wave-wide LDS-DMA: lane i's 16 bytes at `gsrc` land at LDS byte address lds_dst + 16 * i (M0 carries the LDS base and is
// compiler-reserved: saved, set and restored inside ONE statement); counts on vmcnt, the compiler does not know about it
__device__ __forceinline__ void rw_lds_dma16(const void* gsrc, const unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// "this register holds nothing": ends the live range of a ring slot that a (uniform) condition leaves unrequested -- without it
// the slot's previous contents stay live through the conditional load and every ring of the layer loop overlaps the others
__device__ __forceinline__ void rw_kill(u32x4& x)
{
    asm volatile("" : "=v"(x));
}
__device__ __forceinline__ bool rw_give_up(int& spins, int* err, const int code)
{
    if (++spins > RW_SPIN) {
        __hip_atomic_store((__attribute__((address_space(1))) int*)err, code, RW_RLX, RW_AGT);
        return true;
    }
    return (spins & 255) == 0 && __hip_atomic_load((__attribute__((address_space(1))) int*)err, RW_RLX, RW_AGT) != 0;
}
// one wave re-reads the flags [0, n) at byte offset `off` of the hand-off region (four per lane and pass) until every one has
// reached `tag`
__device__ __forceinline__ void rw_poll(const __amdgpu_buffer_rsrc_t r, const int off, const int n, const unsigned tag, const int lane,
                                        int* err, const int code)
{
    for (int base = 0; base < n; base += 256) {
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (base + k * 64 < n) {  // (uniform)
                    const int      i = base + k * 64 + lane;
                    const unsigned v = rw_ld4(r, (i < n ? i : n - 1) * 4, off);
                    ok &= (int)(v - tag) >= 0;
                }
            }
            if (__all(ok)) {
                break;
            }
            if (rw_give_up(spins, err, code)) {
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
}
// ... the same for the n <= 256 flags whose indices sit in the LDS list `idx`
__device__ __forceinline__ void rw_poll_list(const __amdgpu_buffer_rsrc_t r, const int off, const int* idx, const int n, const unsigned tag,
                                             const int lane, int* err, const int code)
{
    int id[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = k * 64 + lane;
        id[k]       = idx[i < n ? i : n - 1];
    }
    int spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k * 64 < n) {  // (uniform)
                ok &= (int)(rw_ld4(r, id[k] * 4, off) - tag) >= 0;
            }
        }
        if (__all(ok)) {
            break;
        }
        if (rw_give_up(spins, err, code)) {
            break;
        }
        __builtin_amdgcn_s_sleep(1);
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
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void rw_lds_set(int* w, const int v, const int lane)
{
    rw_lds_fence();
    if (lane == 0) {
        *(volatile __attribute__((address_space(3))) int*)w = v;
    }
}
__device__ __forceinline__ void rw_lds_bump(int* w, const int lane)
{
    rw_lds_fence();
    if (lane == 0) {
        atomicAdd(w, 1);
    }
}

// owner of column group g when NG groups are dealt to NB workgroups in contiguous ranges [w NG / NB, (w + 1) NG / NB)
__host__ __device__ inline int rw_group_begin(const int w, const int NG, const int NB)
{
    return (int)((long)w * NG / NB);
}
// k-steps [kb, ke) of streamer wave s (0 .. RW_NS - 1) when a workgroup streams k-steps [K0, K1): contiguous shares, weighted
__host__ __device__ inline void rw_wave_ksteps(const int K0, const int K1, const int s, const int (&wcum)[8], int& kb, int& ke)
{
    const int n = K1 - K0;
    kb          = K0 + (int)(((long)n * wcum[s] + wcum[RW_NS] / 2) / wcum[RW_NS]);
    ke          = K0 + (int)(((long)n * wcum[s + 1] + wcum[RW_NS] / 2) / wcum[RW_NS]);
}
__device__ __forceinline__ int rw_sel4(const int (&a)[RW_UMAX], const int i)
{
    return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3]));
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
    // whole-vector arithmetic: four packed instructions per operation (element by element the subtraction of the broadcast mean
    // costs two scalar subtractions and a pack per pair)
    const f16   nm = -mh;  // x - mean == x + (-mean), exactly
    const f16x8 m8 = {nm, nm, nm, nm, nm, nm, nm, nm};
    const f16x8 r8 = {rh, rh, rh, rh, rh, rh, rh, rh};
    const f16x8 c  = (x + m8) * r8;
    f16x8       a  = c * g;
    a              = a + b;
    return a;
}

struct RwSmem {
    f16*   gb;     // [4][Hp]: ln1 gamma, ln1 beta, ln2 gamma, ln2 beta of the current layer (Hp = H rounded up to 512)
    float* part;   // [2][NS][RW_G][4][64] partial sums of the streamer waves (two buffers: consecutive passes alternate)
    float* apart;  // [UMAX][NS][DH + RW_PA] attention partials of the streamer waves
    f16*   att;    // [NW][UMAX][2 * DH] q | k of a wave's units (raw, then rotated in place); wave 0's region: v [UMAX][DH]
    float* stat;   // [16][2] mean, rstd of the layer input's rows
    float* scr;    // [512] scratch of the control wave
    int*   unit;   // [UMAX][8] {cached keys (-1: no such pair), row, head, input length, current token attended, -, -, -}
    int*   lst;    // [256] flags the control wave waits for before the attention (indices of workgroups)
    int*   sync;   // [0] go, [1..2] partial sums written (per buffer), [3] attention partials written
};

// the ring of a weight pass: R k-steps of G weight tiles + the rows' A fragment (shared by the passes of a wave: a
// pass's first ring is requested while the previous pass's sums are still on their way out)
// (RA <= R slots of A fragments: they come from L2 and are requested RA k-steps ahead, the weight tiles R k-steps ahead)
template<bool INT8, int R, int RA, int G>
struct RwRing {
    static_assert(R % RA == 0, "the A slots cycle inside the weight slots");
    u32x4 w[R][G];
    u32x4 a[RA][RwK<INT8>::AV];
};

// ---------------------------------------------------------------------------------------------------------------
// One pass of one wave: k-steps [kb, kb + n) of column groups [g0, g0 + ng) (ng <= G; the groups beyond ng re-read the last one and
// are never stored).  A fragments: row min(lane & 15, M - 1), k = kstep * KS + (lane >> 4) * (KS / 4) .. + KS / 4 of the
// [M][lda] halves at byte offset asoff behind `ar`.  LN: normalised on the fly with the row's statistics and gamma / beta from LDS.
// bind_w / prime_w need nothing but the weights: they run BEFORE the wait for the pass's input.
// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int G, bool LN, int R = RW_R, int RA = R>
struct RwPass {
    static constexpr int KS = RwK<INT8>::KS;
    static constexpr int AV = RwK<INT8>::AV;
    f32x4                  acc[G];
    f16x2                  sc2[G];
    __amdgpu_buffer_rsrc_t wr;       // weight image (a descriptor: the per-lane part of an address is ONE 32-bit register per group)
    unsigned               woff[G];  // byte offset of this lane's 16 bytes of group g's tile at k-step 0
    __amdgpu_buffer_rsrc_t ar;
    int                    aoff;   // byte offset of this lane's fragment at k-step 0
    int                    asoff;  // byte offset of the activations behind `ar` (uniform)
    int                    kb, n;
    f16                    mh, rh;
    const f16 *            lg, *lb;  // LDS gamma / beta (+ the lane's k offset inside a k-step)

    __device__ __forceinline__ void bind_w(const void* w, const int KT, const int NG, const int g0, const int ng, const f16* scale,
                                           const int lane, const int kb_, const int ke_)
    {
        wr = rw_rsrc(w, (size_t)NG * KT * TILE_BYTES);
        kb = kb_;
        n  = ng > 0 ? ke_ - kb_ : 0;
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
    }
    __device__ __forceinline__ void bind_a(const __amdgpu_buffer_rsrc_t ar_, const int asoff_, const int lda_bytes, const int M,
                                           const int lane, const float* stat = nullptr, const f16* gamma = nullptr,
                                           const f16* beta = nullptr)
    {
        ar            = ar_;
        asoff         = asoff_;
        const int row = (lane & 15) < M ? (lane & 15) : M - 1;
        aoff          = row * lda_bytes + (lane >> 4) * (KS / 4) * 2;
        if constexpr (LN) {
            mh = (f16)stat[row * 2];
            rh = (f16)stat[row * 2 + 1];
            lg = gamma + (lane >> 4) * (KS / 4);
            lb = beta + (lane >> 4) * (KS / 4);
        }
    }
    template<class Ring>
    __device__ __forceinline__ void load_w(Ring& q, const int r, const int k)
    {
#pragma unroll
        for (int g = 0; g < G; g++) {
            q.w[r][g] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)woff[g], k * TILE_BYTES, 2);  // (aux 2 = nt)
        }
    }
    template<class Ring>
    __device__ __forceinline__ void load_a(Ring& q, const int r, const int k)
    {
#pragma unroll
        for (int v = 0; v < AV; v++) {
            q.a[r][v] = rw_ld16(ar, aoff + v * 16, asoff + k * KS * 2);
        }
    }
    template<class Ring>
    __device__ __forceinline__ void kill_w(Ring& q, const int r)
    {
#pragma unroll
        for (int g = 0; g < G; g++) {
            rw_kill(q.w[r][g]);
        }
    }
    template<class Ring>
    __device__ __forceinline__ void kill_a(Ring& q, const int r)
    {
#pragma unroll
        for (int v = 0; v < AV; v++) {
            rw_kill(q.a[r][v]);
        }
    }
    template<class Ring>
    __device__ __forceinline__ void consume(const Ring& q, const int r, const int k)
    {
        f16x8 a0 = __builtin_bit_cast(f16x8, q.a[r % RA][0]);
        f16x8 a1 = __builtin_bit_cast(f16x8, q.a[r % RA][AV - 1]);
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
            rw_tile<INT8>(q.w[r][g], a0, a1, sc2[g], acc[g]);
        }
    }
    // the weight tiles of the first ring.  A load is never a duplicate: the ring's conditions are uniform, and only the first ring
    // and the last two rotations of a pass carry them (inside the steady rotations the compiler counts vmcnt exactly)
    template<class Ring>
    __device__ __forceinline__ void prime_w(Ring& q)
    {
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r < n) {
                load_w(q, r, kb + r);
            }
            else {
                kill_w(q, r);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    template<class Ring>
    __device__ __forceinline__ void prime_a(Ring& q)
    {
#pragma unroll
        for (int r = 0; r < RA; r++) {
            if (r < n) {
                load_a(q, r, kb + r);
            }
            else {
                kill_a(q, r);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // streams the pass (its first ring has been requested): a slot is re-requested as soon as it has been consumed
    template<class Ring>
    __device__ __forceinline__ void run(Ring& q)
    {
        if (n <= 0) {
            return;
        }
        const int nrot = (n + R - 1) / R;
        for (int it = 0; it < nrot - 2; it++) {  // every re-request of these rotations is a k-step of the pass
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i = it * R + r;
                consume(q, r, kb + i);
                __builtin_amdgcn_sched_barrier(0);
                load_w(q, r, kb + i + R);
                load_a(q, r % RA, kb + i + RA);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nrot >= 2) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int i = (nrot - 2) * R + r;
                consume(q, r, kb + i);
                __builtin_amdgcn_sched_barrier(0);
                if (i + R < n) {
                    load_w(q, r, kb + i + R);
                }
                else {
                    kill_w(q, r);
                }
                if (i + RA < n) {
                    load_a(q, r % RA, kb + i + RA);
                }
                else {
                    kill_a(q, r % RA);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int last = (nrot - 1) * R;
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (last + r < n) {
                consume(q, r, kb + last + r);
            }
            if constexpr (RA < R) {
                __builtin_amdgcn_sched_barrier(0);
                if (last + r + RA < n) {
                    load_a(q, r % RA, kb + last + r + RA);
                }
                else {
                    kill_a(q, r % RA);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the wave's partial sums -> part[s][g][j][lane] (j = accumulator register: row 4 (lane >> 4) + j, column lane & 15)
    __device__ __forceinline__ void dump(float* part, const int s, const int lane) const
    {
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                part[((s * RW_G + g) * 4 + j) * 64 + lane] = acc[g][j];
            }
        }
    }
};

// The control wave adds the streamer waves' partial sums of one pass in wave order and hands epi(row, group, half, v[8]) eight
// consecutive columns of one row at a time (columns group * 16 + half * 8 ..).
template<typename EPI>
__device__ __forceinline__ void rw_reduce(const float* part, const int ng, const int M, const int lane, EPI&& epi)
{
    const int per = ng * 2;           // items of a row
    const int rpi = 64 / per;         // rows per iteration
    const int rs = lane / per, q = lane - rs * per, g = q >> 1, h8 = q & 1;
    for (int r0 = 0; r0 < M; r0 += rpi) {
        const int r = r0 + rs;
        if (rs < rpi && r < M) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[e] = 0.f;
            }
#pragma unroll
            for (int w = 0; w < RW_NS; w++) {
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
}

__device__ __forceinline__ u32x4 rw_pack8(const f16 (&h)[8])
{
    const f16x8 v = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    return __builtin_bit_cast(u32x4, v);
}

// ---------------------------------------------------------------------------------------------------------------
// The attention stream of ONE streamer wave (decoder_masked_multihead_attention_template.hpp:1099-1919; the arithmetic of
// attn_device.hip.h mmha_partial): of every (row, head) pair of its workgroup the wave takes the 16-key blocks
// s, s + NS, s + 2 NS, ... (s = its streamer index; a block = RW_KVB wave-loads of KPI keys); the ring of RW_KVR blocks runs on
// from one pair to the next.  Lane (grp, sub) = key group lane / LPK, 16-byte piece lane % LPK of a K / V row.  Each key group keeps
// its own running {max, sum, out[8 per lane]}; at the end of a pair the groups are merged and the wave's partial goes to LDS.
// prime() needs the unit table only (not q): it runs before the wait for q | k | v.
// ---------------------------------------------------------------------------------------------------------------
template<int DH, bool PAGED>
struct RwAt {
    static constexpr int LPK = DH / 8;
    static constexpr int KPI = 64 / LPK;
    static constexpr int KB  = RW_KVB * KPI;  // keys of a block
    static_assert(DH == 64 || DH == 128, "size_per_head 64 or 128");
    struct Slot {
        u32x4 k[RW_KVB], v[RW_KVB];
    };
    Slot S[RW_KVR];
    int  nst[RW_UMAX];  // blocks of this wave per unit
    int  U, T;          // units of the workgroup, blocks of this wave over all of them
    int  lu, ls;        // loader cursor (unit, block of the unit)
    int  l_tl, l_b, l_h, l_X, l_x;
    int  s, sub, grp;
    __amdgpu_buffer_rsrc_t rk, rv;  // the layer's K / V cache (non-paged)

    __device__ __forceinline__ void bind(const RowsParams& p, const PersistLayer& lw, const int lane)
    {
        sub = lane % LPK;
        grp = lane / LPK;
        if constexpr (!PAGED) {
            const size_t bytes = (size_t)p.M * p.nh * p.s_max * DH * 2;
            rk                 = rw_rsrc(lw.k_cache, bytes);
            rv                 = rw_rsrc(lw.v_cache, bytes);
        }
    }
    // per kernel: the wave's share of every unit (the unit table is complete).  Block i of a unit's part goes to wave i % NS whatever
    // the unit's place in the workgroup: the reduction tree of a (row, head) pair -- blocks in order inside a wave, waves in order,
    // parts in order -- does not depend on where the pair lands, so identical rows of a batch give bit-identical results (dealing
    // the workgroup's blocks as one round-robin sequence balances the waves a little better and was measured neutral)
    __device__ __forceinline__ void setup(const int* unit, const int U_, const int s_)
    {
        U = U_;
        s = s_;
        T = 0;
#pragma unroll
        for (int u = 0; u < RW_UMAX; u++) {
            // part x of X of the unit's blocks (blocks j = x, x + X, ...)
            const int tl   = u < U ? rw_rfl(unit[u * 8]) : 0;
            const int X    = u < U ? rw_rfl(unit[u * 8 + 5]) : 1;
            const int x    = u < U ? rw_rfl(unit[u * 8 + 6]) : 0;
            const int nblk = tl > 0 ? (tl + KB - 1) / KB : 0;
            const int nbp  = nblk > x ? (nblk - x + X - 1) / X : 0;
            nst[u]         = nbp > s ? (nbp - s + RW_NS - 1) / RW_NS : 0;
            T += nst[u];
        }
    }
    __device__ __forceinline__ void seek(const int* unit)  // the loader's unit -> its parameters
    {
        l_tl = rw_rfl(unit[lu * 8]);
        l_b  = rw_rfl(unit[lu * 8 + 1]);
        l_h  = rw_rfl(unit[lu * 8 + 2]);
        l_X  = rw_rfl(unit[lu * 8 + 5]);
        l_x  = rw_rfl(unit[lu * 8 + 6]);
    }
    // (non-paged: the layer's caches behind two descriptors, a unit's rows at a uniform byte offset; rows_plan() keeps a layer's
    //  cache below 4 GB)
    __device__ __forceinline__ void load(const RowsParams& p, const PersistLayer& lw, Slot& q)
    {
        const int j = l_x + l_X * (s + RW_NS * ls);
        if constexpr (PAGED) {
            // a block of KB keys lies inside ONE page (rows_plan_paged(): page_tokens is a multiple of KB): its page id is a scalar
            // load -- it does not count on vmcnt, so waiting for it does not drain the ring -- and the block's rows sit at a
            // uniform 64-bit base + a 32-bit lane offset
            int t0 = j * KB;
            t0     = t0 < l_tl ? t0 : l_tl - 1;
            const int    pg   = rw_rfl(p.page_table[(size_t)l_b * p.max_pages + t0 / p.page_tokens]);
            const size_t base = ((size_t)pg * p.nh + l_h) * p.page_tokens * DH;  // elements (uniform)
            const f16*   kb   = lw.k_cache + base;
            const f16*   vb   = lw.v_cache + base;
            const int    tp0  = (j * KB) % p.page_tokens;  // the block's first key inside its page
            const int    lim  = l_tl - j * KB;              // keys of the block that exist (>= 1 for a block that is requested)
#pragma unroll
            for (int i = 0; i < RW_KVB; i++) {
                int d = i * KPI + grp;
                d     = d < lim ? d : (lim > 0 ? lim - 1 : 0);  // clamped inside the block; masked in use()
                const unsigned vo = (unsigned)((tp0 + d) * DH + sub * 8);
                q.k[i]            = __builtin_nontemporal_load(RW_GP(u32x4, kb + vo));
                q.v[i]            = __builtin_nontemporal_load(RW_GP(u32x4, vb + vo));
            }
        }
        else {
            const int so = ((l_b * p.nh + l_h) * p.s_max) * (DH * 2);  // uniform
#pragma unroll
            for (int i = 0; i < RW_KVB; i++) {
                int t = j * KB + i * KPI + grp;
                t     = t < l_tl ? t : l_tl - 1;  // clamped; masked in use()
                const int vo = t * (DH * 2) + sub * 16;
                q.k[i]       = __builtin_amdgcn_raw_buffer_load_b128(rk, vo, so, 2);  // (aux 2 = nt)
                q.v[i]       = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, so, 2);
            }
        }
    }
    __device__ __forceinline__ void kill(Slot& q)
    {
#pragma unroll
        for (int i = 0; i < RW_KVB; i++) {
            rw_kill(q.k[i]);
            rw_kill(q.v[i]);
        }
    }
    __device__ __forceinline__ void advance(const int* unit)  // the loader's next block (it stays on the last one at the end)
    {
        if (ls + 1 < rw_sel4(nst, lu)) {
            ls++;
            return;
        }
        int nu = lu + 1;
        while (nu < U && rw_sel4(nst, nu) == 0) {
            nu++;
        }
        if (nu < U) {
            lu = nu;
            ls = 0;
            seek(unit);
        }
    }
    __device__ __forceinline__ void prime(const RowsParams& p, const PersistLayer& lw, const int* unit)
    {
        if (T > 0) {
            lu = 0;
            while (rw_sel4(nst, lu) == 0) {
                lu++;
            }
            ls = 0;
            seek(unit);
#pragma unroll
            for (int r = 0; r < RW_KVR; r++) {
                if (r < T) {  // (uniform; a block is never requested twice)
                    load(p, lw, S[r]);
                    advance(unit);
                }
                else {
                    kill(S[r]);
                }
            }
        }
        else {
#pragma unroll
            for (int r = 0; r < RW_KVR; r++) {
                kill(S[r]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // q | k | v of the new token for the workgroup's units: bias (added here like the reference's MMHA), rotary; the lanes of key
    // group u stage unit u.  aw: this wave's [UMAX][2 DH] halves (q | k per unit), vw: wave 0's [UMAX][DH] (v).  The first
    // streamer also appends k / v to the cache (:1397, :1837).
    __device__ __forceinline__ void stage(const RowsParams& p, const PersistLayer& lw, const __amdgpu_buffer_rsrc_t r_ws, const int* unit,
                                          f16* aw, f16* vw)
    {
        const int  hl   = p.nh * DH;
        const int  u    = grp < U ? grp : U - 1;
        const bool mine = grp < U;
        const int  tl = unit[u * 8], b = unit[u * 8 + 1], h = unit[u * 8 + 2], cur = unit[u * 8 + 4];
        const bool kv = s == 0;  // (uniform)
        const int  qo = (b * 3 * hl + h * DH + sub * 8) * 2;
        f16x8      q8 = __builtin_bit_cast(f16x8, rw_ld16(r_ws, qo, (int)p.o_qkv)), k8 = q8, v8 = q8;
        if (kv) {
            k8 = __builtin_bit_cast(f16x8, rw_ld16(r_ws, qo + hl * 2, (int)p.o_qkv));
            v8 = __builtin_bit_cast(f16x8, rw_ld16(r_ws, qo + 2 * hl * 2, (int)p.o_qkv));
        }
        if (lw.b_qkv) {
            q8 = q8 + *RW_GP(f16x8, lw.b_qkv + h * DH + sub * 8);
            if (kv) {
                k8 = k8 + *RW_GP(f16x8, lw.b_qkv + hl + h * DH + sub * 8);
                v8 = v8 + *RW_GP(f16x8, lw.b_qkv + 2 * hl + h * DH + sub * 8);
            }
        }
        f16* qs = aw + u * 2 * DH;
        if (p.rot > 0) {
            // NeoX pairing (x[j], x[j + rot / 2]) (decoder_masked_multihead_attention_utils.h:1325-1345): through LDS, the partner
            // element may sit in another lane
            if (mine) {
                *reinterpret_cast<f16x8*>(qs + sub * 8)      = q8;
                *reinterpret_cast<f16x8*>(qs + DH + sub * 8) = k8;
            }
            const int hr = p.rot / 2;
            float     qp[8], kp[8], cs[8], sn[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int d = sub * 8 + e;
                const int j = d < hr ? d : d - hr;
                const int jj = j < hr ? j : hr - 1;
                cs[e]        = p.rot_table[((size_t)b * hr + jj) * 2];
                sn[e]        = p.rot_table[((size_t)b * hr + jj) * 2 + 1];
                const int pd = d < hr ? d + hr : (d < p.rot ? d - hr : d);
                qp[e]        = (float)qs[pd];
                kp[e]        = (float)qs[DH + pd];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int d = sub * 8 + e;
                if (d < p.rot) {
                    const float qa = (float)q8[e], ka = (float)k8[e];
                    // first of the pair: cs * a - sn * b ; second: cs * b' + sn * a' (b' = itself, a' = its partner)
                    q8[e] = d < hr ? (f16)(cs[e] * qa - sn[e] * qp[e]) : (f16)(cs[e] * qa + sn[e] * qp[e]);
                    k8[e] = d < hr ? (f16)(cs[e] * ka - sn[e] * kp[e]) : (f16)(cs[e] * ka + sn[e] * kp[e]);
                }
            }
        }
        if (mine) {
            *reinterpret_cast<f16x8*>(qs + sub * 8) = q8;
            if (kv) {
                *reinterpret_cast<f16x8*>(qs + DH + sub * 8)   = k8;
                *reinterpret_cast<f16x8*>(vw + u * DH + sub * 8) = v8;
                if (cur) {
                    size_t ro;
                    if constexpr (PAGED) {
                        const int pg = p.page_table[(size_t)b * p.max_pages + tl / p.page_tokens];
                        ro           = (((size_t)pg * p.nh + h) * p.page_tokens + (tl % p.page_tokens)) * DH + sub * 8;
                    }
                    else {
                        ro = (((size_t)b * p.nh + h) * p.s_max + tl) * DH + sub * 8;
                    }
                    *reinterpret_cast<f16x8*>(lw.k_cache + ro) = k8;
                    *reinterpret_cast<f16x8*>(lw.v_cache + ro) = v8;
                }
            }
        }
    }

    // the stream (its first ring has been requested, q | k | v are staged): partials of every unit -> apart[u][s][DH + RW_PA]
    __device__ __forceinline__ void run(const RowsParams& p, const PersistLayer& lw, const int* unit, const f16* aw, const f16* vw,
                                        float* apart)
    {
        const float inv_sqrt_dh = rsqrtf((float)DH);
        float       m = -INFINITY, l = 0.f, o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = 0.f;
        }
        int cu = 0, cs = 0, c_tl = 0, c_in = 0, c_X = 1, c_x = 0;
        auto qk = [&](const f16x8 q8, const f16x8 kv) {
            float a = 0.f;
            a       = dot2(f16x2{q8[0], q8[1]}, f16x2{kv[0], kv[1]}, a);
            a       = dot2(f16x2{q8[2], q8[3]}, f16x2{kv[2], kv[3]}, a);
            a       = dot2(f16x2{q8[4], q8[5]}, f16x2{kv[4], kv[5]}, a);
            a       = dot2(f16x2{q8[6], q8[7]}, f16x2{kv[6], kv[7]}, a);
            return group_sum_dpp<LPK>(a) * inv_sqrt_dh;
        };
        auto enter = [&]() {  // the consumer's unit -> its parameters
            c_tl = rw_rfl(unit[cu * 8]);
            c_in = rw_rfl(unit[cu * 8 + 3]);
            c_X  = rw_rfl(unit[cu * 8 + 5]);
            c_x  = rw_rfl(unit[cu * 8 + 6]);
        };
        // closes unit cu: the new token (first streamer, from LDS; :1407-1437), the key groups merged, the partial out
        auto finish = [&]() {
            if (s == 0 && rw_rfl(unit[cu * 8 + 4]) != 0) {
                const f16x8 q8 = *reinterpret_cast<const f16x8*>(aw + cu * 2 * DH + sub * 8);
                const f16x8 k8 = *reinterpret_cast<const f16x8*>(aw + cu * 2 * DH + DH + sub * 8);
                const f16x8 v8 = *reinterpret_cast<const f16x8*>(vw + cu * DH + sub * 8);
                const float a  = qk(q8, k8);
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
            float* pa = apart + (size_t)(cu * RW_NS + s) * (DH + RW_PA);
            if (grp == 0) {
                *reinterpret_cast<f32x4*>(pa + sub * 8)     = f32x4{o[0], o[1], o[2], o[3]};
                *reinterpret_cast<f32x4*>(pa + sub * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
                if (sub == 0) {
                    pa[DH]     = mw;
                    pa[DH + 1] = l;
                }
            }
            m = -INFINITY;
            l = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                o[e] = 0.f;
            }
        };
        auto use = [&](const Slot& q) {
            const f16x8 q8 = *reinterpret_cast<const f16x8*>(aw + cu * 2 * DH + sub * 8);
            const int   j  = c_x + c_X * (s + RW_NS * cs);
            float       sc[RW_KVB];
            float       mx = m;
#pragma unroll
            for (int i = 0; i < RW_KVB; i++) {
                const int   t  = j * KB + i * KPI + grp;
                const bool  ok = t < c_tl && !(t >= c_in && t < p.max_input_len);  // padding keys: probability 0 (:1570)
                const float a  = qk(q8, __builtin_bit_cast(f16x8, q.k[i]));
                sc[i]          = ok ? a : -INFINITY;
                mx             = fmaxf(mx, sc[i]);
            }
            if (mx != -INFINITY) {  // (uniform inside the key group: otherwise nothing to add yet)
                const float f = __expf(m - mx);  // m = -inf: 0
                l *= f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    o[e] *= f;
                }
                m = mx;
#pragma unroll
                for (int i = 0; i < RW_KVB; i++) {
                    const float pt = (sc[i] == -INFINITY) ? 0.f : __expf(sc[i] - mx);
                    const f16x8 vv = __builtin_bit_cast(f16x8, q.v[i]);
                    l += pt;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        // (rows beyond the unit were fetched as copies of its last row: pt is exactly 0 for them)
                        o[e] = fmaf(pt, (float)vv[e], o[e]);
                    }
                }
            }
        };
        // units without a block of this wave in front
        while (cu < U && rw_sel4(nst, cu) == 0) {
            finish();
            cu++;
        }
        if (cu < U) {
            enter();
        }
        auto step = [&](const Slot& q) {  // one block of the consumer's unit; closes the units that end with it
            use(q);
            cs++;
            if (cs == rw_sel4(nst, cu)) {
                finish();
                cu++;
                cs = 0;
                while (cu < U && rw_sel4(nst, cu) == 0) {
                    finish();
                    cu++;
                }
                if (cu < U) {
                    enter();
                }
            }
        };
        int used = 0, asked = T < RW_KVR ? T : RW_KVR;
        while (asked + RW_KVR <= T) {  // steady rotations: every re-request is a block of the stream
#pragma unroll
            for (int r = 0; r < RW_KVR; r++) {
                step(S[r]);
                __builtin_amdgcn_sched_barrier(0);
                load(p, lw, S[r]);
                advance(unit);
                __builtin_amdgcn_sched_barrier(0);
            }
            used += RW_KVR;
            asked += RW_KVR;
        }
        while (used < T) {  // the last rotations: what is left to request, then nothing
#pragma unroll
            for (int r = 0; r < RW_KVR; r++) {
                if (used + r < T) {
                    step(S[r]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (asked < T) {
                    load(p, lw, S[r]);
                    advance(unit);
                    asked++;
                }
                else {
                    kill(S[r]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            used += RW_KVR;
        }
    }
};

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
    const int Hp = (H + 511) & ~511;
    const int CB = p.plan.CB, KP2 = p.plan.KP2, KP3 = p.plan.KP3, CR = p.plan.CR, cw = p.plan.cw;
    const int NM = M * CR;  // mergers
    const int      step     = *p.d_step;
    const unsigned tag_base = (unsigned)step * 256u + 1u;

    RwSmem s;
    {
        char* q = smem;
        s.gb    = reinterpret_cast<f16*>(q);
        q += (size_t)4 * Hp * 2;
        s.part = reinterpret_cast<float*>(q);
        q += (size_t)2 * RW_NS * RW_G * 256 * 4;
        s.apart = reinterpret_cast<float*>(q);
        q += (size_t)RW_UMAX * RW_NS * (DH + RW_PA) * 4;
        s.att = reinterpret_cast<f16*>(q);
        q += (size_t)RW_NW * RW_UMAX * 2 * DH * 2;
        s.stat = reinterpret_cast<float*>(q);
        q += 32 * 4;
        s.scr = reinterpret_cast<float*>(q);
        q += 512 * 4;
        s.unit = reinterpret_cast<int*>(q);
        q += RW_UMAX * 8 * 4;
        s.lst = reinterpret_cast<int*>(q);
        q += 256 * 4;
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
    // attention: the (row, head) pairs wg, wg + NB, ... of this workgroup
    // attention: `ufull` whole (row, head) pairs per workgroup (pair = wg + u * NB); each of the `rem` leftover pairs is shared by FX
    // workgroups (part x takes its key blocks x, x + FX, ...; part 0 merges the parts and owns the pair's context)
    const int  npairs = M * p.nh;
    const int  ufull = npairs / NB, rem = npairs - ufull * NB, FX = p.plan.FX;
    const bool has_frac = wg < rem * FX;
    const int  U   = ufull + (has_frac ? 1 : 0);
    const int  npw = ufull > 0 ? NB : rem * FX;  // workgroups that own units

    const __amdgpu_buffer_rsrc_t r_ws = rw_rsrc(p.ws, p.ws_bytes), r_xin = rw_rsrc(p.x_in, (size_t)M * H * 2);

    // ---- kernel start: statistics of x_in's rows, the first layer's LayerNorm parameters, the unit table -- by everybody ----
    {
        const PersistLayer& lw = p.layers[p.l_begin];
        for (int i = tx; i < H / 8; i += RW_NT) {
            reinterpret_cast<u32x4*>(s.gb)[i]                  = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln1_g) + i);
            reinterpret_cast<u32x4*>(s.gb + Hp)[i]             = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln1_b) + i);
            reinterpret_cast<u32x4*>(s.gb + 2 * Hp)[i]         = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln2_g) + i);
            reinterpret_cast<u32x4*>(s.gb + 3 * (size_t)Hp)[i] = *RW_GP(u32x4, reinterpret_cast<const u32x4*>(lw.ln2_b) + i);
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
        if (tx < RW_UMAX) {
            // units of this workgroup: `ufull` whole pairs (pair = wg + u * NB), then -- on the first rem * X workgroups -- part
            // wg % X of leftover pair ufull * NB + wg / X
            const bool frac = tx == ufull && has_frac;
            const bool on   = tx < U;
            const int  pair = frac ? ufull * NB + wg / FX : wg + tx * NB;
            const int  b = on ? pair / p.nh : 0, h = on ? pair - b * p.nh : 0;
            const bool fin = p.finished && p.finished[b];
            const int  X = frac ? FX : 1, x = frac ? wg % FX : 0;
            s.unit[tx * 8]     = on ? (fin ? 0 : p.seq_len[b]) : -1;  // cached keys; the new token goes to index seq_len
            s.unit[tx * 8 + 1] = b;
            s.unit[tx * 8 + 2] = h;
            s.unit[tx * 8 + 3] = p.input_lengths ? p.input_lengths[b] : 0x7fffffff;
            s.unit[tx * 8 + 4] = (on && !fin && x == 0) ? 1 : 0;  // the new token: part 0
            s.unit[tx * 8 + 5] = X;
            s.unit[tx * 8 + 6] = x;
        }
        if (tx >= 64 && tx < 72) {
            s.sync[tx - 64] = 0;
        }
        __syncthreads();
    }
    auto part_of = [&](const int buf) { return s.part + (size_t)buf * RW_NS * RW_G * 256; };

    if (wid != 0) {
        // =========================================== streamer waves ===========================================
        const int sw = wid - 1;
        int       kb1, ke1, kb2, ke2, kb3, ke3;
        rw_wave_ksteps(0, KT1, sw, p.plan.wcum, kb1, ke1);
        rw_wave_ksteps(k2a, k2b, sw, p.plan.wcum, kb2, ke2);
        rw_wave_ksteps(k3a, k3b, sw, p.plan.wcum, kb3, ke3);
        // ONE ring object for every pass of the wave (the QKV pass uses RwQ<G1>::R slots of G1 tiles, the others RW_R slots of
        // RW_G; what a pass does not use is never defined and costs no register)
        RwRing<INT8, (RwQ<G1>::R > RW_R ? RwQ<G1>::R : RW_R), RW_R, RW_G> ring;
        RwAt<DH, PAGED>   at;
        f16*              aw = s.att + (size_t)wid * RW_UMAX * 2 * DH;
        f16*              vw = s.att;  // (wave 0 does not stream: its region holds v of the first streamer)
        at.setup(s.unit, U, sw);
        RwPass<INT8, G1, true, RwQ<G1>::R, RwQ<G1>::RA> pq;
        {
            const PersistLayer& lw = p.layers[p.l_begin];
            pq.bind_w(lw.w_qkv, KT1, NGq, q0, nq, lw.s_qkv, lane, kb1, ke1);
            pq.prime_w(ring);
        }
        for (int l = p.l_begin; l < p.l_end; l++) {
            // (the lane index is laundered once per layer: what a pass derives from it is recomputed here instead of being hoisted out
            //  of the layer loop and kept -- spilled -- across all five streams)
            asm volatile("" : "+v"(tx));
            const int           ln = tx & 63;
            const PersistLayer& lw = p.layers[l];
            const int  li   = l - p.l_begin;
            const bool last = l == p.l_end - 1;
            // the layer input: x_in or the hand-off region's xb[l & 1]
            const __amdgpu_buffer_rsrc_t r_x = li == 0 ? r_xin : r_ws;
            const int                    xso = li == 0 ? 0 : (int)p.o_xb[l & 1];
            stamp(l, 0);
            rw_lds_wait(&s.sync[0], li * RW_PHASES + 1, p.err, 10);
            stamp(l, 1);
            // ---- QKV ----
            pq.bind_a(r_x, xso, H * 2, M, ln, s.stat, s.gb, s.gb + Hp);
            pq.prime_a(ring);
            pq.run(ring);
            RwPass<INT8, RW_G, true> pf;
            pf.bind_w(lw.w_ffn1, KT1, NGf, f0, nf, lw.s_ffn1, ln, kb1, ke1);
            pf.prime_w(ring);
            pq.dump(part_of(0), sw, ln);
            rw_lds_bump(&s.sync[1], ln);
            stamp(l, 2);
            // ---- FFN1 (needs nothing new) ----
            pf.bind_a(r_x, xso, H * 2, M, ln, s.stat, s.gb + 2 * Hp, s.gb + 3 * (size_t)Hp);
            pf.prime_a(ring);
            pf.run(ring);
            at.bind(p, lw, ln);
            at.prime(p, lw, s.unit);
            pf.dump(part_of(1), sw, ln);
            rw_lds_bump(&s.sync[2], ln);
            stamp(l, 3);
            // ---- attention ----
            rw_lds_wait(&s.sync[0], li * RW_PHASES + 2, p.err, 11);
            stamp(l, 4);
            if (U > 0) {
                at.stage(p, lw, r_ws, s.unit, aw, vw);
                at.run(p, lw, s.unit, aw, vw, s.apart);
            }
            RwPass<INT8, RW_G, false> p2;
            p2.bind_w(lw.w_ffn2, KT2, NGo, cb2 * RW_G, ng2, lw.s_ffn2, ln, kb2, ke2);
            p2.prime_w(ring);
            rw_lds_bump(&s.sync[3], ln);
            stamp(l, 5);
            // ---- FFN2 ----
            rw_lds_wait(&s.sync[0], li * RW_PHASES + 3, p.err, 12);
            stamp(l, 6);
            p2.bind_a(r_ws, (int)p.o_mid, Il * 2, M, ln);
            p2.prime_a(ring);
            p2.run(ring);
            RwPass<INT8, RW_G, false> p3;
            p3.bind_w(lw.w_out, KT3, NGo, cb3 * RW_G, ng3, lw.s_out, ln, kb3, ke3);
            p3.prime_w(ring);
            p2.dump(part_of(0), sw, ln);
            rw_lds_bump(&s.sync[1], ln);
            stamp(l, 7);
            // ---- out-proj ----
            rw_lds_wait(&s.sync[0], li * RW_PHASES + 4, p.err, 13);
            stamp(l, 8);
            p3.bind_a(r_ws, (int)p.o_ctx, Hl * 2, M, ln);
            p3.prime_a(ring);
            p3.run(ring);
            if (!last) {
                const PersistLayer& nx = p.layers[l + 1];
                pq.bind_w(nx.w_qkv, KT1, NGq, q0, nq, nx.s_qkv, ln, kb1, ke1);
                pq.prime_w(ring);
            }
            p3.dump(part_of(1), sw, ln);
            rw_lds_bump(&s.sync[2], ln);
            stamp(l, 9);
        }
        return;
    }
    // =============================================== control wave ===============================================
    const unsigned gb_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s.gb;
    // the LayerNorm parameters of layer `nx` by LDS-DMA: arrays a0, a0 + 1 of gb (1 KiB per request; the last chunk re-reads the
    // array's last 16 bytes into the padding behind it)
    auto ln_dma = [&](const f16* g, const f16* b, const int a0) {
        const int n16 = H / 8;
        for (int c = 0; c * 64 < n16; c++) {
            const int i = c * 64 + lane < n16 ? c * 64 + lane : n16 - 1;
            rw_lds_dma16(reinterpret_cast<const u32x4*>(g) + i, (unsigned)rw_rfl((int)(gb_lds + (unsigned)(a0 * Hp * 2 + c * 1024))));
            rw_lds_dma16(reinterpret_cast<const u32x4*>(b) + i,
                         (unsigned)rw_rfl((int)(gb_lds + (unsigned)((a0 + 1) * Hp * 2 + c * 1024))));
        }
    };
    // What this workgroup's NEXT stream really needs, instead of every workgroup's flag (a global wait couples the stream to the chip's
    // slowest workgroup three times per layer):
    //   listA -- before the attention: the owners of the q | k | v column groups of this workgroup's (row, head) pairs;
    //   FFN2 needs the mid columns of its K piece: a contiguous range of FFN1 owners (no list);
    //   out-proj keeps the wait for every pair's context: the owners of the heads inside its K piece alone (160 flags by list)
    //   were measured 5 us per layer SLOWER (3650 against 3445 us per launch; profiles/r05_notes.md).
    // owner of column group g when NG groups are dealt in contiguous ranges (rw_group_begin): floor(((g + 1) NB - 1) / NG)
    auto owner = [&](const int g, const int NG) { return (int)((((long)g + 1) * NB - 1) / NG); };
    int nA = 0;
    {
        int* lA = s.lst;
        // listA (U <= 4 pairs x three ranges of DH / 16 groups)
        bool okA = U > 0;
        for (int u = 0; u < U && okA; u++) {
            const int h = s.unit[u * 8 + 2];
            for (int part = 0; part < 3 && okA; part++) {
                const int g0 = (part * Hl + h * DH) / 16, g1 = g0 + DH / 16 - 1;
                const int w0 = owner(g0, NGq), w1 = owner(g1, NGq);
                if (nA + (w1 - w0 + 1) > 256) {
                    okA = false;
                    break;
                }
                for (int w = w0 + lane; w <= w1; w += 64) {
                    lA[nA + w - w0] = w;
                }
                nA += w1 - w0 + 1;
            }
        }
        if (!okA) {
            nA = 0;  // (the full poll)
        }
        rw_lds_fence();
    }
    // FFN2's K piece [k2a, k2b) k-steps = mid columns [k2a KS, k2b KS): FFN1 column groups and their owners
    int fm_lo = (has2 && k2b > k2a) ? owner((k2a * KS) / 16, NGf) : 0;
    int fm_hi = (has2 && k2b > k2a) ? owner((k2b * KS - 1) / 16, NGf) : -1;
    for (int l = p.l_begin; l < p.l_end; l++) {
        const PersistLayer& lw  = p.layers[l];
        const unsigned      tag = tag_base + (unsigned)l;
        const int           li  = l - p.l_begin;
        const bool          first = li == 0, last = l == p.l_end - 1;
        stamp(l, 0);
        if (!first) {
            // the layer input is complete when every merger's granule carries the previous layer's tag; the same loads bring the
            // rows' statistics: the mergers' partials summed in column-range order
            const int xo    = (int)p.o_xs + (l & 1) * NM * 16;
            int       spins = 0;
            u32x4     g[4];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k * 64 < NM) {  // (uniform)
                        const int i = k * 64 + lane;
                        g[k]        = rw_ld16(r_ws, (i < NM ? i : NM - 1) * 16, xo);
                        ok &= (int)(g[k].x - (tag - 1u)) >= 0;
                    }
                }
                if (__all(ok)) {
                    break;
                }
                if (rw_give_up(spins, p.err, 1)) {
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k * 64 < NM && k * 64 + lane < NM) {
                    s.scr[(k * 64 + lane) * 2]     = __uint_as_float(g[k].y);
                    s.scr[(k * 64 + lane) * 2 + 1] = __uint_as_float(g[k].z);
                }
            }
            rw_lds_fence();
            if (lane < M) {
                float t0 = 0.f, t1 = 0.f;
                for (int c = 0; c < CR; c++) {
                    t0 += s.scr[(lane * CR + c) * 2];
                    t1 += s.scr[(lane * CR + c) * 2 + 1];
                }
                const float mean     = t0 / (float)H;
                s.stat[lane * 2]     = mean;
                s.stat[lane * 2 + 1] = rsqrtf(t1 / (float)H - mean * mean + p.eps);
            }
        }
        rw_drain();  // (the LayerNorm parameters requested during the previous layer have landed)
        rw_lds_set(&s.sync[0], li * RW_PHASES + 1, lane);
        stamp(l, 1);
        // ---- QKV: the waves' sums -> q | k | v (no bias: the attention adds it, like the reference's MMHA) ----
        rw_lds_wait(&s.sync[1], (2 * li + 1) * RW_NS, p.err, 21);
        stamp(l, 2);
        if (nq > 0) {
            rw_reduce(part_of(0), nq, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                f16 hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    hv[e] = (f16)v[e];
                }
                rw_st16(rw_pack8(hv), r_ws, (r * 3 * Hl + (q0 + g) * 16 + h8 * 8) * 2, (int)p.o_qkv);
            });
        }
        rw_drain();
        if (lane == 0) {
            rw_st4(tag, r_ws, wg * 4, (int)p.o_fq);
        }
        stamp(l, 3);
        if (!last) {  // the next layer's ln1 parameters (every streamer of this workgroup is past its QKV pass)
            ln_dma(p.layers[l + 1].ln1_g, p.layers[l + 1].ln1_b, 0);
        }
        // q | k | v of every producer (they travel under the FFN1 stream): the attention may start
        if (nA > 0) {
            rw_poll_list(r_ws, (int)p.o_fq, s.lst, nA, tag, lane, p.err, 2);
        }
        else if (U > 0) {
            rw_poll(r_ws, (int)p.o_fq, NB, tag, lane, p.err, 2);
        }
        rw_lds_set(&s.sync[0], li * RW_PHASES + 2, lane);
        stamp(l, 4);
        // ---- FFN1: mid = gelu(. + bias) ----
        rw_lds_wait(&s.sync[2], (2 * li + 1) * RW_NS, p.err, 22);
        if (nf > 0) {
            rw_reduce(part_of(1), nf, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
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
                rw_st16(rw_pack8(hv), r_ws, (r * Il + col) * 2, (int)p.o_mid);
            });
        }
        rw_drain();
        if (lane == 0) {
            rw_st4(tag, r_ws, wg * 4, (int)p.o_fm);
        }
        stamp(l, 5);
        if (!last) {
            ln_dma(p.layers[l + 1].ln2_g, p.layers[l + 1].ln2_b, 2);
        }
        // mid of every producer (it travels under the attention): FFN2 may start
        if (fm_hi >= fm_lo) {
            rw_poll(r_ws, (int)p.o_fm + fm_lo * 4, fm_hi - fm_lo + 1, tag, lane, p.err, 3);
        }
        rw_lds_set(&s.sync[0], li * RW_PHASES + 3, lane);
        stamp(l, 6);
        // ---- the streamers' attention partials -> ctx of this workgroup's pairs, merged in wave order
        //      (attn_device.hip.h mmha_block's merger; :1632) ----
        rw_lds_wait(&s.sync[3], (li + 1) * RW_NS, p.err, 23);
        if (U > 0) {
            constexpr int LPK = DH / 8;
            const int     u = lane / LPK, sub = lane % LPK;
            const bool    mine = u < U;
            const bool    fr = mine && has_frac && u == ufull;  // the lanes of the shared pair's part
            float         mx = -INFINITY, L = 0.f, o[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                o[e] = 0.f;
            }
            if (mine) {
                const float* pa = s.apart + (size_t)u * RW_NS * (DH + RW_PA);
#pragma unroll
                for (int w = 0; w < RW_NS; w++) {
                    mx = fmaxf(mx, pa[w * (DH + RW_PA) + DH]);
                }
#pragma unroll
                for (int w = 0; w < RW_NS; w++) {
                    const float ms = pa[w * (DH + RW_PA) + DH], ls = pa[w * (DH + RW_PA) + DH + 1];
                    const float ww = (ms == -INFINITY) ? 0.f : __expf(ms - mx);
                    L += ww * ls;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(pa + w * (DH + RW_PA) + sub * 8);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(pa + w * (DH + RW_PA) + sub * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        o[e] += ww * a[e];
                        o[4 + e] += ww * c[e];
                    }
                }
            }
            const int  fslot = wg;  // (has_frac: leftover pair wg / FX, part wg % FX -- the slots of one pair are neighbours)
            const int  po    = fslot * (DH + RW_PA) * 4;
            const bool guest = has_frac && FX > 1 && wg % FX != 0;  // a part that is not the pair's owner
            const bool owner = has_frac && FX > 1 && wg % FX == 0;
            if (guest) {
                // its merged {out, max, sum} travels to part 0
                if (fr) {
                    rw_st16(__builtin_bit_cast(u32x4, f32x4{o[0], o[1], o[2], o[3]}), r_ws, po + sub * 32, (int)p.o_pa);
                    rw_st16(__builtin_bit_cast(u32x4, f32x4{o[4], o[5], o[6], o[7]}), r_ws, po + sub * 32 + 16, (int)p.o_pa);
                    if (sub == 0) {
                        rw_st16(__builtin_bit_cast(u32x4, f32x4{mx, L, 0.f, 0.f}), r_ws, po + DH * 4, (int)p.o_pa);
                    }
                }
                rw_drain();
                if (lane == 0) {
                    rw_st4(tag, r_ws, fslot * 4, (int)p.o_fa);
                }
            }
            if (owner) {
                // the other parts in part order (they publish without waiting for anybody)
                rw_poll(r_ws, (int)p.o_fa + (fslot + 1) * 4, FX - 1, tag, lane, p.err, 4);
                for (int x = 1; x < FX; x++) {
                    const int   qo = (fslot + x) * (DH + RW_PA) * 4;
                    const f32x4 a  = __builtin_bit_cast(f32x4, rw_ld16(r_ws, qo + sub * 32, (int)p.o_pa));
                    const f32x4 c  = __builtin_bit_cast(f32x4, rw_ld16(r_ws, qo + sub * 32 + 16, (int)p.o_pa));
                    const f32x4 ml = __builtin_bit_cast(f32x4, rw_ld16(r_ws, qo + DH * 4, (int)p.o_pa));
                    if (fr) {
                        const float mn = fmaxf(mx, ml[0]);
                        const float wa = (mx == -INFINITY) ? 0.f : __expf(mx - mn);
                        const float wb = (ml[0] == -INFINITY) ? 0.f : __expf(ml[0] - mn);
                        L              = wa * L + wb * ml[1];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            o[e]     = wa * o[e] + wb * a[e];
                            o[4 + e] = wa * o[4 + e] + wb * c[e];
                        }
                        mx = mn;
                    }
                }
            }
            if (mine && !(fr && guest)) {
                const float inv = 1.f / (L + 1.e-6f);
                f16         hv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    hv[e] = (f16)(o[e] * inv);
                }
                const int b = s.unit[u * 8 + 1], h = s.unit[u * 8 + 2];
                rw_st16(rw_pack8(hv), r_ws, (b * Hl + h * DH + sub * 8) * 2, (int)p.o_ctx);
            }
            rw_drain();
            if (lane == 0) {
                rw_st4(tag, r_ws, wg * 4, (int)p.o_fc);
            }
        }
        stamp(l, 7);
        // the context of every pair (it travels under the FFN2 stream): out-proj may start
        rw_poll(r_ws, (int)p.o_fc, npw, tag, lane, p.err, 5);
        rw_lds_set(&s.sync[0], li * RW_PHASES + 4, lane);
        stamp(l, 8);
        // ---- FFN2: the waves' sums -> fp32 partial of this K piece ----
        rw_lds_wait(&s.sync[1], (2 * li + 2) * RW_NS, p.err, 24);
        if (ng2 > 0) {
            rw_reduce(part_of(0), ng2, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                const int o = ((kp2 * M + r) * H + (cb2 * RW_G + g) * 16 + h8 * 8) * 4;
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r_ws, o, (int)p.o_p2);
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r_ws, o + 16, (int)p.o_p2);
            });
        }
        if (has2) {
            rw_drain();
            if (lane == 0) {
                rw_st4(tag, r_ws, wg * 4, (int)p.o_f2);
            }
        }
        stamp(l, 9);
        // ---- out-proj: partial of this K piece ----
        rw_lds_wait(&s.sync[2], (2 * li + 2) * RW_NS, p.err, 25);
        if (ng3 > 0) {
            rw_reduce(part_of(1), ng3, M, lane, [&](const int r, const int g, const int h8, const float(&v)[8]) {
                const int o = ((kp3 * M + r) * H + (cb3 * RW_G + g) * 16 + h8 * 8) * 4;
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), r_ws, o, (int)p.o_p3);
                rw_st16(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), r_ws, o + 16, (int)p.o_p3);
            });
        }
        if (has3) {
            rw_drain();
            if (lane == 0) {
                rw_st4(tag, r_ws, wg * 4, (int)p.o_f3);
            }
        }
        stamp(l, 10);
        // ---- x' = residual(x, attention pieces, FFN pieces, bias) for columns [c cw, (c + 1) cw) of row r: merger r * CR + c
        //      (invokeAddBiasAttentionFfnResidual, add_residual_kernels.cu:116-178) + the row's partial statistics ----
        if (wg < NM) {
            const int r = wg / CR, c = wg - r * CR;
            const int cb_lo = (c * cw) / (RW_G * 16), cb_hi = ((c + 1) * cw - 1) / (RW_G * 16);
            rw_poll(r_ws, (int)p.o_f2 + cb_lo * KP2 * 4, (cb_hi - cb_lo + 1) * KP2, tag, lane, p.err, 6);
            rw_poll(r_ws, (int)p.o_f3 + cb_lo * KP3 * 4, (cb_hi - cb_lo + 1) * KP3, tag, lane, p.err, 7);
            // layer_input/output alias for 0 < l < L-1 in the reference (GptNeoXDecoder.cc:249-250) -> residual form
            const int                    inplace = (l > 0 && l < p.L - 1) ? 1 : 0;
            const __amdgpu_buffer_rsrc_t r_x     = first ? r_xin : r_ws;
            const int                    xso     = first ? 0 : (int)p.o_xb[l & 1];
            float                        q0s = 0.f, q1s = 0.f;
            for (int oc = lane; oc < cw / 8; oc += 64) {
                const int col = c * cw + oc * 8;
                float     sa[8], sb[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    sa[e] = sb[e] = 0.f;
                }
                for (int j = 0; j < KP3; j++) {  // piece order: deterministic
                    const f32x4 a = __builtin_bit_cast(f32x4, rw_ld16(r_ws, ((j * M + r) * H + col) * 4, (int)p.o_p3));
                    const f32x4 d = __builtin_bit_cast(f32x4, rw_ld16(r_ws, ((j * M + r) * H + col) * 4 + 16, (int)p.o_p3));
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sa[e] += a[e];
                        sa[4 + e] += d[e];
                    }
                }
                for (int j = 0; j < KP2; j++) {
                    const f32x4 a = __builtin_bit_cast(f32x4, rw_ld16(r_ws, ((j * M + r) * H + col) * 4, (int)p.o_p2));
                    const f32x4 d = __builtin_bit_cast(f32x4, rw_ld16(r_ws, ((j * M + r) * H + col) * 4 + 16, (int)p.o_p2));
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sb[e] += a[e];
                        sb[4 + e] += d[e];
                    }
                }
                const f16x8 xv = __builtin_bit_cast(f16x8, rw_ld16(r_x, (r * H + col) * 2, xso));
                const f16x8 bv = *RW_GP(f16x8, lw.b_res + col);
                f16         hv[8];
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
                if (last) {
                    *reinterpret_cast<u32x4*>(p.x_out + (size_t)r * H + col) = rw_pack8(hv);
                }
                else {
                    rw_st16(rw_pack8(hv), r_ws, (r * H + col) * 2, (int)p.o_xb[(l + 1) & 1]);
                }
            }
            if (!last) {
                q0s = wave_sum(q0s);
                q1s = wave_sum(q1s);
                rw_drain();
                if (lane == 0) {
                    rw_st16(u32x4{tag, __float_as_uint(q0s), __float_as_uint(q1s), 0u}, r_ws, wg * 16,
                            (int)p.o_xs + ((l + 1) & 1) * NM * 16);
                }
            }
        }
        stamp(l, 11);
    }
}

}  // namespace ftcf
