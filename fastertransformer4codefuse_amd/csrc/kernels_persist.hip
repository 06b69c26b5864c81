// Persistent decode layers (m <= 2 rows): ONE launch runs layers [l_begin, l_end) of the decode step of
// GptNeoXDecoder<T>::forward (models/gptneox/GptNeoXDecoder.cc:245-384) on one resident 8-wave workgroup per CU.
//
// Why: as separate launches every stage pays its own ramp and tail (the last waves of a grid stream alone while the rest
// of the chip idles) and the attention's latency chain sits on the critical path; measured 25 of 75 us per layer.  Inside
// one launch the weight stream of the NEXT stage is already in flight (four register batches per wave, 32 KiB) while the
// vectors of the previous stage are handed over, so HBM stays busy across the dependency edges.
//
// Stages of a layer (all workgroups run all stages, SPMD):
//   S0  gather the layer input x (granules published by the previous layer's mergers), LN1 / LN2
//   P1  stream [QKV u FFN1] column groups -> qkv (no bias, like the reference's MMHA), mid = gelu(.+b)
//   AT  split-KV attention of one (row, head, split) per workgroup -> ctx
//   P3  stream [FFN2 pieces, then out-proj pieces] -> fp32 partials; the owner of a group's last out-proj piece merges
//       them in fixed order, applies invokeAddBiasAttentionFfnResidual and publishes x' for the next layer
// Every hand-off is made of 8-byte {tag, value} GRANULES (cdna_hip_programming.md G16 recipe R2): one relaxed agent-scope
// (sc1, write-through) store per granule, consumers re-read until every tag matches -- no flag, fence, drain or counter
// (a first version with drained counters spent 15 us per edge in the 256 -> 1 fan-in and its pollers).  A consumer only
// sweeps the slice it needs: an attention workgroup its head's q/k/v (192 granules), a P3 workgroup the K range of its
// pieces (mid: 1280, ctx: 640 granules at CodeFuse-13B), every workgroup the layer input x (2560).
// Waves 0..1 are "control" waves: they do the latency-critical sweeps and therefore start their weight prefetch last;
// a wave's memory returns are in order, so a sweep issued behind 32 KiB of prefetch would wait for all of it.  For the
// same reason the per-layer constants (scales, biases, LayerNorm parameters) are fetched into registers one stage
// ahead, before that stage's prefetch is issued.
// The work split is static: a workgroup owns whole runs (a column group, or a K piece of one), its waves cut the runs'
// tiles back to back into contiguous shares, and per-wave tile tables (built once per launch in LDS) drive the stream.
// Every spin is bounded (PS_SPIN) and reports through PersistParams::err instead of hanging the GPU.
#include <type_traits>

#include "attn_device.cuh"
#include "gemv_device.cuh"

namespace ftcf {

constexpr int PS_NW       = 8;           // waves per workgroup (2 per SIMD -> 256 VGPRs each)
constexpr int PS_NT       = PS_NW * 64;
constexpr int PS_NC       = 2;           // control waves
constexpr int PS_U        = 8;           // tiles per register batch
constexpr int PS_NBUF     = 4;           // register batches per wave (3 in flight while one is consumed)
constexpr int PS_RMAX     = 24;          // runs per workgroup and stage
constexpr int PS_MAXMERGE = 8;           // groups merged per workgroup
constexpr int PS_MAXP     = 16;          // PA + PB
constexpr int PS_SPIN     = 1 << 18;
constexpr int PS_UK       = 8;           // attention: K (and V) wave-loads per lane (256 keys per workgroup at dh = 128) ...
constexpr int PS_UK_LONG  = 16;          // ... or 16 (512 keys) for requests whose KV split does not fit 256.  The rows live
                                         // in the weight stream's four register batches, which are idle during the attention
#ifndef PS_FULL_P1_V
#define PS_FULL_P1_V false
#endif
#ifndef PS_FULL_P3_V
#define PS_FULL_P3_V false
#endif
// streamer waves: issue the whole first rotation (32 KiB) before the hand-off instead of half of it
constexpr bool PS_FULL_P1 = PS_FULL_P1_V, PS_FULL_P3 = PS_FULL_P3_V;
constexpr int PS_NLN      = 2;           // LayerNorm parameter vectors (f16x8) per thread and array: H <= 8192

#define PS_RLX __ATOMIC_RELAXED
#define PS_AGT __HIP_MEMORY_SCOPE_AGENT
// pointers that come out of the per-layer table in memory are GLOBAL: say so (a flat access also counts on lgkmcnt)
#define PS_G(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))

// Batch descriptor (one per PS_U consecutive k tiles of ONE run; 8 bytes in LDS):
//   low  word : index of the first tile inside its weight array
//   high word : bit 0 weight array of the stage, 1 x row stride select, 2 wait for the late x vector (ctx), 3..7 run,
//               8..24 LDS half offset of the first tile's x, 25..28 valid tiles (0: padding batch, never consumed)
constexpr unsigned PS_BD_SEL = 1u, PS_BD_XSEL = 2u, PS_BD_WAIT = 4u;

struct RunRec {  // static per launch (LDS)
    int tile0;  // first tile of the run inside its weight array
    int sel;    // weight array of the stage (0 / 1)
    int nt;     // tiles
    int xoff;   // LDS half offset (inside the x region) of the run's first k
    int xsel;   // x row stride select
    int rid;    // stage specific id (P1: combined group, P3: global piece id)
    int grp;    // 16-column group
    int b0;     // first batch of the run in the stage's batch queue (the run has (nt + PS_U - 1) / PS_U batches)
    // how the run's result is published (static per launch, so that the publishing code inside the stream loop needs no
    // kernel parameters: everything it captures would stay live -- and spill -- across the whole loop)
    u64* dst;      // granule of (row 0, column 0)
    int  mstride;  // granules between the rows of x (pairs of halves) -- unused for fp32 pieces
    int  kind;     // PS_PUB_*
};
constexpr int PS_PUB_PLAIN = 0;  // f16(y), pairs of halves          (qkv: the attention adds the bias)
constexpr int PS_PUB_GELU  = 1;  // gelu(y + bias), pairs of halves  (mid)
constexpr int PS_PUB_F32   = 2;  // fp32 partial sums, one granule per value, [M][16] (K pieces)

__device__ __forceinline__ int ps_rfl(int v)
{
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void st_granule_u32(u64* g, unsigned tag, unsigned v)
{
    __hip_atomic_store((gu64*)g, ((u64)tag << 32) | (u64)v, PS_RLX, PS_AGT);
}
__device__ __forceinline__ unsigned short f16_bits(f16 v)
{
    return __builtin_bit_cast(unsigned short, v);
}
__device__ __forceinline__ f16 bits_f16(unsigned v)
{
    return __builtin_bit_cast(f16, (unsigned short)(v & 0xffffu));
}
__device__ __forceinline__ bool ps_give_up(int& spins, int* err, const int code)
{
    if (++spins > PS_SPIN) {
        __hip_atomic_store((__attribute__((address_space(1))) int*)err, code, PS_RLX, PS_AGT);
        return true;
    }
    return (spins & 255) == 0 && __hip_atomic_load((__attribute__((address_space(1))) int*)err, PS_RLX, PS_AGT) != 0;
}
// `nthr` threads (whole waves, tid = 0..nthr-1) re-read granules [0, n) of `g` until every tag matches, NPER granules
// per thread and pass, and hand the 32-bit payloads to sink(index, value)
template<int NPER, typename F>
__device__ __forceinline__ void ps_sweep(const u64* g, const int n, const int tid, const int nthr, const unsigned tag,
                                         int* err, const int code, F&& sink)
{
    for (int base = 0; base < n; base += nthr * NPER) {
        u64 gv[NPER];
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NPER; k++) {
                const int i = base + k * nthr + tid;
                gv[k]       = ld_granule(&g[i < n ? i : n - 1]);
            }
#pragma unroll
            for (int k = 0; k < NPER; k++) {
                ok &= ((unsigned)(gv[k] >> 32) == tag);
            }
            if (__all(ok)) {
                break;
            }
            if (ps_give_up(spins, err, code)) {
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < NPER; k++) {
            const int i = base + k * nthr + tid;
            if (i < n) {
                sink(i, (unsigned)gv[k]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight stream of a workgroup over one stage.  The stage's tiles are cut into BATCHES of PS_U consecutive k tiles of one
// run (8 KiB of contiguous weights per wave-batch); the batches form a QUEUE in LDS (descriptor table + head counter)
// that the eight waves drain dynamically: a wave claims its next batch with one LDS atomic when it issues the loads, i.e.
// PS_NBUF - 1 batches ahead of consuming it.  Static per-wave shares ended up to 7 us apart inside a workgroup (the CU's
// memory pipeline serves whoever issued first, and the control waves join late by design); with the queue every wave
// stops within one batch of the others and there is no share to tune.
// Each batch accumulates into its own fp32 slot part[batch][M*16]; the wave that completes the LAST batch of a run
// (per-run LDS counter) sums the run's slots in batch order -- deterministic whoever computed them -- and PUBLISHES the
// run (qkv / mid / K-piece granules) at once: hand-offs leave as soon as their run is done (qkv after ~1/3 of P1), not
// after an end-of-stage barrier and epilogue.
// Every load in the loop is unconditional (a claim past the end of the queue maps to a padding descriptor whose 8 loads
// read one 16-byte word): the compiler counts vmcnt exactly and three batches stay in flight while one is consumed.
// ---------------------------------------------------------------------------------------------------------------
struct PsStage {
    const u64*    bt;  // [nb + 1] batch descriptors; entry nb is the padding batch
    int           nb;
    int*          qh;  // queue head (LDS)
    int*          rc;  // [runs] completed batches per run (LDS)
    const RunRec* rt;
    const char *  w0, *w1;
    int           xs0, xs1;
};

template<bool INT8, int M>
struct PsStream {
    static constexpr int TK = TileK<INT8>::value;
    PsStage    g;
    const f16* rsc;
    const f16* xs;
    float*     part;
    const int* flag;    // LDS arrival counter of the second x vector
    int        target;  // value it reaches when that vector is staged
    const f16* bias;    // [runs][16] bias of the PS_PUB_GELU runs (LDS)
    unsigned   tag;     // granule tag of this layer
    int        lane, lane16;
    int        i0, i1, i2, i3;      // queue index of the batch in R0..R3
    unsigned   h0, h1, h2, h3;      // and the high word of its descriptor

    __device__ __forceinline__ void bind(const PsStage& g_, const f16* rsc_, const f16* xs_, float* part_, const int tx,
                                         const f16* bias_, const unsigned tag_, const int* flag_ = nullptr,
                                         const int target_ = 0)
    {
        flag   = flag_;
        target = target_;
        bias   = bias_;
        tag    = tag_;
        g      = g_;
        rsc    = rsc_;
        xs     = xs_;
        part   = part_;
        lane   = tx & 63;
        lane16 = lane * 16;
        i0 = i1 = i2 = i3 = g_.nb;
        h0 = h1 = h2 = h3 = 0u;
    }
    // claim the next batch of the queue and request its tiles
    __device__ __forceinline__ void issue(u32x4 (&r)[PS_U], int& idx, unsigned& dhi)
    {
        int i = 0;
        if (lane == 0) {
            i = __hip_atomic_fetch_add(g.qh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        i           = ps_rfl(i);
        i           = i < g.nb ? i : g.nb;
        const u64 d = g.bt[i];
        const unsigned lo = (unsigned)ps_rfl((int)(unsigned)d), hi = (unsigned)ps_rfl((int)(unsigned)(d >> 32));
        const unsigned cnt  = (hi >> 25) & 15u;
        const char*    base = (hi & PS_BD_SEL) ? g.w1 : g.w0;
        const unsigned voff = cnt ? (unsigned)lane16 : 0u;  // padding batch: all lanes read the same 16 bytes (one line)
        const unsigned last = cnt ? cnt - 1u : 0u;
#pragma unroll
        for (int u = 0; u < PS_U; u++) {
            const unsigned uu = (unsigned)u < last ? (unsigned)u : last;  // tiles past the run's end re-read its last one
            r[u] = __builtin_nontemporal_load(
                (const __attribute__((address_space(1))) u32x4*)(base + (size_t)(lo + uu) * TILE_BYTES + voff));
        }
        idx = i;
        dhi = hi;
    }
    // The wave that completed a run's last batch: sum of the run's batch slots in batch order (deterministic whoever
    // computed them), epilogue, granules.  qkv = y (no bias: the attention adds it, like the reference's MMHA),
    // mid = gelu(y + b) (epilogue_helpers.h:52-62 fused fp32 form for int8, activation_kernels.cu:401-426 half form for
    // fp16 weights), K pieces = fp32 partial sums.
    __device__ __forceinline__ void publish(const int j)
    {
        const RunRec* rr   = g.rt + j;
        const int     b0   = ps_rfl(rr->b0), nbj = (ps_rfl(rr->nt) + PS_U - 1) / PS_U, kind = ps_rfl(rr->kind);
        const int     r    = lane < M * 16 ? lane : 0;
        float         v    = 0.f;
        for (int b = 0; b < nbj; b++) {
            v += part[(size_t)(b0 + b) * (M * 16) + r];
        }
        u64* dst = rr->dst;
        if (kind == PS_PUB_F32) {
            if (lane < M * 16) {
                st_granule(dst + lane, tag, v);
            }
            return;
        }
        const int m = r >> 4, c = r & 15;
        f16       o = (f16)v;
        if (kind == PS_PUB_GELU) {
            const f16 bb = bias[j * 16 + c];
            if constexpr (INT8) {
                o = (f16)gelu_f32(v + (float)bb);
            }
            else {
                o = gelu_f16((f16)v + bb);
            }
        }
        const unsigned x0 = f16_bits(o);
        const unsigned x1 = __shfl_down(x0, 1, 64);
        if (lane < M * 16 && (c & 1) == 0) {
            st_granule_u32(dst + (size_t)m * ps_rfl(rr->mstride) + (c >> 1), tag, x0 | (x1 << 16));
        }
    }
    __device__ __forceinline__ void consume(const u32x4 (&r)[PS_U], const int idx, const unsigned hi)
    {
        const unsigned cnt = (hi >> 25) & 15u;
        if (cnt == 0u) {
            return;  // padding batch
        }
        if (hi & PS_BD_WAIT) {  // rare: only a batch that reaches the late vector before it is staged actually spins
            while (ps_rfl(*(const volatile __attribute__((address_space(3))) int*)flag) < target) {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const int  j   = (int)((hi >> 3) & 31u);
        const f16* xr  = a_frag_ptr<INT8, M>(xs + ((hi >> 8) & 0x1ffffu), (hi & PS_BD_XSEL) ? g.xs1 : g.xs0, lane);
        f16x2      sc2 = {(f16)1.0f, (f16)1.0f};
        if constexpr (INT8) {
            const f16 sc = rsc[j * 16 + (lane & 15)];
            sc2          = f16x2{sc, sc};
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (cnt == PS_U) {  // straight-line code like the per-kernel GEMV stream
#pragma unroll
            for (int u = 0; u < PS_U; u++) {
                consume_tile<INT8, M>(r[u], xr + u * TK, sc2, acc);
            }
        }
        else {
#pragma unroll
            for (int u = 0; u < PS_U; u++) {
                if ((unsigned)u < cnt) {
                    consume_tile<INT8, M>(r[u], xr + u * TK, sc2, acc);
                }
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                part[((size_t)idx * M + m) * 16 + lane] = acc_row(acc, m);
            }
        }
        // the slot is written before the run's counter moves (DS operations of a wave execute in order; the fences only
        // keep the compiler from reordering them and wait on lgkmcnt), so whoever sees the final count sees every slot
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        int old = 0;
        if (lane == 0) {
            old = __hip_atomic_fetch_add(&g.rc[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        old = ps_rfl(old);
        const int nbj = (ps_rfl(g.rt[j].nt) + PS_U - 1) / PS_U;
        if (old == nbj - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            publish(j);
        }
    }
    // the first rotation: issued before the hand-off this stage waits for.  The streamer waves issue only half of it
    // there and the rest when they start consuming: a 192 KiB burst per CU sits in FRONT of the control waves' sweeps in
    // the CU's memory pipeline and stretched each hand-off hop to 6-7 us (measured)
    __device__ __forceinline__ void prime_lo(u32x4 (&R0)[PS_U], u32x4 (&R1)[PS_U])
    {
        issue(R0, i0, h0);
        issue(R1, i1, h1);
    }
    __device__ __forceinline__ void prime_hi(u32x4 (&R2)[PS_U], u32x4 (&R3)[PS_U])
    {
        issue(R2, i2, h2);
        issue(R3, i3, h3);
    }
    // HI: the second half of the first rotation is still to be issued.  Compile time: a load under a run-time condition
    // makes the compiler's vmcnt bookkeeping conservative for the whole stream (measured: 340 -> 192 tokens/s)
    template<bool HI>
    __device__ __forceinline__ void run(u32x4 (&R0)[PS_U], u32x4 (&R1)[PS_U], u32x4 (&R2)[PS_U], u32x4 (&R3)[PS_U])
    {
        if constexpr (HI) {
            prime_hi(R2, R3);
            __builtin_amdgcn_sched_barrier(0);
        }
        // claims are handed out in order, so the most recent one (i3 after a full rotation) tells whether the queue has
        // run dry; the up to three padding batches issued on the way out cost eight one-line loads each
        while (i3 < g.nb) {
            consume(R0, i0, h0);
            __builtin_amdgcn_sched_barrier(0);
            issue(R0, i0, h0);
            __builtin_amdgcn_sched_barrier(0);
            consume(R1, i1, h1);
            __builtin_amdgcn_sched_barrier(0);
            issue(R1, i1, h1);
            __builtin_amdgcn_sched_barrier(0);
            consume(R2, i2, h2);
            __builtin_amdgcn_sched_barrier(0);
            issue(R2, i2, h2);
            __builtin_amdgcn_sched_barrier(0);
            consume(R3, i3, h3);
            __builtin_amdgcn_sched_barrier(0);
            issue(R3, i3, h3);
            __builtin_amdgcn_sched_barrier(0);
        }
        consume(R0, i0, h0);
        consume(R1, i1, h1);
        consume(R2, i2, h2);
        consume(R3, i3, h3);
    }
};

// The batch queue of a stage from the workgroup's static run table (runs in queue order, RunRec::b0 set): one thread per
// batch.  Entry nb is the padding batch (the workgroup's first tile of the stage: one address per workgroup, no hot spot).
template<int TK>
__device__ __forceinline__ void ps_build_queue(const RunRec* rt, const int nruns, u64* bt, const int nb, const bool gate_sel1)
{
    for (int bi = threadIdx.x; bi <= nb; bi += PS_NT) {
        u64 d = 0;
        if (bi == nb) {
            d = (u64)(unsigned)(nruns > 0 ? rt[0].tile0 : 0) | ((u64)(unsigned)(nruns > 0 ? rt[0].sel : 0) << 32);
        }
        else {
            for (int j = 0; j < nruns; j++) {
                const RunRec r   = rt[j];
                const int    nbj = (r.nt + PS_U - 1) / PS_U;
                if (bi >= r.b0 && bi < r.b0 + nbj) {
                    const int      t   = (bi - r.b0) * PS_U;
                    const int      cnt = (r.nt - t < PS_U) ? r.nt - t : PS_U;
                    const unsigned hi  = (r.sel ? PS_BD_SEL : 0u) | (r.xsel ? PS_BD_XSEL : 0u)
                                        | ((gate_sel1 && r.xsel) ? PS_BD_WAIT : 0u) | ((unsigned)j << 3)
                                        | ((unsigned)(r.xoff + t * TK) << 8) | ((unsigned)cnt << 25);
                    d = (u64)(unsigned)(r.tile0 + t) | ((u64)hi << 32);
                }
            }
        }
        bt[bi] = d;
    }
}

struct PsSmem {
    f16*      xraw;  // [M][H]
    f16*      xs;    // x region (P1: LN1(x) | LN2(x) ; P3: mid | ctx)
    float*    part;  // [max(nb1, nb3)][M*16] one fp32 slot per batch
    char*     att;   // attention scratch
    RunRec*   rt1;   // [RMAX] P1 runs
    RunRec*   rt3;   // [RMAX] P3 runs
    f16*      rsc;   // [RMAX][16] scales of the current stage
    f16*      b1;    // [RMAX][16] FFN1 bias of the P1 runs
    float*    red;   // 64
    int*      misc;  // 128: [0] nmerge, [1..8] merge groups, [32] ctx arrivals, [33] control pair barrier,
                     //      [34] / [35] queue heads of P1 / P3, [64..64+RMAX) / [96..96+RMAX) run counters of P1 / P3
    u64 *     bt1, *bt3;  // [nb1 + 1], [nb3 + 1]
};
constexpr int PS_MISC_INTS = 128, PS_QH1 = 34, PS_QH3 = 35, PS_RC1 = 64, PS_RC3 = 96;
static_assert(PS_RMAX <= 32, "run counters");

__host__ __device__ inline size_t ps_att_bytes(int dh, int s_max, int nsplit)
{
    const int chunk = ((((s_max + nsplit - 1) / nsplit) + 15) & ~15);
    size_t    a     = (size_t)3 * dh * 2 + (size_t)(2 * PS_NW + PS_NW * dh) * 4 + (size_t)chunk * 4;
    size_t    b     = (size_t)(nsplit * (dh + 2) + nsplit + 4) * 4;
    return ((a > b ? a : b) + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------------------------------------------
// attention of one (row b, head h, split sp) on the whole 8-wave workgroup
// (decoder_masked_multihead_attention_template.hpp:1099-1919; same arithmetic as attn_device.cuh::mmha_partial)
// ---------------------------------------------------------------------------------------------------------------
template<int DH, int UK>
struct PsAttn {
    static constexpr int LPK = DH / 8;
    static constexpr int KPI = 64 / LPK;
    static_assert(UK <= 2 * PS_U, "K rows live in R0|R1, V rows in R2|R3");
    unsigned mask_bits, bias2;
    int      tl, chunk, t_beg;
    float    rot_cs, rot_sn;
    bool     fin;

    // loads that do not depend on this step's qkv: K/V rows of the whole fixed chunk, masks, lengths, rotary table.
    // The rows go into the weight stream's register batches (K: R0 | R1, V: R2 | R3), which are idle between the end of a
    // wave's P1 stream and its first P3 batch: no registers of their own, so they can be requested as soon as the wave
    // leaves the P1 queue -- before the workgroup barrier and the P3 set-up -- without spilling.
    __device__ __forceinline__ void issue(const PersistParams& p, const PersistLayer& lw, int h, int b, int sp, const int tx,
                                          u32x4 (&K0)[PS_U], u32x4 (&K1)[PS_U], u32x4 (&V0)[PS_U], u32x4 (&V1)[PS_U])
    {
        const int lane = tx & 63, wid = tx >> 6;
        const int sub = lane % LPK, grp = lane / LPK;
        chunk = (((p.s_max + p.plan.nsplit - 1) / p.plan.nsplit) + 15) & ~15;
        t_beg = sp * chunk;
        const auto* kc = PS_G(f16, lw.k_cache) + ((size_t)b * p.nh + h) * p.s_max * DH;
        const auto* vc = PS_G(f16, lw.v_cache) + ((size_t)b * p.nh + h) * p.s_max * DH;
        // rows past the end of this split's chunk (the register capacity covers UK * 32 keys, the chunk may be shorter)
        // re-read its last row: a cache hit, not K/V traffic of the neighbouring split
        int t_last = t_beg + chunk - 1;
        t_last     = t_last < p.s_max ? t_last : p.s_max - 1;
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t   = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            t       = t < t_last ? t : t_last;
            (u < PS_U ? K0[u % PS_U] : K1[u % PS_U]) = *PS_G(u32x4, kc + (size_t)t * DH + sub * 8);
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t   = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            t       = t < t_last ? t : t_last;
            (u < PS_U ? V0[u % PS_U] : V1[u % PS_U]) = *PS_G(u32x4, vc + (size_t)t * DH + sub * 8);
        }
        mask_bits = 0u;
        if (p.masked_tokens && sub == 0) {
#pragma unroll
            for (int u = 0; u < UK; u++) {
                int t = t_beg + u * PS_NW * KPI + wid * KPI + grp;
                t     = t < t_last ? t : t_last;
                mask_bits |= (p.masked_tokens[(size_t)b * p.s_max + t] ? 1u : 0u) << u;
            }
        }
        rot_cs = 1.f;
        rot_sn = 0.f;
        if (p.rot > 0 && tx < p.rot / 2) {
            rot_cs = p.rot_table[((size_t)b * (p.rot / 2) + tx) * 2];
            rot_sn = p.rot_table[((size_t)b * (p.rot / 2) + tx) * 2 + 1];
        }
        bias2 = 0u;
        if (tx < 3 * DH / 2) {  // this thread's pair of q / k / v bias values (sweep_qkv)
            const int seg = tx / (DH / 2), i = tx % (DH / 2);
            bias2 = *PS_G(unsigned, reinterpret_cast<const unsigned*>(lw.b_qkv + (size_t)seg * p.nh * DH + h * DH) + i);
        }
        fin = p.finished && p.finished[b];
        tl  = p.seq_len[b];
    }
    // q/k/v of the current token: granules published by the QKV stage of THIS launch (pairs of halves); + bias -> LDS
    __device__ __forceinline__ void sweep_qkv(const PersistParams& p, char* smem, const unsigned tag, int h, int b, const int tx)
    {
        if (fin) {
            return;
        }
        f16* s_q = reinterpret_cast<f16*>(smem);  // [DH] q | [DH] k | [DH] v
        if (tx < 3 * DH / 2) {
            const int  seg = tx / (DH / 2), i = tx % (DH / 2);
            const int  hl  = p.nh * DH;
            const u64* g   = p.gq + ((size_t)b * 3 * hl + (size_t)seg * hl + h * DH) / 2 + i;
            u64        v;
            int        spins = 0;
            for (;;) {
                v = ld_granule(g);
                if ((unsigned)(v >> 32) == tag) {
                    break;
                }
                if (++spins > PS_SPIN) {
                    __hip_atomic_store(p.err, 5, PS_RLX, PS_AGT);
                    break;
                }
                if ((spins & 255) == 0 && __hip_atomic_load(p.err, PS_RLX, PS_AGT) != 0) {
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            const f16 x0 = bits_f16((unsigned)v), x1 = bits_f16((unsigned)v >> 16);
            s_q[seg * DH + 2 * i]     = x0 + bits_f16(bias2);
            s_q[seg * DH + 2 * i + 1] = x1 + bits_f16(bias2 >> 16);
        }
    }
    // returns false when the row is finished (nothing published)
    __device__ __forceinline__ bool compute(const PersistParams& p, const PersistLayer& lw, char* smem, u64* gout,
                                            const unsigned tag, int h, int b, const int tx, const u32x4 (&K0)[PS_U],
                                            const u32x4 (&K1)[PS_U], const u32x4 (&V0)[PS_U], const u32x4 (&V1)[PS_U])
    {
        const int lane = tx & 63, wid = tx >> 6;
        const int sub = lane % LPK, grp = lane / LPK;
        if (fin) {
            return false;  // :1176
        }
        int t_end = t_beg + chunk;
        if (t_end > tl + 1) {
            t_end = tl + 1;
        }
        if (t_beg > tl) {  // empty split
            if (tx < DH) {
                st_granule(&gout[tx], tag, 0.f);
            }
            if (tx == 0) {
                st_granule(&gout[DH], tag, -INFINITY);
                st_granule(&gout[DH + 1], tag, 0.f);
            }
            return true;
        }
        const bool owns_cur     = (tl >= t_beg && tl < t_end);
        const int  t_cached_end = owns_cur ? tl : t_end;
        f16*   s_q   = reinterpret_cast<f16*>(smem);
        f16*   s_k   = s_q + DH;
        f16*   s_v   = s_k + DH;
        float* s_red = reinterpret_cast<float*>(s_v + DH);  // [2*NW + NW*DH]
        float* s_p   = s_red + 2 * PS_NW + PS_NW * DH;      // [chunk]
        __syncthreads();  // q | k | v (+ bias) written by sweep_qkv
        if (p.rot > 0 && tx < p.rot / 2) {
            const int j = tx;
            f16       a = s_q[j], c = s_q[j + p.rot / 2];
            rotary_apply(a, c, rot_cs, rot_sn);
            s_q[j]             = a;
            s_q[j + p.rot / 2] = c;
            if (owns_cur) {
                f16 ka = s_k[j], kc2 = s_k[j + p.rot / 2];
                rotary_apply(ka, kc2, rot_cs, rot_sn);
                s_k[j]             = ka;
                s_k[j + p.rot / 2] = kc2;
            }
        }
        __syncthreads();
        if (owns_cur && tx < DH) {  // append to the cache (:1397, :1837)
            ((__attribute__((address_space(1))) f16*)lw.k_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_k[tx];
            ((__attribute__((address_space(1))) f16*)lw.v_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_v[tx];
        }
        const float inv_sqrt_dh = rsqrtf((float)DH);
        const f16x8 qv          = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
        float       lmax        = -INFINITY;
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const int   t  = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            const f16x8 kv = __builtin_bit_cast(f16x8, u < PS_U ? K0[u % PS_U] : K1[u % PS_U]);
            float       a  = 0.f;
            a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
            a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
            a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
            a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
            a              = group_sum(a, LPK) * inv_sqrt_dh;
            if (t < t_cached_end && sub == 0) {
                const bool m   = ((mask_bits >> u) & 1u) != 0u;
                s_p[t - t_beg] = m ? -INFINITY : a;
                if (!m) {
                    lmax = fmaxf(lmax, a);
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
        float m_loc = s_red[0];
#pragma unroll
        for (int w = 1; w < PS_NW; w++) {
            m_loc = fmaxf(m_loc, s_red[w]);
        }
        float lsum = 0.f;
        for (int i = tx; i < t_end - t_beg; i += PS_NT) {
            const float e = (s_p[i] == -INFINITY) ? 0.f : __expf(s_p[i] - m_loc);
            s_p[i]        = e;
            lsum += e;
        }
        lsum = wave_sum(lsum);
        __syncthreads();
        if (lane == 0) {
            s_red[PS_NW + wid] = lsum;
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const int t = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            if (t < t_cached_end) {  // rows beyond tlength were fetched speculatively and may hold anything
                const float pt = s_p[t - t_beg];
                const f16x8 vv = __builtin_bit_cast(f16x8, u < PS_U ? V0[u % PS_U] : V1[u % PS_U]);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc[j] = fmaf(pt, (float)vv[j], acc[j]);
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
#pragma unroll
        for (int j = 0; j < 8; j++) {
            for (int o = LPK; o < 64; o <<= 1) {
                acc[j] += __shfl_xor(acc[j], o, 64);
            }
        }
        float* s_o = s_red + 2 * PS_NW;  // [NW][DH]
        if (grp == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                s_o[wid * DH + sub * 8 + j] = acc[j];
            }
        }
        __syncthreads();
        if (tx < DH) {
            const int d = tx;
            float     o = 0.f;
#pragma unroll
            for (int w = 0; w < PS_NW; w++) {
                o += s_o[w * DH + d];
            }
            st_granule(&gout[d], tag, o);
        }
        if (tx == 0) {
            float ls = 0.f;
#pragma unroll
            for (int w = 0; w < PS_NW; w++) {
                ls += s_red[PS_NW + w];
            }
            st_granule(&gout[DH], tag, m_loc);
            st_granule(&gout[DH + 1], tag, ls);
        }
        return true;
    }
};

// split-0 workgroup of a (row, head): WAVE 0 alone sweeps the nsplit partials, merges them in split order and publishes
// ctx as granules (one wave: no workgroup barrier, the other waves are already streaming the next stage)
template<int DH>
__device__ __forceinline__ void ps_attn_merge(const PersistParams& p, char* smem, u64* gall, const unsigned tag, int h,
                                              int b, const int tx)
{
    const int ne = DH + 2, ns = p.plan.nsplit;
    const int ng = ns * ne;
    float*    sval = reinterpret_cast<float*>(smem);  // [ns][ne] then [ns] weights + denominator
    ps_sweep<8>(gall, ng, tx, 64, tag, p.err, 2, [&](const int i, const unsigned v) { sval[i] = __uint_as_float(v); });
    // weights (same wave: DS operations of one wave execute in order)
    float ms = -INFINITY, ls = 0.f;
    if (tx < ns) {
        ms = sval[tx * ne + DH];
        ls = sval[tx * ne + DH + 1];
    }
    const float m  = wave_max(ms);
    const float w  = (ms == -INFINITY) ? 0.f : __expf(ms - m);
    float*      sw = sval + ns * ne;
    if (tx < ns) {
        sw[tx] = w;
    }
    float L = 0.f;
    for (int s2 = 0; s2 < ns; s2++) {
        L += __shfl(w * ls, s2, 64);
    }
    const float inv = 1.f / (L + 1.e-6f);  // :1632
    for (int d = tx; d < DH; d += 64) {
        float o = 0.f;
        for (int s2 = 0; s2 < ns; s2++) {
            o += sw[s2] * sval[s2 * ne + d];
        }
        const unsigned b0 = f16_bits((f16)(o * inv));
        const unsigned b1 = __shfl_down(b0, 1, 64);
        if ((d & 1) == 0) {
            st_granule_u32(&p.gc[((size_t)b * p.nh * DH + h * DH + d) >> 1], tag, b0 | (b1 << 16));
        }
    }
}

// finished row: its ctx is never consumed (:1176) but the out-proj stage still waits for the granules
template<int DH>
__device__ __forceinline__ void ps_attn_publish_zero(const PersistParams& p, const unsigned tag, int h, int b, const int tx)
{
    if (tx < DH / 2) {
        st_granule_u32(&p.gc[(((size_t)b * p.nh * DH + h * DH) >> 1) + tx], tag, 0u);
    }
}

// ---------------------------------------------------------------------------------------------------------------
template<bool INT8, int M, int DH, int UK>
__global__ __launch_bounds__(PS_NT) void k_decode_persistent(const PersistParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TK = TileK<INT8>::value;
    const int     H = p.H, Hl = p.Hl, Il = p.Il;
    const int     NB = p.plan.NB, bid = blockIdx.x;
    const int     wid = threadIdx.x >> 6;
    const int     KT = H / TK, KT_a = Hl / TK, KT_b = Il / TK;
    const int     NT0 = 3 * Hl / 16, NG = H / 16;
    const int     PA = p.plan.PA, PB = p.plan.PB, RLa = p.plan.RLa, RLb = p.plan.RLb;

    PsSmem s;
    {
        char* q = smem;
        s.xraw  = reinterpret_cast<f16*>(q);
        q += (size_t)M * H * 2;
        s.xs = reinterpret_cast<f16*>(q);
        q += (size_t)p.plan.xs_halves * 2;
        s.part = reinterpret_cast<float*>(q);
        q += (size_t)p.plan.nbmax * M * 16 * 4;
        s.att = q;
        q += ps_att_bytes(DH, p.s_max, p.plan.nsplit);
        s.rt1 = reinterpret_cast<RunRec*>(q);
        q += sizeof(RunRec) * PS_RMAX;
        s.rt3 = reinterpret_cast<RunRec*>(q);
        q += sizeof(RunRec) * PS_RMAX;
        s.rsc = reinterpret_cast<f16*>(q);
        q += PS_RMAX * 16 * 2;
        s.b1 = reinterpret_cast<f16*>(q);
        q += PS_RMAX * 16 * 2;
        s.red = reinterpret_cast<float*>(q);
        q += 64 * 4;
        s.misc = reinterpret_cast<int*>(q);
        q += PS_MISC_INTS * 4;
        s.bt1 = reinterpret_cast<u64*>(q);
        q += (size_t)(p.plan.nbmax + 1) * 8;
        s.bt3 = reinterpret_cast<u64*>(q);
    }
    const int      step     = *p.d_step;
    const unsigned tag_base = (unsigned)step * 256u + 1u;
    if (p.ts && (threadIdx.x & 63) == 0) {  // kernel entry (slot 15 of the first layer)
        p.ts[(((size_t)blockIdx.x * p.L + p.l_begin) * PS_NW + (threadIdx.x >> 6)) * 16 + 15] = wall_clock64();
    }

    // ---- the workgroup's static share of the streaming stages ----
    // P1: every workgroup owns a range of QKV column groups AND a range of FFN1 column groups; the QKV runs come first in
    // its queue, so qkv is complete -- and published by whoever finishes a run's last batch -- after about a third of the
    // stage, and the attention finds it waiting
    const int NF  = Il / 16;
    const int q0  = (int)((long)NT0 * bid / NB), q1 = (int)((long)NT0 * (bid + 1) / NB);
    const int f0  = (int)((long)NF * bid / NB), f1 = (int)((long)NF * (bid + 1) / NB);
    const int nq  = q1 - q0;
    const int rB0 = (int)((long)NG * PB * bid / NB), rB1 = (int)((long)NG * PB * (bid + 1) / NB);
    const int rA0 = (int)((long)NG * PA * bid / NB), rA1 = (int)((long)NG * PA * (bid + 1) / NB);
    const int nB = rB1 - rB0, nA = rA1 - rA0;
    const int nruns1 = nq + (f1 - f0), nruns3 = nB + nA;
    const int n_items = p.B * p.nh * p.plan.nsplit;
    if (threadIdx.x < PS_MISC_INTS) {
        s.misc[threadIdx.x] = 0;  // nmerge, ctx arrivals, control pair barrier, queue heads, run counters
    }
    __syncthreads();
    if ((int)threadIdx.x < nruns1) {  // P1: QKV column groups q0..q1, then FFN1 column groups f0..f1, full K each
        const int  j   = threadIdx.x;
        const bool seg = j >= nq;
        const int  cg  = seg ? NT0 + f0 + (j - nq) : q0 + j;
        const int  g   = seg ? cg - NT0 : cg;
        RunRec     r;
        r.tile0  = g * KT;
        r.sel    = seg ? 1 : 0;
        r.nt     = KT;
        r.xoff   = seg ? M * (H + XPAD) : 0;
        r.xsel   = 0;
        r.rid    = cg;
        r.grp    = g;
        r.b0     = j * ((KT + PS_U - 1) / PS_U);
        r.dst     = seg ? p.gm + ((size_t)g * 16 >> 1) : p.gq + ((size_t)cg * 16 >> 1);
        r.mstride = seg ? Il / 2 : 3 * Hl / 2;
        r.kind    = seg ? PS_PUB_GELU : PS_PUB_PLAIN;
        s.rt1[j] = r;
    }
    if ((int)threadIdx.x < nruns3) {  // P3: FFN2 K pieces first, then out-proj K pieces (piece-major ids)
        const int  j     = threadIdx.x;
        const bool isA   = j >= nB;
        const int  idx   = isA ? rA0 + (j - nB) : rB0 + j;
        const int  piece = idx / NG, g = idx % NG;
        RunRec     r;
        if (isA) {
            const int t0 = piece * RLa;
            r.tile0      = g * KT_a + t0;
            r.sel        = 1;
            r.nt         = (KT_a - t0 < RLa) ? KT_a - t0 : RLa;
            r.xoff       = M * (Il + XPAD) + t0 * TK;
            r.xsel       = 1;
            r.rid        = NG * PB + idx;
            if (piece == PA - 1) {  // owner of a group's last out-proj piece merges the group
                const int k = atomicAdd(&s.misc[0], 1);
                if (k < PS_MAXMERGE) {
                    s.misc[1 + k] = g;
                }
            }
        }
        else {
            const int t0 = piece * RLb;
            r.tile0      = g * KT_b + t0;
            r.sel        = 0;
            r.nt         = (KT_b - t0 < RLb) ? KT_b - t0 : RLb;
            r.xoff       = t0 * TK;
            r.xsel       = 0;
            r.rid        = idx;
        }
        r.grp     = g;
        r.b0      = 0;
        r.dst     = p.gp + (size_t)r.rid * (M * 16);
        r.mstride = 0;
        r.kind    = PS_PUB_F32;
        s.rt3[j]  = r;
    }
    __syncthreads();
    PsStage sg1{}, sg3{};
    int     mid_lo = 0, mid_hi = 0, ctx_lo = 0, ctx_hi = 0;  // K ranges (halves) of mid / ctx this workgroup consumes
    {
        int  nb3 = 0;
        bool fb = true, fa = true;
        for (int j = 0; j < nruns3; j++) {  // (every thread: uniform results, b0 written by thread 0)
            const RunRec r = s.rt3[j];
            if (threadIdx.x == 0) {
                s.rt3[j].b0 = nb3;
            }
            nb3 += (r.nt + PS_U - 1) / PS_U;
            if (r.sel == 0) {
                const int lo = r.xoff, hi = r.xoff + r.nt * TK;
                mid_lo = fb ? lo : (lo < mid_lo ? lo : mid_lo);
                mid_hi = fb ? hi : (hi > mid_hi ? hi : mid_hi);
                fb     = false;
            }
            else {
                const int lo = r.xoff - M * (Il + XPAD), hi = lo + r.nt * TK;
                ctx_lo = fa ? lo : (lo < ctx_lo ? lo : ctx_lo);
                ctx_hi = fa ? hi : (hi > ctx_hi ? hi : ctx_hi);
                fa     = false;
            }
        }
        mid_lo = ps_rfl(mid_lo);
        mid_hi = ps_rfl(mid_hi);
        ctx_lo = ps_rfl(ctx_lo);
        ctx_hi = ps_rfl(ctx_hi);
        sg1.bt = s.bt1;
        sg1.nb = ps_rfl(nruns1 * ((KT + PS_U - 1) / PS_U));
        sg1.qh = &s.misc[PS_QH1];
        sg1.rc = &s.misc[PS_RC1];
        sg1.rt = s.rt1;
        sg3.bt = s.bt3;
        sg3.nb = ps_rfl(nb3);
        sg3.qh = &s.misc[PS_QH3];
        sg3.rc = &s.misc[PS_RC3];
        sg3.rt = s.rt3;
        sg1.xs0 = sg1.xs1 = H + XPAD;  // LDS rows of x are padded: see XPAD
        sg3.xs0 = Il + XPAD;
        sg3.xs1 = Hl + XPAD;
    }
    __syncthreads();  // rt3[].b0
    ps_build_queue<TK>(s.rt1, nruns1, s.bt1, sg1.nb, false);
    ps_build_queue<TK>(s.rt3, nruns3, s.bt3, sg3.nb, true);
    __syncthreads();

    // Control waves and streamer waves run SEPARATE instantiations of the layer loop (same barriers, in the same order):
    // with a shared body the register batches of the role that primes early stay live, as far as the compiler can tell,
    // through every section of the other role and spill.  Whole waves take one side, s_barrier only counts arrivals.
    auto body = [&](auto role) {
        constexpr bool    CTRL = decltype(role)::value;
        int               tid  = threadIdx.x;
        PsStream<INT8, M> st;
        u32x4             R0[PS_U], R1[PS_U], R2[PS_U], R3[PS_U];  // weight batches; K / V rows during the attention
        auto stamp = [&](const int l, const int k) {
            const int lane = tid & 63, wid = tid >> 6;
            if (p.ts && lane == 0) {
                p.ts[(((size_t)bid * p.L + l) * PS_NW + wid) * 16 + k] = wall_clock64();
            }
        };
        // ---- per-layer constants, fetched one stage ahead into registers (before that stage's prefetch) ----
        f16   r_sc1 = (f16)1.f, r_sc3 = (f16)1.f;  // scale of (run tid/16, column tid%16) of P1 / P3
        f16   r_b1 = (f16)0.f;                     // ffn1 bias of (run tid/16, column tid%16) of P1
        f16   r_bres[2];                           // residual bias of the merge items
        f16x8 r_ln[4][PS_NLN];                     // ln1_g, ln1_b, ln2_g, ln2_b vectors tid, tid + 512
        auto  load_sc1 = [&](const int l) {
            if constexpr (INT8) {
                r_sc1 = (f16)1.f;
                if (tid < nruns1 * 16) {
                    const PersistLayer& lw = p.layers[l];
                    const RunRec&       r  = s.rt1[tid >> 4];
                    r_sc1 = PS_G(f16, r.sel ? lw.s_ffn1 : lw.s_qkv)[r.grp * 16 + (tid & 15)];
                }
            }
        };
        auto load_p1_consts = [&](const int l) {  // LN parameters, ffn1 bias, P3 scales of layer l
            const PersistLayer& lw = p.layers[l];
            // (unconditional, clamped: a conditional assignment would carry the previous layer's values -- 32 VGPRs --
            // through the whole loop body, weight streams included)
#pragma unroll
            for (int k = 0; k < PS_NLN; k++) {
                const int v = tid + k * PS_NT;
                const int o = (v * 8 < H) ? v * 8 : 0;
                r_ln[0][k]  = *PS_G(f16x8, lw.ln1_g + o);
                r_ln[1][k]  = *PS_G(f16x8, lw.ln1_b + o);
                r_ln[2][k]  = *PS_G(f16x8, lw.ln2_g + o);
                r_ln[3][k]  = *PS_G(f16x8, lw.ln2_b + o);
            }
            r_b1 = (f16)0.f;
            if (tid < nruns1 * 16) {
                const int cg = s.rt1[tid >> 4].rid;
                if (cg >= NT0) {
                    r_b1 = PS_G(f16, lw.b_ffn1)[(cg - NT0) * 16 + (tid & 15)];
                }
            }
            if constexpr (INT8) {
                r_sc3 = (f16)1.f;
                if (tid < nruns3 * 16) {
                    const RunRec& r = s.rt3[tid >> 4];
                    r_sc3 = PS_G(f16, r.sel ? lw.s_out : lw.s_ffn2)[r.grp * 16 + (tid & 15)];
                }
            }
        };
        auto load_p3_consts = [&](const int l) {  // residual bias of layer l, P1 scales of layer l + 1
            if constexpr (CTRL) {
                const PersistLayer& lw = p.layers[l];
                const int           nm = s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int t = tid + k * PS_NC * 64;
                    r_bres[k]   = (f16)0.f;
                    if (t < nm * M * 16) {
                        r_bres[k] = PS_G(f16, lw.b_res)[s.misc[1 + t / (M * 16)] * 16 + (t & 15)];
                    }
                }
            }
            // (clamped, not conditional: values assigned under a condition are carried around the layer loop, i.e. stay live
            // through both weight streams; after the last layer the loads are harmless re-reads)
            load_sc1(l + 1 < p.l_end ? l + 1 : l);
        };
        // per layer: scales of the stage's runs -> LDS, the OTHER stage's queue state back to zero (nobody is inside it:
        // a workgroup barrier lies between its last use and here, and another one before its next use), bind the stream
        auto setup_p1 = [&](const int l) {
            const PersistLayer& lw = p.layers[l];
            if constexpr (INT8) {
                if (tid < nruns1 * 16) {
                    s.rsc[tid] = r_sc1;
                }
            }
            if (tid < PS_RMAX) {
                s.misc[PS_RC3 + tid] = 0;
            }
            if (tid == PS_RMAX) {
                s.misc[PS_QH3] = 0;
            }
            load_p1_consts(l);
            sg1.w0 = reinterpret_cast<const char*>(lw.w_qkv);
            sg1.w1 = reinterpret_cast<const char*>(lw.w_ffn1);
            st.bind(sg1, s.rsc, s.xs, s.part, tid, s.b1, tag_base + (unsigned)l);
            if constexpr (!CTRL) {
                st.prime_lo(R0, R1);
                if constexpr (PS_FULL_P1) {
                    st.prime_hi(R2, R3);
                }
            }
        };
        auto setup_p3 = [&](const int l) {
            const PersistLayer& lw = p.layers[l];
            if constexpr (INT8) {
                if (tid < nruns3 * 16) {
                    s.rsc[tid] = r_sc3;
                }
            }
            if (tid < PS_RMAX) {
                s.misc[PS_RC1 + tid] = 0;
            }
            if (tid == PS_RMAX) {
                s.misc[PS_QH1] = 0;
            }
            load_p3_consts(l);
            sg3.w0 = reinterpret_cast<const char*>(lw.w_ffn2);
            sg3.w1 = reinterpret_cast<const char*>(lw.w_out);
            st.bind(sg3, s.rsc, s.xs, s.part, tid, s.b1, tag_base + (unsigned)l, &s.misc[32], (l - p.l_begin + 1) * PS_NC);
        };

        load_sc1(p.l_begin);
        setup_p1(p.l_begin);
        for (int l = p.l_begin; l < p.l_end; l++) {
            // opaque copies: keeps per-thread address arithmetic from being hoisted out of the layer loop, where it
            // becomes dozens of long-lived VGPRs that spill around the register batches
            asm volatile("" : "+v"(tid));
            const int           lane = tid & 63, wid = tid >> 6;
            const PersistLayer& lw  = p.layers[l];
            const unsigned      tag = tag_base + (unsigned)l;
            stamp(l, 0);
            // =========================== S0: layer input -> xraw (control waves) =================================
            if constexpr (CTRL) {
                if (l == p.l_begin) {
                    for (int i = tid * 8; i < M * H; i += PS_NC * 64 * 8) {
                        *reinterpret_cast<f16x8*>(s.xraw + i) = *reinterpret_cast<const f16x8*>(p.x_in + i);
                    }
                }
                else {
                    ps_sweep<20>(p.gx, M * H / 2, tid, PS_NC * 64, tag_base + (unsigned)(l - 1), p.err, 3,
                                [&](const int i, const unsigned v) { reinterpret_cast<unsigned*>(s.xraw)[i] = v; });
                }
            }
            __syncthreads();
            stamp(l, 1);
            // =========================== P1: LN1 / LN2, [QKV u FFN1] ===============================================
            {
                // LayerNorm x2 (layernorm_kernels.cu:157-286 arithmetic: fp32 statistics, var = E[x^2] - mean^2, half
                // normalise); both norms share the statistics of x
                float s0[M], s1[M];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    s0[m] = 0.f;
                    s1[m] = 0.f;
#pragma unroll
                    for (int k = 0; k < PS_NLN; k++) {
                        const int v = tid + k * PS_NT;
                        if (v * 8 < H) {
                            const f16x8 x8 = *reinterpret_cast<const f16x8*>(s.xraw + (size_t)m * H + v * 8);
#pragma unroll
                            for (int e = 0; e < 8; e++) {
                                const float f = (float)x8[e];
                                s0[m] += f;
                                s1[m] += f * f;
                            }
                        }
                    }
                    s0[m] = wave_sum(s0[m]);
                    s1[m] = wave_sum(s1[m]);
                    if (lane == 0) {
                        s.red[(m * PS_NW + wid) * 2]     = s0[m];
                        s.red[(m * PS_NW + wid) * 2 + 1] = s1[m];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < M; m++) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int w = 0; w < PS_NW; w++) {
                        a0 += s.red[(m * PS_NW + w) * 2];
                        a1 += s.red[(m * PS_NW + w) * 2 + 1];
                    }
                    const float mean = a0 / (float)H;
                    const float rstd = rsqrtf(a1 / (float)H - mean * mean + p.eps);
                    const f16   mh = (f16)mean, rh = (f16)rstd;
#pragma unroll
                    for (int k = 0; k < PS_NLN; k++) {
                        const int v = tid + k * PS_NT;
                        if (v * 8 < H) {
                            const f16x8 x8 = *reinterpret_cast<const f16x8*>(s.xraw + (size_t)m * H + v * 8);
                            f16x8       o1, o2;
#pragma unroll
                            for (int e = 0; e < 8; e++) {
                                const f16 nrm = (x8[e] - mh) * rh;
                                o1[e]         = (nrm * r_ln[0][k][e]) + r_ln[1][k][e];
                                o2[e]         = (nrm * r_ln[2][k][e]) + r_ln[3][k][e];
                            }
                            *reinterpret_cast<f16x8*>(s.xs + (size_t)m * (H + XPAD) + v * 8)       = o1;
                            *reinterpret_cast<f16x8*>(s.xs + (size_t)(M + m) * (H + XPAD) + v * 8) = o2;
                        }
                    }
                }
                if (tid < nruns1 * 16) {
                    s.b1[tid] = r_b1;  // (fetched in setup_p1, ahead of the prefetch: the oldest load in flight)
                }
                if constexpr (CTRL) {
                    st.prime_lo(R0, R1);
                }
                stamp(l, 2);
                __syncthreads();
                st.template run<CTRL || !PS_FULL_P1>(R0, R1, R2, R3);
                stamp(l, 3);
            }

            // =========================== attention ===============================================================
            asm volatile("" : "+v"(tid));
            PsAttn<DH, UK> at;
            const bool has_item = bid < n_items;
            int        a_sp = 0, a_h = 0, a_b = 0;
            if (has_item) {
                a_sp         = bid % p.plan.nsplit;
                const int hb = bid / p.plan.nsplit;
                a_h          = hb % p.nh;
                a_b          = hb / p.nh;
                // the wave has left the P1 queue: its register batches are free, the K / V rows can be on their way while
                // the other waves finish
#ifndef PS_EXP_LATE_KV
                at.issue(p, lw, a_h, a_b, a_sp, tid, R0, R1, R2, R3);
#endif
            }
            stamp(l, 4);
            __syncthreads();  // every wave is out of P1: slots / scales / queue state can be reused
            setup_p3(l);
            stamp(l, 5);
            bool live = false;
            u64* gall = p.ga + ((size_t)a_b * p.nh + a_h) * p.plan.nsplit * (DH + 2);
            if (has_item) {
#ifdef PS_EXP_LATE_KV
                at.issue(p, lw, a_h, a_b, a_sp, tid, R0, R1, R2, R3);
#endif
                at.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                stamp(l, 6);
                live = at.compute(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, R0, R1, R2, R3);
            }
            if constexpr (CTRL) {
                // the K range of mid this workgroup's FFN2 pieces read -> LDS.  Before the barrier, i.e. before the
                // streamer waves' prefetch burst (a sweep queued behind the burst took 5 us), and after the attention
                // (ahead of it, it made the attention wait for the slowest FFN1)
#pragma unroll
                for (int m = 0; m < M; m++) {
                    ps_sweep<10>(p.gm + (((size_t)m * Il + mid_lo) >> 1), (mid_hi - mid_lo) >> 1, tid, PS_NC * 64, tag,
                                 p.err, 6, [&](const int i, const unsigned v) {
                                     reinterpret_cast<unsigned*>(s.xs + (size_t)m * (Il + XPAD) + mid_lo)[i] = v;
                                 });
                }
            }
            stamp(l, 7);
            __syncthreads();  // mid staged, attention scratch free
            stamp(l, 8);
            if constexpr (!CTRL) {
                // AFTER the barrier: issuing 32 KiB per wave takes ~5 us (the CU's memory pipeline throttles the issue)
                // and the control waves, which carry the attention's critical path, must not wait for it
                st.prime_lo(R0, R1);  // the streamer waves issue no other load until the end of the P3 stream
                if constexpr (PS_FULL_P3) {
                    st.prime_hi(R2, R3);
                }
            }
            // =========================== P3: [FFN2 u out-proj] -> residual ========================================
            // The streamer waves start on the FFN2 pieces at once; the control waves finish the attention (merge of the
            // split partials by wave 0 of the split-0 workgroups), stage the K range of ctx the out-proj pieces read and
            // announce it through an LDS counter that gates every batch touching ctx (those sit at the END of the queue),
            // then join the queue.
            if constexpr (CTRL) {
                if (has_item && a_sp == 0 && wid == 0) {
                    if (live) {
                        ps_attn_merge<DH>(p, s.att, gall, tag, a_h, a_b, tid);
                    }
                    else {
                        ps_attn_publish_zero<DH>(p, tag, a_h, a_b, tid);
                    }
                    stamp(l, 13);
                }
#pragma unroll
                for (int m = 0; m < M; m++) {
                    ps_sweep<5>(p.gc + (((size_t)m * Hl + ctx_lo) >> 1), (ctx_hi - ctx_lo) >> 1, tid, PS_NC * 64, tag,
                                p.err, 7, [&](const int i, const unsigned v) {
                                    reinterpret_cast<unsigned*>(s.xs + (size_t)M * (Il + XPAD) + (size_t)m * (Hl + XPAD) + ctx_lo)[i] = v;
                                });
                }
                stamp(l, 14);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) {
                    atomicAdd(&s.misc[32], 1);  // DS operations of a wave execute in order: the writes above are visible
                }
                st.prime_lo(R0, R1);
            }
            stamp(l, 9);
            st.template run<CTRL || !PS_FULL_P3>(R0, R1, R2, R3);
            stamp(l, 10);
            const int  nmerge = s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE;
            const bool last   = (l == p.l_end - 1);
            // layer_input/output alias for 0 < l < L-1 in the reference (GptNeoXDecoder.cc:249-250) -> residual form
            const int inplace = (l > 0 && l < p.L - 1) ? 1 : 0;
            __syncthreads();  // every wave is out of P3: the streamer waves start the next layer's weight stream now
            asm volatile("" : "+v"(tid));
            // (unconditional, clamped -- see load_p3_consts: after the last layer this re-requests two batches of its weights)
            setup_p1(l + 1 < p.l_end ? l + 1 : l);
            stamp(l, 11);
            // merge the groups this workgroup owns (control waves) -> x'
            if constexpr (CTRL) {
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const int t = tid + k2 * PS_NC * 64;
                    if (t < nmerge * M * 16) {
                        const int k = t / (M * 16), r = t % (M * 16), m = r >> 4, c = r & 15;
                        const int g = s.misc[1 + k];
                        u64       gv[PS_MAXP];
                        int       spins = 0;
                        for (;;) {
                            bool ok = true;
#pragma unroll
                            for (int q = 0; q < PS_MAXP; q++) {
                                if (q < PA + PB) {
                                    const int rid = (q < PA) ? NG * PB + q * NG + g : (q - PA) * NG + g;
                                    gv[q]         = ld_granule(&p.gp[(size_t)rid * (M * 16) + r]);
                                }
                            }
#pragma unroll
                            for (int q = 0; q < PS_MAXP; q++) {
                                if (q < PA + PB) {
                                    ok &= ((unsigned)(gv[q] >> 32) == tag);
                                }
                            }
                            if (ok || ps_give_up(spins, p.err, 4)) {
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        float sa = 0.f, sb = 0.f;
#pragma unroll
                        for (int q = 0; q < PS_MAXP; q++) {  // piece order: deterministic
                            if (q < PA) {
                                sa += __uint_as_float((unsigned)gv[q]);
                            }
                            else if (q < PA + PB) {
                                sb += __uint_as_float((unsigned)gv[q]);
                            }
                        }
                        const int    n    = g * 16 + c;
                        const size_t oidx = (size_t)m * H + n;
                        const f16    attn = (f16)sa, ffn = (f16)sb;
                        const f16    xin  = (f16)((float)s.xraw[oidx] / (float)p.tp);
                        const f16    bb   = r_bres[k2];
                        f16          o;
                        if (inplace) {
                            o = (f16)((float)xin + (float)ffn + (float)attn + (float)bb);  // add_residual_kernels.cu:116-152
                        }
                        else {
                            o = ((ffn + attn) + bb) + xin;
                        }
                        if (last) {
                            p.x_out[oidx] = o;
                        }
                        else {
                            const unsigned b0 = f16_bits(o);
                            const unsigned b1 = __shfl_down(b0, 1, 64);
                            if ((c & 1) == 0) {
                                st_granule_u32(&p.gx[oidx >> 1], tag, b0 | (b1 << 16));
                            }
                        }
                    }
                }
            }
            stamp(l, 12);
            // xraw is rewritten by the next layer's gather: only the two control waves touch it between here and the
            // barrier after that gather, so they synchronise among themselves (the streamer waves are busy issuing
            // their prefetch; a workgroup barrier here would make the gather wait for that)
            if constexpr (CTRL) {
                if (lane == 0) {
                    atomicAdd(&s.misc[33], 1);
                }
                const int want = (l - p.l_begin + 1) * PS_NC;
                while (ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[33]) < want) {
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        }
    };
    if (wid < PS_NC) {
        body(std::true_type{});
    }
    else {
        body(std::false_type{});
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static size_t ps_smem_bytes(int M, int H, int xs_halves, int dh, int s_max, int nsplit, int nbmax)
{
    return (size_t)M * H * 2 + (size_t)xs_halves * 2 + (size_t)nbmax * M * 16 * 4 + ps_att_bytes(dh, s_max, nsplit)
           + 2 * sizeof(RunRec) * PS_RMAX + 2 * PS_RMAX * 16 * 2 + 64 * 4 + PS_MISC_INTS * 4 + 2 * (size_t)(nbmax + 1) * 8;
}

PersistPlan persist_plan(int B, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb)
{
    PersistPlan pl{};
    const int   M  = B;
    const int   TK = int8 ? TILE_K_I8 : TILE_K_F16;
    if (M < 1 || M > 2 || (dh != 64 && dh != 128) || H % TK || Hl % TK || Il % TK || H % 16 || Hl % 16 || Il % 16) {
        return pl;
    }
    if (H > PS_NLN * PS_NT * 8) {
        return pl;
    }
    const int NB = force_nb > 0 ? force_nb : num_cu;
    if (NB < 1 || B * nh > NB) {
        return pl;
    }
    const int KT = H / TK, KT_a = Hl / TK, KT_b = Il / TK, NG = H / 16, NT0h = 3 * Hl / 16, NFh = Il / 16;
    if ((NT0h + NB - 1) / NB + (NFh + NB - 1) / NB > PS_RMAX) {
        return pl;
    }
    int nsplit = NB / (B * nh);
    nsplit     = nsplit > MMHA_MAX_SPLIT ? MMHA_MAX_SPLIT : nsplit;
    const int chunk = ((((s_max + nsplit - 1) / nsplit) + 15) & ~15);
    if (chunk > PS_NW * (64 / (dh / 8)) * PS_UK_LONG) {
        return pl;  // the K/V rows of a split must fit the registers of one trip
    }
    pl.uk = chunk > PS_NW * (64 / (dh / 8)) * PS_UK ? PS_UK_LONG : PS_UK;
    // K pieces: best balanced tile count per workgroup
    double best = 1e30;
    for (int PA = 1; PA <= 8; PA *= 2) {
        for (int PB = 1; PB <= 16; PB *= 2) {
            if (PA + PB > PS_MAXP) {
                continue;
            }
            const int RLa = (KT_a + PA - 1) / PA, RLb = (KT_b + PB - 1) / PB;
            if ((PA - 1) * RLa >= KT_a || (PB - 1) * RLb >= KT_b || (PA > 1 && RLa < 4) || (PB > 1 && RLb < 4)) {
                continue;
            }
            long mx = 0, tot = 0;
            bool ok = true;
            for (int b = 0; b < NB && ok; b++) {
                const int rB0 = (int)((long)NG * PB * b / NB), rB1 = (int)((long)NG * PB * (b + 1) / NB);
                const int rA0 = (int)((long)NG * PA * b / NB), rA1 = (int)((long)NG * PA * (b + 1) / NB);
                if (rB1 - rB0 + rA1 - rA0 > PS_RMAX) {
                    ok = false;
                }
                long t  = 0;
                int  nm = 0;
                for (int i = rB0; i < rB1; i++) {
                    const int t0 = (i / NG) * RLb;
                    t += std::min(RLb, KT_b - t0);
                }
                for (int i = rA0; i < rA1; i++) {
                    const int t0 = (i / NG) * RLa;
                    t += std::min(RLa, KT_a - t0);
                    nm += (i / NG == PA - 1);
                }
                if (nm > PS_MAXMERGE) {
                    ok = false;
                }
                mx = std::max(mx, t);
                tot += t;
            }
            if (!ok || tot == 0) {
                continue;
            }
            const double cost = (double)mx * NB / (double)tot + 0.004 * (PA + PB);
            if (cost < best) {
                best   = cost;
                pl.PA  = PA;
                pl.PB  = PB;
                pl.RLa = RLa;
                pl.RLb = RLb;
            }
        }
    }
    if (best > 1e29) {
        return pl;
    }
    // batches per stage: the maximum over the workgroups (every run is padded to whole batches)
    int nbmax = 1;
    for (int b = 0; b < NB; b++) {
        const int nr1 = (int)((long)NT0h * (b + 1) / NB) - (int)((long)NT0h * b / NB)
                        + (int)((long)NFh * (b + 1) / NB) - (int)((long)NFh * b / NB);
        const int rB0 = (int)((long)NG * pl.PB * b / NB), rB1 = (int)((long)NG * pl.PB * (b + 1) / NB);
        const int rA0 = (int)((long)NG * pl.PA * b / NB), rA1 = (int)((long)NG * pl.PA * (b + 1) / NB);
        int       nb3 = 0;
        for (int i = rB0; i < rB1; i++) {
            nb3 += (std::min(pl.RLb, KT_b - (i / NG) * pl.RLb) + PS_U - 1) / PS_U;
        }
        for (int i = rA0; i < rA1; i++) {
            nb3 += (std::min(pl.RLa, KT_a - (i / NG) * pl.RLa) + PS_U - 1) / PS_U;
        }
        nbmax = std::max(nbmax, std::max(nr1 * ((KT + PS_U - 1) / PS_U), nb3));
    }
    pl.nbmax      = nbmax;
    pl.NB         = NB;
    pl.nsplit     = nsplit;
    pl.xs_halves  = M * std::max(2 * (H + XPAD), Il + Hl + 2 * XPAD);
    if (pl.xs_halves > 0x1ffff) {
        return pl;
    }
    pl.smem = ps_smem_bytes(M, H, pl.xs_halves, dh, s_max, nsplit, pl.nbmax);
    if (pl.smem > 160 * 1024) {
        return pl;
    }
    pl.ok = 1;
    return pl;
}

template<bool INT8, int M, int DH, int UK>
static const void* ps_kernel()
{
    return reinterpret_cast<const void*>(&k_decode_persistent<INT8, M, DH, UK>);
}
static const void* ps_kernel_for(bool int8, int M, int dh, int uk)
{
#define PS_SEL(I8, MM, D)                                                                                              \
    if (int8 == I8 && M == MM && dh == D) {                                                                            \
        return uk == PS_UK_LONG ? ps_kernel<I8, MM, D, PS_UK_LONG>() : ps_kernel<I8, MM, D, PS_UK>();                  \
    }
    PS_SEL(true, 1, 128)
#ifndef PS_ONLY_ONE
    PS_SEL(true, 2, 128)
    PS_SEL(true, 1, 64)
    PS_SEL(true, 2, 64)
    PS_SEL(false, 1, 128)
    PS_SEL(false, 2, 128)
    PS_SEL(false, 1, 64)
    PS_SEL(false, 2, 64)
#endif
#undef PS_SEL
    return nullptr;
}

// The hand-offs only work if EVERY workgroup of the grid is resident at once (one per CU at up to 160 KB of LDS).  A
// plain launch checks nothing, so the engine asks here before it chooses the persistent path: dynamic-LDS limit raised on
// THIS device (the attribute is per device, and launches of a multi-device process must not race on a static flag), and
// the occupancy query must admit the grid.  A cooperative launch would add the same check at every launch for +15-19 us
// per token (MI355X_MICROARCH.md "coop-launch"); checking once per plan is free.
bool persist_resident(const PersistPlan& pl, bool int8, int M, int dh, int num_cu)
{
    const void* k = ps_kernel_for(int8, M, dh, pl.uk);
    if (!pl.ok || !k) {
        return false;
    }
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, PS_NT, pl.smem) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return per_cu >= 1 && (long)per_cu * num_cu >= pl.NB;
}

void launch_decode_persistent(const PersistParams& p, bool int8, hipStream_t s)
{
    FTCF_CHECK_ARG(p.plan.ok && p.B >= 1 && p.B <= 2, "persistent decode: shape not eligible");
    FTCF_CHECK_ARG(p.dh == 64 || p.dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh && (p.rot == 0 || p.rot_table != nullptr), "bad rotary configuration");
    FTCF_CHECK_ARG(p.L <= 255, "at most 255 layers");
    const void* k = ps_kernel_for(int8, p.B, p.dh, p.plan.uk);
    FTCF_CHECK_ARG(k != nullptr, "persistent decode: no kernel for this shape");
    PersistParams pp     = p;
    void*         args[] = {&pp};
    FTCF_HIP_CHECK(hipLaunchKernel(k, dim3(p.plan.NB), dim3(PS_NT), args, p.plan.smem, s));
}

}  // namespace ftcf
