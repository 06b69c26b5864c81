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
#pragma once
#include <type_traits>

#include "attn_device.hip.h"
#include "gemv_device.hip.h"

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
constexpr int PS_UK       = 8;           // attention: K (and V) wave-loads per lane (256 keys per workgroup) ...
constexpr int PS_UK_LONG  = 16;          // ... or 16 (512 keys: 3072 tokens of context at six splits) for requests whose KV
                                         // split does not fit 256.  A second instantiation: 128 VGPRs of rows do not fit beside
                                         // the weight stream's register batches, so the long form keeps its rows IN those
                                         // batches (idle between the two streams) -- and the short form, the headline's, keeps
                                         // the code it was tuned with
#ifndef PS_FULL_P1_V
#define PS_FULL_P1_V true
#endif
#ifndef PS_FULL_P3_V
#define PS_FULL_P3_V false
#endif
// streamer waves: issue the whole first rotation (32 KiB) before the hand-off instead of half of it
constexpr bool PS_FULL_P1 = PS_FULL_P1_V, PS_FULL_P3 = PS_FULL_P3_V;
// Round 3 (profiles/r03_notes.md).  PS_EARLY_P3: the streamer waves request the first FFN2 weight batches (1: half a
// rotation, 2: the whole rotation) BEFORE the attention -- right after the K/V rows, while the workgroup waits for the
// slowest QKV producer anyway -- instead of after it; q/k/v are then swept by the control waves alone (a poll of a
// streamer wave would return behind its own 16-32 KiB).  Short attention form only (the long form keeps its K/V rows in
// the register batches).  PS_CTRL_EARLY: the control waves request their P3 share's first batches (1: half, 2: whole
// rotation) before they wait for ctx instead of after.
#ifndef PS_EARLY_P3
#define PS_EARLY_P3 0
#endif
#ifndef PS_CTRL_EARLY
#define PS_CTRL_EARLY 0
#endif
// Round 4.  PS_QKV_EARLY: every wave streams its slice of the QKV runs FIRST and its slice of the FFN1 runs after it (the first
// form cut the flat [QKV | FFN1] tile space into contiguous shares: all runs ended together, q/k/v were published with mid at the
// end of the stream and the attention waited 4.5 us for the slowest of its ~9 producer workgroups); the wave that flushes the
// LAST QKV partial sums of the workgroup publishes q/k/v right there, from inside the stream, ~55 % of the stream before the
// attention needs them: the hop disappears.  PS_KV_EARLY: the K rows of the workgroup's KV split are requested into register
// batch R0 when its last batch has been consumed, the V rows into R1 after its last batch (two batches before the stream
// ends): they land under the stream's tail and the epilogue instead of after it (26 MB per layer: 4 us at the full HBM rate).
#ifndef PS_QKV_EARLY
#define PS_QKV_EARLY 0
#endif
#ifndef PS_KV_EARLY
#define PS_KV_EARLY 0
#endif
// PS_ATT_WP: the eight-wave attention with wave-private soft-max statistics (PsAttn::compute_wp: three barriers instead of five)
#ifndef PS_ATT_WP
#define PS_ATT_WP 0
#endif
// PS_NF: register batches a wave keeps IN FLIGHT in the steady state (the fourth / third / second one has landed and
// waits to be consumed).  Everything a compute unit has outstanding sits in ONE in-order return queue, and whatever the
// chip has outstanding beyond bandwidth x unloaded latency only adds to the latency of every request -- the hand-off polls
// included (3 batches x 8 waves = 192 KiB per CU = 50 MB on the chip = 7.5 us of HBM time).
// PS_PACE: the first rotation is requested with at most PS_PACE batches of a wave in flight (0: back to back).
#ifndef PS_NF
#define PS_NF 3
#endif
#ifndef PS_PACE
#define PS_PACE 0
#endif
// one wave-wide LDS-DMA: lane i's 16 bytes at `gsrc` land at LDS byte address lds_dst + 16 * i (MI355X guide, section 5.7:
// M0 carries the LDS base and is compiler-reserved, so it is saved, set and restored inside ONE statement); the request
// counts on vmcnt like any load, the compiler does not know about it
__device__ __forceinline__ void ps_lds_dma16(const void* gsrc, const unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template<int N>
__device__ __forceinline__ void ps_wait_vm()  // at most N vector-memory operations of this wave outstanding
{
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
constexpr int PS_NLN      = 2;           // LayerNorm parameter vectors (f16x8) per thread and array: H <= 8192

typedef const PersistLayer PsLayerC;
#define PS_LAYER(p, l) ((p).layers[l])
#define PS_RLX __ATOMIC_RELAXED
#define PS_AGT __HIP_MEMORY_SCOPE_AGENT
// pointers that come out of the per-layer table in memory are GLOBAL: say so (a flat access also counts on lgkmcnt)
#define PS_G(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))

// batch-table entry (one per PS_U tiles of ONE run, consecutive k): bit 0 valid, 1 flush after the batch, 2..6 run,
// 8..24 LDS half offset of the first tile's x, 25 x stride select, 26 wait for the late x vector, 27..30 valid tiles
constexpr unsigned PS_BT_FAST = 1u, PS_BT_FLUSH = 2u, PS_BT_XSEL = 1u << 25, PS_BT_WAIT = 1u << 26;
// bit 31 (persist4_device.hip.h): after this batch's flush the wave bumps an LDS counter -- its last batch of the QKV runs
constexpr unsigned PS_BT_SIGNAL = 1u << 31;

struct RunRec {  // static per launch (LDS)
    int tile0;  // first tile of the run inside its weight array
    int sel;    // weight array of the stage (0 / 1)
    int nt;     // tiles
    int xoff;   // LDS half offset (inside the x region) of the run's first k
    int xsel;   // x row stride select
    int rid;    // stage specific id (P1: combined group, P3: global piece id)
    int grp;    // 16-column group
    int pad;
};

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

// wave w's share of T tiles: control waves get cs/16 of a streamer wave's share and sit at the END of the flat space
// (P3 puts the out-proj pieces there: the control waves are the ones that wait for ctx anyway).  Shares start at whole
// batches, so that with run lengths that are multiples of PS_U every batch lies inside one run.
__host__ __device__ inline void ps_wave_range(const int T, const int w, const int cs, int& tb, int& te)
{
    const int total = PS_NC * cs + (PS_NW - PS_NC) * 16;
    const int c0    = (w >= PS_NC) ? (w - PS_NC) * 16 : (PS_NW - PS_NC) * 16 + w * cs;
    const int c1    = c0 + ((w < PS_NC) ? cs : 16);
    tb              = (int)((long)T * c0 / total) / PS_U * PS_U;
    te              = (c1 == total) ? T : (int)((long)T * c1 / total) / PS_U * PS_U;
}
// The same with one weight per wave (wt[w], in 1/16 of a nominal share; flat order: waves PS_NC.. first, control waves last).
// Round 4: the waves of a workgroup do NOT stream at one rate -- the stamps show the first streamer waves ending a stage 2-6 us
// before the last ones on equal shares (issue arbitration favours the older wave), and the stage ends with the slowest wave.
__host__ __device__ inline void ps_wave_range_w(const int T, const int w, const int* wt, int& tb, int& te)
{
    int total = 0, c0 = 0;
    for (int i = 0; i < PS_NW; i++) {
        total += wt[i];
    }
    for (int k = 0; k < PS_NW; k++) {  // k-th wave of the flat order
        const int i = (k + PS_NC) % PS_NW;
        if (i == w) {
            break;
        }
        c0 += wt[i];
    }
    const int c1 = c0 + wt[w];
    total        = total > 0 ? total : 1;
    tb           = (int)((long)T * c0 / total) / PS_U * PS_U;
    te           = (c1 == total) ? T : (int)((long)T * c1 / total) / PS_U * PS_U;
}
// PS_QKV_EARLY: the two slices of wave w -- [qb, qe) of the Tq tiles of the QKV runs, then [fb, fe) of the Tf tiles of the FFN1
// runs, sized so that the waves' TOTALS follow the weights; slices start on batch boundaries
__host__ __device__ inline void ps_wave_range2_w(const int Tq, const int Tf, const int w, const int* wt, int& qb, int& qe, int& fb,
                                                 int& fe)
{
    int total = 0;
    for (int i = 0; i < PS_NW; i++) {
        total += wt[i];
    }
    total = total > 0 ? total : 1;
    auto al   = [](long v) { return (int)(v / PS_U * PS_U); };
    auto qend = [&](long c) { return c >= total ? Tq : al((long)Tq * c / total); };
    auto fend = [&](long c) {
        if (c >= total) {
            return Tf;
        }
        long f = (long)(Tq + Tf) * c / total - qend(c);
        f      = f < 0 ? 0 : (f > Tf ? Tf : f);
        return al(f);
    };
    int c0 = 0;
    for (int k = 0; k < PS_NW; k++) {  // flat order: waves PS_NC.. first, control waves last
        const int i = (k + PS_NC) % PS_NW;
        if (i == w) {
            break;
        }
        c0 += wt[i];
    }
    const int c1 = c0 + wt[w];
    qb = qend(c0);
    qe = qend(c1);
    fb = fend(c0);
    fe = fend(c1);
    if (fe < fb) {
        fe = fb;
    }
}
// table entries a wave needs for [tb, te) over runs of the given lengths: every run piece is padded to whole batches
template<typename NT>
__host__ __device__ inline int ps_wave_entries(const int nruns, NT&& run_nt, const int tb, const int te)
{
    int e = 0, pre = 0;
    for (int j = 0; j < nruns; j++) {
        const int nt = run_nt(j);
        const int a = tb > pre ? tb : pre, b = te < pre + nt ? te : pre + nt;
        if (b > a) {
            e += (b - a + PS_U - 1) / PS_U * PS_U;
        }
        pre += nt;
    }
    return e;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight stream of one wave over its share [tb, te) of the workgroup's flat tile space, driven by two per-wave LDS
// tables: lt[i] = {weight array select, tile index}, ct[i] = {x offset, run, flush, valid}.  PS_NBUF register batches
// of PS_U tiles rotate; the tables are padded to a whole number of rotations with entries that re-read tile 0 of the
// stage (an L2 / MALL hit, never HBM) and are not consumed, so the loop has NO conditional load: the compiler counts
// vmcnt exactly and three batches stay in flight while one is consumed.  The accumulator is flushed to
// part[run][wave] after the last tile of the wave's piece of a run (a wave meets a run in ONE contiguous piece).
// ---------------------------------------------------------------------------------------------------------------
struct PsStage {
    const unsigned *lt, *bt;  // per tile: {array select, tile index}; per batch: descriptor (see PS_BT_*)
    int             nrot;  // rotations (PS_NBUF batches each), >= 1
    const char *    w0, *w1;
    int             xs0, xs1;
};

template<bool INT8, int M, bool SIG = false>
struct PsStream {
    static constexpr int TK = TileK<INT8>::value;
    int* sig = nullptr;  // SIG: LDS counter bumped after a PS_BT_SIGNAL batch
    // PS_QKV_EARLY (first form of the kernel): the wave whose bump completes the workgroup's count publishes q/k/v
    int             sig_full = 0;      // count that completes this layer
    int             eq_n = 0;          // QKV runs of the workgroup x 16
    const RunRec*   eq_rt = nullptr;   // their records (column group ids)
    u64*            eq_g = nullptr;    // granule slab of q/k/v
    unsigned        eq_tag = 0;
    int             eq_hl3 = 0;        // 3 * Hl (row stride of the slab, halves)
    u32x4      R0[PS_U], R1[PS_U], R2[PS_U], R3[PS_U];
    f32x4      acc;
    PsStage    g;
    const f16* rsc;
    const f16* xs;
    float*     part;
    const int* flag;    // LDS arrival counter of the second x vector
    int        target;  // value it reaches when that vector is staged
    int        lane, wid;

    __device__ __forceinline__ void bind(const PsStage& g_, const f16* rsc_, const f16* xs_, float* part_, const int tx,
                                         const int* flag_ = nullptr, const int target_ = 0)
    {
        flag   = flag_;
        target = target_;
        g    = g_;
        rsc  = rsc_;
        xs   = xs_;
        part = part_;
        lane = tx & 63;
        wid  = ps_rfl(tx >> 6);
        acc  = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void load(u32x4 (&r)[PS_U], const int i)
    {
#pragma unroll
        for (int u = 0; u < PS_U; u++) {
            const unsigned e    = g.lt[i * PS_U + u];
            const char*    base = (e >> 31) ? g.w1 : g.w0;
            r[u] = __builtin_nontemporal_load(
                (const __attribute__((address_space(1))) u32x4*)(base + ((size_t)(e & 0x7fffffffu) * 64 + lane) * 16));
        }
    }
    __device__ __forceinline__ void flush(const int j)
    {
        if (lane < 16) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                part[((size_t)j * PS_NW + wid) * (M * 16) + m * 16 + lane] = acc_row(acc, m);
            }
        }
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void consume(const u32x4 (&r)[PS_U], const int i)
    {
        const unsigned bd = (unsigned)ps_rfl((int)g.bt[i]);
        if (!(bd & PS_BT_FAST)) {
            return;  // padding batch
        }
        if (bd & PS_BT_WAIT) {  // rare: only the first batch of a wave that touches the late vector actually spins
            while (ps_rfl(*(const volatile __attribute__((address_space(3))) int*)flag) < target) {
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const int  j   = (bd >> 2) & 31;
        const int  cnt = (bd >> 27) & 15;
        const f16* xr  = a_frag_ptr<INT8, M>(xs + ((bd >> 8) & 0x1ffffu), (bd & PS_BT_XSEL) ? g.xs1 : g.xs0, lane);
        f16x2      sc2 = {(f16)1.0f, (f16)1.0f};
        if constexpr (INT8) {
            const f16 sc = rsc[j * 16 + (lane & 15)];
            sc2          = f16x2{sc, sc};
        }
        if (cnt == PS_U) {  // straight-line code like the per-kernel GEMV stream
#pragma unroll
            for (int u = 0; u < PS_U; u++) {
                consume_tile<INT8, M>(r[u], xr + u * TK, sc2, acc);
            }
        }
        else {
#pragma unroll
            for (int u = 0; u < PS_U; u++) {
                if (u < cnt) {
                    consume_tile<INT8, M>(r[u], xr + u * TK, sc2, acc);
                }
            }
        }
        if (bd & PS_BT_FLUSH) {
            flush(j);
        }
        if constexpr (SIG) {
            if (bd & PS_BT_SIGNAL) {  // (DS operations of a wave execute in order: the flush above is visible first)
                if (eq_g == nullptr) {
                    if (lane == 0) {
                        atomicAdd(sig, 1);
                    }
                }
                else {
                    // the wave that completes the count has every wave's QKV partial sums in LDS (each bumped behind its own
                    // flushes): q/k/v = y (no bias: the attention adds it, like the reference's) -> granules, from inside the stream
                    int old = 0;
                    if (lane == 0) {
                        old = atomicAdd(sig, 1);
                    }
                    if (ps_rfl(old) + 1 == sig_full) {
                        for (int idx = lane; idx < eq_n; idx += 64) {
                            const int j = idx / (M * 16), r = idx % (M * 16), m = r >> 4, c = r & 15;
                            float     v = 0.f;
#pragma unroll
                            for (int w = 0; w < PS_NW; w++) {
                                v += part[((size_t)j * PS_NW + w) * (M * 16) + r];
                            }
                            const unsigned b0 = f16_bits((f16)v);
                            const unsigned b1 = next_lane_u32(b0);
                            if ((c & 1) == 0) {
                                st_granule_u32(eq_g + (((size_t)m * eq_hl3 + eq_rt[j].rid * 16 + c) >> 1), eq_tag, b0 | (b1 << 16));
                            }
                        }
                    }
                }
            }
        }
    }
    // the first rotation: issued before the hand-off this stage waits for.  The streamer waves issue only half of it
    // there and the rest when they start consuming: a 192 KiB burst per CU sits in FRONT of the control waves' sweeps in
    // the CU's memory pipeline and stretched each hand-off hop to 6-7 us (measured)
    __device__ __forceinline__ void prime_lo()
    {
        load(R0, 0);
        load(R1, 1);
    }
    __device__ __forceinline__ void prime_hi()
    {
        if constexpr (PS_PACE > 0 && PS_PACE < 3) {
            __builtin_amdgcn_sched_barrier(0);
            ps_wait_vm<PS_U*(PS_PACE - 1)>();
            load(R2, 2);
            __builtin_amdgcn_sched_barrier(0);
            ps_wait_vm<PS_U*(PS_PACE - 1)>();
            load(R3, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        else {
            load(R2, 2);
            load(R3, 3);
        }
    }
    __device__ __forceinline__ void prime()
    {
        prime_lo();
        prime_hi();
    }
    // HI: the second half of the first rotation is still to be issued.  Compile time: a load under a run-time condition
    // makes the compiler's vmcnt bookkeeping conservative for the whole stream (measured: 340 -> 192 tokens/s)
    template<bool HI>
    __device__ __forceinline__ void run()
    {
        if constexpr (HI) {
            prime_hi();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int last = (g.nrot - 1) * PS_NBUF;
        for (int i = 0; i < last; i += PS_NBUF) {
            consume(R0, i);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PS_NF < 3) {
                ps_wait_vm<PS_U*(PS_NF - 1)>();
            }
            load(R0, i + 4);
            __builtin_amdgcn_sched_barrier(0);
            consume(R1, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PS_NF < 3) {
                ps_wait_vm<PS_U*(PS_NF - 1)>();
            }
            load(R1, i + 5);
            __builtin_amdgcn_sched_barrier(0);
            consume(R2, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PS_NF < 3) {
                ps_wait_vm<PS_U*(PS_NF - 1)>();
            }
            load(R2, i + 6);
            __builtin_amdgcn_sched_barrier(0);
            consume(R3, i + 3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PS_NF < 3) {
                ps_wait_vm<PS_U*(PS_NF - 1)>();
            }
            load(R3, i + 7);
            __builtin_amdgcn_sched_barrier(0);
        }
        consume(R0, last);
        consume(R1, last + 1);
        consume(R2, last + 2);
        consume(R3, last + 3);
    }
    // the same stream with two hooks in its tail: after0() runs when register batch R0 has been consumed for the last time,
    // after1() when R1 has -- PS_KV_EARLY requests the K rows of the attention into R0 and the V rows into R1 there, two
    // batches before the stream ends
    template<bool HI, typename F0, typename F1>
    __device__ __forceinline__ void run_hooked(F0&& after0, F1&& after1)
    {
        if constexpr (HI) {
            prime_hi();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int last = (g.nrot - 1) * PS_NBUF;
        for (int i = 0; i < last; i += PS_NBUF) {
            consume(R0, i);
            __builtin_amdgcn_sched_barrier(0);
            load(R0, i + 4);
            __builtin_amdgcn_sched_barrier(0);
            consume(R1, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            load(R1, i + 5);
            __builtin_amdgcn_sched_barrier(0);
            consume(R2, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            load(R2, i + 6);
            __builtin_amdgcn_sched_barrier(0);
            consume(R3, i + 3);
            __builtin_amdgcn_sched_barrier(0);
            load(R3, i + 7);
            __builtin_amdgcn_sched_barrier(0);
        }
        consume(R0, last);
        __builtin_amdgcn_sched_barrier(0);
        after0();
        __builtin_amdgcn_sched_barrier(0);
        consume(R1, last + 1);
        __builtin_amdgcn_sched_barrier(0);
        after1();
        __builtin_amdgcn_sched_barrier(0);
        consume(R2, last + 2);
        consume(R3, last + 3);
    }
};

// One wave fills its tables for a stage from the workgroup's static run table: every piece of a run the wave owns is
// padded to whole batches (padding tiles re-read the wave's own first tile -- one shared address would be a hot spot --
// and are never consumed), so a batch never spans two runs.  One lane per batch.
template<int TK>
__device__ __forceinline__ void ps_build_tables(const RunRec* rt, const int nruns, const int tb, const int te,
                                                unsigned* lt, unsigned* bt, const int entries, const int jbase = 0)
{
    const int lane = threadIdx.x & 63;
    // the wave's first tile (padding address): uniform
    unsigned pad = 0u;
    {
        int pre = 0;
        for (int j = 0; j < nruns; j++) {
            const int nt = rt[j].nt;
            if (tb < te && tb >= pre && tb < pre + nt) {
                pad = ((unsigned)rt[j].sel << 31) | (unsigned)(rt[j].tile0 + tb - pre);
            }
            pre += nt;
        }
    }
    for (int bi = lane; bi < entries / PS_U; bi += 64) {
        // locate batch bi: pieces of the runs intersecting [tb, te), each padded to whole batches
        int      pre = 0, eb = 0;  // tiles before run j, batches before run j's piece
        unsigned bd  = 0u;
        int      first = 0, cnt = 0, sel = 0;
        for (int j = 0; j < nruns; j++) {
            const RunRec r  = rt[j];
            const int    a  = tb > pre ? tb : pre, b = te < pre + r.nt ? te : pre + r.nt;
            const int    nb = b > a ? (b - a + PS_U - 1) / PS_U : 0;
            if (bi >= eb && bi < eb + nb) {
                const int t   = a + (bi - eb) * PS_U;
                const int off = t - pre;
                cnt   = (b - t < PS_U) ? b - t : PS_U;
                first = r.tile0 + off;
                sel   = r.sel;
                bd    = PS_BT_FAST | ((t + cnt == b) ? PS_BT_FLUSH : 0u) | ((unsigned)(j + jbase) << 2)
                     | ((unsigned)(r.xoff + off * TK) << 8) | (r.xsel ? (PS_BT_XSEL | PS_BT_WAIT) : 0u)
                     | ((unsigned)cnt << 27);
            }
            eb += nb;
            pre += r.nt;
        }
        for (int u = 0; u < PS_U; u++) {
            lt[bi * PS_U + u] = (u < cnt) ? (((unsigned)sel << 31) | (unsigned)(first + u)) : pad;
        }
        bt[bi] = bd;
    }
}

struct PsSmem {
    f16*      xraw;  // [M][H]
    f16*      xs;    // x region (P1: LN1(x) | LN2(x) ; P3: mid | ctx)
    float*    part;  // [RMAX][NW][M*16]
    float*    part3; // P3's own partial sums (one row, not A3): no barrier between the P1 epilogue and the P3 set-up
    char*     att;   // attention scratch
    RunRec*   rt1;   // [RMAX] P1 runs
    RunRec*   rt3;   // [RMAX] P3 runs
    f16*      rsc;   // [RMAX][16] scales of the current stage
    f16*      rsc3;  // A3: P3's scales in their own array (written before the barrier that ends P1)
    float*    red;   // 64
    int*      misc;  // 64: [0] nmerge, [1..8] merge groups
    unsigned *lt1, *lt3;  // [NW][e1], [NW][e3]
    unsigned *bt1, *bt3;  // [NW][e1 / PS_U], [NW][e3 / PS_U]
    char*     kbuf;       // A3: K rows of the workgroup's KV split, [UK][NW][1 KiB] (LDS-DMA destination)
};

__host__ __device__ inline size_t ps_att_bytes(int dh, int s_max, int nsplit)
{
    const int chunk = ((((s_max + nsplit - 1) / nsplit) + 15) & ~15);
    size_t    a     = (size_t)3 * dh * 2 + (size_t)(2 * PS_NW + PS_NW * dh) * 4 + (size_t)chunk * 4;
    size_t    b     = (size_t)(nsplit * (dh + 2) + nsplit + 4) * 4;
    return ((a > b ? a : b) + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------------------------------------------
// attention of one (row b, head h, split sp) on the whole 8-wave workgroup
// (decoder_masked_multihead_attention_template.hpp:1099-1919; same arithmetic as attn_device.hip.h::mmha_partial)
// ---------------------------------------------------------------------------------------------------------------
// A8 (PS_KV_EARLY, UK == PS_U): the rows live in the stream's register batches too -- K in R0, V in R1 -- requested from inside
// the first stream's tail (issue_k / issue_v) instead of after it
template<int DH, int UK, int NSW = 3 * DH / 2, bool A8 = false>
struct PsAttn {
    static constexpr int LPK = DH / 8;
    static constexpr int NQ  = 3 * DH / 2;             // q | k | v granules (pairs of halves) of one head
    static constexpr int NB2 = (NQ + NSW - 1) / NSW;   // granules per sweeping thread (NSW threads sweep)
    static constexpr int KPI = 64 / LPK;
    static constexpr bool ALIAS = UK > PS_U || A8;  // rows live in the stream's register batches: K in R0 | R1, V in R2 | R3
    static_assert(UK <= 2 * PS_U, "K rows live in R0|R1, V rows in R2|R3");
    static_assert(!A8 || UK == PS_U, "A8: one register batch of K rows, one of V rows");
    u32x4 kreg[ALIAS ? 1 : UK], vreg[ALIAS ? 1 : UK];
    template<typename ST>
    __device__ __forceinline__ u32x4& kr(ST& st, const int u)
    {
        if constexpr (ALIAS) {
            return u < PS_U ? st.R0[u % PS_U] : st.R1[u % PS_U];
        }
        else {
            return kreg[u];
        }
    }
    template<typename ST>
    __device__ __forceinline__ u32x4& vr(ST& st, const int u)
    {
        if constexpr (A8) {
            return st.R1[u];
        }
        else if constexpr (ALIAS) {
            return u < PS_U ? st.R2[u % PS_U] : st.R3[u % PS_U];
        }
        else {
            return vreg[u];
        }
    }
    // A8: the rows alone, K and V apart (the masks, lengths, rotary coefficients and bias follow with issue_impl<false>)
    template<bool VROWS, typename ST>
    __device__ __forceinline__ void issue_rows(const PersistParams& p, PsLayerC& lw, int h, int b, int sp, const int tx, ST& st,
                                               const bool item)
    {
        const int lane = tx & 63, wid = tx >> 6;
        const int sub = lane % LPK, grp = lane / LPK;
        const int ck = (((p.s_max + p.plan.nsplit - 1) / p.plan.nsplit) + 15) & ~15;
        const int tb = sp * ck;
        const auto* rc = PS_G(f16, VROWS ? lw.v_cache : lw.k_cache) + ((size_t)b * p.nh + h) * p.s_max * DH;
        int t_last = tb + ck - 1;
        t_last     = t_last < p.s_max ? t_last : p.s_max - 1;
        t_last     = item ? t_last : tb;
#pragma unroll
        for (int u = 0; u < UK; u++) {
            int t = tb + u * PS_NW * KPI + wid * KPI + grp;
            t     = t < t_last ? t : t_last;
            if constexpr (VROWS) {
                vr(st, u) = *PS_G(u32x4, rc + (size_t)t * DH + sub * 8);
            }
            else {
                kr(st, u) = *PS_G(u32x4, rc + (size_t)t * DH + sub * 8);
            }
        }
    }
    unsigned mask_bits, bias2[NB2];
    int      tl, chunk, t_beg;
    float    rot_cs, rot_sn;
    bool     fin;

    // loads that do not depend on this step's qkv: K/V rows of the whole fixed chunk, masks, lengths, rotary table
    template<typename ST>
    __device__ __forceinline__ void issue(const PersistParams& p, PsLayerC& lw, int h, int b, int sp, const int tx, ST& st,
                                          const bool item = true)
    {
        issue_impl<true>(p, lw, h, b, sp, tx, st, item);
    }
    // `item` false: a workgroup without a (row, head, split) that must not assign the row registers under a condition (they
    // would be carried around the layer loop): it requests ONE cached row over and over
    template<bool ROWS, typename ST>
    __device__ __forceinline__ void issue_impl(const PersistParams& p, PsLayerC& lw, int h, int b, int sp, const int tx, ST& st,
                                               const bool item = true)
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
        t_last     = item ? t_last : t_beg;
        if constexpr (ROWS) {
#pragma unroll
            for (int u = 0; u < UK; u++) {
                int t   = t_beg + u * PS_NW * KPI + wid * KPI + grp;
                t       = t < t_last ? t : t_last;
                kr(st, u) = *PS_G(u32x4, kc + (size_t)t * DH + sub * 8);
            }
#pragma unroll
            for (int u = 0; u < UK; u++) {
                int t   = t_beg + u * PS_NW * KPI + wid * KPI + grp;
                t       = t < t_last ? t : t_last;
                vr(st, u) = *PS_G(u32x4, vc + (size_t)t * DH + sub * 8);
            }
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
#pragma unroll
        for (int k = 0; k < NB2; k++) {  // this thread's pairs of q / k / v bias values (sweep_qkv)
            const int gi = tx + k * NSW;
            bias2[k]     = 0u;
            if (tx < NSW && gi < NQ) {
                const int seg = gi / (DH / 2), i = gi % (DH / 2);
                bias2[k] = *PS_G(unsigned, reinterpret_cast<const unsigned*>(lw.b_qkv + (size_t)seg * p.nh * DH + h * DH) + i);
            }
        }
        fin = p.finished && p.finished[b];
        tl  = p.seq_len[b];
    }
    // ---- A3: the whole split on the two control waves -----------------------------------------------------------
    // The split's rows are cut into 1-KiB blocks (64 / LPK keys each, the unit of one wave-wide request); control wave c
    // owns blocks j = u * PS_NW + c * (PS_NW / PS_NC) + q  (u < UK, q < PS_NW / PS_NC): it requests their K rows into
    // LDS block j of kbuf (the rows land as [key][DH] halves, lane i's 16 bytes at 16 i -- the order it reads them back
    // in), their V rows into its register batches R0..R3, and later computes exactly those keys.
    static constexpr int NBLK = UK * PS_NW / PS_NC;
    static constexpr int BPC  = PS_NW / PS_NC;
    template<typename ST>
    __device__ __forceinline__ u32x4& vrow(ST& st, const int i)
    {
        return i < PS_U ? st.R0[i % PS_U] : i < 2 * PS_U ? st.R1[i % PS_U] : i < 3 * PS_U ? st.R2[i % PS_U] : st.R3[i % PS_U];
    }
    // `item`: false for a workgroup without a (row, head, split) -- it requests one cached row over and over (registers
    // assigned under a condition would be carried around the layer loop)
    template<typename ST>
    __device__ __forceinline__ void issue_ctrl(const PersistParams& p, PsLayerC& lw, int h, int b, int sp, const int tx,
                                               ST& st, const unsigned kbuf_lds, const bool item)
    {
        static_assert(NBLK <= 32 && NBLK == PS_NBUF * PS_U, "V rows of a control wave fill its four register batches");
        const int lane = tx & 63, c = tx >> 6;
        const int sub = lane % LPK, grp = lane / LPK;
        chunk = (((p.s_max + p.plan.nsplit - 1) / p.plan.nsplit) + 15) & ~15;
        t_beg = sp * chunk;
        const f16*  kc = lw.k_cache + ((size_t)b * p.nh + h) * p.s_max * DH;
        const auto* vc = PS_G(f16, lw.v_cache) + ((size_t)b * p.nh + h) * p.s_max * DH;
        int t_last = t_beg + chunk - 1;
        t_last     = t_last < p.s_max ? t_last : p.s_max - 1;
        t_last     = item ? t_last : t_beg;
        // (a block's K request and V row together, four blocks at a time: requested in two passes, the 32 clamped row indices
        // -- two registers each -- stay live from the first pass to the second, beside the 128 registers of V rows: the role
        // spilled)
#pragma unroll
        for (int i = 0; i < NBLK; i++) {
            const int j = (i / BPC) * PS_NW + c * BPC + i % BPC;
            int       t = t_beg + j * KPI + grp;
            t           = t < t_last ? t : t_last;
            ps_lds_dma16(kc + (size_t)t * DH + sub * 8, (unsigned)ps_rfl((int)(kbuf_lds + (unsigned)j * 1024u)));
            vrow(st, i) = *PS_G(u32x4, vc + (size_t)t * DH + sub * 8);
            if (i % 4 == 3) {
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mask_bits = 0u;
        if (p.masked_tokens && sub == 0) {
#pragma unroll
            for (int i = 0; i < NBLK; i++) {
                const int j = (i / BPC) * PS_NW + c * BPC + i % BPC;
                int       t = t_beg + j * KPI + grp;
                t           = t < t_last ? t : t_last;
                mask_bits |= (p.masked_tokens[(size_t)b * p.s_max + t] ? 1u : 0u) << i;
            }
        }
        rot_cs = 1.f;
        rot_sn = 0.f;
        if (p.rot > 0 && tx < p.rot / 2) {
            rot_cs = p.rot_table[((size_t)b * (p.rot / 2) + tx) * 2];
            rot_sn = p.rot_table[((size_t)b * (p.rot / 2) + tx) * 2 + 1];
        }
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            const int gi = tx + k * NSW;
            bias2[k]     = 0u;
            if (tx < NSW && gi < NQ) {
                const int seg = gi / (DH / 2), i = gi % (DH / 2);
                bias2[k] = *PS_G(unsigned, reinterpret_cast<const unsigned*>(lw.b_qkv + (size_t)seg * p.nh * DH + h * DH) + i);
            }
        }
        fin = p.finished && p.finished[b];
        tl  = p.seq_len[b];
    }
    // same arithmetic as compute() (decoder_masked_multihead_attention_template.hpp:1099-1919) on PS_NC waves; bar2 is the
    // control waves' own barrier.  Returns false when the row is finished (nothing published).
    template<typename ST, typename BAR>
    __device__ __forceinline__ bool compute_ctrl(const PersistParams& p, PsLayerC& lw, char* smem, const char* kbuf, u64* gout,
                                                 const unsigned tag, int h, int b, const int tx, ST& st, BAR&& bar2)
    {
        constexpr int NTC = PS_NC * 64;
        const int     lane = tx & 63, c = tx >> 6;
        const int     sub = lane % LPK, grp = lane / LPK;
        if (fin) {
            return false;  // :1176
        }
        int t_end = t_beg + chunk;
        if (t_end > tl + 1) {
            t_end = tl + 1;
        }
        if (t_beg > tl) {  // empty split
            for (int d = tx; d < DH; d += NTC) {
                st_granule(&gout[d], tag, 0.f);
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
        float* s_red = reinterpret_cast<float*>(s_v + DH);  // [2*NW + NW*DH] (sized for the eight-wave form)
        float* s_p   = s_red + 2 * PS_NW + PS_NW * DH;      // [chunk]
        bar2();  // q | k | v (+ bias) written by sweep_qkv
        if (p.rot > 0 && tx < p.rot / 2) {
            const int j = tx;
            f16       a = s_q[j], c2 = s_q[j + p.rot / 2];
            rotary_apply(a, c2, rot_cs, rot_sn);
            s_q[j]             = a;
            s_q[j + p.rot / 2] = c2;
            if (owns_cur) {
                f16 ka = s_k[j], kc2 = s_k[j + p.rot / 2];
                rotary_apply(ka, kc2, rot_cs, rot_sn);
                s_k[j]             = ka;
                s_k[j + p.rot / 2] = kc2;
            }
        }
        bar2();
        if (owns_cur) {  // append to the cache (:1397, :1837)
            for (int d = tx; d < DH; d += NTC) {
                ((__attribute__((address_space(1))) f16*)lw.k_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + d] = s_k[d];
                ((__attribute__((address_space(1))) f16*)lw.v_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + d] = s_v[d];
            }
        }
        // Each control wave runs its own keys through the whole soft-max (scores in a wave-private part of s_p, its own
        // maximum, sum and un-normalised output); the two partials are combined like the splits of a (row, head) are:
        // no exchange between the waves before the end, no LDS crossbar in the reductions.
        const float inv_sqrt_dh = rsqrtf((float)DH);
        const f16x8 qv          = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
        float       lmax        = -INFINITY;
        static_assert(BPC == 4, "four key blocks per trip");
#pragma unroll 1
        for (int u = 0; u < UK; u++) {  // (rolled: nothing of it can be hoisted above the V rows' long lives)
            const char* kb = kbuf + (size_t)(u * PS_NW + c * BPC) * 1024 + lane * 16;
            // (two key blocks in flight, pinned: with all four -- 16 registers of K beside the 128 of V rows -- the role spilled)
#pragma unroll
            for (int q = 0; q < BPC; q++) {
                if (q % 2 == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f16x8 kq = *reinterpret_cast<const f16x8*>(kb + q * 1024);
                const int   t = t_beg + (u * PS_NW + c * BPC + q) * KPI + grp;
                float       a = 0.f;
                a             = dot2(f16x2{qv[0], qv[1]}, f16x2{kq[0], kq[1]}, a);
                a             = dot2(f16x2{qv[2], qv[3]}, f16x2{kq[2], kq[3]}, a);
                a             = dot2(f16x2{qv[4], qv[5]}, f16x2{kq[4], kq[5]}, a);
                a             = dot2(f16x2{qv[6], qv[7]}, f16x2{kq[6], kq[7]}, a);
                a             = group_sum_dpp<LPK>(a) * inv_sqrt_dh;
                const bool m = ((mask_bits >> (u * BPC + q)) & 1u) != 0u;
                a            = m ? -INFINITY : a;
                if (t < t_cached_end && sub == 0) {
                    s_p[t - t_beg] = a;
                    lmax           = fmaxf(lmax, a);
                }
            }
        }
        float cur_p = -INFINITY;  // wave 0: score of the current token (:1407-1437), from LDS
        if (owns_cur && c == 0) {
            float a = 0.f;
            if (lane < LPK) {
                const f16x8 kv = *reinterpret_cast<const f16x8*>(s_k + lane * 8);
                const f16x8 q8 = *reinterpret_cast<const f16x8*>(s_q + lane * 8);
                a              = dot2(f16x2{q8[0], q8[1]}, f16x2{kv[0], kv[1]}, a);
                a              = dot2(f16x2{q8[2], q8[3]}, f16x2{kv[2], kv[3]}, a);
                a              = dot2(f16x2{q8[4], q8[5]}, f16x2{kv[4], kv[5]}, a);
                a              = dot2(f16x2{q8[6], q8[7]}, f16x2{kv[6], kv[7]}, a);
            }
            cur_p = wave_sum_dpp(a) * inv_sqrt_dh;
            lmax  = fmaxf(lmax, cur_p);
        }
        const float m_c = wave_max_dpp(lmax);
        float       acc[8];
        float       lsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            acc[e] = 0.f;
        }
#pragma unroll
        for (int i0 = 0; i0 < NBLK; i0 += 4) {
            float sc[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = ((i0 + q) / BPC) * PS_NW + c * BPC + (i0 + q) % BPC;
                const int t = t_beg + j * KPI + grp;
                // (rows beyond tlength were fetched speculatively and may hold anything: weight 0)
                sc[q] = (t < t_cached_end) ? s_p[t - t_beg] : -INFINITY;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float pt = (sc[q] == -INFINITY) ? 0.f : __expf(sc[q] - m_c);
                lsum += (sub == 0) ? pt : 0.f;
                const f16x8 vv = __builtin_bit_cast(f16x8, vrow(st, i0 + q));
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    acc[e] = (sc[q] == -INFINITY) ? acc[e] : fmaf(pt, (float)vv[e], acc[e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // (one group of four key blocks at a time: interleaved, the groups' scores,
                                                // weights and converted rows pile up beside the 128 registers of V rows)
        }
        if (owns_cur && c == 0) {
            const float pt = __expf(cur_p - m_c);
            lsum += (lane == 0) ? pt : 0.f;
            if (grp == 0) {
                const f16x8 vv = *reinterpret_cast<const f16x8*>(s_v + sub * 8);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    acc[e] = fmaf(pt, (float)vv[e], acc[e]);
                }
            }
        }
        const float l_c = wave_sum_dpp(lsum);
        // un-normalised outputs of the wave's four key groups (DH = 64: eight groups, folded in pairs first) -> LDS
        constexpr int NG4 = 4;
        if constexpr (LPK == 8) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                acc[e] += dpp_read<0x128>(acc[e]);  // row_ror:8 -- the other key group of the row
            }
        }
        float*    s_o = s_red + 2 * PS_NW;  // [NC][4][DH]
        const int g4  = lane >> 4;
        if (LPK == 16 || (lane & 8) == 0) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                s_o[(size_t)(c * NG4 + g4) * DH + sub * 8 + e] = acc[e];
            }
        }
        if (lane == 0) {
            s_red[c]         = m_c;
            s_red[PS_NW + c] = l_c;
        }
        bar2();
        float m_loc = s_red[0];
#pragma unroll
        for (int w = 1; w < PS_NC; w++) {
            m_loc = fmaxf(m_loc, s_red[w]);
        }
        float wgt[PS_NC], ls = 0.f;
#pragma unroll
        for (int w = 0; w < PS_NC; w++) {
            wgt[w] = (s_red[w] == -INFINITY) ? 0.f : __expf(s_red[w] - m_loc);
            ls += wgt[w] * s_red[PS_NW + w];
        }
        for (int d = tx; d < DH; d += NTC) {
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < PS_NC; w++) {
                float ow = 0.f;
#pragma unroll
                for (int g = 0; g < NG4; g++) {
                    ow += s_o[(size_t)(w * NG4 + g) * DH + d];
                }
                o += wgt[w] * ow;
            }
            st_granule(&gout[d], tag, o);
        }
        if (tx == 0) {
            st_granule(&gout[DH], tag, m_loc);
            st_granule(&gout[DH + 1], tag, ls);
        }
        bar2();  // the scratch is free (the merge of a split-0 workgroup reuses it)
        return true;
    }
    // q/k/v of the current token: granules published by the QKV stage of THIS launch (pairs of halves); + bias -> LDS
    __device__ __forceinline__ void sweep_qkv(const PersistParams& p, char* smem, const unsigned tag, int h, int b, const int tx)
    {
        if (fin) {
            return;
        }
        f16* s_q = reinterpret_cast<f16*>(smem);  // [DH] q | [DH] k | [DH] v
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            const int gi = tx + k * NSW;
            if (tx < NSW && gi < NQ) {
                const int  seg = gi / (DH / 2), i = gi % (DH / 2);
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
                s_q[seg * DH + 2 * i]     = x0 + bits_f16(bias2[k]);
                s_q[seg * DH + 2 * i + 1] = x1 + bits_f16(bias2[k] >> 16);
            }
        }
    }
    // the current token's rotated K and its V (LDS, left there by compute<false>) -> the cache (:1397, :1837)
    __device__ __forceinline__ void append_current(const PersistParams& p, PsLayerC& lw, char* smem, int h, int b, const int tx)
    {
        if (fin || t_beg > tl) {
            return;
        }
        int t_end = t_beg + chunk;
        t_end     = t_end > tl + 1 ? tl + 1 : t_end;
        if (tl >= t_beg && tl < t_end && tx < DH) {
            const f16* s_k = reinterpret_cast<const f16*>(smem) + DH;
            const f16* s_v = s_k + DH;
            ((__attribute__((address_space(1))) f16*)lw.k_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_k[tx];
            ((__attribute__((address_space(1))) f16*)lw.v_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_v[tx];
        }
    }
    // Round 4 (PS_ATT_WP): the same split on the eight waves with WAVE-PRIVATE soft-max statistics.  compute() below takes five
    // workgroup barriers (q/k/v staged, rotary, maxima, exponentials, outputs) and sends every score through LDS twice; here a
    // wave runs its own keys -- 4 (dh = 128) or 8 (dh = 64) per register row, UK rows -- through the whole soft-max against ITS
    // OWN maximum (scores, probabilities and the weighted V rows never leave its registers), and the eight {maximum, sum,
    // un-normalised output} triples are combined like the splits of a (row, head) are: three barriers, no score buffer.  The
    // arithmetic is compute_ctrl's (decoder_masked_multihead_attention_template.hpp:1099-1919 with fp32 probabilities);
    // against compute() only the reference point of the exponentials differs (own maximum, rescaled at the end).
    template<bool APPEND = true, typename ST>
    __device__ __forceinline__ bool compute_wp(const PersistParams& p, PsLayerC& lw, char* smem, u64* gout, const unsigned tag, int h,
                                               int b, const int tx, ST& st)
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
        float* s_red = reinterpret_cast<float*>(s_v + DH);  // [NW] maxima | [NW] sums | [NW][DH] outputs
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
        if (APPEND && owns_cur && tx < DH) {  // append to the cache (:1397, :1837)
            ((__attribute__((address_space(1))) f16*)lw.k_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_k[tx];
            ((__attribute__((address_space(1))) f16*)lw.v_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_v[tx];
        }
        const float inv_sqrt_dh = rsqrtf((float)DH);
        const f16x8 qv          = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
        float       sc[UK];
        float       lmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const f16x8 kv = __builtin_bit_cast(f16x8, kr(st, u));
            float       a  = 0.f;
            a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
            a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
            a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
            a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
            a              = group_sum_dpp<LPK>(a) * inv_sqrt_dh;
            const int  t   = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            // (rows beyond tlength were fetched speculatively and may hold anything; masked tokens: weight 0)
            const bool ok = t < t_cached_end && ((mask_bits >> u) & 1u) == 0u;
            sc[u]         = ok ? a : -INFINITY;
            lmax          = fmaxf(lmax, sc[u]);
        }
        float cur_p = -INFINITY;  // wave 0: the current token from LDS (:1407-1437)
        if (owns_cur && wid == 0) {
            float a = 0.f;
            if (lane < LPK) {
                const f16x8 kv = *reinterpret_cast<const f16x8*>(s_k + lane * 8);
                const f16x8 q8 = *reinterpret_cast<const f16x8*>(s_q + lane * 8);
                a              = dot2(f16x2{q8[0], q8[1]}, f16x2{kv[0], kv[1]}, a);
                a              = dot2(f16x2{q8[2], q8[3]}, f16x2{kv[2], kv[3]}, a);
                a              = dot2(f16x2{q8[4], q8[5]}, f16x2{kv[4], kv[5]}, a);
                a              = dot2(f16x2{q8[6], q8[7]}, f16x2{kv[6], kv[7]}, a);
            }
            cur_p = wave_sum_dpp(a) * inv_sqrt_dh;
            lmax  = fmaxf(lmax, cur_p);
        }
        const float m_w = wave_max_dpp(lmax);
        float       acc[8];
        float       lsum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const bool  ok  = sc[u] != -INFINITY;
            const float pt  = ok ? __expf(sc[u] - m_w) : 0.f;
            const u32x4 raw = vr(st, u);
            const u32x4 vz  = {ok ? raw.x : 0u, ok ? raw.y : 0u, ok ? raw.z : 0u, ok ? raw.w : 0u};
            const f16x8 vv  = __builtin_bit_cast(f16x8, vz);
            lsum += (sub == 0) ? pt : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                acc[j] = fmaf(pt, (float)vv[j], acc[j]);
            }
        }
        if (owns_cur && wid == 0) {
            const float pt = __expf(cur_p - m_w);
            lsum += (lane == 0) ? pt : 0.f;
            if (grp == 0) {
                const f16x8 vv = *reinterpret_cast<const f16x8*>(s_v + sub * 8);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc[j] = fmaf(pt, (float)vv[j], acc[j]);
                }
            }
        }
        const float l_w = wave_sum_dpp(lsum);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = across_groups_sum<LPK>(acc[j]);
        }
        float* s_o = s_red + 2 * PS_NW;  // [NW][DH]
        if (grp == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                s_o[wid * DH + sub * 8 + j] = acc[j];
            }
        }
        if (lane == 0) {
            s_red[wid]         = m_w;
            s_red[PS_NW + wid] = l_w;
        }
        __syncthreads();
        float m_loc = s_red[0];
#pragma unroll
        for (int w = 1; w < PS_NW; w++) {
            m_loc = fmaxf(m_loc, s_red[w]);
        }
        float wgt[PS_NW], ls = 0.f;
#pragma unroll
        for (int w = 0; w < PS_NW; w++) {  // wave order: deterministic
            wgt[w] = (s_red[w] == -INFINITY) ? 0.f : __expf(s_red[w] - m_loc);
            ls += wgt[w] * s_red[PS_NW + w];
        }
        if (tx < DH) {
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < PS_NW; w++) {
                o += wgt[w] * s_o[w * DH + tx];
            }
            st_granule(&gout[tx], tag, o);
        }
        if (tx == 0) {
            st_granule(&gout[DH], tag, m_loc);
            st_granule(&gout[DH + 1], tag, ls);
        }
        return true;
    }
    // returns false when the row is finished (nothing published)
    // APPEND false: the caller appends the current token's K / V to the cache itself, after the call (append_current) -- the
    // attention then contains no global-memory operation before its last barrier, hence no vmcnt wait the compiler would
    // place for one (P3L: such a wait would also wait for the LDS-DMA requests in flight)
    template<bool APPEND = true, typename ST>
    __device__ __forceinline__ bool compute(const PersistParams& p, PsLayerC& lw, char* smem, u64* gout,
                                            const unsigned tag, int h, int b, const int tx, ST& st)
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
        if (APPEND && owns_cur && tx < DH) {  // append to the cache (:1397, :1837)
            ((__attribute__((address_space(1))) f16*)lw.k_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_k[tx];
            ((__attribute__((address_space(1))) f16*)lw.v_cache)[(((size_t)b * p.nh + h) * p.s_max + tl) * DH + tx] = s_v[tx];
        }
        const float inv_sqrt_dh = rsqrtf((float)DH);
        const f16x8 qv          = *reinterpret_cast<const f16x8*>(s_q + sub * 8);
        float       lmax        = -INFINITY;
        // (two passes: the UK dot products first, without control flow -- their dependent dot2 / DPP chains interleave --
        // then the stores of the valid ones)
        float sc[UK];
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const f16x8 kv = __builtin_bit_cast(f16x8, kr(st, u));
            float       a  = 0.f;
            a              = dot2(f16x2{qv[0], qv[1]}, f16x2{kv[0], kv[1]}, a);
            a              = dot2(f16x2{qv[2], qv[3]}, f16x2{kv[2], kv[3]}, a);
            a              = dot2(f16x2{qv[4], qv[5]}, f16x2{kv[4], kv[5]}, a);
            a              = dot2(f16x2{qv[6], qv[7]}, f16x2{kv[6], kv[7]}, a);
            a              = group_sum_dpp<LPK>(a) * inv_sqrt_dh;
            sc[u]          = ((mask_bits >> u) & 1u) != 0u ? -INFINITY : a;
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const int t = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            if (t < t_cached_end && sub == 0) {
                s_p[t - t_beg] = sc[u];
                lmax           = fmaxf(lmax, sc[u]);
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
            a = wave_sum_dpp(a) * inv_sqrt_dh;
            if (lane == 0) {
                s_p[tl - t_beg] = a;
                lmax            = fmaxf(lmax, a);
            }
        }
        lmax = wave_max_dpp(lmax);
        if (lane == 0) {
            s_red[wid] = lmax;
        }
#ifdef PS_ATT_STAMPS
        if (p.ts && lane == 0 && wid >= PS_NC) {
            p.ts[(((size_t)blockIdx.x * p.L + (tag & 255u) - 1u) * PS_NW + wid) * 16 + 13] = wall_clock64();
        }
#endif
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
        lsum = wave_sum_dpp(lsum);
        __syncthreads();
#ifdef PS_ATT_STAMPS
        if (p.ts && lane == 0 && wid >= PS_NC) {
            p.ts[(((size_t)blockIdx.x * p.L + (tag & 255u) - 1u) * PS_NW + wid) * 16 + 14] = wall_clock64();
        }
#endif
        if (lane == 0) {
            s_red[PS_NW + wid] = lsum;
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[j] = 0.f;
        }
        // (rows beyond tlength were fetched speculatively and may hold anything: weight 0 AND value 0, without control flow)
        float pw[UK];
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const int  t     = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            const bool valid = t < t_cached_end;
            const int  idx   = valid ? t - t_beg : 0;
            pw[u]            = valid ? s_p[idx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UK; u++) {
            const int   t     = t_beg + u * PS_NW * KPI + wid * KPI + grp;
            const bool  valid = t < t_cached_end;
            const u32x4 raw   = vr(st, u);
            const u32x4 vz    = {valid ? raw.x : 0u, valid ? raw.y : 0u, valid ? raw.z : 0u, valid ? raw.w : 0u};
            const f16x8 vv    = __builtin_bit_cast(f16x8, vz);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                acc[j] = fmaf(pw[u], (float)vv[j], acc[j]);
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
            acc[j] = across_groups_sum<LPK>(acc[j]);
        }
#ifdef PS_ATT_STAMPS
        if (p.ts && lane == 0 && wid >= PS_NC) {
            p.ts[(((size_t)blockIdx.x * p.L + (tag & 255u) - 1u) * PS_NW + wid) * 16 + 15] = wall_clock64();
        }
#endif
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
#ifndef PS_MERGE_NPER
#define PS_MERGE_NPER 8
#endif
template<int DH, int NPER = PS_MERGE_NPER>
__device__ __forceinline__ void ps_attn_merge(const PersistParams& p, char* smem, u64* gall, const unsigned tag, int h,
                                              int b, const int tx)
{
    const int ne = DH + 2, ns = p.plan.nsplit;
    const int ng = ns * ne;
    float*    sval = reinterpret_cast<float*>(smem);  // [ns][ne] then [ns] weights + denominator
#ifndef PS_MERGE_V2
#define PS_MERGE_V2 1  // (round 4: ctx arrives ~2.5 us earlier, the launch is 1.5 % shorter; 0 = round 3's single-wave merge)
#endif
    if constexpr (PS_MERGE_V2 != 0) {
        // The partials were swept into LDS by BOTH control waves (ps_attn_merge_sweep: 128 lanes x 8 granules = six splits of a
        // 128-wide head in ONE round trip; the first form's single wave needed two) and the merge runs out of registers: every
        // lane reads the splits' maxima and sums from LDS (broadcast reads), derives the weights itself and combines its own
        // columns -- no cross-lane step, no LDS round trip for the weights (the first form: 1.7 us between "partials swept" and
        // "merged", on the path every out-proj piece waits for)
        // (every LDS value this lane needs is requested up front -- loops over `ns` with dependent LDS reads cost a round trip each)
        constexpr int MS = 8;
        if (ns <= MS) {
            // (three batches of LDS reads -- statistics, first column, second column -- so that 16 values are live, not 32: the
            // control role has no registers to spare)
            float wgt[MS], t0[MS];
#pragma unroll
            for (int s2 = 0; s2 < MS; s2++) {
                const int q = s2 < ns ? s2 : 0;
                wgt[s2]     = sval[q * ne + DH];
                t0[s2]      = sval[q * ne + DH + 1];
            }
            float m = -INFINITY;
#pragma unroll
            for (int s2 = 0; s2 < MS; s2++) {
                m = s2 < ns ? fmaxf(m, wgt[s2]) : m;
            }
            float L = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < MS; s2++) {  // split order
                wgt[s2] = (s2 >= ns || wgt[s2] == -INFINITY) ? 0.f : __expf(wgt[s2] - m);
                L += __fmul_rn(wgt[s2], t0[s2]);  // (the product rounded on its own, as the first form's shuffled sum has it)
            }
#pragma unroll
            for (int s2 = 0; s2 < MS; s2++) {
                t0[s2] = sval[(s2 < ns ? s2 : 0) * ne + tx];
            }
#pragma unroll
            for (int s2 = 0; s2 < MS; s2++) {
                o0 += wgt[s2] * t0[s2];
            }
            if constexpr (DH > 64) {
#pragma unroll
                for (int s2 = 0; s2 < MS; s2++) {
                    t0[s2] = sval[(s2 < ns ? s2 : 0) * ne + 64 + (tx < DH - 64 ? tx : 0)];
                }
#pragma unroll
                for (int s2 = 0; s2 < MS; s2++) {
                    o1 += wgt[s2] * t0[s2];
                }
            }
            const float inv = 1.f / (L + 1.e-6f);  // :1632
            static_assert(DH == 64 || DH == 128, "one or two columns per lane");
#pragma unroll
            for (int k = 0; k < DH / 64; k++) {
                const int      d  = tx + k * 64;
                const unsigned b0 = f16_bits((f16)((k == 0 ? o0 : o1) * inv));
                const unsigned b1 = next_lane_u32(b0);
                if ((d & 1) == 0) {
                    st_granule_u32(&p.gc[((size_t)b * p.nh * DH + h * DH + d) >> 1], tag, b0 | (b1 << 16));
                }
            }
            return;
        }
        float m = -INFINITY;
        for (int s2 = 0; s2 < ns; s2++) {
            m = fmaxf(m, sval[s2 * ne + DH]);
        }
        float L = 0.f;
        for (int s2 = 0; s2 < ns; s2++) {  // (split order, like the first form's lane-ordered sum)
            const float ms = sval[s2 * ne + DH];
            L += ((ms == -INFINITY) ? 0.f : __expf(ms - m)) * sval[s2 * ne + DH + 1];
        }
        const float inv = 1.f / (L + 1.e-6f);  // :1632
        for (int d = tx; d < DH; d += 64) {
            float o = 0.f;
            for (int s2 = 0; s2 < ns; s2++) {
                const float ms = sval[s2 * ne + DH];
                o += ((ms == -INFINITY) ? 0.f : __expf(ms - m)) * sval[s2 * ne + d];
            }
            const unsigned b0 = f16_bits((f16)(o * inv));
            const unsigned b1 = next_lane_u32(b0);
            if ((d & 1) == 0) {
                st_granule_u32(&p.gc[((size_t)b * p.nh * DH + h * DH + d) >> 1], tag, b0 | (b1 << 16));
            }
        }
        return;
    }
    ps_sweep<NPER>(gall, ng, tx, 64, tag, p.err, 2, [&](const int i, const unsigned v) { sval[i] = __uint_as_float(v); });
    if (p.ts && tx == 0) {  // (debug stamp 15: partials swept)
        p.ts[(((size_t)blockIdx.x * p.L + (tag & 255u) - 1u) * PS_NW + 0) * 16 + 15] = wall_clock64();
    }
    // weights (same wave: DS operations of one wave execute in order)
    float ms = -INFINITY, ls = 0.f;
    if (tx < ns) {
        ms = sval[tx * ne + DH];
        ls = sval[tx * ne + DH + 1];
    }
    const float m  = wave_max_dpp(ms);
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
        const unsigned b1 = next_lane_u32(b0);
        if ((d & 1) == 0) {
            st_granule_u32(&p.gc[((size_t)b * p.nh * DH + h * DH + d) >> 1], tag, b0 | (b1 << 16));
        }
    }
}

// PS_MERGE_V2: both control waves of a split-0 workgroup sweep the (row, head)'s partials into LDS
__device__ __forceinline__ void ps_attn_merge_sweep(const PersistParams& p, char* smem, const u64* gall, const unsigned tag,
                                                    const int dh, const int tid2)
{
    float* sval = reinterpret_cast<float*>(smem);
    ps_sweep<8>(gall, p.plan.nsplit * (dh + 2), tid2, PS_NC * 64, tag, p.err, 2,
                [&](const int i, const unsigned v) { sval[i] = __uint_as_float(v); });
    if (p.ts && tid2 == 0) {  // (debug stamp 15: partials swept)
        p.ts[(((size_t)blockIdx.x * p.L + (tag & 255u) - 1u) * PS_NW + 0) * 16 + 15] = wall_clock64();
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
// The tensor-parallel all-reduce of x' (GptNeoXDecoder.cc:357-359), inside the launch: this rank's partial x' (x / TP + its
// attention and FFN shares + bias / TP, rounded to half like the tensor the reference hands to NCCL) goes as {tag, pair of
// halves} granules into slot [rank] of EVERY rank's exchange window -- peer memory over xGMI, system-scope stores -- and
// the same lanes wait for the TP partials of their pair in their OWN window, add them in rank order in fp32 (the same
// value on every rank) and publish the result exactly as the single-GPU kernel publishes x'.  No kernel boundary, no RCCL
// call, no host involvement per layer: the hop rides on the merge the kernel does anyway.
__device__ __forceinline__ void ps_tp_exchange(const PersistParams& p, const unsigned tag, const size_t oidx,
                                               const size_t slab, const f16 o, const int c, const bool last)
{
    const unsigned b0 = f16_bits(o);
    const unsigned b1 = next_lane_u32(b0);
    if ((c & 1) == 0) {
        const size_t   gi   = oidx >> 1;
        const unsigned pair = b0 | (b1 << 16);
        // Two planes, by layer parity (the tag's low bit is the layer's, tag = step * 256 + 1 + l): a peer writes plane
        // parity(l) again at layer l + 2 only, and it cannot finish layer l + 1 without THIS rank's layer l + 1 partial,
        // which this rank produces after it has read layer l -- so a slot is never overwritten under a reader that was
        // delayed (with one plane a peer a full layer ahead could, and the reader would spin until it gave up).
        const size_t plane = (size_t)((tag - 1u) & 1u) * (size_t)p.tp * slab;
        for (int r2 = 0; r2 < p.tp; r2++) {
            __hip_atomic_store((gu64*)(p.xw[r2] + plane + (size_t)p.tp_rank * slab + gi), ((u64)tag << 32) | (u64)pair, PS_RLX,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // (all ranks' slots are requested together and re-read until every tag matches: one round trip when the partials are
        // in, not tp dependent ones -- round 3 polled them one after the other: ~4 us of a TP = 8 layer)
        float lo = 0.f, hi = 0.f;
        u64   pv[PERSIST_MAX_TP];
        int   sp2 = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int r2 = 0; r2 < PERSIST_MAX_TP; r2++) {
                if (r2 < p.tp) {
                    pv[r2] = __hip_atomic_load((const gu64*)(p.xw[p.tp_rank] + plane + (size_t)r2 * slab + gi), PS_RLX,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
#pragma unroll
            for (int r2 = 0; r2 < PERSIST_MAX_TP; r2++) {
                if (r2 < p.tp) {
                    ok &= ((unsigned)(pv[r2] >> 32) == tag);
                }
            }
            if (ok || ps_give_up(sp2, p.err, 9)) {
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int r2 = 0; r2 < PERSIST_MAX_TP; r2++) {  // rank order: the same sum on every rank
            if (r2 < p.tp) {
                lo += (float)bits_f16((unsigned)pv[r2]);
                hi += (float)bits_f16((unsigned)pv[r2] >> 16);
            }
        }
        const unsigned fin = (unsigned)f16_bits((f16)lo) | ((unsigned)f16_bits((f16)hi) << 16);
        if (last) {
            *reinterpret_cast<unsigned*>(&p.x_out[oidx]) = fin;
        }
        else {
            st_granule_u32(&p.gx[gi], tag, fin);
        }
    }
}

// TP: tensor parallel -- the per-layer all-reduce (GptNeoXDecoder.cc:357-359) happens INSIDE the launch, in the merge of x'
// (see there); a separate instantiation so that the TP = 1 kernel's code is exactly what it was.
// GROUP (test infrastructure, see engine.hip): all ranks of a LOCAL tensor-parallel group in ONE launch on one device --
// workgroups [r * nb, (r + 1) * nb) are rank r -- so that every rank's workgroups are resident together by construction.
// A3 ("attention apart", round 3; one row, short attention form): the attention leaves the streamer waves.  The two control
// waves request the K rows of the workgroup's KV split by LDS-DMA (64 KiB of LDS, no registers) and its V rows into their own
// register batches (idle between their P1 and P3 shares), sweep q/k/v, run the whole split on 128 threads and publish the
// partial; the streamer waves go from the P1 epilogue straight to the mid sweep and the FFN2 stream: their window between
// the two weight streams is one hop (mid) instead of hop + attention + mid sweep (11.4 -> ~6 us per layer, measured with a
// timing probe before this was built: profiles/r03_notes.md).
// P3L (round 3; one row, short attention form, not A3): the control waves' share of the P3 stream -- the out-proj pieces at the
// end of the workgroup's tile space, which wait for ctx anyway -- is requested into LDS (kbuf, 32 tiles per control wave) by
// LDS-DMA right after q/k/v are staged, eight requests per wave: it lands during the attention, when the K/V rows are in and
// the HBM has nothing else to do, and is consumed from LDS the moment ctx arrives.  64 KiB per CU = 16 MB per layer leave
// the P3 stream, and the control waves no longer finish it last.
template<bool INT8, int M, int DH, int UK, bool TP, bool GROUP = false, bool A3 = false, bool P3L = false>
__global__ __launch_bounds__(PS_NT) void k_decode_persistent(
    const typename std::conditional<GROUP, PersistGroupParams, PersistParams>::type pa)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PersistParams& p = [&]() -> const PersistParams& {
        if constexpr (GROUP) {
            return pa.p[blockIdx.x / pa.nb];
        }
        else {
            return pa;
        }
    }();
    const int bid = [&]() -> int {
        if constexpr (GROUP) {
            return (int)(blockIdx.x % pa.nb);
        }
        else {
            return (int)blockIdx.x;
        }
    }();
    constexpr int TK = TileK<INT8>::value;
    // Values assigned under a condition inside the layer loop (the LayerNorm parameters of the NEXT layer, fetched under
    // `l + 1 < l_end`) are carried around the loop by the compiler: 32 VGPRs live through both weight streams.  The long
    // attention form needs them back, so it makes those assignments unconditional (clamped; after the last layer the
    // loads are harmless re-reads); so do the two-row and tensor-parallel forms, which spilled 35-41 VGPRs without it.
    // The short one-row form -- the headline's -- keeps the code it was tuned with: the same change
    // there measured -0.7 % (profiles/r02_notes.md: hipcc's allocation of this kernel moves +-2 % with anything).
#ifndef PS_NOCARRY_ALL
#define PS_NOCARRY_ALL 0
#endif
#ifndef PS_PART3
#define PS_PART3 0
#endif
    constexpr bool PART3 = PS_PART3 != 0 && M == 1 && !A3;
    constexpr bool NOCARRY = UK > PS_U || M > 1 || TP || A3 || P3L || PS_NOCARRY_ALL;
    const int     H = p.H, Hl = p.Hl, Il = p.Il;
    const int     NB = p.plan.NB;
    const int     wid = threadIdx.x >> 6;
    const int     KT = H / TK, KT_a = Hl / TK, KT_b = Il / TK;
    const int     NT0 = 3 * Hl / 16, NG = H / 16;
    const int     PA = p.plan.PA, PB = p.plan.PB, RLa = p.plan.RLa, RLb = p.plan.RLb;
    const int     E1 = p.plan.e1, E3 = p.plan.e3;

    PsSmem s;
    {
        char* q = smem;
        s.xraw  = reinterpret_cast<f16*>(q);
        q += (size_t)M * H * 2;
        s.xs = reinterpret_cast<f16*>(q);
        q += (size_t)p.plan.xs_halves * 2;
        s.part = reinterpret_cast<float*>(q);
        q += (size_t)PS_RMAX * PS_NW * M * 16 * 4;
        s.part3 = s.part;
        if constexpr (PART3) {
            s.part3 = reinterpret_cast<float*>(q);
            q += (size_t)PS_RMAX * PS_NW * M * 16 * 4;
        }
        s.att = q;
        q += ps_att_bytes(DH, p.s_max, p.plan.nsplit);
        s.rt1 = reinterpret_cast<RunRec*>(q);
        q += sizeof(RunRec) * PS_RMAX;
        s.rt3 = reinterpret_cast<RunRec*>(q);
        q += sizeof(RunRec) * PS_RMAX;
        s.rsc = reinterpret_cast<f16*>(q);
        q += PS_RMAX * 16 * 2;
        s.rsc3 = reinterpret_cast<f16*>(q);
        q += PS_RMAX * 16 * 2;
        s.red = reinterpret_cast<float*>(q);
        q += 64 * 4;
        s.misc = reinterpret_cast<int*>(q);
        q += 64 * 4;
        s.lt1 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * E1 * 4;
        s.lt3 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * E3 * 4;
        s.bt1 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * (E1 / PS_U) * 4;
        s.bt3 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * (E3 / PS_U) * 4;
        s.kbuf = smem + (((size_t)(q - smem) + 1023) & ~(size_t)1023);
    }
    if (p.d_stop && *p.d_stop) {
        return;  // every row has finished (a token of a multi-token graph behind the request's last one): uniform over the grid
    }
    const int      step     = *p.d_step;
    const unsigned tag_base = (unsigned)step * 256u + 1u;
    if (p.ts && (threadIdx.x & 63) == 0) {  // kernel entry (slot 15 of the first layer)
        // (blockIdx.x, not `bid`: the time stamps are a single-rank tool, and this very line written with `bid` costs the
        // kernel 20 spilled VGPRs -- hipcc's allocation of this kernel is that fragile)
        p.ts[(((size_t)blockIdx.x * p.L + p.l_begin) * PS_NW + (threadIdx.x >> 6)) * 16 + 15] = wall_clock64();
    }

    // ---- the workgroup's static share of the streaming stages ----
    // P1: every workgroup owns a range of QKV column groups AND a range of FFN1 column groups (QKV runs first in its run
    // table: they are streamed first, so qkv is complete -- and published -- well before the stage ends)
    const int NF  = Il / 16;
    // (NT0 / NB is not whole at 13B: 3.75 -> three of four workgroups stream 4 QKV groups, one streams 3, i.e. 720 vs 640
    // tiles.  plan.qrot (FTCF_PERSIST_QROT, default 0) rotates WHICH workgroups get the light share: workgroup b runs on XCD
    // b % 8 and the stamps show XCDs 2 / 6 ending P1 1.3 us after the others on equal shares; moving the light share onto
    // them measured +-0.3 %: whoever is last, the next hand-off waits for it)
    const int qb  = (bid + p.plan.qrot) % NB;
    const int q0  = (int)((long)NT0 * qb / NB), q1 = (int)((long)NT0 * (qb + 1) / NB);
    const int f0  = (int)((long)NF * bid / NB), f1 = (int)((long)NF * (bid + 1) / NB);
    const int nq  = q1 - q0;
    const int rB0 = (int)((long)NG * PB * bid / NB), rB1 = (int)((long)NG * PB * (bid + 1) / NB);
    const int rA0 = (int)((long)NG * PA * bid / NB), rA1 = (int)((long)NG * PA * (bid + 1) / NB);
    const int nB = rB1 - rB0, nA = rA1 - rA0;
    const int nruns1 = nq + (f1 - f0), nruns3 = nB + nA;
    const int n_items = p.B * p.nh * p.plan.nsplit;
    // The run tables and the per-wave tile / batch tables depend on the plan and the workgroup only -- not on the layer, the
    // token or the request.  Building them took 17.5 us of every launch (0.7 % of a token): the engine now builds them ONCE
    // per plan with a launch over no layers (tab_mode 1: build, then copy the table region of LDS out) and every real launch
    // copies them in (tab_mode 2: ~13 KB per workgroup from L2).  tab_mode 0 builds them in place as before.
    const size_t tab_bytes = (size_t)(reinterpret_cast<char*>(s.bt3 + (size_t)PS_NW * (E3 / PS_U)) - reinterpret_cast<char*>(s.rt1));
    if (p.tab_mode == 2) {
        const auto* src = PS_G(u32x4, p.tab + (size_t)bid * tab_bytes);
        u32x4*      dst = reinterpret_cast<u32x4*>(s.rt1);
        for (int i = threadIdx.x; i < (int)(tab_bytes / 16); i += PS_NT) {
            dst[i] = src[i];
        }
    }
    else {
        PsStage sg1{}, sg3{};
        int     mid_lo = 0, mid_hi = 0, ctx_lo = 0, ctx_hi = 0;
        if (threadIdx.x == 0) {
            s.misc[0]  = 0;
            s.misc[32] = 0;  // ctx arrival counter (+PS_NC per layer)
            s.misc[33] = 0;  // control-wave pair barrier (+PS_NC per layer)
            s.misc[34] = 0;  // A3: streamer-wave barrier (+(PS_NW - PS_NC) per layer)
            s.misc[35] = 0;  // A3: control-wave barrier inside the attention
            s.misc[39] = 0;  // control pair barrier of the partials' sweep (PS_MERGE_V2)
            s.misc[36] = 0;  // PS_QKV_EARLY: waves that have flushed their QKV slice (+ per layer)
            s.misc[37] = 0;  // PS_QKV_EARLY: waves without a QKV slice
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
            r.pad    = 0;
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
            r.grp    = g;
            r.pad    = 0;
            s.rt3[j] = r;
        }
        __syncthreads();
        {
            int T1 = 0, T3 = 0;
            for (int j = 0; j < nruns1; j++) {
                T1 += s.rt1[j].nt;
            }
            bool fb = true, fa = true;
            for (int j = 0; j < nruns3; j++) {
                const RunRec r = s.rt3[j];
                T3 += r.nt;
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
            T1 = ps_rfl(T1);
            T3 = ps_rfl(T3);
            mid_lo = ps_rfl(mid_lo);
            mid_hi = ps_rfl(mid_hi);
            ctx_lo = ps_rfl(ctx_lo);
            ctx_hi = ps_rfl(ctx_hi);
            const int w = ps_rfl(wid);
            int       tb, te;
            int ent;
            sg1.lt = s.lt1 + (size_t)w * E1;
            sg1.bt = s.bt1 + (size_t)w * (E1 / PS_U);
            if constexpr (PS_QKV_EARLY != 0) {
                // this wave's slice of the QKV runs first, then its slice of the FFN1 runs; the batch that flushes its last QKV
                // partial sum carries PS_BT_SIGNAL (a wave without QKV tiles is counted in misc[37] instead)
                int qa, qz, fa2, fz;
                ps_wave_range2_w(nq * KT, (f1 - f0) * KT, w, p.plan.wt1, qa, qz, fa2, fz);
                auto      ntk = [&](int) { return KT; };
                const int eq  = ps_wave_entries(nq, ntk, qa, qz);
                ent           = eq + ps_wave_entries(f1 - f0, ntk, fa2, fz);
                sg1.nrot      = (ent + PS_U * PS_NBUF - 1) / (PS_U * PS_NBUF);
                sg1.nrot      = sg1.nrot < 1 ? 1 : sg1.nrot;
                unsigned* lt  = s.lt1 + (size_t)w * E1;
                unsigned* bt  = s.bt1 + (size_t)w * (E1 / PS_U);
                ps_build_tables<TK>(s.rt1, nq, qa, qz, lt, bt, eq, 0);
                ps_build_tables<TK>(s.rt1 + nq, f1 - f0, fa2, fz, lt + eq, bt + eq / PS_U, sg1.nrot * PS_U * PS_NBUF - eq, nq);
                if ((threadIdx.x & 63) == 0) {
                    if (eq > 0) {
                        bt[eq / PS_U - 1] |= PS_BT_SIGNAL;  // (the builder's lanes wrote it: same wave, DS order)
                    }
                    else {
                        atomicAdd(&s.misc[37], 1);
                    }
                }
            }
            else {
                ps_wave_range_w(T1, w, p.plan.wt1, tb, te);
                ent      = ps_wave_entries(nruns1, [&](int j) { return s.rt1[j].nt; }, tb, te);
                sg1.nrot = (ent + PS_U * PS_NBUF - 1) / (PS_U * PS_NBUF);
                sg1.nrot = sg1.nrot < 1 ? 1 : sg1.nrot;
                ps_build_tables<TK>(s.rt1, nruns1, tb, te, s.lt1 + (size_t)w * E1, s.bt1 + (size_t)w * (E1 / PS_U),
                                    sg1.nrot * PS_U * PS_NBUF);
            }
            ps_wave_range_w(T3, w, p.plan.wt3, tb, te);
            ent      = ps_wave_entries(nruns3, [&](int j) { return s.rt3[j].nt; }, tb, te);
            sg3.lt   = s.lt3 + (size_t)w * E3;
            sg3.bt   = s.bt3 + (size_t)w * (E3 / PS_U);
            sg3.nrot = (ent + PS_U * PS_NBUF - 1) / (PS_U * PS_NBUF);
            sg3.nrot = sg3.nrot < 1 ? 1 : sg3.nrot;
            ps_build_tables<TK>(s.rt3, nruns3, tb, te, s.lt3 + (size_t)w * E3, s.bt3 + (size_t)w * (E3 / PS_U),
                                sg3.nrot * PS_U * PS_NBUF);
            sg1.xs0 = sg1.xs1 = H + XPAD;  // LDS rows of x are padded: see XPAD
            sg3.xs0 = Il + XPAD;
            sg3.xs1 = Hl + XPAD;
        }
        if ((threadIdx.x & 63) == 0) {  // (every later use reads these back from LDS: one source for both modes)
            s.misc[40 + wid] = sg1.nrot;
            s.misc[48 + wid] = sg3.nrot;
        }
        if (threadIdx.x == 0) {
            s.misc[56] = mid_lo;
            s.misc[57] = mid_hi;
            s.misc[58] = ctx_lo;
            s.misc[59] = ctx_hi;
        }
        if (p.tab_mode == 1) {
            __syncthreads();
            auto*        dst = (__attribute__((address_space(1))) u32x4*)(p.tab + (size_t)bid * tab_bytes);
            const u32x4* src = reinterpret_cast<const u32x4*>(s.rt1);
            for (int i = threadIdx.x; i < (int)(tab_bytes / 16); i += PS_NT) {
                dst[i] = src[i];
            }
        }
    }
    __syncthreads();
    PsStage sg1{}, sg3{};
    {
        const int w = ps_rfl(wid);
        sg1.lt      = s.lt1 + (size_t)w * E1;
        sg1.bt      = s.bt1 + (size_t)w * (E1 / PS_U);
        sg3.lt      = s.lt3 + (size_t)w * E3;
        sg3.bt      = s.bt3 + (size_t)w * (E3 / PS_U);
        sg1.nrot    = ps_rfl(s.misc[40 + w]);
        sg3.nrot    = ps_rfl(s.misc[48 + w]);
        sg1.xs0 = sg1.xs1 = H + XPAD;  // LDS rows of x are padded: see XPAD
        sg3.xs0 = Il + XPAD;
        sg3.xs1 = Hl + XPAD;
    }
    const int mid_lo = ps_rfl(s.misc[56]), mid_hi = ps_rfl(s.misc[57]);  // K ranges (halves) of mid / ctx this workgroup consumes
    const int ctx_lo = ps_rfl(s.misc[58]), ctx_hi = ps_rfl(s.misc[59]);
    if (p.tab_mode == 1) {
        return;  // the launch that only builds the tables
    }

    // Control waves and streamer waves run SEPARATE instantiations of the layer loop (same barriers, in the same order):
    // with a shared body the register batches of the role that primes early stay live, as far as the compiler can tell,
    // through every section of the other role and spill.  Whole waves take one side, s_barrier only counts arrivals.
    auto body = [&](auto role) {
        constexpr bool    CTRL = decltype(role)::value;
        int               tid  = threadIdx.x;
        PsStream<INT8, M, PS_QKV_EARLY != 0> st;
        const int n_sig = PS_NW - ps_rfl(s.misc[37]);  // (PS_QKV_EARLY) waves that flush QKV partial sums
        auto stamp = [&](const int l, const int k) {
            const int lane = tid & 63, wid = tid >> 6;
            if (p.ts && lane == 0) {
                p.ts[(((size_t)bid * p.L + l) * PS_NW + wid) * 16 + k] = wall_clock64();
            }
        };
        // A3: barrier of the control waves among themselves (LDS counter; DS operations of a wave execute in order)
        int  cb_want      = 0;
        auto ctrl_barrier = [&]() {
            cb_want += PS_NC;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((tid & 63) == 0) {
                atomicAdd(&s.misc[35], 1);
            }
            for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[35]) < cb_want;) {
                if (++spins > (PS_SPIN << 6)) {  // (bounded like every other wait of this kernel)
                    __hip_atomic_store(p.err, 11, PS_RLX, PS_AGT);
                    break;
                }
                __builtin_amdgcn_s_sleep(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        // ---- per-layer constants, fetched one stage ahead into registers (before that stage's prefetch) ----
        f16   r_sc1 = (f16)1.f, r_sc3 = (f16)1.f;  // scale of (run tid/16, column tid%16) of P1 / P3
        f16   r_b1[2], r_bres[2];                  // ffn1 bias of the P1 epilogue items / residual bias of the merge items
        f16x8 r_ln[4][PS_NLN];                     // ln1_g, ln1_b, ln2_g, ln2_b vectors tid, tid + 512
        auto  load_sc1 = [&](const int l) {
            if constexpr (INT8) {
                if (tid < nruns1 * 16) {
                    PsLayerC& lw = PS_LAYER(p, l);
                    const RunRec&       r  = s.rt1[tid >> 4];
                    r_sc1 = PS_G(f16, r.sel ? lw.s_ffn1 : lw.s_qkv)[r.grp * 16 + (tid & 15)];
                }
            }
        };
        auto load_p1_consts = [&](const int l) {  // LN parameters, ffn1 bias, P3 scales of layer l
            PsLayerC& lw = PS_LAYER(p, l);
#pragma unroll
            for (int k = 0; k < PS_NLN; k++) {
                const int v = tid + k * PS_NT;
                if constexpr (NOCARRY) {  // unconditional, clamped (see NOCARRY)
                    const int o = (v * 8 < H) ? v * 8 : 0;
                    r_ln[0][k]  = *PS_G(f16x8, lw.ln1_g + o);
                    r_ln[1][k]  = *PS_G(f16x8, lw.ln1_b + o);
                    r_ln[2][k]  = *PS_G(f16x8, lw.ln2_g + o);
                    r_ln[3][k]  = *PS_G(f16x8, lw.ln2_b + o);
                }
                else if (v * 8 < H) {
                    r_ln[0][k] = *PS_G(f16x8, lw.ln1_g + v * 8);
                    r_ln[1][k] = *PS_G(f16x8, lw.ln1_b + v * 8);
                    r_ln[2][k] = *PS_G(f16x8, lw.ln2_g + v * 8);
                    r_ln[3][k] = *PS_G(f16x8, lw.ln2_b + v * 8);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int idx = tid + k * PS_NT;
                r_b1[k]       = (f16)0.f;
                if (idx < nruns1 * M * 16) {
                    const int cg = s.rt1[idx / (M * 16)].rid;
                    if (cg >= NT0) {
                        r_b1[k] = PS_G(f16, lw.b_ffn1)[(cg - NT0) * 16 + (idx & 15)];
                    }
                }
            }
            if constexpr (INT8) {
                if (tid < nruns3 * 16) {
                    const RunRec& r = s.rt3[tid >> 4];
                    r_sc3 = PS_G(f16, r.sel ? lw.s_out : lw.s_ffn2)[r.grp * 16 + (tid & 15)];
                }
            }
        };
        auto load_p3_consts = [&](const int l) {  // residual bias of layer l, P1 scales of layer l + 1
            if constexpr (CTRL) {
                PsLayerC& lw = PS_LAYER(p, l);
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
            if constexpr (NOCARRY) {
                load_sc1(l + 1 < p.l_end ? l + 1 : l);
            }
            else if (l + 1 < p.l_end) {
                load_sc1(l + 1);
            }
        };
        // per layer: scales of the stage's runs -> LDS, zero the partial buffer, bind the stream
        auto setup_p1 = [&](const int l) {
            PsLayerC& lw = PS_LAYER(p, l);
            if constexpr (INT8) {
                if (tid < nruns1 * 16) {
                    s.rsc[tid] = r_sc1;
                }
            }
            for (int i = tid; i < nruns1 * PS_NW * M * 16; i += PS_NT) {
                s.part[i] = 0.f;
            }
            load_p1_consts(l);
            sg1.w0 = reinterpret_cast<const char*>(lw.w_qkv);
            sg1.w1 = reinterpret_cast<const char*>(lw.w_ffn1);
            st.bind(sg1, s.rsc, s.xs, s.part, tid);
            if constexpr (PS_QKV_EARLY != 0) {
                st.sig      = &s.misc[36];
                st.sig_full = (l - p.l_begin + 1) * n_sig;
                st.eq_n     = nq * M * 16;
                st.eq_rt    = s.rt1;
                st.eq_g     = p.gq;
                st.eq_tag   = tag_base + (unsigned)l;
                st.eq_hl3   = 3 * Hl;
            }
            if constexpr (!CTRL) {
                st.prime_lo();
                if constexpr (PS_FULL_P1) {
                    st.prime_hi();
                }
            }
        };
        auto setup_p3 = [&](const int l) {
            PsLayerC& lw = PS_LAYER(p, l);
            if constexpr (A3 || PART3) {
                // no workgroup barrier between here and the P3 stream: a wave zeroes the slots it flushes itself (the scales
                // were stored before the barrier that ended P1)
                for (int i = tid & 63; i < nruns3 * M * 16; i += 64) {
                    s.part3[((size_t)(i / (M * 16)) * PS_NW + (tid >> 6)) * (M * 16) + i % (M * 16)] = 0.f;
                }
            }
            else {
                if constexpr (INT8) {
                    if (tid < nruns3 * 16) {
                        s.rsc[tid] = r_sc3;
                    }
                }
                for (int i = tid; i < nruns3 * PS_NW * M * 16; i += PS_NT) {
                    s.part[i] = 0.f;
                }
            }
            sg3.w0 = reinterpret_cast<const char*>(lw.w_ffn2);
            sg3.w1 = reinterpret_cast<const char*>(lw.w_out);
            st.bind(sg3, (A3 || PART3) ? s.rsc3 : s.rsc, s.xs, s.part3, tid, &s.misc[32], (l - p.l_begin + 1) * PS_NC);
            if constexpr (PS_QKV_EARLY != 0) {
                st.eq_g = nullptr;  // (P3's tables carry no PS_BT_SIGNAL)
            }
        };

        load_sc1(p.l_begin);
        setup_p1(p.l_begin);
        for (int l = p.l_begin; l < p.l_end; l++) {
            // opaque copies: keeps per-thread address arithmetic from being hoisted out of the layer loop, where it
            // becomes dozens of long-lived VGPRs that spill around the register batches
            asm volatile("" : "+v"(tid));
            const int           lane = tid & 63, wid = tid >> 6;
            PsLayerC& lw  = PS_LAYER(p, l);
            const unsigned      tag = tag_base + (unsigned)l;
            static_assert(!A3 || (M == 1 && UK == PS_U), "A3: one row, short attention form");
            static_assert(!P3L || (M == 1 && UK == PS_U && !A3 && !TP), "P3L: one row, short attention form, one GPU");
            constexpr bool      A3F = A3;
#ifndef PS_A3_KV_LATE
#define PS_A3_KV_LATE 0
#endif
            // A3: the K/V rows are requested when the control waves have finished their P1 share (their register batches are
            // free from there to their P3 share, and the rows land under the streamer waves' P1 tail), not next to the
            // q/k/v polls, which would return behind them
            using AttnA = PsAttn<DH, UK, PS_NC * 64>;
            AttnA      at3;
            const bool i_has = bid < n_items;
            const int  i_sp = i_has ? bid % p.plan.nsplit : 0;
            const int  i_h = i_has ? (bid / p.plan.nsplit) % p.nh : 0, i_b = i_has ? (bid / p.plan.nsplit) / p.nh : 0;
            // PS_KV_EARLY: the attention's K/V rows are requested from inside the first stream's tail into R0 / R1
            constexpr bool KVE = PS_KV_EARLY != 0 && UK == PS_U && !A3 && !P3L && PS_EARLY_P3 == 0;
            using AttnE = PsAttn<DH, UK, 3 * DH / 2, KVE>;
            AttnE      ate;
            const bool e_has = bid < n_items;
            const int  e_sp = e_has ? bid % p.plan.nsplit : 0;
            const int  e_h = e_has ? (bid / p.plan.nsplit) % p.nh : 0, e_b = e_has ? (bid / p.plan.nsplit) / p.nh : 0;
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
                    s0[m] = wave_sum_dpp(s0[m]);
                    s1[m] = wave_sum_dpp(s1[m]);
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
                if constexpr (CTRL) {
                    st.prime_lo();
                }
                stamp(l, 2);
                __syncthreads();
                if constexpr (KVE) {
                    st.template run_hooked<CTRL || !PS_FULL_P1>(
                        [&]() { ate.template issue_rows<false>(p, lw, e_h, e_b, e_sp, tid, st, e_has); },
                        [&]() { ate.template issue_rows<true>(p, lw, e_h, e_b, e_sp, tid, st, e_has); });
                }
                else {
                    st.template run<CTRL || !PS_FULL_P1>();
                }
                if constexpr (A3F && CTRL && !PS_A3_KV_LATE) {
                    at3.issue_ctrl(p, lw, i_h, i_b, i_sp, tid, st,
                                   (unsigned)(size_t)(__attribute__((address_space(3))) char*)s.kbuf, i_has);
                }
                stamp(l, 3);
                __syncthreads();
                // epilogue: qkv = y (bias is added by the attention), mid = gelu(y + b) ; pairs of halves -> granules
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int idx = tid + k * PS_NT;
                    if (idx < nruns1 * M * 16) {
                        const int j = idx / (M * 16), r = idx % (M * 16), m = r >> 4, c = r & 15;
                        float     v = 0.f;
#pragma unroll
                        for (int w = 0; w < PS_NW; w++) {
                            v += s.part[((size_t)j * PS_NW + w) * (M * 16) + r];
                        }
                        const int cg = s.rt1[j].rid;
                        if (PS_QKV_EARLY != 0 && cg < NT0) {
                            continue;  // (published from inside the stream by the wave that flushed the last QKV partial sum)
                        }
                        f16       o;
                        if (cg < NT0) {
                            o = (f16)v;
                        }
                        else {
                            if constexpr (INT8) {
                                o = (f16)gelu_f32(v + (float)r_b1[k]);  // epilogue_helpers.h:52-62
                            }
                            else {
                                o = gelu_f16((f16)v + r_b1[k]);  // activation_kernels.cu:401-426
                            }
                        }
                        const unsigned b0 = f16_bits(o);
                        const unsigned b1 = next_lane_u32(b0);
                        if ((c & 1) == 0) {
                            // (two stores, not one through a selected pointer: the compiler turns that select into a
                            // table in scratch memory, and a kernel that uses scratch pays for it at every dispatch)
                            if (cg < NT0) {
                                st_granule_u32(p.gq + (((size_t)m * 3 * Hl + cg * 16 + c) >> 1), tag, b0 | (b1 << 16));
                            }
                            else {
                                st_granule_u32(p.gm + (((size_t)m * Il + (cg - NT0) * 16 + c) >> 1), tag, b0 | (b1 << 16));
                            }
                        }
                    }
                }
                if constexpr ((A3 || PART3) && INT8) {
                    if (tid < nruns3 * 16) {
                        s.rsc3[tid] = r_sc3;
                    }
                }
                stamp(l, 4);
            }

            // =========================== attention ===============================================================
            asm volatile("" : "+v"(tid));
            // (EARLY: see PS_EARLY_P3; the long form's K/V rows occupy the register batches)
            constexpr int EARLY = (UK > PS_U) ? 0 : PS_EARLY_P3;
            using Attn          = typename std::conditional<KVE, AttnE, PsAttn<DH, UK, (EARLY || A3F) ? PS_NC * 64 : 3 * DH / 2>>::type;
            Attn       at_own;
            Attn&      at = [&]() -> Attn& {
                if constexpr (KVE) {
                    return ate;
                }
                else {
                    return at_own;
                }
            }();
            const bool has_item = bid < n_items;
            int        a_sp = 0, a_h = 0, a_b = 0;
            if (has_item) {
                a_sp         = bid % p.plan.nsplit;
                const int hb = bid / p.plan.nsplit;
                a_h          = hb % p.nh;
                a_b          = hb / p.nh;
            }
            if constexpr (!PART3) {
                __syncthreads();  // part / scales reuse
            }
            setup_p3(l);
            stamp(l, 5);
            bool live = false;
            u64* gall = p.ga + ((size_t)a_b * p.nh + a_h) * p.plan.nsplit * (DH + 2);
            if constexpr (KVE) {
                // (the rows were requested in the first stream's tail; masks, lengths, rotary coefficients, bias follow here)
                at.template issue_impl<false>(p, lw, a_h, a_b, a_sp, tid, st, has_item);
            }
            else if constexpr (Attn::ALIAS) {
                // (rows in the stream's register batches: requested UNCONDITIONALLY -- a workgroup without an item reads
                // item 0's rows for nothing -- because registers assigned under a condition carry their previous contents,
                // here all four weight batches, around the layer loop: 30 spilled VGPRs)
                at.issue(p, lw, a_h, a_b, a_sp, tid, st);
            }
            if constexpr (A3F) {
                if constexpr (CTRL) {
                    if constexpr (PS_A3_KV_LATE) {
                        at3.issue_ctrl(p, lw, a_h, a_b, a_sp, tid, st,
                                       (unsigned)(size_t)(__attribute__((address_space(3))) char*)s.kbuf, has_item);
                    }
                    if (has_item) {
                        __builtin_amdgcn_s_setprio(2);  // (the streamer wave on this SIMD is in its weight stream)
                        at3.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                        stamp(l, 6);
                        ps_wait_vm<0>();  // this wave's K blocks have landed in LDS (the compiler does not know about them)
                        live = at3.compute_ctrl(p, lw, s.att, s.kbuf, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st,
                                                ctrl_barrier);
                        __builtin_amdgcn_s_setprio(0);
                    }
                    stamp(l, 7);
                }
                else {
                    // the streamer waves stage the K range of mid themselves and synchronise among themselves
                    const int stid = tid - PS_NC * 64;
                    ps_sweep<4>(p.gm + ((size_t)mid_lo >> 1), (mid_hi - mid_lo) >> 1, stid, (PS_NW - PS_NC) * 64, tag, p.err, 6,
                                [&](const int i, const unsigned v) { reinterpret_cast<unsigned*>(s.xs + mid_lo)[i] = v; });
                    stamp(l, 7);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) {
                        atomicAdd(&s.misc[34], 1);
                    }
                    const int want = (l - p.l_begin + 1) * (PS_NW - PS_NC);
                    for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[34]) < want;) {
                        if (++spins > (PS_SPIN << 6)) {
                            __hip_atomic_store(p.err, 12, PS_RLX, PS_AGT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    stamp(l, 8);
                    load_p3_consts(l);
                    st.prime_lo();
                    if constexpr (PS_FULL_P3) {
                        st.prime_hi();
                    }
                }
            }
            else if constexpr (EARLY != 0) {
                if (has_item) {
                    at.issue(p, lw, a_h, a_b, a_sp, tid, st);
                }
                if constexpr (!CTRL) {
                    // behind the K/V rows in this wave's return order, ahead of everything else: in flight through the
                    // whole attention window (the q/k/v hop alone is 4-5 us for the median workgroup)
                    st.prime_lo();
                    if constexpr (EARLY == 2) {
                        st.prime_hi();
                    }
                }
                if (has_item) {
                    at.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                    stamp(l, 6);
                    live = at.compute(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st);
                }
            }
            else if constexpr (P3L) {
                at.issue(p, lw, a_h, a_b, a_sp, tid, st, has_item);
                if (has_item) {
                    at.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                    stamp(l, 6);
                }
                // q/k/v are in (nothing of this compute unit polls any more) and the K/V rows were requested long ago: the
                // control waves' P3 tiles -> LDS now, PS_NC * 32 / PS_NW wave-wide requests per wave
                __syncthreads();
                // (hipcc does not know about the LDS-DMA requests below, and vmcnt counts them: its wait for the K/V rows --
                // vmcnt(0), the rows being the youngest loads it knows -- would wait for all of them too.  Naming the rows as
                // inputs here makes it place that wait BEFORE the requests, where the rows have long landed)
                asm volatile("" ::"v"(at.kreg[0]), "v"(at.kreg[1]), "v"(at.kreg[2]), "v"(at.kreg[3]), "v"(at.kreg[4]), "v"(at.kreg[5]),
                             "v"(at.kreg[6]), "v"(at.kreg[7]), "v"(at.vreg[0]), "v"(at.vreg[1]), "v"(at.vreg[2]), "v"(at.vreg[3]),
                             "v"(at.vreg[4]), "v"(at.vreg[5]), "v"(at.vreg[6]), "v"(at.vreg[7]), "v"(at.mask_bits), "v"(at.tl),
                             "v"(at.rot_cs), "v"(at.rot_sn));
                {
                    const unsigned kb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s.kbuf;
                    const int      w8 = ps_rfl(tid >> 6);
#ifndef PS_P3L_DMA
#define PS_P3L_DMA 1
#endif
#pragma unroll
                    for (int k = 0; k < (PS_P3L_DMA ? PS_NC * PS_U * PS_NBUF / PS_NW : 0); k++) {
                        const int      e   = w8 + k * PS_NW;
                        const unsigned ent = (unsigned)ps_rfl((int)s.lt3[(size_t)(e / (PS_U * PS_NBUF)) * E3 + e % (PS_U * PS_NBUF)]);
                        const char*    wb  = reinterpret_cast<const char*>((ent >> 31) ? lw.w_out : lw.w_ffn2);
                        ps_lds_dma16(wb + ((size_t)(ent & 0x7fffffffu) * 64 + (tid & 63)) * 16, (unsigned)ps_rfl((int)(kb + (unsigned)e * 1024u)));
                    }
                }
                if (has_item) {
                    live = at.template compute<false>(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st);
                    at.append_current(p, lw, s.att, a_h, a_b, tid);
                }
            }
            else if (has_item) {
                // (issued here and not before the barrier above: K/V rows held across setup_p3 spill, measured)
                if constexpr (!Attn::ALIAS) {
                    at.issue(p, lw, a_h, a_b, a_sp, tid, st);
                }
                at.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                stamp(l, 6);
                if constexpr (PS_ATT_WP != 0) {
                    live = at.compute_wp(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st);
                }
                else {
                    live = at.compute(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st);
                }
            }
            if constexpr (!A3F) {
#ifndef PS_MID_ALL
#define PS_MID_ALL 1
#endif
                // the K range of mid this workgroup's FFN2 pieces read -> LDS (published at the end of P1: long there), after
                // the attention (ahead of it, it made the attention wait for the slowest FFN1).
                // PS_MID_ALL 0: by the control waves, before the barrier, i.e. before the streamer waves' prefetch burst;
                // >= 1: by all eight waves (a quarter of the passes per thread); 2 / 3: the streamer waves request half /
                // all of their first rotation BEFORE their part of the sweep (their polls return behind it, when it has landed)
                if constexpr (PS_MID_ALL >= 2 && !CTRL && EARLY == 0) {
                    st.prime_lo();
                    if constexpr (PS_MID_ALL == 3) {
                        st.prime_hi();
                    }
                }
                if constexpr (PS_MID_ALL >= 1) {
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        ps_sweep<3>(p.gm + (((size_t)m * Il + mid_lo) >> 1), (mid_hi - mid_lo) >> 1, tid, PS_NT, tag, p.err, 6,
                                    [&](const int i, const unsigned v) {
                                        reinterpret_cast<unsigned*>(s.xs + (size_t)m * (Il + XPAD) + mid_lo)[i] = v;
                                    });
                    }
                }
                else if constexpr (CTRL) {
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        ps_sweep<10>(p.gm + (((size_t)m * Il + mid_lo) >> 1), (mid_hi - mid_lo) >> 1, tid, PS_NC * 64, tag,
                                     p.err, 6, [&](const int i, const unsigned v) {
                                         reinterpret_cast<unsigned*>(s.xs + (size_t)m * (Il + XPAD) + mid_lo)[i] = v;
                                     });
                    }
                }
                stamp(l, 7);
                if constexpr (P3L) {
                    ps_wait_vm<0>();  // this wave's LDS-DMA requests have landed (the compiler does not know about them)
                }
                __syncthreads();  // mid staged, attention scratch free (P3L: the control waves' P3 tiles are in LDS)
                stamp(l, 8);
                if constexpr (!CTRL) {
                    // (the constants of the layer's end -- residual bias, next layer's scales -- are fetched HERE and not in
                    // the P3 set-up: their table reads are synchronous round trips, and the set-up sits on the attention's
                    // critical path, ahead of the K/V request)
                    load_p3_consts(l);
                    // AFTER the barrier: issuing 32 KiB per wave takes ~5 us (the CU's memory pipeline throttles the issue)
                    // and the control waves, which carry the attention's critical path, must not wait for it
                    if constexpr (EARLY == 0 && PS_MID_ALL < 2) {
                        st.prime_lo();  // the streamer waves issue no other load until the end of the P3 stream
                        if constexpr (PS_FULL_P3) {
                            st.prime_hi();
                        }
                    }
                }
            }
            // =========================== P3: [FFN2 u out-proj] -> residual ========================================
            // The streamer waves start on the FFN2 pieces at once; the control waves finish the attention (merge of the
            // split partials by wave 0 of the split-0 workgroups), stage the K range of ctx the out-proj pieces read and
            // announce it through an LDS counter that gates every batch touching ctx; their own share is the END of the
            // workgroup's tile space, i.e. the out-proj pieces.
            if constexpr (CTRL) {
                if constexpr (PS_CTRL_EARLY == 1 || PS_CTRL_EARLY == 2) {
                    // the weights of the control waves' share (the out-proj pieces) need nothing: requested before the
                    // wait for ctx (polls of this wave return behind them -- ctx is 6-10 us away anyway)
                    st.prime_lo();
                    if constexpr (PS_CTRL_EARLY == 2) {
                        st.prime_hi();
                    }
                }
                if constexpr (PS_MERGE_V2 != 0) {
                    if (has_item && a_sp == 0) {  // (both control waves; `live` is uniform over the workgroup)
                        if (live) {
                            ps_attn_merge_sweep(p, s.att, gall, tag, DH, tid);
                            // pair barrier (an LDS counter of its own: + PS_NC per layer)
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) {
                                atomicAdd(&s.misc[39], 1);
                            }
                            const int want = (l - p.l_begin + 1) * PS_NC;
                            for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[39]) < want;) {
                                if (++spins > (PS_SPIN << 6)) {
                                    __hip_atomic_store(p.err, 15, PS_RLX, PS_AGT);
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(0);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            if (wid == 0) {
                                ps_attn_merge<DH>(p, s.att, gall, tag, a_h, a_b, tid);
                            }
                        }
                        else {
                            if (lane == 0) {
                                atomicAdd(&s.misc[39], 1);  // (keeps the counter in step with the layers)
                            }
                            if (wid == 0) {
                                ps_attn_publish_zero<DH>(p, tag, a_h, a_b, tid);
                            }
                        }
                        stamp(l, 13);
                    }
                }
                else if (has_item && a_sp == 0 && wid == 0) {
                    if (live) {
                        ps_attn_merge<DH>(p, s.att, gall, tag, a_h, a_b, tid);
                    }
                    else {
                        ps_attn_publish_zero<DH>(p, tag, a_h, a_b, tid);
                    }
                    stamp(l, 13);
                }
                if constexpr (PS_CTRL_EARLY == 3 || PS_CTRL_EARLY == 4) {
                    // ... requested AFTER the merge of the splits (round 3's PS_CTRL_EARLY 1 / 2 put the merging wave's polls of
                    // the partials behind its own prefetch: ctx arrived 3 us later for everybody) and before the sweep of ctx,
                    // which is several microseconds away: the share's first rotation lands under that wait, in the window where
                    // the HBM has little else to do, instead of being requested when ctx has arrived (the control waves
                    // finished P3 2-7 us after the streamer waves)
                    st.prime_lo();
                    if constexpr (PS_CTRL_EARLY == 3) {
                        st.prime_hi();
                    }
                }
#pragma unroll
                for (int m = 0; m < M; m++) {
                    ps_sweep<5>(p.gc + (((size_t)m * Hl + ctx_lo) >> 1), (ctx_hi - ctx_lo) >> 1, tid, PS_NC * 64, tag,
                                p.err, 7, [&](const int i, const unsigned v) {
                                    reinterpret_cast<unsigned*>(s.xs + (size_t)M * (Il + XPAD) + (size_t)m * (Hl + XPAD) + ctx_lo)[i] = v;
                                });
                }
                stamp(l, 14);
                if (lane == 0) {
                    atomicAdd(&s.misc[32], 1);  // DS operations of a wave execute in order: the writes above are visible
                }
                load_p3_consts(l);
                if constexpr (P3L) {
                    // the whole share out of LDS: a batch of tiles into a register batch, then the same consume as the stream's
                    const int c0 = ps_rfl(tid >> 6) * (PS_U * PS_NBUF);
                    for (int b2 = 0; b2 < sg3.nrot * PS_NBUF; b2++) {
#pragma unroll
                        for (int u = 0; u < PS_U; u++) {
                            st.R0[u] = *reinterpret_cast<const u32x4*>(s.kbuf + ((size_t)(c0 + b2 * PS_U + u) * 64 + (tid & 63)) * 16);
                        }
                        st.consume(st.R0, b2);
                    }
                }
                else if constexpr (PS_CTRL_EARLY == 0) {
                    st.prime_lo();
                }
            }
            stamp(l, 9);
            if constexpr (!(CTRL && P3L)) {
                st.template run<CTRL ? (PS_CTRL_EARLY != 2 && PS_CTRL_EARLY != 3) : ((!A3F && PS_MID_ALL >= 2) ? PS_MID_ALL != 3 : EARLY == 0 ? !PS_FULL_P3 : EARLY != 2)>();
            }
            stamp(l, 10);
            __syncthreads();
            asm volatile("" : "+v"(tid));
            // K pieces -> granules
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int idx = tid + k * PS_NT;
                if (idx < nruns3 * M * 16) {
                    const int j = idx / (M * 16), r = idx % (M * 16);
                    float     v = 0.f;
#pragma unroll
                    for (int w = 0; w < PS_NW; w++) {
                        v += s.part3[((size_t)j * PS_NW + w) * (M * 16) + r];
                    }
                    st_granule(&p.gp[(size_t)s.rt3[j].rid * (M * 16) + r], tag, v);
                }
            }
            const int  nmerge = s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE;
            const bool last   = (l == p.l_end - 1);
            const bool lm_tail = !TP && p.lm_w != nullptr && p.l_end == p.L;  // the LM head follows in this launch
            // layer_input/output alias for 0 < l < L-1 in the reference (GptNeoXDecoder.cc:249-250) -> residual form
            const int inplace = (l > 0 && l < p.L - 1) ? 1 : 0;
            __syncthreads();  // part / scales are free: the streamer waves start the next layer's weight stream now
            // (the control waves carry the layer boundary's critical path -- pieces -> merge -> x' -> gather: they merge
            // FIRST and fetch the next layer's constants afterwards, under the gather's wait)
            if constexpr (!CTRL) {
                if constexpr (NOCARRY) {
                    setup_p1(l + 1 < p.l_end ? l + 1 : l);
                }
                else if (l + 1 < p.l_end) {
                    setup_p1(l + 1);
                }
            }
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
                        if constexpr (TP) {
                            ps_tp_exchange(p, tag, oidx, (size_t)M * H / 2, o, c, last);
                        }
                        else if (last && !lm_tail) {
                            p.x_out[oidx] = o;
                        }
                        else {
                            const unsigned b0 = f16_bits(o);
                            const unsigned b1 = next_lane_u32(b0);
                            if ((c & 1) == 0) {
                                st_granule_u32(&p.gx[oidx >> 1], tag, b0 | (b1 << 16));
                            }
                        }
                    }
                }
            }
            if constexpr (CTRL) {
                if constexpr (NOCARRY) {
                    setup_p1(l + 1 < p.l_end ? l + 1 : l);
                }
                else if (l + 1 < p.l_end) {
                    setup_p1(l + 1);
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
        // =========================== LM head (the launch that ran the last layer, one GPU) =============================
        // final LayerNorm + logits = h . W^T over the [V][H] fp16 tensor, read in place (GptNeoX.cc:853-925; the same
        // arithmetic as k_lm_head: lanes along k, 8 halves per lane and 512-half step, dot2 chains in k order, the same
        // wave sum).  A workgroup takes V / NB consecutive rows, its waves consecutive row ranges: a wave's share is ONE
        // contiguous byte range, streamed through the same four register batches as the weights; the streamer waves request
        // their first rotation right here, while the control waves still merge and gather the last layer's x'.
// (compiled only with -DPS_EXPERIMENTS: the mere presence of this tail costs the layer loop 2 % -- hipcc's register
// allocation of this kernel -- and the LM head streams at the same 6.6 TB/s inside or outside the launch: profiles/r03_notes.md)
#ifdef PS_EXPERIMENTS
#define PS_LM_CODE 1
#else
#define PS_LM_CODE 0
#endif
        if constexpr (!TP && PS_LM_CODE) {
            if (p.lm_w != nullptr && p.l_end == p.L) {
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63;
                const int w    = ps_rfl(tid >> 6);
                const int KL   = H / 512;  // wave-loads per row (the host checks H % 512 == 0)
                const int V    = p.lm_rows;
                const int wr0 = (int)((long)V * bid / NB), wr1 = (int)((long)V * (bid + 1) / NB);
                const int ra = wr0 + (int)((long)(wr1 - wr0) * w / PS_NW), rb = wr0 + (int)((long)(wr1 - wr0) * (w + 1) / PS_NW);
                const int nl = (rb - ra) * KL;  // wave-loads of this wave (may be 0)
                const int nb = (nl + PS_U - 1) / PS_U;
                const auto* base = (const __attribute__((address_space(1))) char*)p.lm_w
                                   + (size_t)(ra < V ? ra : V - 1) * H * 2 + (size_t)lane * 16;
                auto lm_load = [&](u32x4 (&r)[PS_U], const int b) {
#pragma unroll
                    for (int u = 0; u < PS_U; u++) {
                        int n = b * PS_U + u;
                        n     = n < nl ? n : (nl > 0 ? nl - 1 : 0);  // (padding: re-reads of the wave's last KiB, never consumed)
                        r[u]  = __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4*)(base + (size_t)n * 1024));
                    }
                };
                const f16* hx = s.xs;  // [M][H + XPAD] normalised hidden state
                float      lacc[M];
#pragma unroll
                for (int m = 0; m < M; m++) {
                    lacc[m] = 0.f;
                }
                int  ci = 0, row = ra;
                auto lm_consume = [&](const u32x4 (&r)[PS_U], const int b) {
#pragma unroll
                    for (int u = 0; u < PS_U; u++) {
                        if (b * PS_U + u < nl) {  // (uniform)
                            const f16x8 wv = __builtin_bit_cast(f16x8, r[u]);
#pragma unroll
                            for (int m = 0; m < M; m++) {
                                const f16x8 xv = *reinterpret_cast<const f16x8*>(hx + (size_t)m * (H + XPAD) + ci * 512 + lane * 8);
                                float       a  = lacc[m];
                                a              = dot2(f16x2{wv[0], wv[1]}, f16x2{xv[0], xv[1]}, a);
                                a              = dot2(f16x2{wv[2], wv[3]}, f16x2{xv[2], xv[3]}, a);
                                a              = dot2(f16x2{wv[4], wv[5]}, f16x2{xv[4], xv[5]}, a);
                                a              = dot2(f16x2{wv[6], wv[7]}, f16x2{xv[6], xv[7]}, a);
                                lacc[m]        = a;
                            }
                            if (++ci == KL) {
#pragma unroll
                                for (int m = 0; m < M; m++) {
                                    const float v = wave_sum(lacc[m]);
                                    if (lane == 0) {
                                        p.lm_logits[(size_t)m * p.lm_ldc + row] = v;
                                    }
                                    lacc[m] = 0.f;
                                }
                                ci = 0;
                                row++;
                            }
                        }
                    }
                };
                if constexpr (!CTRL) {
                    lm_load(st.R0, 0);
                    lm_load(st.R1, 1);
                    lm_load(st.R2, 2);
                    lm_load(st.R3, 3);
                }
                f16x8 fg[PS_NLN], fb[PS_NLN];
#pragma unroll
                for (int k = 0; k < PS_NLN; k++) {
                    const int v = tid + k * PS_NT;
                    const int o = (v * 8 < H) ? v * 8 : 0;
                    fg[k]       = *PS_G(f16x8, p.lm_g + o);
                    fb[k]       = *PS_G(f16x8, p.lm_b + o);
                }
                if constexpr (CTRL) {
                    ps_sweep<20>(p.gx, M * H / 2, tid, PS_NC * 64, tag_base + (unsigned)(p.L - 1), p.err, 13,
                                [&](const int i, const unsigned v) { reinterpret_cast<unsigned*>(s.xraw)[i] = v; });
                }
                __syncthreads();
                {  // final LayerNorm (layernorm_kernels.cu:157-286 arithmetic, like the layers' and k_lm_head's)
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
                        s0[m] = wave_sum_dpp(s0[m]);
                        s1[m] = wave_sum_dpp(s1[m]);
                        if (lane == 0) {
                            s.red[(m * PS_NW + (tid >> 6)) * 2]     = s0[m];
                            s.red[(m * PS_NW + (tid >> 6)) * 2 + 1] = s1[m];
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < PS_NW; w2++) {
                            a0 += s.red[(m * PS_NW + w2) * 2];
                            a1 += s.red[(m * PS_NW + w2) * 2 + 1];
                        }
                        const float mean = a0 / (float)H;
                        const float rstd = rsqrtf(a1 / (float)H - mean * mean + p.eps);
                        const f16   mh = (f16)mean, rh = (f16)rstd;
#pragma unroll
                        for (int k = 0; k < PS_NLN; k++) {
                            const int v = tid + k * PS_NT;
                            if (v * 8 < H) {
                                const f16x8 x8 = *reinterpret_cast<const f16x8*>(s.xraw + (size_t)m * H + v * 8);
                                f16x8       o1;
#pragma unroll
                                for (int e = 0; e < 8; e++) {
                                    o1[e] = (((x8[e] - mh) * rh) * fg[k][e]) + fb[k][e];
                                }
                                *reinterpret_cast<f16x8*>(s.xs + (size_t)m * (H + XPAD) + v * 8) = o1;
                            }
                        }
                    }
                }
                if constexpr (CTRL) {
                    lm_load(st.R0, 0);
                    lm_load(st.R1, 1);
                    lm_load(st.R2, 2);
                    lm_load(st.R3, 3);
                }
                __syncthreads();
                const int lastb = ((nb + PS_NBUF - 1) / PS_NBUF - 1) * PS_NBUF;  // (nb = 0: one rotation of padding)
                for (int i = 0; i < (lastb < 0 ? 0 : lastb); i += PS_NBUF) {
                    lm_consume(st.R0, i);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_load(st.R0, i + 4);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_consume(st.R1, i + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_load(st.R1, i + 5);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_consume(st.R2, i + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_load(st.R2, i + 6);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_consume(st.R3, i + 3);
                    __builtin_amdgcn_sched_barrier(0);
                    lm_load(st.R3, i + 7);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int lb = lastb < 0 ? 0 : lastb;
                lm_consume(st.R0, lb);
                lm_consume(st.R1, lb + 1);
                lm_consume(st.R2, lb + 2);
                lm_consume(st.R3, lb + 3);
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


}  // namespace ftcf
