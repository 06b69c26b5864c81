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
// streamer waves: the whole first rotation (32 KiB) of the P1 stream is requested before the hand-off at the layer boundary
// (+0.8 %, round 2); at the attention seam only half of it (the whole one there was neutral: profiles/r02_notes.md)
constexpr bool PS_FULL_P1 = true, PS_FULL_P3 = false;
constexpr int PS_NLN      = 2;           // LayerNorm parameter vectors (f16x8) per thread and array: H <= 8192

typedef const PersistLayer PsLayerC;
#define PS_LAYER(p, l) (lay[l])  // (`lay`: the per-layer table -- global memory, or the workgroup's LDS copy of it)
#define PS_RLX __ATOMIC_RELAXED
#define PS_AGT __HIP_MEMORY_SCOPE_AGENT
// pointers that come out of the per-layer table in memory are GLOBAL: say so (a flat access also counts on lgkmcnt)
#define PS_G(T, ptr) ((const __attribute__((address_space(1))) T*)(ptr))

// batch-table entry (one per PS_U tiles of ONE run, consecutive k): bit 0 valid, 1 flush after the batch, 2..6 run,
// 8..24 LDS half offset of the first tile's x, 25 x stride select, 26 wait for the late x vector, 27..30 valid tiles
constexpr unsigned PS_BT_FAST = 1u, PS_BT_FLUSH = 2u, PS_BT_XSEL = 1u << 25, PS_BT_WAIT = 1u << 26;

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

// The same for a slice of up to NMAX granules per thread in ONE round trip (own-group layout: all of mid, all of ctx): the k-th
// load of a thread exists when k * nthr < n (uniform), so a short slice -- a tensor-parallel shard's -- issues what it needs.
// Addresses are a uniform base per k plus the thread's index (one VGPR for all loads: twenty clamped 64-bit addresses, live
// across the retry loop beside the control waves' prefetched batches, spilled); a pass may therefore read up to nthr - 1 granules
// PAST n -- the slabs swept this way are followed by other slabs of the same allocation -- and ignores what it finds there.
template<int NMAX, typename F>
__device__ __forceinline__ void ps_sweep_wide(const u64* g, const int n, const int tid, const int nthr, const unsigned tag,
                                              int* err, const int code, F&& sink)
{
    for (int base = 0; base < n; base += nthr * NMAX) {
        u64       gv[NMAX];
        const int kn    = (n - base + nthr - 1) / nthr;  // loads of this pass (uniform), <= NMAX of them issued
        int       spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < NMAX; k++) {
                if (k < kn) {
                    const u64* gk = g + (base + k * nthr);  // uniform
                    gv[k]         = ld_granule(&gk[tid]);
                }
            }
#pragma unroll
            for (int k = 0; k < NMAX; k++) {
                if (k < kn) {
                    ok &= ((unsigned)(gv[k] >> 32) == tag) || (base + k * nthr + tid >= n);
                }
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
        for (int k = 0; k < NMAX; k++) {
            const int i = base + k * nthr + tid;
            if (k < kn && i < n) {
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
// ---- P3 in the own-group layout (round 6; plan.own) ------------------------------------------------------------------
// Until round 5 every 16-column group of [FFN2 | out-proj] was cut along K into PB + PA pieces dealt over the workgroups
// (each workgroup: the same K range of ten groups -- a short sweep of mid), the pieces met as fp32 partials, the owner of a
// group's last piece merged them and published x', and every workgroup gathered x': TWO chip-wide hops at the layer boundary,
// 3.4 of a layer's 62.7 us on one GPU and 3 of 25 on a TP 8 shard.  Here a workgroup owns `no` = NG / NB column groups over
// their WHOLE K extent -- it finishes them alone and publishes x' (or this rank's share into the exchange windows) the moment
// its stream ends: one hop.  The NS = NG - no * NB groups left over (64 of 320 at 256 workgroups) are cut into PC = NB / NS
// pieces of their flat [FFN2 | out-proj] tile space, one per workgroup; the pieces without ctx-dependent tiles come FIRST in
// their workgroup's stream and every wave publishes its partial of such a run when it flushes it (PS_BT_PUB), so the owner of the
// last piece (the merger: it has the out-proj tiles, which are late anyway) finds them waiting when its own stream ends.
// Price: a workgroup reads all of mid and all of ctx (wide sweeps), not one K range of them.
constexpr unsigned PS_BT_PUB    = 1u << 7;  // batch descriptor: the run's partial of this wave is published when flushed
constexpr int      PS_OWN_MAXG  = 12;       // groups a workgroup finishes (own + one merged)
constexpr int      PS_OWN_ITEMS = 3;        // (group, row, column) items per control-wave thread
constexpr int      PS_OWN_MAXR  = 16;       // remote partial slots a merger adds
// own table (LDS ints): [0] groups finished here, [1] remote slots, [2] index of the merged group (-1: none),
// [4..15] group ids, [16..27] local runs of a group (FFN2 run | out-proj run << 8, 0xff: none), [28..43] remote slots
// ((slot * PS_NW + wave) << 1 | out-proj)
constexpr int PS_OT_G = 4, PS_OT_RUNS = 16, PS_OT_REM = 28, PS_OT_N = 64;
// RunRec::pad in this layout: 1 piece of a shared group merged elsewhere (published at flush), 2 own group, 4 the merger's piece
__host__ __device__ inline int ps_own_bound(const int KTT, const int PC, const int q)
{
    if (q <= 0) {
        return 0;
    }
    if (q >= PC) {
        return KTT;
    }
    long v = (long)KTT * q / PC;
    if (KTT / PC >= 4 * PS_U) {
        v = (v + PS_U / 2) / PS_U * PS_U;  // whole batches where a piece is long enough for it to matter
    }
    return (int)v;
}
__host__ __device__ inline bool ps_own_ok(const int NB, const int NG, const int M)
{
    if (NB < 1 || NG < NB) {
        return false;
    }
    const int no = NG / NB, NS = NG - no * NB;
    if (NS > 0 && NB % NS != 0) {
        return false;
    }
    return 2 * no + 2 <= PS_RMAX && no + 1 <= PS_OWN_MAXG && (no + 1) * M * 16 <= PS_OWN_ITEMS * PS_NC * 64;
}
// the P3 runs of workgroup b in stream order (r may be null: count only)
__host__ __device__ inline int ps_own_runs(const int b, const int NB, const int NG, const int KT_a, const int KT_b, const int TK,
                                           const int M, const int Il, RunRec* r)
{
    const int no = NG / NB, NS = NG - no * NB, PC = NS > 0 ? NB / NS : 0, KTT = KT_a + KT_b;
    int       n  = 0;
    auto      put = [&](const int g, const int sel, const int t0, const int nt, const int rid, const int flags) {
        if (r) {
            RunRec x;
            x.tile0 = g * (sel ? KT_a : KT_b) + t0;
            x.sel   = sel;
            x.nt    = nt;
            x.xoff  = sel ? M * (Il + XPAD) + t0 * TK : t0 * TK;
            x.xsel  = sel;
            x.rid   = rid;
            x.grp   = g;
            x.pad   = flags;
            r[n]    = x;
        }
        n++;
    };
    int sg = 0, lo = 0, hi = 0, q = 0;
    if (NS > 0) {
        sg = no * NB + b / PC;
        q  = b % PC;
        lo = ps_own_bound(KTT, PC, q);
        hi = ps_own_bound(KTT, PC, q + 1);
    }
    const bool merger = NS > 0 && q == PC - 1;
    const int  fl = lo < KT_b ? lo : KT_b, fh = hi < KT_b ? hi : KT_b;                 // FFN2 tiles [fl, fh) of the piece
    const int  ol = (lo > KT_b ? lo : KT_b) - KT_b, oh = (hi > KT_b ? hi : KT_b) - KT_b;  // out-proj tiles [ol, oh)
    const int  slot = NS > 0 ? ((sg - no * NB) * PC + q) * 2 : 0;
    if (NS > 0 && !merger && fh > fl) {
        put(sg, 0, fl, fh - fl, slot, 1);
    }
    for (int k = 0; k < no; k++) {
        put(b * no + k, 0, 0, KT_b, -1, 2);
    }
    if (merger && fh > fl) {
        put(sg, 0, fl, fh - fl, slot, 4);
    }
    if (NS > 0 && oh > ol) {
        put(sg, 1, ol, oh - ol, slot + 1, merger ? 4 : 1);
    }
    for (int k = 0; k < no; k++) {
        put(b * no + k, 1, 0, KT_a, -1, 2);
    }
    return n;
}
// merger b: the (slot, wave) partials the other pieces of its shared group publish, in the order they are added
// (tmp: room for PS_RMAX runs; ent may be null: count only)
__host__ __device__ inline int ps_own_remote(const int b, const int NB, const int NG, const int KT_a, const int KT_b, const int TK,
                                             const int M, const int Il, const int cs3, RunRec* tmp, int* ent, const int maxent)
{
    const int no = NG / NB, NS = NG - no * NB, PC = NS > 0 ? NB / NS : 0;
    if (NS == 0 || b % PC != PC - 1) {
        return 0;
    }
    int n = 0;
    for (int q = 0; q < PC - 1; q++) {
        const int nr = ps_own_runs((b / PC) * PC + q, NB, NG, KT_a, KT_b, TK, M, Il, tmp);
        int       T  = 0;
        for (int j = 0; j < nr; j++) {
            T += tmp[j].nt;
        }
        int pre = 0;
        for (int j = 0; j < nr; j++) {
            if (tmp[j].pad & 1) {
                for (int w = 0; w < PS_NW; w++) {
                    int tb, te;
                    ps_wave_range(T, w, cs3, tb, te);
                    const int a = tb > pre ? tb : pre, e = te < pre + tmp[j].nt ? te : pre + tmp[j].nt;
                    if (e > a) {
                        if (ent && n < maxent) {
                            ent[n] = ((tmp[j].rid * PS_NW + w) << 1) | tmp[j].sel;
                        }
                        n++;
                    }
                }
            }
            pre += tmp[j].nt;
        }
    }
    return n;
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

template<bool INT8, int M, bool OWN = false>
struct PsStream {
    static constexpr int TK = TileK<INT8>::value;
    u32x4      R0[PS_U], R1[PS_U], R2[PS_U], R3[PS_U];
    f32x4      acc;
    PsStage    g;
    const f16* rsc;
    const f16* xs;
    float*     part;
    const int* flag;    // LDS arrival counter of the second x vector
    int        target;  // value it reaches when that vector is staged
    int        lane, wid;
    // own-group layout: runs of the stage (LDS), the slab of published partials and the layer's tag (PS_BT_PUB)
    const RunRec* rt;
    u64*          gpub;
    unsigned      ptag;

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
    __device__ __forceinline__ void flush(const int j, const unsigned bd)
    {
        if (lane < 16) {
#pragma unroll
            for (int m = 0; m < M; m++) {
                part[((size_t)j * PS_NW + wid) * (M * 16) + m * 16 + lane] = acc_row(acc, m);
            }
        }
        if constexpr (OWN) {
            // a piece of a shared group whose merger is another workgroup: this wave's partial leaves NOW (the merger adds the
            // waves' partials in a fixed order), long before the merger's own stream ends
            if (bd & PS_BT_PUB) {
                const int slot = ps_rfl(rt[j].rid);
                if (lane < 16) {
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        st_granule(&gpub[((size_t)slot * PS_NW + wid) * (M * 16) + m * 16 + lane], ptag, acc_row(acc, m));
                    }
                }
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
            flush(j, bd);
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
        load(R2, 2);
        load(R3, 3);
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
        consume(R1, last + 1);
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
                     | ((unsigned)cnt << 27) | ((r.pad & 1) ? PS_BT_PUB : 0u);
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
    char*     att;   // attention scratch
    RunRec*   rt1;   // [RMAX] P1 runs
    RunRec*   rt3;   // [RMAX] P3 runs
    f16*      rsc;   // [RMAX][16] scales of the current stage
    float*    red;   // 64
    int*      misc;  // 64: [0] nmerge, [1..8] merge groups
    int*      otab;  // PS_OT_N: the own-group layout's table (see PS_OT_*)
    u64*      layt;  // own-group layout: the launch's per-layer pointer table, copied in at kernel entry
    unsigned *lt1, *lt3;  // [NW][e1], [NW][e3]
    unsigned *bt1, *bt3;  // [NW][e1 / PS_U], [NW][e3 / PS_U]
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
template<int DH, int UK, int NSW = 3 * DH / 2, bool AL = (UK > PS_U)>
struct PsAttn {
    static constexpr int LPK = DH / 8;
    static constexpr int NQ  = 3 * DH / 2;             // q | k | v granules (pairs of halves) of one head
    static constexpr int NB2 = (NQ + NSW - 1) / NSW;   // granules per sweeping thread (NSW threads sweep)
    static constexpr int KPI = 64 / LPK;
    static constexpr bool ALIAS = AL;  // rows live in the stream's register batches: K in R0 | R1, V in R2 | R3
    static_assert(UK <= 2 * PS_U, "K rows live in R0|R1, V rows in R2|R3");
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
        if constexpr (ALIAS) {
            return u < PS_U ? st.R2[u % PS_U] : st.R3[u % PS_U];
        }
        else {
            return vreg[u];
        }
    }
    unsigned mask_bits, bias2[NB2];
    int      tl, chunk, t_beg;
    float    rot_cs, rot_sn;
    bool     fin;

    // loads that do not depend on this step's qkv: K/V rows of the whole fixed chunk, masks, lengths, rotary table
    // `item` false: a workgroup without a (row, head, split) that must not assign the row registers under a condition (they
    // would be carried around the layer loop): it requests ONE cached row over and over
    template<typename ST>
    __device__ __forceinline__ void issue(const PersistParams& p, PsLayerC& lw, int h, int b, int sp, const int tx, ST& st,
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
    // Round 4 (PS_ATT_WP): the same split on the eight waves with WAVE-PRIVATE soft-max statistics.  compute() below takes five
    // workgroup barriers (q/k/v staged, rotary, maxima, exponentials, outputs) and sends every score through LDS twice; here a
    // wave runs its own keys -- 4 (dh = 128) or 8 (dh = 64) per register row, UK rows -- through the whole soft-max against ITS
    // OWN maximum (scores, probabilities and the weighted V rows never leave its registers), and the eight {maximum, sum,
    // un-normalised output} triples are combined like the splits of a (row, head) are: three barriers, no score buffer.  The
    // arithmetic is compute_ctrl's (decoder_masked_multihead_attention_template.hpp:1099-1919 with fp32 probabilities);
    // against compute() only the reference point of the exponentials differs (own maximum, rescaled at the end).
    template<typename ST>
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
        if (owns_cur && tx < DH) {  // append to the cache (:1397, :1837)
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
    template<typename ST>
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
        if (owns_cur && tx < DH) {  // append to the cache (:1397, :1837)
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

// split-0 workgroup of a (row, head): the nsplit partials -- swept into LDS by both control waves (ps_attn_merge_sweep) --
// are merged in split order by WAVE 0 and published as ctx granules (no workgroup barrier: the other waves are already
// streaming the next stage)
template<int DH>
__device__ __forceinline__ void ps_attn_merge(const PersistParams& p, char* smem, const unsigned tag, int h, int b, const int tx)
{
    const int ne = DH + 2, ns = p.plan.nsplit;
    float*    sval = reinterpret_cast<float*>(smem);  // [ns][ne]
    {
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
    }
}

// both control waves of a split-0 workgroup sweep the (row, head)'s partials into LDS
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
// Rejected forms of this kernel (the attention on the control waves alone, P3 tiles prefetched into LDS, the LM head as the launch's
// tail, q/k/v published from inside the stream, a second kernel with the attention branch under the FFN streams) are described
// with their measurements in profiles/r03_notes.md and profiles/r04_notes.md; their code is no longer in the tree.
template<bool INT8, int M, int DH, int UK, bool TP, bool GROUP = false, bool OWN = false>
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
    constexpr bool NOCARRY = UK > PS_U || M > 1 || TP || OWN;
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
        s.att = q;
        q += ps_att_bytes(DH, p.s_max, p.plan.nsplit);
        s.rt1 = reinterpret_cast<RunRec*>(q);
        q += sizeof(RunRec) * PS_RMAX;
        s.rt3 = reinterpret_cast<RunRec*>(q);
        q += sizeof(RunRec) * PS_RMAX;
        s.rsc = reinterpret_cast<f16*>(q);
        q += PS_RMAX * 16 * 2;
        s.red = reinterpret_cast<float*>(q);
        q += 64 * 4;
        s.misc = reinterpret_cast<int*>(q);
        q += 64 * 4;
        s.otab = reinterpret_cast<int*>(q);
        q += PS_OT_N * 4;
        s.lt1 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * E1 * 4;
        s.lt3 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * E3 * 4;
        s.bt1 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * (E1 / PS_U) * 4;
        s.bt3 = reinterpret_cast<unsigned*>(q);
        q += (size_t)PS_NW * (E3 / PS_U) * 4;
        s.layt = reinterpret_cast<u64*>(q);
    }
    // The per-layer table (17 pointers per layer) lives in global memory and every use of it is a DEPENDENT vector load: on a
    // compute unit whose other waves stream, ~3-4 us each -- once per layer at the boundary, in front of the gather of x', and
    // again in front of the streamer waves' first requests.  The own-group kernels copy it into LDS when they start.
    const PersistLayer* lay = p.layers;
    if constexpr (OWN) {
        const auto* src = PS_G(u64, p.layers);
        for (int i = threadIdx.x; i < p.L * (int)(sizeof(PersistLayer) / 8); i += PS_NT) {
            s.layt[i] = src[i];
        }
        lay = reinterpret_cast<const PersistLayer*>(s.layt);
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
    // tiles.  plan.qrot (1: kernels_persist.hip) rotates WHICH workgroups get the light share: workgroup b runs on XCD
    // b % 8 and the stamps show XCDs 2 / 6 ending P1 1.3 us after the others on equal shares; moving the light share onto
    // them measured +-0.3 %: whoever is last, the next hand-off waits for it)
    const int qb  = (bid + p.plan.qrot) % NB;
    const int q0  = (int)((long)NT0 * qb / NB), q1 = (int)((long)NT0 * (qb + 1) / NB);
    const int f0  = (int)((long)NF * bid / NB), f1 = (int)((long)NF * (bid + 1) / NB);
    const int nq  = q1 - q0;
    const int rB0 = (int)((long)NG * PB * bid / NB), rB1 = (int)((long)NG * PB * (bid + 1) / NB);
    const int rA0 = (int)((long)NG * PA * bid / NB), rA1 = (int)((long)NG * PA * (bid + 1) / NB);
    const int nB = rB1 - rB0, nA = rA1 - rA0;
    const int nruns1 = nq + (f1 - f0);
    const int nruns3 = OWN ? ps_own_runs(bid, NB, NG, KT_a, KT_b, TK, M, Il, nullptr) : nB + nA;
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
            s.misc[39] = 0;  // control pair barrier of the partials' sweep
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
        if constexpr (OWN) {  // P3, own-group layout: the runs in stream order, the groups finished here, the merger's remote slots
            if (threadIdx.x == 0) {
                ps_own_runs(bid, NB, NG, KT_a, KT_b, TK, M, Il, s.rt3);
                int ng = 0, mg = -1;
                for (int j = 0; j < nruns3; j++) {  // groups in the order of their first run; a group has at most one run per matrix here
                    const RunRec r = s.rt3[j];
                    if (r.pad & 1) {
                        continue;  // merged elsewhere
                    }
                    int k = 0;
                    while (k < ng && s.otab[PS_OT_G + k] != r.grp) {
                        k++;
                    }
                    if (k == ng) {
                        s.otab[PS_OT_G + k]    = r.grp;
                        s.otab[PS_OT_RUNS + k] = 0xffff;
                        ng++;
                    }
                    if (r.pad & 4) {
                        mg = k;
                    }
                    const int old = s.otab[PS_OT_RUNS + k];
                    s.otab[PS_OT_RUNS + k] = r.sel ? ((old & 0xff) | (j << 8)) : ((old & 0xff00) | j);
                }
                s.otab[0] = ng;
                s.otab[2] = mg;
                // (the temporary run list of the other pieces' workgroups lives in the partial buffer, free until the first layer)
                s.otab[1] = ps_own_remote(bid, NB, NG, KT_a, KT_b, TK, M, Il, p.plan.cs3, reinterpret_cast<RunRec*>(s.part),
                                          &s.otab[PS_OT_REM], PS_OWN_MAXR);
            }
        }
        else if ((int)threadIdx.x < nruns3) {  // P3: FFN2 K pieces first, then out-proj K pieces (piece-major ids)
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
            ps_wave_range(T1, w, p.plan.cs1, tb, te);
            ent      = ps_wave_entries(nruns1, [&](int j) { return s.rt1[j].nt; }, tb, te);
            sg1.nrot = (ent + PS_U * PS_NBUF - 1) / (PS_U * PS_NBUF);
            sg1.nrot = sg1.nrot < 1 ? 1 : sg1.nrot;
            ps_build_tables<TK>(s.rt1, nruns1, tb, te, s.lt1 + (size_t)w * E1, s.bt1 + (size_t)w * (E1 / PS_U),
                                sg1.nrot * PS_U * PS_NBUF);
            ps_wave_range(T3, w, p.plan.cs3, tb, te);
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
        PsStream<INT8, M, OWN> st;
        auto stamp = [&](const int l, const int k) {
            const int lane = tid & 63, wid = tid >> 6;
            if (p.ts && lane == 0) {
                p.ts[(((size_t)bid * p.L + l) * PS_NW + wid) * 16 + k] = wall_clock64();
            }
        };
        // ---- per-layer constants, fetched one stage ahead into registers (before that stage's prefetch) ----
        f16   r_sc1 = (f16)1.f, r_sc3 = (f16)1.f;  // scale of (run tid/16, column tid%16) of P1 / P3
        constexpr int NBR = OWN ? PS_OWN_ITEMS : 2;
        f16   r_b1[2], r_bres[NBR];                // ffn1 bias of the P1 epilogue items / residual bias of the merge items
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
                const int           nm = OWN ? s.otab[0] : (s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE);
#pragma unroll
                for (int k = 0; k < NBR; k++) {
                    const int t = tid + k * PS_NC * 64;
                    r_bres[k]   = (f16)0.f;
                    if (t < nm * M * 16) {
                        const int g = OWN ? s.otab[PS_OT_G + t / (M * 16)] : s.misc[1 + t / (M * 16)];
                        r_bres[k]   = PS_G(f16, lw.b_res)[g * 16 + (t & 15)];
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
            if constexpr (!CTRL) {
                st.prime_lo();
                if constexpr (PS_FULL_P1) {
                    st.prime_hi();
                }
            }
        };
        auto setup_p3 = [&](const int l) {
            PsLayerC& lw = PS_LAYER(p, l);
            if constexpr (INT8) {
                if (tid < nruns3 * 16) {
                    s.rsc[tid] = r_sc3;
                }
            }
            for (int i = tid; i < nruns3 * PS_NW * M * 16; i += PS_NT) {
                s.part[i] = 0.f;
            }
            sg3.w0 = reinterpret_cast<const char*>(lw.w_ffn2);
            sg3.w1 = reinterpret_cast<const char*>(lw.w_out);
            st.bind(sg3, s.rsc, s.xs, s.part, tid, &s.misc[32], (l - p.l_begin + 1) * PS_NC);
            if constexpr (OWN) {
                st.rt   = s.rt3;
                st.gpub = p.gp;
                st.ptag = tag_base + (unsigned)l;
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
                st.template run<CTRL || !PS_FULL_P1>();
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
                stamp(l, 4);
            }

            // =========================== attention ===============================================================
            asm volatile("" : "+v"(tid));
            // Tensor-parallel shards (a rank's share of a stream is shorter than one rotation of prefetch, the layer is hand-offs):
            // CTRL_EARLY -- the control waves request their whole P3 share, the ctx-dependent out-proj pieces at the end of the
            // workgroup's tile space, after the merge of the splits and BEFORE they wait for ctx, so that it lands under that wait
            // (TP 2 / 4 / 8 shard +1.7 / +1.8 / +4.4 %, profiles/r06_notes.md; on one GPU the sweep of ctx returns behind the
            // prefetch and the share still needs a second loaded latency: slower, profiles/r04_notes.md); ATT_WP -- the attention
            // with wave-private soft-max statistics (three workgroup barriers instead of five: +1 % on a shard, -1 % on one GPU).
            constexpr bool CTRL_EARLY = TP, ATT_WP = TP;
            // (own-group layout: the short form keeps its rows in the idle stream batches too -- 64 VGPRs that the wide sweeps need)
            using Attn          = PsAttn<DH, UK, 3 * DH / 2, (UK > PS_U) || OWN>;
            Attn       at;
            const bool has_item = bid < n_items;
            int        a_sp = 0, a_h = 0, a_b = 0;
            if (has_item) {
                a_sp         = bid % p.plan.nsplit;
                const int hb = bid / p.plan.nsplit;
                a_h          = hb % p.nh;
                a_b          = hb / p.nh;
            }
            __syncthreads();  // part / scales reuse
            setup_p3(l);
            stamp(l, 5);
            bool live = false;
            u64* gall = p.ga + ((size_t)a_b * p.nh + a_h) * p.plan.nsplit * (DH + 2);
            if constexpr (Attn::ALIAS) {
                // (rows in the stream's register batches: requested UNCONDITIONALLY -- a workgroup without an item reads
                // item 0's rows for nothing -- because registers assigned under a condition carry their previous contents,
                // here all four weight batches, around the layer loop: 30 spilled VGPRs)
                at.issue(p, lw, a_h, a_b, a_sp, tid, st);
            }
            if (has_item) {
                // (issued here and not before the barrier above: K/V rows held across setup_p3 spill, measured)
                if constexpr (!Attn::ALIAS) {
                    at.issue(p, lw, a_h, a_b, a_sp, tid, st);
                }
                at.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                stamp(l, 6);
                if constexpr (ATT_WP) {
                    live = at.compute_wp(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st);
                }
                else {
                    live = at.compute(p, lw, s.att, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st);
                }
            }
            // the K range of mid this workgroup's FFN2 pieces read -> LDS, by all eight waves (published at the end of P1: long
            // there), after the attention (ahead of it, it made the attention wait for the slowest FFN1)
            if constexpr (OWN) {  // all of mid (the own groups run over the whole K), every row, one round trip
                ps_sweep_wide<(TP ? 10 : 20)>(p.gm, (M * Il) >> 1, tid, PS_NT, tag, p.err, 6, [&](const int i, const unsigned v) {
                    const int m = (M > 1 && i >= (Il >> 1)) ? 1 : 0, c = i - m * (Il >> 1);
                    reinterpret_cast<unsigned*>(s.xs + (size_t)m * (Il + XPAD))[c] = v;
                });
            }
            else {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    ps_sweep<3>(p.gm + (((size_t)m * Il + mid_lo) >> 1), (mid_hi - mid_lo) >> 1, tid, PS_NT, tag, p.err, 6,
                                [&](const int i, const unsigned v) {
                                    reinterpret_cast<unsigned*>(s.xs + (size_t)m * (Il + XPAD) + mid_lo)[i] = v;
                                });
                }
            }
            stamp(l, 7);
            __syncthreads();  // mid staged, attention scratch free
            stamp(l, 8);
            if constexpr (!CTRL) {
                // (the constants of the layer's end -- residual bias, next layer's scales -- are fetched HERE and not in
                // the P3 set-up: their table reads are synchronous round trips, and the set-up sits on the attention's
                // critical path, ahead of the K/V request)
                load_p3_consts(l);
                // AFTER the barrier: issuing 32 KiB per wave takes ~5 us (the CU's memory pipeline throttles the issue)
                // and the control waves, which carry the attention's critical path, must not wait for it
                st.prime_lo();  // the streamer waves issue no other load until the end of the P3 stream
                if constexpr (PS_FULL_P3) {
                    st.prime_hi();
                }
            }
            // =========================== P3: [FFN2 u out-proj] -> residual ========================================
            // The streamer waves start on the FFN2 pieces at once; the control waves finish the attention (merge of the
            // split partials by wave 0 of the split-0 workgroups), stage the K range of ctx the out-proj pieces read and
            // announce it through an LDS counter that gates every batch touching ctx; their own share is the END of the
            // workgroup's tile space, i.e. the out-proj pieces.
            if constexpr (CTRL) {
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
                            ps_attn_merge<DH>(p, s.att, tag, a_h, a_b, tid);
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
                if constexpr (CTRL_EARLY) {
                    st.prime();
                }
                if constexpr (OWN) {
                    // All of ctx, 20 granules per thread and pass on one GPU.  (Waiting on ONE granule per head first and sweeping when
                    // those carry the tag -- a quarter of the polling traffic -- costs a second loaded round trip: +3 us, measured.)
                    ps_sweep_wide<(TP ? 10 : 20)>(p.gc, (M * Hl) >> 1, tid, PS_NC * 64, tag, p.err, 7, [&](const int i, const unsigned v) {
                        const int m = (M > 1 && i >= (Hl >> 1)) ? 1 : 0, c = i - m * (Hl >> 1);
                        reinterpret_cast<unsigned*>(s.xs + (size_t)M * (Il + XPAD) + (size_t)m * (Hl + XPAD))[c] = v;
                    });
                }
                else {
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        ps_sweep<5>(p.gc + (((size_t)m * Hl + ctx_lo) >> 1), (ctx_hi - ctx_lo) >> 1, tid, PS_NC * 64, tag,
                                    p.err, 7, [&](const int i, const unsigned v) {
                                        reinterpret_cast<unsigned*>(s.xs + (size_t)M * (Il + XPAD) + (size_t)m * (Hl + XPAD) + ctx_lo)[i] = v;
                                    });
                    }
                }
                stamp(l, 14);
                if (lane == 0) {
                    atomicAdd(&s.misc[32], 1);  // DS operations of a wave execute in order: the writes above are visible
                }
                load_p3_consts(l);
                if constexpr (!CTRL_EARLY) {
                    st.prime_lo();
                }
            }
            stamp(l, 9);
            st.template run<CTRL ? !CTRL_EARLY : !PS_FULL_P3>();
            stamp(l, 10);
            __syncthreads();
            asm volatile("" : "+v"(tid));
            // own-group layout: the sums of the groups this workgroup finishes, out of the waves' partials (control waves; the
            // partial buffer is the streamer waves' again behind the next barrier)
            float own_sa[NBR], own_sb[NBR];
            if constexpr (OWN) {
                if constexpr (CTRL) {
                    const int ng = s.otab[0];
#pragma unroll
                    for (int k2 = 0; k2 < NBR; k2++) {
                        const int t = tid + k2 * PS_NC * 64;
                        own_sa[k2]  = 0.f;
                        own_sb[k2]  = 0.f;
                        if (t < ng * M * 16) {
                            const int k = t / (M * 16), r = t % (M * 16);
                            const int runs = s.otab[PS_OT_RUNS + k], jf = runs & 0xff, jo = (runs >> 8) & 0xff;
                            if (jf != 0xff) {
#pragma unroll
                                for (int w = 0; w < PS_NW; w++) {  // wave order: deterministic
                                    own_sb[k2] += s.part[((size_t)jf * PS_NW + w) * (M * 16) + r];
                                }
                            }
                            if (jo != 0xff) {
#pragma unroll
                                for (int w = 0; w < PS_NW; w++) {
                                    own_sa[k2] += s.part[((size_t)jo * PS_NW + w) * (M * 16) + r];
                                }
                            }
                        }
                    }
                }
            }
            else {
                // K pieces -> granules
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int idx = tid + k * PS_NT;
                    if (idx < nruns3 * M * 16) {
                        const int j = idx / (M * 16), r = idx % (M * 16);
                        float     v = 0.f;
#pragma unroll
                        for (int w = 0; w < PS_NW; w++) {
                            v += s.part[((size_t)j * PS_NW + w) * (M * 16) + r];
                        }
                        st_granule(&p.gp[(size_t)s.rt3[j].rid * (M * 16) + r], tag, v);
                    }
                }
            }
            const int  nmerge = OWN ? s.otab[0] : (s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE);
            const bool last   = (l == p.l_end - 1);
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
                for (int k2 = 0; k2 < NBR; k2++) {
                    const int t = tid + k2 * PS_NC * 64;
                    if (t < nmerge * M * 16) {
                        const int k = t / (M * 16), r = t % (M * 16), m = r >> 4, c = r & 15;
                        const int g = OWN ? s.otab[PS_OT_G + k] : s.misc[1 + k];
                        float     sa = 0.f, sb = 0.f;
                        if constexpr (OWN) {
                            // the group of which this workgroup holds the last piece: the other pieces' partials, one per (run,
                            // wave) that streamed a part of it, were published when those waves flushed them -- long ago
                            const int nrem = s.otab[1];
                            if (nrem > 0 && k == s.otab[2]) {
                                u64 gv[PS_OWN_MAXR];
                                int spins = 0;
                                for (;;) {
                                    bool ok = true;
#pragma unroll
                                    for (int q = 0; q < PS_OWN_MAXR; q++) {
                                        if (q < nrem) {
                                            gv[q] = ld_granule(&p.gp[(size_t)(s.otab[PS_OT_REM + q] >> 1) * (M * 16) + r]);
                                        }
                                    }
#pragma unroll
                                    for (int q = 0; q < PS_OWN_MAXR; q++) {
                                        if (q < nrem) {
                                            ok &= ((unsigned)(gv[q] >> 32) == tag);
                                        }
                                    }
                                    if (ok || ps_give_up(spins, p.err, 4)) {
                                        break;
                                    }
                                    __builtin_amdgcn_s_sleep(1);
                                }
#pragma unroll
                                for (int q = 0; q < PS_OWN_MAXR; q++) {  // slot order: deterministic
                                    if (q < nrem) {
                                        const float v = __uint_as_float((unsigned)gv[q]);
                                        if (s.otab[PS_OT_REM + q] & 1) {
                                            sa += v;
                                        }
                                        else {
                                            sb += v;
                                        }
                                    }
                                }
                            }
                            sa += own_sa[k2];
                            sb += own_sb[k2];
                        }
                        else {
                            u64 gv[PS_MAXP];
                            int spins = 0;
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
#pragma unroll
                            for (int q = 0; q < PS_MAXP; q++) {  // piece order: deterministic
                                if (q < PA) {
                                    sa += __uint_as_float((unsigned)gv[q]);
                                }
                                else if (q < PA + PB) {
                                    sb += __uint_as_float((unsigned)gv[q]);
                                }
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
                        else if (last) {
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
    };
    if (wid < PS_NC) {
        body(std::true_type{});
    }
    else {
        body(std::false_type{});
    }
}


}  // namespace ftcf
