// Persistent decode layers, second form (round 4): the ATTENTION BRANCH LEAVES THE WEIGHT STREAM'S CRITICAL PATH.
// GptNeoXDecoder<T>::forward (models/gptneox/GptNeoXDecoder.cc:245-384), one row, <= 256 keys per KV split.
//
// persist_device.hip.h runs a layer as  [QKV u FFN1] -> attention -> [FFN2 u out-proj]  on all eight waves of every
// workgroup: between the two weight streams the whole chip waits for a chain of dependent hand-offs (q/k/v hop 4.7 us,
// attention 3.5, mid sweep 1.5, first rotation's latency 2-4: profiles/r03_notes.md, 11-14 us of a 61 us layer in which
// only the 26 MB of K/V rows move).  A parallel-residual layer has TWO independent branches, though:
//     attention branch:  LN1 -> QKV -> attention -> out-proj            FFN branch:  LN2 -> FFN1 -> gelu -> FFN2
// Here the six STREAMER waves of a workgroup stream  QKV -> FFN1 -> (mid hop) -> FFN2 -> out-proj  back to back -- QKV first
// in TIME on every wave, so that q/k/v are complete after ~40 % of the first stream -- and the two CONTROL waves carry the
// attention branch alone, under the FFN streams: K rows of the workgroup's KV split by LDS-DMA into 64 KiB of LDS and V
// rows into their (idle) register batches, requested when the layer begins; they publish q/k/v the moment the streamer
// waves have flushed the QKV runs (an LDS counter bumped from inside the stream), sweep their head's q/k/v, run the split
// on 128 threads (PsAttn::compute_ctrl), merge the splits, stage ctx -- all of it while FFN1 and FFN2 stream -- and then
// take their share of the stream's tail (the out-proj pieces, which need ctx and sit at the end of the tile space).
// The only window left between the streams is the mid hop (FFN1 -> FFN2), bridged by the FFN2 prefetch.
// Everything else is persist_device.hip.h's: granule hand-offs, static run / tile tables (built once per request),
// four rotating register batches, bounded spins, the in-launch tensor-parallel exchange of x'.
#pragma once
#include "persist_device.hip.h"

namespace ftcf {

constexpr int PS4_NS = PS_NW - PS_NC;  // streamer waves

#ifndef PS4_MID_PRIME
#define PS4_MID_PRIME 2  // FFN2 batches the streamer waves request BEFORE they sweep mid (0: none, 1: half, 2: whole rotation)
#endif
#ifndef PS4_COND_TAIL
#define PS4_COND_TAIL 1  // the streamer waves skip the next layer's set-up (and its 32 KiB of prefetch per wave) after the last layer
#endif

// The P1 shares of a workgroup with Tq QKV tiles and Tf FFN1 tiles.  Streamer wave i (0..PS4_NS-1): a slice of the QKV runs
// FIRST, then a slice of the FFN1 runs sized so that the waves' totals balance.  The control waves are idle from the
// LayerNorm until q/k/v are complete: they stream cs/16 of a streamer wave's share from the END of the FFN1 runs first (their
// register batches hold the V rows only afterwards).  Slices start on batch boundaries.
__host__ __device__ inline int ps4_ctrl_tiles(const int Tq, const int Tf, const int cs)
{
    const long W = (long)PS4_NS * 16 + (long)PS_NC * cs;
    long       t = (long)(Tq + Tf) * cs / W / PS_U * PS_U;
    const long cap = (long)Tf / PS_NC / PS_U * PS_U;
    return (int)(t < cap ? t : cap);
}
__host__ __device__ inline void ps4_p1_ranges(const int Tq, const int Tf_all, const int cs, const int i, int& qb, int& qe, int& fb,
                                              int& fe)
{
    const int Tf = Tf_all - PS_NC * ps4_ctrl_tiles(Tq, Tf_all, cs);  // the streamer waves' part of FFN1
    auto al = [](long v) { return (int)(v / PS_U * PS_U); };
    auto qend = [&](int k) { return k >= PS4_NS ? Tq : al((long)Tq * k / PS4_NS); };
    auto fend = [&](int k) {  // FFN1 tiles owned by streamer waves 0..k-1
        if (k >= PS4_NS) {
            return Tf;
        }
        if (k <= 0) {
            return 0;
        }
        long c = (long)(Tq + Tf) * k / PS4_NS - qend(k);
        c      = c < 0 ? 0 : (c > Tf ? Tf : c);
        return al(c);
    };
    qb = qend(i);
    qe = qend(i + 1);
    fb = fend(i);
    fe = fend(i + 1);
    if (fe < fb) {
        fe = fb;
    }
}
// control wave c (0..PS_NC-1): [fb, fe) of the FFN1 runs
__host__ __device__ inline void ps4_p1_ctrl_range(const int Tq, const int Tf_all, const int cs, const int c, int& fb, int& fe)
{
    const int tc = ps4_ctrl_tiles(Tq, Tf_all, cs);
    fb           = Tf_all - (PS_NC - c) * tc;
    fe           = fb + tc;
}
// table entries of a wave's share (every run piece padded to whole batches), KT tiles per run; w: wave of the workgroup
__host__ __device__ inline int ps4_p1_entries(const int nq, const int nf, const int KT, const int cs, const int w, int& eq)
{
    auto nt = [&](int) { return KT; };
    if (w < PS_NC) {
        int fb, fe;
        ps4_p1_ctrl_range(nq * KT, nf * KT, cs, w, fb, fe);
        eq = 0;
        return ps_wave_entries(nf, nt, fb, fe);
    }
    int qb, qe, fb, fe;
    ps4_p1_ranges(nq * KT, nf * KT, cs, w - PS_NC, qb, qe, fb, fe);
    eq = ps_wave_entries(nq, nt, qb, qe);
    return eq + ps_wave_entries(nf, nt, fb, fe);
}

template<bool INT8, int DH, bool TP, bool GROUP = false>
__global__ __launch_bounds__(PS_NT) void k_decode_persistent4(
    const typename std::conditional<GROUP, PersistGroupParams, PersistParams>::type pa)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int M  = 1;
    constexpr int UK = PS_UK;
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
    const int     H = p.H, Hl = p.Hl, Il = p.Il;
    const int     NB = p.plan.NB;
    const int     wid = threadIdx.x >> 6;
    const int     KT = H / TK, KT_a = Hl / TK, KT_b = Il / TK;
    const int     NT0 = 3 * Hl / 16, NG = H / 16;
    const int     PA = p.plan.PA, PB = p.plan.PB, RLa = p.plan.RLa, RLb = p.plan.RLb;
    const int     E1 = p.plan.e1, E3 = p.plan.e3;
    const int     R1 = p.plan.r1max, R3 = p.plan.r3max;       // runs per workgroup (P1 / P3), the maximum over workgroups
    // x region of P3: [mid range of this workgroup | ... | ctx]; ctx lies BEHIND LN1(x) | LN2(x): the control waves stage it
    // while the streamer waves still read LN2(x)
    const int     ctx_off = p.plan.ctx_off;

    PsSmem s;
    {
        char* q = smem;
        s.xraw  = reinterpret_cast<f16*>(q);
        q += (size_t)H * 2;
        s.xs = reinterpret_cast<f16*>(q);
        q += (size_t)p.plan.xs_halves * 2;
        s.part = reinterpret_cast<float*>(q);
        q += (size_t)R1 * PS_NW * 16 * 4;
        s.part3 = reinterpret_cast<float*>(q);
        q += (size_t)R3 * PS_NW * 16 * 4;
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
        p.ts[(((size_t)blockIdx.x * p.L + p.l_begin) * PS_NW + (threadIdx.x >> 6)) * 16 + 15] = wall_clock64();
    }

    // ---- the workgroup's static share (the same split of column groups and K pieces as persist_device.hip.h) ----
    const int NF  = Il / 16;
    const int qb  = (bid + p.plan.qrot) % NB;
    const int q0  = (int)((long)NT0 * qb / NB), q1 = (int)((long)NT0 * (qb + 1) / NB);
    const int f0  = (int)((long)NF * bid / NB), f1 = (int)((long)NF * (bid + 1) / NB);
    const int nq  = q1 - q0, nf = f1 - f0;
    const int rB0 = (int)((long)NG * PB * bid / NB), rB1 = (int)((long)NG * PB * (bid + 1) / NB);
    const int rA0 = (int)((long)NG * PA * bid / NB), rA1 = (int)((long)NG * PA * (bid + 1) / NB);
    const int nB = rB1 - rB0, nA = rA1 - rA0;
    const int nruns1 = nq + nf, nruns3 = nB + nA;
    const int n_items = p.B * p.nh * p.plan.nsplit;
    const size_t tab_bytes = (size_t)(reinterpret_cast<char*>(s.bt3 + (size_t)PS_NW * (E3 / PS_U)) - reinterpret_cast<char*>(s.rt1));
    if (p.tab_mode == 2) {
        const auto* src = PS_G(u32x4, p.tab + (size_t)bid * tab_bytes);
        u32x4*      dst = reinterpret_cast<u32x4*>(s.rt1);
        for (int i = threadIdx.x; i < (int)(tab_bytes / 16); i += PS_NT) {
            dst[i] = src[i];
        }
    }
    else {
        for (int i = threadIdx.x; i < 64; i += PS_NT) {
            s.misc[i] = 0;  // [0] merge groups, [32] ctx arrivals, [33] control pair, [34] streamer barrier, [35] control
        }                   // barrier inside the attention, [36] QKV flushes, [37] streamer waves without QKV tiles, [38] control waves' FFN1 share flushed
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
            r.xoff   = seg ? (H + XPAD) : 0;
            r.xsel   = 0;
            r.rid    = cg;
            r.grp    = g;
            r.pad    = 0;
            s.rt1[j] = r;
        }
        // the K range of mid this workgroup's FFN2 pieces read (piece-major ids: consecutive pieces are adjacent ranges)
        int mid_lo = 0, mid_hi = 0, ctx_lo = 0, ctx_hi = 0;
        {
            bool fb = true, fa = true;
            for (int idx = rB0; idx < rB1; idx++) {
                const int t0 = (idx / NG) * RLb, nt = (KT_b - t0 < RLb) ? KT_b - t0 : RLb;
                const int lo = t0 * TK, hi = (t0 + nt) * TK;
                mid_lo = fb ? lo : (lo < mid_lo ? lo : mid_lo);
                mid_hi = fb ? hi : (hi > mid_hi ? hi : mid_hi);
                fb     = false;
            }
            for (int idx = rA0; idx < rA1; idx++) {
                const int t0 = (idx / NG) * RLa, nt = (KT_a - t0 < RLa) ? KT_a - t0 : RLa;
                const int lo = t0 * TK, hi = (t0 + nt) * TK;
                ctx_lo = fa ? lo : (lo < ctx_lo ? lo : ctx_lo);
                ctx_hi = fa ? hi : (hi > ctx_hi ? hi : ctx_hi);
                fa     = false;
            }
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
                r.xoff       = ctx_off + t0 * TK;  // ctx is staged whole-range relative (xs + ctx_off + k)
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
                r.xoff       = t0 * TK - mid_lo;  // mid is staged relative to the workgroup's own K range
                r.xsel       = 0;
                r.rid        = idx;
            }
            r.grp    = g;
            r.pad    = 0;
            s.rt3[j] = r;
        }
        __syncthreads();
        {
            int T3 = 0;
            for (int j = 0; j < nruns3; j++) {
                T3 += s.rt3[j].nt;
            }
            T3          = ps_rfl(T3);
            const int w = ps_rfl(wid);
            int       nrot1 = 1;
            {   // P1: a streamer wave's QKV slice, then its FFN1 slice; a control wave's FFN1 slice
                int a = 0, b = 0, c, d;
                if (w >= PS_NC) {
                    ps4_p1_ranges(nq * KT, nf * KT, p.plan.cs1, w - PS_NC, a, b, c, d);
                }
                else {
                    ps4_p1_ctrl_range(nq * KT, nf * KT, p.plan.cs1, w, c, d);
                }
                int       eq;
                const int ent = ps4_p1_entries(nq, nf, KT, p.plan.cs1, w, eq);
                nrot1         = (ent + PS_U * PS_NBUF - 1) / (PS_U * PS_NBUF);
                nrot1         = nrot1 < 1 ? 1 : nrot1;
                unsigned* lt  = s.lt1 + (size_t)w * E1;
                unsigned* bt  = s.bt1 + (size_t)w * (E1 / PS_U);
                ps_build_tables<TK>(s.rt1, nq, a, b, lt, bt, eq, 0);
                ps_build_tables<TK>(s.rt1 + nq, nf, c, d, lt + eq, bt + eq / PS_U, nrot1 * PS_U * PS_NBUF - eq, nq);
                if (w >= PS_NC && (threadIdx.x & 63) == 0) {
                    if (eq > 0) {
                        bt[eq / PS_U - 1] |= PS_BT_SIGNAL;  // (the builder's lanes wrote it: same wave, DS order)
                    }
                    else {
                        atomicAdd(&s.misc[37], 1);
                    }
                }
            }
            int tb, te;
            ps_wave_range_w(T3, w, p.plan.wt3, tb, te);
            const int ent3 = ps_wave_entries(nruns3, [&](int j) { return s.rt3[j].nt; }, tb, te);
            int       nrot3 = (ent3 + PS_U * PS_NBUF - 1) / (PS_U * PS_NBUF);
            nrot3           = nrot3 < 1 ? 1 : nrot3;
            ps_build_tables<TK>(s.rt3, nruns3, tb, te, s.lt3 + (size_t)w * E3, s.bt3 + (size_t)w * (E3 / PS_U),
                                nrot3 * PS_U * PS_NBUF);
            if ((threadIdx.x & 63) == 0) {
                s.misc[40 + wid] = nrot1;
                s.misc[48 + wid] = nrot3;
            }
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
    if (p.tab_mode == 1) {
        return;  // the launch that only builds the tables
    }
    PsStage sg1{}, sg3{};
    {
        const int w = ps_rfl(wid);
        sg1.lt      = s.lt1 + (size_t)w * E1;
        sg1.bt      = s.bt1 + (size_t)w * (E1 / PS_U);
        sg3.lt      = s.lt3 + (size_t)w * E3;
        sg3.bt      = s.bt3 + (size_t)w * (E3 / PS_U);
        sg1.nrot    = ps_rfl(s.misc[40 + w]);
        sg3.nrot    = ps_rfl(s.misc[48 + w]);
        sg1.xs0 = sg1.xs1 = H + XPAD;
        sg3.xs0 = sg3.xs1 = 0;  // (one row: the row stride is never used)
    }
    const int mid_lo = ps_rfl(s.misc[56]), mid_hi = ps_rfl(s.misc[57]);
    const int ctx_lo = ps_rfl(s.misc[58]), ctx_hi = ps_rfl(s.misc[59]);
    const int n_noq  = ps_rfl(s.misc[37]);  // streamer waves that own no QKV tile (they never signal)

    auto body = [&](auto role) {
        constexpr bool          CTRL = decltype(role)::value;
        int                     tid  = threadIdx.x;
        PsStream<INT8, M, true> st;
        auto stamp = [&](const int l, const int k) {
            if (p.ts && (tid & 63) == 0) {
                p.ts[(((size_t)blockIdx.x * p.L + l) * PS_NW + (tid >> 6)) * 16 + k] = wall_clock64();
            }
        };
        // barrier of the control waves / of the streamer waves among themselves (LDS counters; DS operations of a wave
        // execute in order)
        int  cb_want = 0, sb_want = 0;
        auto ctrl_barrier = [&]() {
            cb_want += PS_NC;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((tid & 63) == 0) {
                atomicAdd(&s.misc[35], 1);
            }
            for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[35]) < cb_want;) {
                if (++spins > (PS_SPIN << 6)) {
                    __hip_atomic_store(p.err, 11, PS_RLX, PS_AGT);
                    break;
                }
                __builtin_amdgcn_s_sleep(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        auto strm_barrier = [&]() {
            sb_want += PS4_NS;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((tid & 63) == 0) {
                atomicAdd(&s.misc[34], 1);
            }
            for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[34]) < sb_want;) {
                if (++spins > (PS_SPIN << 6)) {
                    __hip_atomic_store(p.err, 12, PS_RLX, PS_AGT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        // ---- per-layer constants, fetched one stage ahead into registers ----
        f16   r_sc1 = (f16)1.f, r_sc3 = (f16)1.f;  // scale of (run tid/16, column tid%16) of P1 / P3
        f16   r_b1 = (f16)0.f, r_bres[2];          // ffn1 bias of the mid epilogue item / residual bias of the merge items
        f16x8 r_ln[4][PS_NLN];                     // ln1_g, ln1_b, ln2_g, ln2_b vectors tid, tid + 512
        auto  load_sc1 = [&](const int l) {
            if constexpr (INT8) {
                if (tid < nruns1 * 16) {
                    PsLayerC&     lw = PS_LAYER(p, l);
                    const RunRec& r  = s.rt1[tid >> 4];
                    r_sc1 = PS_G(f16, r.sel ? lw.s_ffn1 : lw.s_qkv)[r.grp * 16 + (tid & 15)];
                }
            }
        };
        auto load_p1_consts = [&](const int l) {  // LN parameters, ffn1 bias, P3 scales of layer l
            PsLayerC& lw = PS_LAYER(p, l);
#pragma unroll
            for (int k = 0; k < PS_NLN; k++) {
                const int v = tid + k * PS_NT;
                const int o = (v * 8 < H) ? v * 8 : 0;  // (unconditional, clamped: nothing is carried around the layer loop)
                r_ln[0][k]  = *PS_G(f16x8, lw.ln1_g + o);
                r_ln[1][k]  = *PS_G(f16x8, lw.ln1_b + o);
                r_ln[2][k]  = *PS_G(f16x8, lw.ln2_g + o);
                r_ln[3][k]  = *PS_G(f16x8, lw.ln2_b + o);
            }
            if constexpr (!CTRL) {  // the mid epilogue runs on the streamer waves' threads: item = tid - 128
                const int idx = tid - PS_NC * 64;
                r_b1          = (f16)0.f;
                if (idx < nf * 16) {
                    r_b1 = PS_G(f16, lw.b_ffn1)[(f0 + idx / 16) * 16 + (idx & 15)];
                }
            }
            if constexpr (INT8) {
                if (tid < nruns3 * 16) {
                    const RunRec& r = s.rt3[tid >> 4];
                    r_sc3 = PS_G(f16, r.sel ? lw.s_out : lw.s_ffn2)[r.grp * 16 + (tid & 15)];
                }
            }
        };
        // (control waves: nothing is carried through their share of the stream -- the residual bias is fetched behind it, under
        // the barrier's wait, and the next layer's P1 scales right before they are stored: the role is at its register limit)
        auto load_bres = [&](const int l) {  // residual bias of layer l's merge items
            PsLayerC& lw = PS_LAYER(p, l);
            const int nm = s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int t = tid + k * PS_NC * 64;
                r_bres[k]   = (f16)0.f;
                if (t < nm * 16) {
                    r_bres[k] = PS_G(f16, lw.b_res)[s.misc[1 + t / 16] * 16 + (t & 15)];
                }
            }
        };
        // P1 of layer l: scales -> LDS, this wave's partial slots zeroed, constants, stream bound and (streamers) primed
        auto setup_p1 = [&](const int l) {
            PsLayerC& lw = PS_LAYER(p, l);
            if constexpr (CTRL) {
                load_sc1(l);
            }
            if constexpr (INT8) {
                if (tid < nruns1 * 16) {
                    s.rsc[tid] = r_sc1;
                }
            }
            for (int i = tid & 63; i < nruns1 * 16; i += 64) {  // (a wave zeroes the slots it flushes itself)
                s.part[((size_t)(i / 16) * PS_NW + (tid >> 6)) * 16 + i % 16] = 0.f;
            }
            load_p1_consts(l);
            sg1.w0 = reinterpret_cast<const char*>(lw.w_qkv);
            sg1.w1 = reinterpret_cast<const char*>(lw.w_ffn1);
            st.bind(sg1, s.rsc, s.xs, s.part, tid);
            st.sig = &s.misc[36];
            if constexpr (!CTRL) {
                st.prime();  // (the control waves request theirs after the LayerNorm: the gather's polls would return behind it)
            }
        };
        auto setup_p3 = [&](const int l) {
            PsLayerC& lw = PS_LAYER(p, l);
            for (int i = tid & 63; i < nruns3 * 16; i += 64) {
                s.part3[((size_t)(i / 16) * PS_NW + (tid >> 6)) * 16 + i % 16] = 0.f;
            }
            sg3.w0 = reinterpret_cast<const char*>(lw.w_ffn2);
            sg3.w1 = reinterpret_cast<const char*>(lw.w_out);
            st.bind(sg3, s.rsc3, s.xs, s.part3, tid, &s.misc[32], (l - p.l_begin + 1) * PS_NC);
            st.sig = &s.misc[36];
        };

        using Attn = PsAttn<DH, UK, PS_NC * 64>;
        const bool has_item = bid < n_items;
        const int  a_sp = has_item ? bid % p.plan.nsplit : 0;
        const int  a_h = has_item ? (bid / p.plan.nsplit) % p.nh : 0, a_b = has_item ? (bid / p.plan.nsplit) / p.nh : 0;
        const unsigned kbuf_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s.kbuf;

        if constexpr (!CTRL) {
            load_sc1(p.l_begin);
        }
        setup_p1(p.l_begin);
        for (int l = p.l_begin; l < p.l_end; l++) {
            asm volatile("" : "+v"(tid));  // (keeps per-thread address arithmetic inside the layer loop)
            const int      lane = tid & 63, wid = tid >> 6;
            PsLayerC&      lw  = PS_LAYER(p, l);
            const unsigned tag = tag_base + (unsigned)l;
            const int      li  = l - p.l_begin;
            Attn           at;
            stamp(l, 0);
            // =========================== S0: layer input -> xraw (control waves) =================================
            if constexpr (CTRL) {
                if (l == p.l_begin) {
                    for (int i = tid * 8; i < H; i += PS_NC * 64 * 8) {
                        *reinterpret_cast<f16x8*>(s.xraw + i) = *reinterpret_cast<const f16x8*>(p.x_in + i);
                    }
                }
                else {
                    ps_sweep<20>(p.gx, H / 2, tid, PS_NC * 64, tag_base + (unsigned)(l - 1), p.err, 3,
                                [&](const int i, const unsigned v) { reinterpret_cast<unsigned*>(s.xraw)[i] = v; });
                }
            }
            __syncthreads();
            stamp(l, 1);
            // =========================== LN1 / LN2 (layernorm_kernels.cu:157-286 arithmetic) ======================
            {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < PS_NLN; k++) {
                    const int v = tid + k * PS_NT;
                    if (v * 8 < H) {
                        const f16x8 x8 = *reinterpret_cast<const f16x8*>(s.xraw + v * 8);
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float f = (float)x8[e];
                            s0 += f;
                            s1 += f * f;
                        }
                    }
                }
                s0 = wave_sum_dpp(s0);
                s1 = wave_sum_dpp(s1);
                if (lane == 0) {
                    s.red[wid * 2]     = s0;
                    s.red[wid * 2 + 1] = s1;
                }
                if constexpr (INT8) {  // P3's scales: every P3 consumer is at least one barrier away
                    if (tid < nruns3 * 16) {
                        s.rsc3[tid] = r_sc3;
                    }
                }
                __syncthreads();
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int w = 0; w < PS_NW; w++) {
                    a0 += s.red[w * 2];
                    a1 += s.red[w * 2 + 1];
                }
                const float mean = a0 / (float)H;
                const float rstd = rsqrtf(a1 / (float)H - mean * mean + p.eps);
                const f16   mh = (f16)mean, rh = (f16)rstd;
#pragma unroll
                for (int k = 0; k < PS_NLN; k++) {
                    const int v = tid + k * PS_NT;
                    if (v * 8 < H) {
                        const f16x8 x8 = *reinterpret_cast<const f16x8*>(s.xraw + v * 8);
                        f16x8       o1, o2;
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const f16 nrm = (x8[e] - mh) * rh;
                            o1[e]         = (nrm * r_ln[0][k][e]) + r_ln[1][k][e];
                            o2[e]         = (nrm * r_ln[2][k][e]) + r_ln[3][k][e];
                        }
                        *reinterpret_cast<f16x8*>(s.xs + v * 8)              = o1;
                        *reinterpret_cast<f16x8*>(s.xs + (H + XPAD) + v * 8) = o2;
                    }
                }
                stamp(l, 2);
                __syncthreads();
            }
            if constexpr (!CTRL) {
                // =========================== streamer waves: QKV -> FFN1 -> mid hop -> FFN2 -> out-proj ===========
                st.template run<false>();
                stamp(l, 3);
                strm_barrier();  // every streamer wave has flushed its FFN1 runs (and is done with LN2(x) in LDS)
                {   // (the control waves finished their share of FFN1 long ago: they bump this counter behind it)
                    const int want = (li + 1) * PS_NC;
                    for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[38]) < want;) {
                        if (++spins > (PS_SPIN << 6)) {
                            __hip_atomic_store(p.err, 14, PS_RLX, PS_AGT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                {   // mid = gelu(y + b) of this workgroup's FFN1 columns -> granules
                    const int idx = tid - PS_NC * 64;
                    if (idx < nf * 16) {
                        const int j = nq + idx / 16, c = idx & 15;
                        float     v = 0.f;
#pragma unroll
                        for (int w = 0; w < PS_NW; w++) {
                            v += s.part[((size_t)j * PS_NW + w) * 16 + c];
                        }
                        f16 o;
                        if constexpr (INT8) {
                            o = (f16)gelu_f32(v + (float)r_b1);  // epilogue_helpers.h:52-62
                        }
                        else {
                            o = gelu_f16((f16)v + r_b1);  // activation_kernels.cu:401-426
                        }
                        const unsigned b0 = f16_bits(o);
                        const unsigned b1 = next_lane_u32(b0);
                        if ((c & 1) == 0) {
                            st_granule_u32(p.gm + (((size_t)(f0 + idx / 16) * 16 + c) >> 1), tag, b0 | (b1 << 16));
                        }
                    }
                }
                stamp(l, 4);
                setup_p3(l);
                if constexpr (PS4_MID_PRIME >= 1) {
                    st.prime_lo();
                }
                if constexpr (PS4_MID_PRIME >= 2) {
                    st.prime_hi();
                }
                load_sc1(l + 1 < p.l_end ? l + 1 : l);
                // the K range of mid this workgroup's FFN2 pieces read -> LDS (relative to the range's start)
                ps_sweep<4>(p.gm + ((size_t)mid_lo >> 1), (mid_hi - mid_lo) >> 1, tid - PS_NC * 64, PS4_NS * 64, tag, p.err, 6,
                            [&](const int i, const unsigned v) { reinterpret_cast<unsigned*>(s.xs)[i] = v; });
                stamp(l, 7);
                strm_barrier();
                stamp(l, 8);
                if constexpr (PS4_MID_PRIME == 0) {
                    st.prime_lo();
                }
                stamp(l, 9);
                st.template run<(PS4_MID_PRIME < 2)>();
                stamp(l, 10);
            }
            else {
                // =========================== control waves: a share of FFN1, then the attention branch ============
                st.prime_lo();
                st.template run<true>();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) {
                    atomicAdd(&s.misc[38], 1);  // this wave's FFN1 partial sums are flushed
                }
                stamp(l, 8);
                at.issue_ctrl(p, lw, a_h, a_b, a_sp, tid, st, kbuf_lds, has_item);
                // q/k/v of this workgroup's QKV columns: complete when every streamer wave has flushed its QKV slice
                {
                    const int want = (li + 1) * (PS4_NS - n_noq);
                    for (int spins = 0; ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[36]) < want;) {
                        if (++spins > (PS_SPIN << 6)) {
                            __hip_atomic_store(p.err, 13, PS_RLX, PS_AGT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                stamp(l, 3);
                for (int idx = tid; idx < nq * 16; idx += PS_NC * 64) {  // (no bias: the attention adds it, like the reference's)
                    const int j = idx / 16, c = idx & 15;
                    float     v = 0.f;
#pragma unroll
                    for (int w = PS_NC; w < PS_NW; w++) {
                        v += s.part[((size_t)j * PS_NW + w) * 16 + c];
                    }
                    const unsigned b0 = f16_bits((f16)v);
                    const unsigned b1 = next_lane_u32(b0);
                    if ((c & 1) == 0) {
                        st_granule_u32(p.gq + (((size_t)(q0 + j) * 16 + c) >> 1), tag, b0 | (b1 << 16));
                    }
                }
                stamp(l, 4);
                bool live = false;
                u64* gall = p.ga + ((size_t)a_b * p.nh + a_h) * p.plan.nsplit * (DH + 2);
                if (has_item) {
                    at.sweep_qkv(p, s.att, tag, a_h, a_b, tid);
                    stamp(l, 6);
                    ps_wait_vm<0>();  // this wave's K blocks have landed in LDS (the compiler does not know about them)
                    live = at.compute_ctrl(p, lw, s.att, s.kbuf, gall + (size_t)a_sp * (DH + 2), tag, a_h, a_b, tid, st,
                                           ctrl_barrier);
                }
                stamp(l, 5);
                if (has_item && a_sp == 0) {  // (both control waves; `live` is uniform over the pair)
                    if (live) {
                        if constexpr (PS_MERGE_V2 != 0) {
                            ps_attn_merge_sweep(p, s.att, gall, tag, DH, tid);
                            ctrl_barrier();
                        }
                        if (wid == 0) {
                            ps_attn_merge<DH>(p, s.att, gall, tag, a_h, a_b, tid);
                        }
                    }
                    else if (wid == 0) {
                        ps_attn_publish_zero<DH>(p, tag, a_h, a_b, tid);
                    }
                    stamp(l, 13);
                }
                ps_sweep<5>(p.gc + ((size_t)ctx_lo >> 1), (ctx_hi - ctx_lo) >> 1, tid, PS_NC * 64, tag, p.err, 7,
                            [&](const int i, const unsigned v) {
                                reinterpret_cast<unsigned*>(s.xs + ctx_off + ctx_lo)[i] = v;
                            });
                stamp(l, 14);
                if (lane == 0) {
                    atomicAdd(&s.misc[32], 1);  // DS operations of a wave execute in order: the writes above are visible
                }
                setup_p3(l);
                st.prime_lo();
                stamp(l, 9);
                st.template run<true>();
                stamp(l, 10);
                load_bres(l);
            }
            __syncthreads();
            asm volatile("" : "+v"(tid));
            // K pieces -> granules
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int idx = tid + k * PS_NT;
                if (idx < nruns3 * 16) {
                    const int j = idx / 16, r = idx % 16;
                    float     v = 0.f;
#pragma unroll
                    for (int w = 0; w < PS_NW; w++) {
                        v += s.part3[((size_t)j * PS_NW + w) * 16 + r];
                    }
                    st_granule(&p.gp[(size_t)s.rt3[j].rid * 16 + r], tag, v);
                }
            }
            const int  nmerge  = s.misc[0] < PS_MAXMERGE ? s.misc[0] : PS_MAXMERGE;
            const bool last    = (l == p.l_end - 1);
            const int  inplace = (l > 0 && l < p.L - 1) ? 1 : 0;  // GptNeoXDecoder.cc:249-250 -> residual form
            if constexpr (!CTRL) {
                // (no barrier: P1's partial slots and scales were last read before the barrier above, and each wave zeroes
                // its own slots)
#if PS4_COND_TAIL
                if (l + 1 < p.l_end) {  // (after the last layer there is nothing to prefetch: 49 MB of HBM reads per token)
                    setup_p1(l + 1);
                }
#else
                setup_p1(l + 1 < p.l_end ? l + 1 : l);
#endif
            }
            stamp(l, 11);
            if constexpr (CTRL) {
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const int t = tid + k2 * PS_NC * 64;
                    if (t < nmerge * 16) {
                        const int k = t / 16, c = t & 15;
                        const int g = s.misc[1 + k];
                        u64       gv[PS_MAXP];
                        int       spins = 0;
                        for (;;) {
                            bool ok = true;
#pragma unroll
                            for (int q = 0; q < PS_MAXP; q++) {
                                if (q < PA + PB) {
                                    const int rid = (q < PA) ? NG * PB + q * NG + g : (q - PA) * NG + g;
                                    gv[q]         = ld_granule(&p.gp[(size_t)rid * 16 + c]);
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
                        const size_t oidx = (size_t)n;
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
                            ps_tp_exchange(p, tag, oidx, (size_t)H / 2, o, c, last);
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
                setup_p1(l + 1 < p.l_end ? l + 1 : l);
                stamp(l, 12);
                // xraw is rewritten by the next layer's gather: only the two control waves touch it between here and the
                // barrier after that gather, so they synchronise among themselves
                if (lane == 0) {
                    atomicAdd(&s.misc[33], 1);
                }
                const int want = (li + 1) * PS_NC;
                while (ps_rfl(*(const volatile __attribute__((address_space(3))) int*)&s.misc[33]) < want) {
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            else {
                stamp(l, 12);
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
