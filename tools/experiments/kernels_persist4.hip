// Host side and instantiations of the second form of the persistent decode layers (persist4_device.hip.h,
// k_decode_persistent4): plan, kernel table.  Launch and residency go through kernels_persist.hip (PersistPlan::a4).
#include "persist4_device.hip.h"

namespace ftcf {

static size_t ps4_smem_bytes(int H, int xs_halves, int r1, int r3, int dh, int s_max, int nsplit, int e1, int e3)
{
    return (size_t)H * 2 + (size_t)xs_halves * 2 + (size_t)(r1 + r3) * PS_NW * 16 * 4 + ps_att_bytes(dh, s_max, nsplit)
           + 2 * sizeof(RunRec) * PS_RMAX + 2 * PS_RMAX * 16 * 2 + 64 * 4 + 64 * 4 + (size_t)PS_NW * (e1 + e3) * 4
           + (size_t)PS_NW * (e1 + e3) / PS_U * 4 + 1024 + (size_t)PS_UK * PS_NW * 1024;
}

PersistPlan persist_plan4(const PersistPlan& base, int B, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8)
{
    // share of a control wave in the first stream (FFN1 tiles it streams before it turns to the attention), in 1/16 of a
    // streamer wave's: the control waves are idle until q/k/v are complete (~40 % of the stream)
    static const int cs1_env = getenv("FTCF_PERSIST4_CS1") ? atoi(getenv("FTCF_PERSIST4_CS1")) : 8;
    static const int on = getenv("FTCF_PERSIST_A4") ? atoi(getenv("FTCF_PERSIST_A4")) : 0;  // (opt-in: slower than the first form)
    PersistPlan      pl = base;
    if (!on || !pl.ok || B != 1 || pl.uk != PS_UK || pl.a3 || pl.p3l) {
        return base;
    }
    const int TK = int8 ? TILE_K_I8 : TILE_K_F16;
    const int NB = pl.NB, KT = H / TK, KT_b = Il / TK, NG = H / 16, NT0 = 3 * Hl / 16, NF = Il / 16;
    // the fewest KV splits whose chunk fits the 256 keys of one trip (at least two: a one-split request still runs the merge)
    {
        int ns = 1;
        while (ns < pl.nsplit && ((((s_max + ns - 1) / ns) + 15) & ~15) > PS_NW * (64 / (dh / 8)) * PS_UK) {
            ns++;
        }
        pl.nsplit = std::min(pl.nsplit, std::max(ns, 2));
    }
    pl.cs1 = std::max(0, std::min(16, cs1_env));
    int e1 = 0, r1 = 0, r3 = 0, span = 0;
    for (int b = 0; b < NB; b++) {
        const int qb = (b + pl.qrot) % NB;
        const int nq = (int)((long)NT0 * (qb + 1) / NB) - (int)((long)NT0 * qb / NB);
        const int nf = (int)((long)NF * (b + 1) / NB) - (int)((long)NF * b / NB);
        r1           = std::max(r1, nq + nf);
        for (int w = 0; w < PS_NW; w++) {
            int eq;
            e1 = std::max(e1, ps4_p1_entries(nq, nf, KT, pl.cs1, w, eq));
        }
        const int rB0 = (int)((long)NG * pl.PB * b / NB), rB1 = (int)((long)NG * pl.PB * (b + 1) / NB);
        const int rA0 = (int)((long)NG * pl.PA * b / NB), rA1 = (int)((long)NG * pl.PA * (b + 1) / NB);
        r3            = std::max(r3, rB1 - rB0 + rA1 - rA0);
        int lo = 0, hi = 0;
        for (int idx = rB0; idx < rB1; idx++) {
            const int t0 = (idx / NG) * pl.RLb, nt = std::min(pl.RLb, KT_b - t0);
            lo = idx == rB0 ? t0 * TK : std::min(lo, t0 * TK);
            hi = idx == rB0 ? (t0 + nt) * TK : std::max(hi, (t0 + nt) * TK);
        }
        span = std::max(span, hi - lo);
    }
    if (r1 > PS_RMAX || r3 > PS_RMAX || r1 > 32 || r1 < 1 || r3 < 1) {
        return base;  // (the batch descriptor carries the run in five bits)
    }
    const int rot = PS_U * PS_NBUF;
    pl.e1         = std::max(rot, (e1 + rot - 1) / rot * rot);
    pl.r1max      = r1;
    pl.r3max      = r3;
    pl.mid_span   = std::max(span, 2);
    pl.ctx_off    = std::max(pl.mid_span + XPAD, 2 * (H + XPAD));
    pl.xs_halves  = pl.ctx_off + Hl + XPAD;
    if (pl.xs_halves > 0x1ffff) {
        return base;
    }
    pl.smem = ps4_smem_bytes(H, pl.xs_halves, r1, r3, dh, s_max, pl.nsplit, pl.e1, pl.e3);
    if (pl.smem > 160 * 1024 || !persist4_kernel(int8, dh, false, false)) {
        return base;
    }
    pl.a4 = 1;
    return pl;
}

const void* persist4_kernel(bool int8, int dh, bool tp, bool group)
{
#ifdef PS_ONLY_ONE  // (kernel experiments, tools/build_variant4.sh: the headline instantiation alone)
    if (int8 && dh == 128 && !tp && !group) {
        return reinterpret_cast<const void*>(&k_decode_persistent4<true, 128, false, false>);
    }
    return nullptr;
#endif
#define PS4_SEL(I8, D)                                                                                                 \
    if (int8 == I8 && dh == D) {                                                                                       \
        return group ? reinterpret_cast<const void*>(&k_decode_persistent4<I8, D, true, true>)                         \
               : tp  ? reinterpret_cast<const void*>(&k_decode_persistent4<I8, D, true, false>)                        \
                     : reinterpret_cast<const void*>(&k_decode_persistent4<I8, D, false, false>);                      \
    }
#ifndef PS_ONLY_ONE
    PS4_SEL(true, 128)
    PS4_SEL(false, 128)
    PS4_SEL(true, 64)
    PS4_SEL(false, 64)
#endif
#undef PS4_SEL
    return nullptr;
}

}  // namespace ftcf
