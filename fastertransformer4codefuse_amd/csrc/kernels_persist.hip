// Host side of the persistent decode layers (device code: persist_device.hip.h): plan, residency check, launchers of the
// TP = 1 instantiations.  The tensor-parallel instantiations live in kernels_persist_tp.hip (built in parallel).
#include "persist_device.hip.h"

namespace ftcf {

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static size_t ps_smem_bytes(int M, int H, int xs_halves, int dh, int s_max, int nsplit, int e1, int e3)
{
    return (size_t)M * H * 2 + (size_t)xs_halves * 2 + (size_t)PS_RMAX * PS_NW * M * 16 * 4 + ps_att_bytes(dh, s_max, nsplit)
           + 2 * sizeof(RunRec) * PS_RMAX + PS_RMAX * 16 * 2 + 64 * 4 + 64 * 4 + PS_OT_N * 4 + (size_t)PS_NW * (e1 + e3) * 4
           + (size_t)PS_NW * (e1 + e3) / PS_U * 4;
}

PersistPlan persist_plan(int B, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb,
                         int cs1, int cs3, int own, int L)
{
    PersistPlan pl{};
    const int   M  = B;
    const int   TK = int8 ? TILE_K_I8 : TILE_K_F16;
    if (M < 1 || M > 2 || (dh != 64 && dh != 128) || H % TK || Hl % TK || Il % TK || H % 16 || Hl % 16 || Il % 16) {
        return pl;
    }
    if (cs1 < 1 || cs1 > 16 || cs3 < 1 || cs3 > 16 || H > PS_NLN * PS_NT * 8) {
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
    // The attention of a split costs the same for 16 keys and for 256 (one trip of fixed size), its partial costs the merge
    // 130 granules to sweep and a term to add: with fewer heads than workgroups (tensor-parallel shards: 5 heads per rank at
    // TP = 8) take the FEWEST splits whose chunk fits one short trip, at least two (so that small test models still merge).
    // 13B at TP = 1: 6 either way.  TP = 8 shard: 16 -> 6 splits, "partials swept -> merged" 5.6 -> 2 us per layer.
    {
        int ns = 1;
        while (ns < nsplit && ((((s_max + ns - 1) / ns) + 15) & ~15) > PS_NW * (64 / (dh / 8)) * PS_UK) {
            ns++;
        }
        if (((((s_max + ns - 1) / ns) + 15) & ~15) <= PS_NW * (64 / (dh / 8)) * PS_UK) {
            nsplit = std::min(nsplit, std::max(ns, 2));
        }
    }
    const int chunk = ((((s_max + nsplit - 1) / nsplit) + 15) & ~15);
    if (chunk > PS_NW * (64 / (dh / 8)) * PS_UK_LONG) {
        return pl;  // the K/V rows of a split must fit the registers of one trip
    }
    pl.uk = chunk > PS_NW * (64 / (dh / 8)) * PS_UK ? PS_UK_LONG : PS_UK;
    // K pieces: best balanced tile count per workgroup
    double best = 1e30;
    long   t3max = 0;
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
                t3max  = mx;
            }
        }
    }
    if (best > 1e29) {
        return pl;
    }
    // P3 in the own-group layout (persist_device.hip.h "own-group layout"): whole column groups per workgroup, one hop at the
    // layer boundary instead of two; where the shape does not divide that way the K pieces above stay
    // own = 2 (auto): where it measured faster -- one rank's shard of TP 2 / 4 at the 13B shape +2.7 / +2.3 %, one GPU +0.3 %; a
    // TP 8 shard (63 tiles per workgroup: the exchange's round trip is the boundary, the wide sweeps only add) -2 %, so not below
    // 100 tiles of [FFN2 | out-proj] per workgroup (profiles/r06_notes.md)
    pl.own = 0;
    if (own == 2 && (long)NG * (KT_a + KT_b) < 100L * NB) {
        own = 0;
    }
    if (own != 0 && L > 0 && ps_own_ok(NB, NG, M)) {
        RunRec tmp[PS_RMAX];
        const int cs = cs3;
        bool ok = cs >= 1;
        for (int b = 0; b < NB && ok; b++) {
            ok = ps_own_remote(b, NB, NG, KT_a, KT_b, TK, M, Il, cs, tmp, nullptr, 0) <= PS_OWN_MAXR;
        }
        pl.own = ok ? 1 : 0;
    }
    // tile-table entries per wave: the exact maximum over workgroups and waves, in whole rotations (e3c: control waves, P3)
    int  e1 = 0, e3 = 0;
    {
        for (int b = 0; b < NB; b++) {
            const int nr1 = (int)((long)NT0h * (b + 1) / NB) - (int)((long)NT0h * b / NB)
                            + (int)((long)NFh * (b + 1) / NB) - (int)((long)NFh * b / NB);
            const int rB0 = (int)((long)NG * pl.PB * b / NB), rB1 = (int)((long)NG * pl.PB * (b + 1) / NB);
            const int rA0 = (int)((long)NG * pl.PA * b / NB), rA1 = (int)((long)NG * pl.PA * (b + 1) / NB);
            const int nB = rB1 - rB0, nA = rA1 - rA0;
            auto nt1 = [&](int) { return KT; };
            auto nt3 = [&](int j) {
                if (j < nB) {
                    const int t0 = ((rB0 + j) / NG) * pl.RLb;
                    return std::min(pl.RLb, KT_b - t0);
                }
                const int t0 = ((rA0 + j - nB) / NG) * pl.RLa;
                return std::min(pl.RLa, KT_a - t0);
            };
            RunRec    own_r[PS_RMAX];
            const int own_n = pl.own ? ps_own_runs(b, NB, NG, KT_a, KT_b, TK, M, Il, own_r) : 0;
            auto      nt3o  = [&](int j) { return own_r[j].nt; };
            int       T3    = 0;
            for (int j = 0; j < (pl.own ? own_n : nB + nA); j++) {
                T3 += pl.own ? nt3o(j) : nt3(j);
            }
            for (int w = 0; w < PS_NW; w++) {
                int tb, te;
                ps_wave_range(nr1 * KT, w, cs1, tb, te);
                e1 = std::max(e1, ps_wave_entries(nr1, nt1, tb, te));
                ps_wave_range(T3, w, cs3, tb, te);
                e3 = std::max(e3, pl.own ? ps_wave_entries(own_n, nt3o, tb, te) : ps_wave_entries(nB + nA, nt3, tb, te));
            }
        }
    }
    const int rot = PS_U * PS_NBUF;
    pl.e1         = std::max(rot, (e1 + rot - 1) / rot * rot);
    pl.e3         = std::max(rot, (e3 + rot - 1) / rot * rot);
    (void)t3max;
    pl.NB         = NB;
    pl.nsplit     = nsplit;
    pl.cs1        = cs1;
    pl.cs3        = cs3;
    pl.xs_halves  = M * std::max(2 * (H + XPAD), Il + Hl + 2 * XPAD);
    if (pl.xs_halves > 0x1ffff) {
        return pl;
    }
    pl.smem = ps_smem_bytes(M, H, pl.xs_halves, dh, s_max, nsplit, pl.e1, pl.e3);
    if (pl.own) {  // the kernel's LDS copy of the per-layer pointer table
        const size_t with = pl.smem + (((size_t)L * sizeof(PersistLayer) + 15) & ~(size_t)15);
        if (with > 160 * 1024) {
            return persist_plan(B, H, Hl, Il, nh, dh, s_max, int8, num_cu, force_nb, cs1, cs3, 0, L);
        }
        pl.smem = with;
    }
    if (pl.smem > 160 * 1024) {
        return pl;
    }
    // (round 3, on the kernel with the DPP reductions: the light share on the odd XCDs -- rotation 1 or 3 -- measures 2468-2469 us per
    // launch against 2480-2486 with it on the even ones, in two builds: profiles/r03_notes.md)
    pl.qrot = 1 % NB;
    // K-piece partials: [NG * (PA + PB)][M * 16] ; own-group layout: [NS * PC pieces][2 matrices][PS_NW waves][M * 16]
    {
        const int no = NG / NB, NS = NG - no * NB;
        pl.gp_n      = pl.own ? (long)(NS > 0 ? NB : 0) * 2 * PS_NW * M * 16 + 16 : (long)NG * (pl.PA + pl.PB) * M * 16;
    }
    pl.ok = 1;
    return pl;
}

size_t persist_table_bytes(const PersistPlan& pl)
{
    // rt1 | rt3 | rsc | red | misc | lt1 | lt3 | bt1 | bt3 (persist_device.hip.h, the carve of the kernel's LDS)
    return 2 * sizeof(RunRec) * PS_RMAX + PS_RMAX * 16 * 2 + 64 * 4 + 64 * 4 + PS_OT_N * 4 + (size_t)PS_NW * (pl.e1 + pl.e3) * 4
           + (size_t)PS_NW * (pl.e1 + pl.e3) / PS_U * 4;
}

template<bool INT8, int M, int DH, int UK>
static const void* ps_kernel()
{
    return reinterpret_cast<const void*>(&k_decode_persistent<INT8, M, DH, UK, false, false>);
}
static const void* ps_kernel_for(bool int8, int M, int dh, int uk, int own)
{
    if (own) {
        return persist_own_kernel(int8, M, dh, uk);
    }
#define PS_SEL(I8, MM, D)                                                                                              \
    if (int8 == I8 && M == MM && dh == D) {                                                                            \
        return uk == PS_UK_LONG ? ps_kernel<I8, MM, D, PS_UK_LONG>() : ps_kernel<I8, MM, D, PS_UK>();                  \
    }
    PS_SEL(true, 1, 128)
#ifndef PS_ONLY_ONE  // (tools/build_variant.sh: kernel-variant builds instantiate the 13B int8 one-row form only)
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
static bool ps_kernel_resident(const void* k, const PersistPlan& pl, int num_cu, long grid)
{
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
    return per_cu >= 1 && (long)per_cu * num_cu >= grid;
}
bool persist_group_resident(const PersistPlan& pl, bool int8, int M, int dh, int num_cu, int world)
{
    return ps_kernel_resident(persist_tp_kernel(int8, M, dh, pl.uk, true, pl.own), pl, num_cu, (long)pl.NB * world);
}
bool persist_resident(const PersistPlan& pl, bool int8, int M, int dh, int num_cu, int tp)
{
    if (tp > 1) {
        return ps_kernel_resident(persist_tp_kernel(int8, M, dh, pl.uk, false, pl.own), pl, num_cu, pl.NB);
    }
    return ps_kernel_resident(ps_kernel_for(int8, M, dh, pl.uk, pl.own), pl, num_cu, pl.NB);
}

void launch_decode_persistent(const PersistParams& p, bool int8, hipStream_t s)
{
    FTCF_CHECK_ARG(p.plan.ok && p.B >= 1 && p.B <= 2, "persistent decode: shape not eligible");
    FTCF_CHECK_ARG(p.dh == 64 || p.dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh && (p.rot == 0 || p.rot_table != nullptr), "bad rotary configuration");
    FTCF_CHECK_ARG(p.L <= 255, "at most 255 layers");
    const void* k = p.tp > 1 ? persist_tp_kernel(int8, p.B, p.dh, p.plan.uk, false, p.plan.own)
                             : ps_kernel_for(int8, p.B, p.dh, p.plan.uk, p.plan.own);
    FTCF_CHECK_ARG(k != nullptr, "persistent decode: no kernel for this shape");
    FTCF_CHECK_ARG(p.tp >= 1 && p.tp <= PERSIST_MAX_TP && p.tp_rank >= 0 && p.tp_rank < p.tp, "bad tensor-parallel rank");
    PersistParams pp     = p;
    void*         args[] = {&pp};
    FTCF_HIP_CHECK(hipLaunchKernel(k, dim3(p.plan.NB), dim3(PS_NT), args, p.plan.smem, s));
}

void launch_decode_persistent_group(const PersistGroupParams& g, bool int8, hipStream_t s)
{
    FTCF_CHECK_ARG(g.world >= 2 && g.world <= PERSIST_MAX_TP && g.nb >= 1, "bad local group");
    const PersistParams& p = g.p[0];
    const void*          k = persist_tp_kernel(int8, p.B, p.dh, p.plan.uk, true, p.plan.own);
    FTCF_CHECK_ARG(k != nullptr && p.plan.ok && p.plan.NB == g.nb, "persistent decode: no group kernel for this shape");
    PersistGroupParams gg     = g;
    void*              args[] = {&gg};
    FTCF_HIP_CHECK(hipLaunchKernel(k, dim3(g.nb * g.world), dim3(PS_NT), args, p.plan.smem, s));
}

}  // namespace ftcf
