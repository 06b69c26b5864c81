// Host side of the persistent decode layers (device code: persist_device.hip.h): plan, residency check, launchers of the
// TP = 1 instantiations.  The tensor-parallel instantiations live in kernels_persist_tp.hip (built in parallel).
#include "persist_device.hip.h"

// A3 / P3L / the LM-head tail are measured experiments that did not pay (profiles/r03_notes.md): instantiated only with
// -DPS_EXPERIMENTS, never selected otherwise
#ifdef PS_EXPERIMENTS
#define PS_EXPERIMENTS_ON 1
#else
#define PS_EXPERIMENTS_ON 0
#endif

namespace ftcf {

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static size_t ps_smem_bytes(int M, int H, int xs_halves, int dh, int s_max, int nsplit, int e1, int e3, bool a3)
{
    // (a3: the K rows of the KV split, 1 KiB aligned behind everything else)
    return (size_t)M * H * 2 + (size_t)xs_halves * 2 + (size_t)PS_RMAX * PS_NW * M * 16 * 4 + ps_att_bytes(dh, s_max, nsplit)
           + 2 * sizeof(RunRec) * PS_RMAX + 2 * PS_RMAX * 16 * 2 + 64 * 4 + 64 * 4 + (size_t)PS_NW * (e1 + e3) * 4
           + (size_t)PS_NW * (e1 + e3) / PS_U * 4 + (a3 ? 1024 + (size_t)PS_UK * PS_NW * 1024 : 0)
           + ((M == 1 && !a3 && PS_PART3) ? (size_t)PS_RMAX * PS_NW * M * 16 * 4 : 0);
}

PersistPlan persist_plan(int B, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb,
                         int cs1, int cs3, bool allow_a3)
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
        static const int few = getenv("FTCF_PERSIST_FEW_SPLITS") ? atoi(getenv("FTCF_PERSIST_FEW_SPLITS")) : 1;
        int ns = 1;
        while (ns < nsplit && ((((s_max + ns - 1) / ns) + 15) & ~15) > PS_NW * (64 / (dh / 8)) * PS_UK) {
            ns++;
        }
        if (few && ((((s_max + ns - 1) / ns) + 15) & ~15) <= PS_NW * (64 / (dh / 8)) * PS_UK) {
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
    static const int qrot_env = getenv("FTCF_PERSIST_QROT") ? atoi(getenv("FTCF_PERSIST_QROT")) : 1;
    const int        qrot_plan = ((qrot_env % NB) + NB) % NB;
    // per-wave shares of the two streams (1/16 of a nominal share): the control waves' from cs1 / cs3, the streamer waves' from
    // FTCF_PERSIST_WT1 / WT3 = six comma-separated integers for waves 2..7 (default 16 each)
    int  wt1[8], wt3[8];
    auto fill_wt = [](int* wt, const int cs, const char* env) {
        for (int i = 0; i < PS_NW; i++) {
            wt[i] = i < PS_NC ? cs : 16;
        }
        if (const char* e = getenv(env)) {
            int v[PS_NW - PS_NC];
            if (sscanf(e, "%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6) {
                for (int i = PS_NC; i < PS_NW; i++) {
                    wt[i] = std::max(1, std::min(64, v[i - PS_NC]));
                }
            }
        }
    };
    // tile-table entries per wave: the exact maximum over workgroups and waves, in whole rotations (e3c: control waves, P3)
    int  e1 = 0, e3 = 0, e3c = 0;
    auto count_entries = [&](const int c1, const int c3) {
        fill_wt(wt1, c1, "FTCF_PERSIST_WT1");
        fill_wt(wt3, c3, "FTCF_PERSIST_WT3");
        e1 = e3 = e3c = 0;
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
            int T3 = 0;
            for (int j = 0; j < nB + nA; j++) {
                T3 += nt3(j);
            }
            for (int w = 0; w < PS_NW; w++) {
                int tb, te;
                if (PS_QKV_EARLY != 0) {  // (the wave's QKV slice, then its FFN1 slice: persist_device.hip.h ps_wave_range2_w)
                    const int qbk = (b + qrot_plan) % NB;
                    const int nqb = (int)((long)NT0h * (qbk + 1) / NB) - (int)((long)NT0h * qbk / NB);
                    const int nfb = (int)((long)NFh * (b + 1) / NB) - (int)((long)NFh * b / NB);
                    int       qa, qz, fa, fz;
                    ps_wave_range2_w(nqb * KT, nfb * KT, w, wt1, qa, qz, fa, fz);
                    e1 = std::max(e1, ps_wave_entries(nqb, nt1, qa, qz) + ps_wave_entries(nfb, nt1, fa, fz));
                }
                else {
                    ps_wave_range_w(nr1 * KT, w, wt1, tb, te);
                    e1 = std::max(e1, ps_wave_entries(nr1, nt1, tb, te));
                }
                ps_wave_range_w(T3, w, wt3, tb, te);
                const int en = ps_wave_entries(nB + nA, nt3, tb, te);
                e3           = std::max(e3, en);
                if (w < PS_NC) {
                    e3c = std::max(e3c, en);
                }
            }
        }
    };
    count_entries(cs1, cs3);
    // P3L (one row, not A3): the control waves' P3 share -- the ctx-dependent out-proj pieces at the end of the tile space --
    // is requested into LDS (64 KiB, 32 tiles per control wave) by LDS-DMA during the attention, when the HBM has nothing else
    // to do, and consumed from there when ctx arrives: the share is shrunk (cs3) until it fits
    static const int p3l_env = PS_EXPERIMENTS_ON && getenv("FTCF_PERSIST_P3L") ? atoi(getenv("FTCF_PERSIST_P3L")) : 0;
    pl.p3l = 0;
    if (p3l_env != 0 && allow_a3 && M == 1 && pl.uk == PS_UK) {
        int c3 = cs3;
        while (c3 > 1 && e3c > PS_U * PS_NBUF) {
            c3--;
            count_entries(cs1, c3);
        }
        if (e3c <= PS_U * PS_NBUF) {
            cs3    = c3;
            pl.p3l = 1;
        }
        else {
            count_entries(cs1, cs3);
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
    for (int i = 0; i < PS_NW; i++) {
        pl.wt1[i] = wt1[i];
        pl.wt3[i] = wt3[i];
    }
    pl.xs_halves  = M * std::max(2 * (H + XPAD), Il + Hl + 2 * XPAD);
    if (pl.xs_halves > 0x1ffff) {
        return pl;
    }
    // A3 (the attention on the control waves, K rows by LDS-DMA): one row, the short attention form, and 64 KiB more LDS
    static const int a3_env = PS_EXPERIMENTS_ON && getenv("FTCF_PERSIST_A3") ? atoi(getenv("FTCF_PERSIST_A3")) : 0;
    pl.a3 = (a3_env != 0 && !pl.p3l && allow_a3 && M == 1 && pl.uk == PS_UK
             && ps_smem_bytes(M, H, pl.xs_halves, dh, s_max, nsplit, pl.e1, pl.e3, true) <= 160 * 1024) ? 1 : 0;
    if (pl.p3l && ps_smem_bytes(M, H, pl.xs_halves, dh, s_max, nsplit, pl.e1, pl.e3, true) > 160 * 1024) {
        return pl;  // (cannot happen for a shape the plain kernel fits with 64 KiB to spare; the caller retries without P3L)
    }
    pl.smem = ps_smem_bytes(M, H, pl.xs_halves, dh, s_max, nsplit, pl.e1, pl.e3, pl.a3 != 0 || pl.p3l != 0);
    if (pl.smem > 160 * 1024) {
        return pl;
    }
    // (round 3, on the kernel with the DPP reductions: the light share on the odd XCDs -- rotation 1 or 3 -- measures 2468-2469 us per
    // launch against 2480-2486 with it on the even ones, in two builds: profiles/r03_notes.md)
    static const int qrot = getenv("FTCF_PERSIST_QROT") ? atoi(getenv("FTCF_PERSIST_QROT")) : 1;
    pl.qrot = ((qrot % NB) + NB) % NB;
    pl.ok = 1;
    return pl;
}

bool persist_lm_tail_built()
{
    return PS_LM_CODE != 0;
}

size_t persist_table_bytes(const PersistPlan& pl)
{
    // rt1 | rt3 | rsc | rsc3 | red | misc | lt1 | lt3 | bt1 | bt3 (persist_device.hip.h, the carve of the kernel's LDS)
    return 2 * sizeof(RunRec) * PS_RMAX + 2 * PS_RMAX * 16 * 2 + 64 * 4 + 64 * 4 + (size_t)PS_NW * (pl.e1 + pl.e3) * 4
           + (size_t)PS_NW * (pl.e1 + pl.e3) / PS_U * 4;
}

template<bool INT8, int M, int DH, int UK, bool A3 = false, bool P3L = false>
static const void* ps_kernel()
{
    return reinterpret_cast<const void*>(&k_decode_persistent<INT8, M, DH, UK, false, false, A3, P3L>);
}
static const void* ps_kernel_for(bool int8, int M, int dh, int uk, bool a3, bool p3l = false)
{
#define PS_SEL(I8, MM, D)                                                                                              \
    if (int8 == I8 && M == MM && dh == D) {                                                                            \
        if constexpr (MM == 1 && PS_EXPERIMENTS_ON) {                                                                  \
            if (p3l && uk == PS_UK) {                                                                                  \
                return ps_kernel<I8, MM, D, PS_UK, false, true>();                                                     \
            }                                                                                                          \
            if (a3 && uk == PS_UK) {                                                                                   \
                return ps_kernel<I8, MM, D, PS_UK, true>();                                                            \
            }                                                                                                          \
        }                                                                                                              \
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
#ifdef PS_EXPERIMENTS
    if (pl.a4) {
        return ps_kernel_resident(persist4_kernel(int8, dh, true, true), pl, num_cu, (long)pl.NB * world);
    }
#endif
    return ps_kernel_resident(persist_tp_kernel(int8, M, dh, pl.uk, true), pl, num_cu, (long)pl.NB * world);
}
bool persist_resident(const PersistPlan& pl, bool int8, int M, int dh, int num_cu, int tp)
{
#ifdef PS_EXPERIMENTS
    if (pl.a4) {
        return ps_kernel_resident(persist4_kernel(int8, dh, tp > 1, false), pl, num_cu, pl.NB);
    }
#endif
    if (tp > 1) {
        return ps_kernel_resident(persist_tp_kernel(int8, M, dh, pl.uk, false), pl, num_cu, pl.NB);
    }
    return ps_kernel_resident(ps_kernel_for(int8, M, dh, pl.uk, pl.a3 != 0, pl.p3l != 0), pl, num_cu, pl.NB);
}

void launch_decode_persistent(const PersistParams& p, bool int8, hipStream_t s)
{
    FTCF_CHECK_ARG(p.plan.ok && p.B >= 1 && p.B <= 2, "persistent decode: shape not eligible");
    FTCF_CHECK_ARG(p.dh == 64 || p.dh == 128, "size_per_head must be 64 or 128");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh && (p.rot == 0 || p.rot_table != nullptr), "bad rotary configuration");
    FTCF_CHECK_ARG(p.L <= 255, "at most 255 layers");
    const void* k = p.tp > 1 ? persist_tp_kernel(int8, p.B, p.dh, p.plan.uk, false)
                             : ps_kernel_for(int8, p.B, p.dh, p.plan.uk, p.plan.a3 != 0, p.plan.p3l != 0);
#ifdef PS_EXPERIMENTS
    if (p.plan.a4) {
        k = persist4_kernel(int8, p.dh, p.tp > 1, false);
    }
#endif
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
    const void*          k = persist_tp_kernel(int8, p.B, p.dh, p.plan.uk, true);
#ifdef PS_EXPERIMENTS
    if (p.plan.a4) {
        k = persist4_kernel(int8, p.dh, true, true);
    }
#endif
    FTCF_CHECK_ARG(k != nullptr && p.plan.ok && p.plan.NB == g.nb, "persistent decode: no group kernel for this shape");
    PersistGroupParams gg     = g;
    void*              args[] = {&gg};
    FTCF_HIP_CHECK(hipLaunchKernel(k, dim3(g.nb * g.world), dim3(PS_NT), args, p.plan.smem, s));
}

}  // namespace ftcf
