// Host side of the persistent decode layers for 3..16 rows (rows_device.hip.h): the static plan, the hand-off region, the launch.
#include "rows_device.hip.h"

#include <stdlib.h>

namespace ftcf {

// k-steps the control wave streams itself out of the n of a workgroup's piece: what seven equal shares leave over, unless that is
// more than about half a streamer wave's share (the control wave starts its stream behind its polls)
static int rows_control_share(const int n)
{
    const int c = n % RW_NS;
    return (c <= n / RW_NS / 2 + 1) ? c : 0;
}

RowsPlan rows_plan(int M, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb)
{
    RowsPlan pl{};
    (void)s_max;
    const int KS = int8 ? TILE_K_I8 : TILE_K_F16;
    if (M < 1 || M > 16 || (dh != 64 && dh != 128) || nh * dh != Hl || H % 64 != 0 || Hl % 64 != 0 || Il % 64 != 0 || H > 8192) {
        return pl;
    }
    const int NB = force_nb > 0 ? std::min(force_nb, num_cu) : num_cu;
    if (NB < 1) {
        return pl;
    }
    const int NGq = 3 * Hl / 16, NGf = Il / 16, NGo = H / 16;
    const int gq = (NGq + NB - 1) / NB, gf = (NGf + NB - 1) / NB;
    if (gq > RW_G || gf > RW_G) {
        return pl;  // more than RW_G column groups per workgroup and pass
    }
    pl.g1 = gq <= 4 ? 4 : 5;
    pl.CB = (NGo + RW_G - 1) / RW_G;
    if (pl.CB > NB) {
        return pl;
    }
    const int KT1 = H / KS, KT2 = Il / KS, KT3 = Hl / KS;
    pl.KP2 = std::max(1, std::min(NB / pl.CB, KT2 / 8));
    pl.KP3 = std::max(1, std::min(std::min(NB / pl.CB, KT3 / 8), 16));
    // attention: one wave per (row, head, split); the splits fill the streamer waves
    const int pairs = M * nh;
    if (pairs > NB * RW_NS) {
        return pl;
    }
    pl.nsplit = std::max(1, std::min(RW_MAXSPLIT, NB * RW_NS / pairs));
    static const int ns_env = getenv("FTCF_ROWS_NSPLIT") ? atoi(getenv("FTCF_ROWS_NSPLIT")) : 0;
    if (ns_env > 0 && ns_env <= RW_MAXSPLIT && pairs * ns_env <= NB * RW_NS) {
        pl.nsplit = ns_env;
    }
    static const int nc_env = getenv("FTCF_ROWS_NC") ? atoi(getenv("FTCF_ROWS_NC")) : -1;
    pl.nc1 = nc_env >= 0 ? nc_env : rows_control_share(KT1);
    pl.nc2 = nc_env >= 0 ? nc_env : rows_control_share((KT2 + pl.KP2 - 1) / pl.KP2);
    pl.nc3 = nc_env >= 0 ? nc_env : rows_control_share((KT3 + pl.KP3 - 1) / pl.KP3);
    // the merger's scratch: (rows of a merger) x (octets of a column block) items
    if (((M + pl.KP3 - 1) / pl.KP3) * RW_G * 2 > 256) {
        return pl;
    }
    pl.NB   = NB;
    pl.smem = (size_t)4 * H * 2 + (size_t)2 * RW_NW * RW_G * 256 * 4 + 32 * 4 + 512 * 4 + (size_t)RW_NW * 3 * dh * 2 + 64;
    if (pl.smem > 160 * 1024) {
        return pl;
    }
    pl.ok = 1;
    return pl;
}

static size_t al256(size_t v)
{
    return (v + 255) & ~(size_t)255;
}
size_t rows_flag_bytes(const RowsPlan& pl, int M, int nh)
{
    // err | fq fm f2 f3 fx [NB each] | fa [M nh nsplit] | fc [M nh]
    return al256(256 + (size_t)(5 * pl.NB + M * nh * pl.nsplit + M * nh) * 4);
}
size_t rows_workspace_bytes(const RowsPlan& pl, int M, int H, int Hl, int Il, int nh, int dh)
{
    size_t b = rows_flag_bytes(pl, M, nh);
    b += al256((size_t)2 * M * pl.CB * 8);                          // stats
    b += 2 * al256((size_t)M * H * 2);                              // xb
    b += al256((size_t)M * 3 * Hl * 2) + al256((size_t)M * Il * 2) + al256((size_t)M * Hl * 2);  // qkv, mid, ctx
    b += al256((size_t)pl.KP2 * M * H * 4) + al256((size_t)pl.KP3 * M * H * 4);                   // p2, p3
    b += al256((size_t)M * nh * pl.nsplit * (dh + RW_PA_PAD) * 4);                                // pa
    return b;
}
void rows_carve(RowsParams& p, void* workspace)
{
    const RowsPlan& pl = p.plan;
    char*           q  = static_cast<char*>(workspace);
    p.err              = reinterpret_cast<int*>(q);
    unsigned* f        = reinterpret_cast<unsigned*>(q + 256);
    p.fq               = f;
    p.fm               = f + pl.NB;
    p.f2               = f + 2 * pl.NB;
    p.f3               = f + 3 * pl.NB;
    p.fx               = f + 4 * pl.NB;
    p.fa               = f + 5 * pl.NB;
    p.fc               = p.fa + (size_t)p.M * p.nh * pl.nsplit;
    q += rows_flag_bytes(pl, p.M, p.nh);
    p.stats = reinterpret_cast<unsigned long long*>(q);
    q += al256((size_t)2 * p.M * pl.CB * 8);
    p.xb[0] = reinterpret_cast<f16*>(q);
    q += al256((size_t)p.M * p.H * 2);
    p.xb[1] = reinterpret_cast<f16*>(q);
    q += al256((size_t)p.M * p.H * 2);
    p.qkv = reinterpret_cast<f16*>(q);
    q += al256((size_t)p.M * 3 * p.Hl * 2);
    p.mid = reinterpret_cast<f16*>(q);
    q += al256((size_t)p.M * p.Il * 2);
    p.ctx = reinterpret_cast<f16*>(q);
    q += al256((size_t)p.M * p.Hl * 2);
    p.p2 = reinterpret_cast<float*>(q);
    q += al256((size_t)pl.KP2 * p.M * p.H * 4);
    p.p3 = reinterpret_cast<float*>(q);
    q += al256((size_t)pl.KP3 * p.M * p.H * 4);
    p.pa = reinterpret_cast<float*>(q);
}

template<bool INT8, int DH, int G1, bool PAGED>
static const void* rw_kernel()
{
    return reinterpret_cast<const void*>(&k_decode_rows<INT8, DH, G1, PAGED>);
}
static const void* rw_kernel_for(bool int8, int dh, int g1, bool paged)
{
#define RW_SEL(I8, D, G, P)                                                                                            \
    if (int8 == I8 && dh == D && g1 == G && paged == P) {                                                              \
        return rw_kernel<I8, D, G, P>();                                                                               \
    }
    RW_SEL(true, 128, 4, false)
#ifndef RW_FEW
    RW_SEL(true, 128, 5, false)
    RW_SEL(true, 64, 4, false)
    RW_SEL(true, 64, 5, false)
    RW_SEL(false, 128, 4, false)
    RW_SEL(false, 128, 5, false)
    RW_SEL(false, 64, 4, false)
    RW_SEL(false, 64, 5, false)
    RW_SEL(true, 128, 4, true)
    RW_SEL(true, 128, 5, true)
    RW_SEL(true, 64, 4, true)
    RW_SEL(true, 64, 5, true)
    RW_SEL(false, 128, 4, true)
    RW_SEL(false, 128, 5, true)
    RW_SEL(false, 64, 4, true)
    RW_SEL(false, 64, 5, true)
#endif
#undef RW_SEL
    return nullptr;
}

// every workgroup of the grid must be resident at once (the hand-offs spin): asked once per plan, like persist_resident()
bool rows_resident(const RowsPlan& pl, bool int8, int dh, int num_cu)
{
    if (!pl.ok) {
        return false;
    }
    for (int paged = 0; paged < 2; paged++) {
        const void* k = rw_kernel_for(int8, dh, pl.g1, paged != 0);
        if (!k) {
            return false;
        }
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, RW_NT, pl.smem) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (per_cu < 1 || (long)per_cu * num_cu < pl.NB) {
            return false;
        }
    }
    return true;
}

void launch_decode_rows(const RowsParams& p, bool int8, hipStream_t s)
{
    FTCF_CHECK_ARG(p.plan.ok && p.M >= 1 && p.M <= 16, "rows kernel: shape not eligible");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh && (p.rot == 0 || p.rot_table != nullptr), "bad rotary configuration");
    FTCF_CHECK_ARG(p.L <= 255 && p.l_begin >= 0 && p.l_begin < p.l_end && p.l_end <= p.L, "bad layer range");
    const void* k = rw_kernel_for(int8, p.dh, p.plan.g1, p.page_table != nullptr);
    FTCF_CHECK_ARG(k != nullptr, "rows kernel: no instantiation for this shape");
    RowsParams pp     = p;
    void*      args[] = {&pp};
    FTCF_HIP_CHECK(hipLaunchKernel(k, dim3(p.plan.NB), dim3(RW_NT), args, p.plan.smem, s));
}

}  // namespace ftcf
