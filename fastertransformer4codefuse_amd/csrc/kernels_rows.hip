// Host side of the persistent decode layers for 3..16 rows (rows_device.hip.h): the static plan, the hand-off region, the launch.
#include "rows_device.hip.h"

#include <stdio.h>
#include <stdlib.h>

namespace ftcf {

RowsPlan rows_plan(int M, int H, int Hl, int Il, int nh, int dh, int s_max, bool int8, int num_cu, int force_nb)
{
    RowsPlan pl{};
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
    const int KS  = int8 ? TILE_K_I8 : TILE_K_F16;
    const int KT2 = Il / KS, KT3 = Hl / KS;
    pl.KP2 = std::max(1, std::min(NB / pl.CB, KT2 / 8));
    pl.KP3 = std::max(1, std::min(std::min(NB / pl.CB, KT3 / 8), 16));
    // (32-bit byte offsets behind the kernel's descriptors: a weight matrix, a layer's K cache)
    if ((size_t)H * std::max(3 * Hl, Il) * 2 >= ((size_t)1 << 31) || (size_t)M * nh * (size_t)s_max * dh * 2 >= ((size_t)1 << 31)) {
        return pl;
    }
    // attention: the (row, head) pairs are dealt round robin, a pair stays inside its workgroup
    {
        const int full = M * nh / NB, rem = M * nh % NB;
        // a leftover pair is shared by FX workgroups only when EVERY pair is a leftover pair (fewer pairs than workgroups: small
        // batches, tensor-parallel shards): then all pairs have the same reduction tree.  With whole pairs next to shared ones
        // (bs 16 at 40 heads: 2.5 pairs per workgroup) the shared pairs' tree would differ from the whole pairs' and identical
        // rows of a batch would stop being bit-identical; sharing them anyway balances the attention stream and was measured
        // neutral on the step (profiles/r05_notes.md)
        pl.FX = (rem > 0 && full == 0) ? std::max(1, std::min(NB / rem, 8)) : 1;
        pl.U = full + (rem > 0 ? 1 : 0);
    }
    if (pl.U > RW_UMAX) {
        return pl;
    }
    // the layer boundary: merger (row, column range); a range is a whole number of 8-column pieces
    if (M > NB) {
        return pl;
    }
    pl.CR = std::max(1, std::min(NB / M, H / 64));
    while (H % pl.CR != 0 || (H / pl.CR) % 8 != 0) {
        pl.CR--;
    }
    pl.cw = H / pl.CR;
    if (M * pl.CR > 256) {
        pl.CR = 256 / M;  // (the control wave sweeps at most 256 granules)
        while (H % pl.CR != 0 || (H / pl.CR) % 8 != 0) {
            pl.CR--;
        }
        pl.cw = H / pl.CR;
    }
    // K shares of the streamer waves: equal (unequal weights measured no better: profiles/r05_notes.md)
    {
        const int w[RW_NS] = {16, 16, 16, 16, 16, 16, 16};
        pl.wcum[0] = 0;
        for (int i = 0; i < RW_NS; i++) {
            pl.wcum[i + 1] = pl.wcum[i] + w[i];
        }
    }
    pl.NB = NB;
    const int Hp = (H + 511) & ~511;
    pl.smem      = (size_t)4 * Hp * 2 + (size_t)2 * RW_NS * RW_G * 256 * 4 + (size_t)RW_UMAX * RW_NS * (dh + RW_PA) * 4
              + (size_t)RW_NW * RW_UMAX * 2 * dh * 2 + 32 * 4 + 512 * 4 + RW_UMAX * 8 * 4 + 256 * 4 + 64;
    if (pl.smem > 160 * 1024) {
        return pl;
    }
    pl.ok = 1;
    return pl;
}

// keys of an attention block of the rows kernel (RW_KVB wave-loads of 64 / (dh / 8) keys): a paged K/V block must lie inside one page
int rows_paged_block(int dh)
{
    return RW_KVB * (64 / (dh / 8));
}

static size_t al256(size_t v)
{
    return (v + 255) & ~(size_t)255;
}
size_t rows_flag_bytes(const RowsPlan& pl, int M, int nh)
{
    (void)nh;
    // err | fq fm fc f2 f3 fa [NB each] | x' granules [2][M * CR] of 16 bytes
    return al256(256 + (size_t)6 * pl.NB * 4) + al256((size_t)2 * M * pl.CR * 16);
}
size_t rows_workspace_bytes(const RowsPlan& pl, int M, int H, int Hl, int Il, int nh, int dh)
{
    size_t b = rows_flag_bytes(pl, M, nh);
    b += al256((size_t)pl.NB * (dh + RW_PA) * 4);  // pa
    b += 2 * al256((size_t)M * H * 2);                                                            // xb
    b += al256((size_t)M * 3 * Hl * 2) + al256((size_t)M * Il * 2) + al256((size_t)M * Hl * 2);  // qkv, mid, ctx
    b += al256((size_t)pl.KP2 * M * H * 4) + al256((size_t)pl.KP3 * M * H * 4);                   // p2, p3
    return b;
}
void rows_carve(RowsParams& p, void* workspace)
{
    const RowsPlan& pl = p.plan;
    p.ws               = static_cast<char*>(workspace);
    p.err              = reinterpret_cast<int*>(workspace);
    size_t o           = 256;
    p.o_fq             = (unsigned)o;
    p.o_fm             = (unsigned)(o + (size_t)pl.NB * 4);
    p.o_fc             = (unsigned)(o + (size_t)2 * pl.NB * 4);
    p.o_f2             = (unsigned)(o + (size_t)3 * pl.NB * 4);
    p.o_f3             = (unsigned)(o + (size_t)4 * pl.NB * 4);
    p.o_fa             = (unsigned)(o + (size_t)5 * pl.NB * 4);
    o                  = al256(256 + (size_t)6 * pl.NB * 4);
    p.o_xs             = (unsigned)o;
    o += al256((size_t)2 * p.M * pl.CR * 16);
    p.o_xb[0] = (unsigned)o;
    o += al256((size_t)p.M * p.H * 2);
    p.o_xb[1] = (unsigned)o;
    o += al256((size_t)p.M * p.H * 2);
    p.o_qkv = (unsigned)o;
    o += al256((size_t)p.M * 3 * p.Hl * 2);
    p.o_mid = (unsigned)o;
    o += al256((size_t)p.M * p.Il * 2);
    p.o_ctx = (unsigned)o;
    o += al256((size_t)p.M * p.Hl * 2);
    p.o_p2 = (unsigned)o;
    o += al256((size_t)pl.KP2 * p.M * p.H * 4);
    p.o_p3 = (unsigned)o;
    o += al256((size_t)pl.KP3 * p.M * p.H * 4);
    p.o_pa = (unsigned)o;
    o += al256((size_t)pl.NB * (p.dh + RW_PA) * 4);
    p.ws_bytes = (unsigned)o;
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
    // paged K/V (the continuous-batching front end)
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
bool rows_resident(const RowsPlan& pl, bool int8, int dh, int num_cu, bool paged)
{
    if (!pl.ok) {
        return false;
    }
    const void* k = rw_kernel_for(int8, dh, pl.g1, paged);
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
    return per_cu >= 1 && (long)per_cu * num_cu >= pl.NB;
}

void launch_decode_rows(const RowsParams& p, bool int8, hipStream_t s)
{
    FTCF_CHECK_ARG(p.plan.ok && p.M >= 1 && p.M <= 16, "rows kernel: shape not eligible");
    FTCF_CHECK_ARG(p.rot % 2 == 0 && p.rot <= p.dh && (p.rot == 0 || p.rot_table != nullptr), "bad rotary configuration");
    FTCF_CHECK_ARG(p.L <= 255 && p.l_begin >= 0 && p.l_begin < p.l_end && p.l_end <= p.L, "bad layer range");
    FTCF_CHECK_ARG(p.page_table == nullptr || (p.page_tokens > 0 && p.page_tokens % rows_paged_block(p.dh) == 0),
                   "rows kernel: page_tokens must be a multiple of the attention block (16 keys at size_per_head 128, 32 at 64)");
    const void* k = rw_kernel_for(int8, p.dh, p.plan.g1, p.page_table != nullptr);
    FTCF_CHECK_ARG(k != nullptr, "rows kernel: no instantiation for this shape");
    RowsParams pp     = p;
    void*      args[] = {&pp};
    FTCF_HIP_CHECK(hipLaunchKernel(k, dim3(p.plan.NB), dim3(RW_NT), args, p.plan.smem, s));
}

}  // namespace ftcf
