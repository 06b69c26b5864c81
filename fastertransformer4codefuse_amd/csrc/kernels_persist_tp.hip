// Tensor-parallel instantiations of the persistent decode layers (persist_device.hip.h): the product kernel of a rank
// (in-launch all-reduce through the ranks' exchange windows) and the local-group kernel that runs all ranks of a job in one
// launch on one device (test infrastructure).  K/V register depth PS_UK only: a shard has 1/TP of the heads, so the KV
// splits are TP times finer than on one GPU and the short form covers the same contexts.
#include "persist_device.hip.h"

namespace ftcf {

const void* persist_tp_own_kernel(bool int8, int M, int dh, int uk, bool group);  // kernels_persist_tp_own.hip

const void* persist_tp_kernel(bool int8, int M, int dh, int uk, bool group, int own)
{
    if (own) {
        return persist_tp_own_kernel(int8, M, dh, uk, group);
    }
    if (uk != PS_UK) {
        return nullptr;
    }
#define PS_SEL(I8, MM, D)                                                                                              \
    if (int8 == I8 && M == MM && dh == D) {                                                                            \
        return group ? reinterpret_cast<const void*>(&k_decode_persistent<I8, MM, D, PS_UK, true, true>)               \
                     : reinterpret_cast<const void*>(&k_decode_persistent<I8, MM, D, PS_UK, true, false>);             \
    }
    PS_SEL(true, 1, 128)
#ifndef PS_ONLY_ONE  // (tools/build_variant_tu.sh: kernel-variant builds instantiate the 13B int8 one-row form only)
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

}  // namespace ftcf
