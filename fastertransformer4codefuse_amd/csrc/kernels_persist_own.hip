// Persistent decode layers, P3 in the own-group layout (persist_device.hip.h "own-group layout": whole column groups per
// workgroup, one hop at the layer boundary): the instantiations for one GPU.  A translation unit of its own so that the layouts
// build in parallel and the K-piece form keeps the code it was tuned with.
#include "persist_device.hip.h"

namespace ftcf {

const void* persist_own_kernel(bool int8, int M, int dh, int uk)
{
#define PS_SEL(I8, MM, D)                                                                                              \
    if (int8 == I8 && M == MM && dh == D) {                                                                            \
        return uk == PS_UK_LONG                                                                                        \
                   ? reinterpret_cast<const void*>(&k_decode_persistent<I8, MM, D, PS_UK_LONG, false, false, true>)    \
                   : reinterpret_cast<const void*>(&k_decode_persistent<I8, MM, D, PS_UK, false, false, true>);        \
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

}  // namespace ftcf
