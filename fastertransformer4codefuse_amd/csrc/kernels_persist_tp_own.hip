// Tensor-parallel instantiations of the persistent decode layers with P3 in the own-group layout (persist_device.hip.h): the
// product kernel of a rank and the local-group kernel (test infrastructure); see kernels_persist_tp.hip.
#include "persist_device.hip.h"

namespace ftcf {

const void* persist_tp_own_kernel(bool int8, int M, int dh, int uk, bool group)
{
    if (uk != PS_UK) {
        return nullptr;
    }
#define PS_SEL(I8, MM, D)                                                                                              \
    if (int8 == I8 && M == MM && dh == D) {                                                                            \
        return group ? reinterpret_cast<const void*>(&k_decode_persistent<I8, MM, D, PS_UK, true, true, true>)         \
                     : reinterpret_cast<const void*>(&k_decode_persistent<I8, MM, D, PS_UK, true, false, true>);       \
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

}  // namespace ftcf
