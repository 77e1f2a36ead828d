// Device code of the linear-scan engines (ISECT 0: scene resident in LDS, 1: scene streamed through LDS), both
// broad-phase forms.  Its own translation unit so that it can be compiled with its own flags (build.py): these kernels
// profit from SLP vectorisation into packed FP32 ops, the traversal kernels lose by it.
// The square roots here stay the compiler's IEEE sequence: the shorter correctly-rounded core with its branch for odd operands
// (sqrt_rn, rt_kernel.hip.h) gains 1 % in the LDS-tree kernel and 0.4 % in the quantised walk but costs these kernels 2 % (c2).
#define RT_IEEE_SQRT_PLAIN
#include "rt_kernel.hip.h"

namespace rtk {
KernelFn kernel_linear(bool streamed, bool expanded) {
    if (streamed) return expanded ? rt_tile_kernel<1, true> : rt_tile_kernel<1, false>;
    return expanded ? rt_tile_kernel<0, true> : rt_tile_kernel<0, false>;
}
}  // namespace rtk
