// Device code of the linear-scan engines (ISECT 0: scene resident in LDS, 1: scene streamed through LDS), both
// broad-phase forms.  Its own translation unit so that it can be compiled with its own flags (build.py): these kernels
// profit from SLP vectorisation into packed FP32 ops, the traversal kernels lose by it.
#include "rt_kernel.hip.h"

namespace rtk {
KernelFn kernel_linear(bool streamed, bool expanded) {
    if (streamed) return expanded ? rt_tile_kernel<1, true> : rt_tile_kernel<1, false>;
    return expanded ? rt_tile_kernel<0, true> : rt_tile_kernel<0, false>;
}
}  // namespace rtk
