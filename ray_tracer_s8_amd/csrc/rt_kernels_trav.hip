// Device code of the BVH-traversal engines (ISECT 2: exact 64-byte nodes, 3: quantised 32-byte nodes, 4: the same
// with a capped LDS stack, 5: the exact tree resident in LDS, 7: quantised nodes walked nearer child first with distance culling).  Its own
// translation unit: compiled with -fno-slp-vectorize (build.py) — packed FP32 pairs made by the SLP vectoriser in the
// ray-generation / shading code cost these kernels 1.5 % (register pairs, v_pk_mov), while the linear kernels gain 3 %.
#include "rt_kernel.hip.h"

namespace rtk {
KernelFn kernel_traverse(int variant, bool stats) {
    if (stats) {
        if (variant == 3) return rt_tile_kernel<5, false, LTREE_BLOCK, true>;
        if (variant == 5) return rt_tile_kernel<7, false, BLOCK, true>;
        if (variant == 6) return rt_tile_kernel<8, false, BLOCK, true>;
        if (variant == 7) return rt_tile_kernel<9, false, BLOCK, true>;
        return variant == 2   ? rt_tile_kernel<4, false, BLOCK, true>
               : variant == 1 ? rt_tile_kernel<3, false, BLOCK, true>
                              : rt_tile_kernel<2, false, BLOCK, true>;
    }
    if (variant == 3) return rt_tile_kernel<5, false, LTREE_BLOCK>;
    if (variant == 5) return rt_tile_kernel<7, false>;
    if (variant == 6) return rt_tile_kernel<8, false>;
    if (variant == 7) return rt_tile_kernel<9, false>;
    return variant == 2 ? rt_tile_kernel<4, false> : variant == 1 ? rt_tile_kernel<3, false> : rt_tile_kernel<2, false>;
}
}  // namespace rtk
