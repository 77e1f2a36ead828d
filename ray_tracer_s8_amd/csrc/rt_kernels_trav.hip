// Device code of the BVH-traversal engines (ISECT 2: exact 64-byte nodes, 3: quantised 32-byte nodes, 4: the same
// with a capped LDS stack, 5: the exact tree resident in LDS, 7: quantised nodes walked nearer child first with distance culling).  Its own
// translation unit: compiled with -fno-slp-vectorize (build.py) — packed FP32 pairs made by the SLP vectoriser in the
// ray-generation / shading code cost these kernels 1.5 % (register pairs, v_pk_mov), while the linear kernels gain 3 %.
#include "rt_kernel.hip.h"

namespace rtk {
KernelFn kernel_traverse(int variant, bool stats) {
    if (stats) {
        if (variant == 3) return rt_tile_kernel<5, false, LTREE_BLOCK, true>;
        if (variant == 4) return rt_tile_kernel<6, false, LTREE_BLOCK, true>;
        if (variant == 5) return rt_tile_kernel<7, false, BLOCK, true>;
        if (variant == 6) return rt_tile_kernel<8, false, BLOCK, true>;
        if (variant == 7) return rt_tile_kernel<9, false, BLOCK, true>;
        return variant == 2   ? rt_tile_kernel<4, false, BLOCK, true>
               : variant == 1 ? rt_tile_kernel<3, false, BLOCK, true>
                              : rt_tile_kernel<2, false, BLOCK, true>;
    }
    if (variant == 3) return rt_tile_kernel<5, false, LTREE_BLOCK>;
    if (variant == 4) return rt_tile_kernel<6, false, LTREE_BLOCK>;
    if (variant == 5) return rt_tile_kernel<7, false>;
    if (variant == 6) return rt_tile_kernel<8, false>;
    if (variant == 7) return rt_tile_kernel<9, false>;
    return variant == 2 ? rt_tile_kernel<4, false> : variant == 1 ? rt_tile_kernel<3, false> : rt_tile_kernel<2, false>;
}

// Self-test of sqrt_rn as the traversal kernels use it (tests/test_gpu_parity.py, rt_debug_sqrt_selftest): every f32 bit pattern in
// [from, from + n) through sqrt_rn and through the compiler's IEEE sequence; counts the patterns whose results differ in any bit
// (NaN results count as equal when both are NaN: their payloads are never stored).
__global__ void sqrt_selftest_kernel(uint32_t from, unsigned long long n, unsigned long long* bad) {
    unsigned long long mine = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        float x = __uint_as_float(from + (uint32_t)i);
        asm volatile("" : "+v"(x));
        const float ref = __builtin_sqrtf(x);
        const float a = sqrt_rn(x), b = sqrt_rn<true>(x);
        const bool both_nan_a = (a != a) && (ref != ref), both_nan_b = (b != b) && (ref != ref);
        if (__float_as_uint(a) != __float_as_uint(ref) && !both_nan_a) mine++;
        if (__float_as_uint(b) != __float_as_uint(ref) && !both_nan_b) mine++;
    }
    if (mine) atomicAdd(bad, mine);
}
void sqrt_selftest_launch(uint32_t from, unsigned long long n, unsigned long long* d_bad, hipStream_t st) {
    hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(4096), dim3(256), 0, st, from, n, d_bad);
}
}  // namespace rtk
