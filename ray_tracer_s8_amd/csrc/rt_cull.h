// rt_cull.h — the distance-culling bounds of the culled walks (product code, shared by the HIP kernels and by the CPU
// harness tests/host/cull_host.cpp, which checks the inequalities they claim over 10^7 seeded and adversarial cases:
// tests/test_cull_lemma.py).  Plain f32 arithmetic, no contraction (-ffp-contract=off on both sides).
#pragma once
#if defined(__HIPCC__)
#define RT_HD __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtk {

// Distance culling bound of the culled walk (ISECT 7).  A sphere X whose own AABB the ray enters at t_X
// and whose reference root test returns x satisfies x >= t_X - sqrt(2) r_X - 2^-9 |o - c_X| - 2^-21.4 |o - c_X|^2 (genuine
// roots lie in the box up to the rounding of the reference's quadratic; false roots of a near miss lie within 2^-9 |oc| of
// the closest approach, which is at most sqrt(2) r behind the box entry), |o - c_X| <= 1.01 (x + 2 r_X), and the compared
// distance |P - o| is x up to 2^-20 x + 2^-22 |o|_1.  Hence nothing entered beyond the value returned here can reach
// `best` or tie with it.  (The walk's boxes contain the exact ones, so their entry is not later than t_X.)
RT_HD float cull_bound(float best, float ox, float oy, float oz, float r_slack) {
    const float a = 1.5f * r_slack + 0x1p-18f * (__builtin_fabsf(ox) + __builtin_fabsf(oy) + __builtin_fabsf(oz));
    const float q = best + 2.0f * r_slack;
    return (best + a) * (1.0f + 0x1p-8f) + 0x1p-19f * (q * q);
}

// The same for TRIANGLES (mesh.rs:109-161, Moller-Trumbore in f32; u = 2^-24).  With e1, e2 the rounded edges, K = |e1||e2| and
// |det| >= 1e-5 (the reference's own rejection threshold), the computed determinant is the true one up to a factor
// 1 +- 0.0602 K, so for K <= 0.25 the returned root x and the computed barycentrics are those of the true ray-plane
// intersection t*, u*, v* up to |t* - x| <= K (0.134 x + 0.072 Es) and a point at most D = K (0.138 x + 0.199 Es) + 3 u E outside the
// triangle (Es = |e1| + |e2|, E the longest edge; |o - A| <= 1.07 (x + Es) was used).  A point that close to the triangle is
// that close to its AABB, and a ray that enters the AABB at T_X > t* has travelled at most the diameter of the AABB inflated
// by D between the two: t* >= T_X - diag_X - 3.47 D.  Together: x (1 + 0.613 K) >= T_X - diag_X - 0.763 K Es - 11 u E.  The bound
// below carries twice these K terms, the same (1 + 2^-8) and |o|_1 terms as cull_bound for the float evaluation of the box
// entries and of the compared distance, and the maxima over the triangles that are not in the scene's `big` list
// (K > 0.25 or a box far larger than the rest: tested at every query start instead).
RT_HD float cull_bound_tri(float best, float ox, float oy, float oz, float k, float diag, float es, float e) {
    const float o1 = __builtin_fabsf(ox) + __builtin_fabsf(oy) + __builtin_fabsf(oz);
    return (best + 0x1p-21f * o1) * (1.0f + 1.25f * k) * (1.0f + 0x1p-8f) + diag + 1.6f * k * es + 0x1p-18f * o1 + 1e-6f * e;
}

}  // namespace rtk
