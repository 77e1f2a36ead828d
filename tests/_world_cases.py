"""Worlds in which the ORDER of `RenderInfo.world` is observable (shared by tests/test_world_order.py, tests/test_oracle_cross.py and
tests/golden/make_golden.py)."""
import numpy as np

from ray_tracer_s8_amd import _abi


def tie_world(seed=0, n_groups=6, dup=4, with_tris=True):
    """Groups of `dup` IDENTICAL spheres (same centre and radius: coincident centroids, every hit an exact distance tie)
    with different albedos, and pairs of identical triangles likewise: which copy a ray 'hits' is decided by the order of
    the world list alone."""
    g = np.random.default_rng(seed)
    sph = np.zeros(n_groups * dup, _abi.SPHERE_DTYPE)
    for k in range(n_groups):
        c = (g.uniform(-1.6, 1.6), g.uniform(-1.0, 1.0), g.uniform(-5.0, -3.0))
        r = g.uniform(0.3, 0.6)
        for j in range(dup):
            s = sph[k * dup + j]
            s["cx"], s["cy"], s["cz"], s["radius"] = c[0], c[1], c[2], r
            s["albedo_r"], s["albedo_g"], s["albedo_b"] = g.uniform(0.05, 0.95, 3)
            s["roughness"] = (0.0, 1.0, 0.3, 0.0)[j % 4]
    tri = np.zeros(0, _abi.TRIANGLE_DTYPE)
    if with_tris:
        tri = np.zeros(8, _abi.TRIANGLE_DTYPE)
        quad = [((-3, -1.2, -2), (3, -1.2, -2), (3, -1.2, -8)), ((-3, -1.2, -2), (3, -1.2, -8), (-3, -1.2, -8))]
        for i in range(8):
            a, b, c = quad[i % 2]                                   # four identical copies of each half of the floor
            t = tri[i]
            t["a"], t["b"], t["c"] = a, b, c
            t["albedo_r"], t["albedo_g"], t["albedo_b"] = g.uniform(0.1, 0.9, 3)
            t["roughness"] = 0.0 if i % 3 else 0.8
    return sph, tri


def interleave(ns, nt, seed):
    """A world_index that interleaves the two arrays AND permutes the copies inside each."""
    g = np.random.default_rng(seed)
    return g.permutation(ns + nt).astype(np.uint32)


