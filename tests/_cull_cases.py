"""Scene generators of the culled-walk soaks (engines 5 and 6 against the oracle), shared by the reduced deterministic
versions in the GPU suite (tests/test_gpu_cull_soaks.py) and the long runs under tools/ (cull_soak.py, tricull_soak.py,
grazing_soak.py, sliver_soak.py)."""
import numpy as np

from ray_tracer_s8_amd import _abi as F
from ray_tracer_s8_amd import scenes


def sphere_field_cases(sizes=(65536, 30000, 12000, 65536, 120000, 8000), squeezed=(9000, 20000)):
    """c5-recipe fields of several sizes / seeds, plus squeezed (densely overlapping) ones.  Yields (spheres, k)."""
    for k, n in enumerate(sizes):
        yield scenes.rand65536(n=n, seed=0x5EED1000 + k), k
    for k, n in enumerate(squeezed):
        s = scenes.rand65536(n=n, seed=77 + k)
        s["cx"] *= 0.08
        s["cy"] *= 0.2
        s["cz"] = -3 + (s["cz"] + 3) * 0.1
        yield s, 10 + k


def terrain_case(nx, scale):
    """Terrain of 2 nx^2 triangles; `scale` moves |e1||e2| from 0.002 to the bound's limit of 0.25 and beyond (where the
    host must switch culling off)."""
    t = scenes.mesh_world(nx, nx).copy()
    for k in ("a", "b", "c"):
        t[k] = (t[k] * scale).astype(np.float32)
    return f"terrain {2 * nx * nx} x{scale}", np.zeros(0, F.SPHERE_DTYPE), t


def soup_case(g, n, ext, edge):
    """Dense triangle soup with 400 spheres (one huge: the `big` list) in the same box."""
    t = np.zeros(n, F.TRIANGLE_DTYPE)
    a = g.uniform([-ext, -2, -3 - 2 * ext], [ext, ext / 2, -3], (n, 3))
    t["a"], t["b"], t["c"] = a, a + g.uniform(-edge, edge, (n, 3)), a + g.uniform(-edge, edge, (n, 3))
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        t[ch] = g.uniform(0.1, 0.95, n)
    t["roughness"] = g.choice([0.0, 0.3, 1.0], n)
    t["emission"] = np.where(g.uniform(size=n) < 0.02, 4.0, 0.0)
    s = np.zeros(400, F.SPHERE_DTYPE)
    s["cx"], s["cy"], s["cz"] = g.uniform(-ext, ext, 400), g.uniform(-1, ext / 3, 400), g.uniform(-3 - 2 * ext, -3, 400)
    s["radius"] = g.uniform(0.1, 0.5, 400)
    s["cx"][0], s["cy"][0], s["cz"][0], s["radius"][0] = 0, -502, -20, 500
    s["albedo_r"] = s["albedo_g"] = s["albedo_b"] = 0.7
    return f"soup {n} edge {edge} + 400 spheres", s, t


def grazing_case(case):
    """Layers / walls of small triangles seen at grazing angles (the worst case of the triangle bound): the reference's
    determinant is a few times its 1e-5 threshold and its roots are off by per cent.  Returns (spheres, triangles, request)."""
    g = np.random.default_rng(1000 + case)
    e = float(g.choice([0.05, 0.2, 0.35, 0.49]))
    axis = int(g.integers(0, 2))                       # 0: floors (planes y = const), 1: side walls (planes x = const)
    tris = []
    for layer in range(int(g.integers(2, 5))):
        off = -float(g.choice([1e-4, 1e-3, 3e-3, 1e-2, 5e-2])) * (1 + layer) * (1 if g.uniform() < 0.8 else -1)
        alb = tuple(g.uniform(0.2, 0.9, 3))
        tilt = float(g.choice([0.0, 0.0, 1e-4, 1e-3]))
        for i in range(-6, 6):
            for k in range(2, int(30 / e) if e > 0.1 else 200):
                if g.uniform() < 0.3:
                    continue
                u0, z0 = e * i, -e * k
                h0 = off + tilt * z0
                if axis == 0:
                    A, B, C, D = (u0, h0, z0), (u0 + e, h0, z0), (u0, h0 + tilt * -e, z0 - e), (u0 + e, h0 + tilt * -e, z0 - e)
                else:
                    A, B, C, D = (h0, u0, z0), (h0, u0 + e, z0), (h0 + tilt * -e, u0, z0 - e), (h0 + tilt * -e, u0 + e, z0 - e)
                tris.append((A, B, C, *alb, float(g.choice([0.0, 1.0])), 0.0))
                tris.append((D, C, B, *alb, 0.0, 0.0))
    tr = np.array(tris, dtype=F.TRIANGLE_DTYPE)
    ns = int(g.integers(0, 40))
    sph = np.zeros(ns, F.SPHERE_DTYPE)
    if ns:
        sph["cx"], sph["cy"], sph["cz"] = g.uniform(-3, 3, ns), g.uniform(-0.5, 0.8, ns), g.uniform(-28, -3, ns)
        sph["radius"] = g.uniform(0.05, 0.4, ns)
        sph["albedo_r"] = sph["albedo_g"] = sph["albedo_b"] = 0.7
        sph["roughness"] = g.choice([0.0, 1.0], ns)
    rq = F.default_request(width=int(g.choice([64, 97])), height=int(g.choice([49, 81])), divisions=1, spp=2,
                           max_bounces=int(g.choice([2, 5])), seed=int(g.integers(0, 2**62)),
                           aperture=float(g.choice([0.0, 0.0, 0.01])), fov=float(g.choice([0.01, 0.03, 0.2])), t_max=500.0)
    return sph, tr, rq


def sliver_case(case):
    """Thousands of long thin triangles (length 2..15, width 1e-4..1e-2) in a box in front of the camera."""
    g = np.random.default_rng(7000 + case)
    n = int(g.choice([2000, 8000]))
    L, Wd = float(g.choice([2.0, 6.0, 15.0])), float(g.choice([1e-4, 1e-3, 1e-2]))
    t = np.zeros(n, F.TRIANGLE_DTYPE)
    a = g.uniform([-10, -3, -40], [10, 6, -3], (n, 3))
    dirv = g.normal(size=(n, 3))
    dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
    side = g.normal(size=(n, 3))
    side -= (side * dirv).sum(1, keepdims=True) * dirv
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    t["a"], t["b"], t["c"] = a, a + dirv * L * g.uniform(0.3, 1.0, (n, 1)), a + side * Wd
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        t[ch] = g.uniform(0.2, 0.9, n)
    t["roughness"] = g.choice([0.0, 1.0], n)
    t["emission"] = np.where(g.uniform(size=n) < 0.05, 3.0, 0.0)
    rq = F.default_request(width=160, height=90, divisions=1, spp=2, max_bounces=4, seed=case, t_max=500.0)
    return t, rq, (n, L, Wd)


XCULL = F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK     # engine 6 where valid
QCULL = F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_CULL_WALK                            # engine 5 where valid
