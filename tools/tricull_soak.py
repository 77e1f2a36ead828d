#!/usr/bin/env python3
"""Culled walk over the exact nodes (triangles) against the oracle: terrains at several resolutions and scales (the scale moves
|e1||e2| from 0.002 to the bound's limit of 0.25 and beyond, where the host must switch culling off), a dense triangle soup,
mixed scenes; 1280x720, 4 spp."""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
from oracle import oracle as orc
rt.init()
F = _abi
bad = 0
cases = []
for nx, scale in ((64, 1.0), (64, 0.35), (128, 1.0), (128, 2.0), (224, 1.0), (224, 2.4), (224, 3.3), (32, 0.6)):
    t = scenes.mesh_world(nx, nx).copy()
    for k in ("a", "b", "c"):
        t[k] = (t[k] * scale).astype(np.float32)
    cases.append((f"terrain {2 * nx * nx} x{scale}", np.zeros(0, F.SPHERE_DTYPE), t))
g = np.random.default_rng(5)
for n, ext, edge in ((20000, 12.0, 0.5), (60000, 30.0, 0.3), (8000, 4.0, 0.7)):
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
    cases.append((f"soup {n} edge {edge} + 400 spheres", s, t))
for k, (name, sph, tri) in enumerate(cases):
    rq = F.default_request(width=1280, height=720, divisions=1, spp=4, max_bounces=2 + 2 * (k % 3), seed=500 + k)
    ref, _, info = orc.render(rq, sph if len(sph) else None, tri, backend=1)
    for fl in (F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK, 0):
        r = rq.copy(); r.flags = fl
        with rt.Scene(0, rt.World(sph, tri)) as sc:
            rgb, _, st = sc.render_tile(r)
        ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
        bad += 0 if ok else 1
        print(f"{name:36s} depth {rq.max_bounces} flags {fl:5d} engine {st.engine} segments {st.ray_segments:9d} {st.ray_segments / st.kernel_ms / 1e3:7.0f} Mrays/s exact {ok}", flush=True)
print("FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
