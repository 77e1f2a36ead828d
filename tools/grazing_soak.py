#!/usr/bin/env python3
"""Random variations of test_grazing_rays_over_triangle_floors: layers / walls of small triangles seen at grazing angles (the worst
case of the triangle bound of the culled walk), engine 6 against the oracle.  usage: tools/grazing_soak.py [cases]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi
from oracle import oracle as orc
rt.init()
F = _abi
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = used = 0
for case in range(n_cases):
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
    rq = F.default_request(width=int(g.choice([64, 97])), height=int(g.choice([49, 81])), divisions=1, spp=2, max_bounces=int(g.choice([2, 5])),
                           seed=int(g.integers(0, 2**62)), aperture=float(g.choice([0.0, 0.0, 0.01])), fov=float(g.choice([0.01, 0.03, 0.2])), t_max=500.0)
    ref, _, info = orc.render(rq, sph if ns else None, tr, backend=1)
    r = rq.copy(); r.flags = F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK
    with rt.Scene(0, rt.World(sph, tr)) as sc:
        rgb, _, st = sc.render_tile(r)
    ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
    used += st.engine == 6
    bad += 0 if ok else 1
    if not ok or case % 10 == 0:
        print(f"case {case}: {len(tr)} triangles edge {e} axis {axis} engine {st.engine} segments {st.ray_segments} exact {ok}", flush=True)
print(f"{n_cases} cases, {used} through the culled walk over the exact nodes:", "FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
