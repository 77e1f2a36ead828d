#!/usr/bin/env python3
"""Scenes that MIX spheres and triangles (a terrain mesh under a sphere field): every engine that can take them against the host's
pick; kernel Mrays/s, 1080p, 4 spp, depth 6.   usage (GPU box): python3 tools/mixed_matrix.py"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F, scenes
from small_scene_matrix import field

ENG = [("linear", F.RT_FLAG_LINEAR_SCAN), ("LDS tree", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES),
       ("L2 exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_NO_CULL_WALK),
       ("L2 exact culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK),
       ("L2 quant", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_LDS_TREE), ("default", 0)]
g = np.random.default_rng(23)
print(f"{'scene':24s} " + " ".join(f"{e[0]:>16s}" for e in ENG))
for nx, ns in ((4, 20), (8, 100), (12, 400), (20, 400), (20, 2000), (40, 1000), (60, 6000), (100, 300)):
    tri = scenes.mesh_world(nx, nx)
    sph = field(ns, g, [-20, 0.5, -60], [20, 8, -4], (0.15, 0.6), ground=False)
    row = []
    with rt.Scene(0, rt.World(sph, tri)) as sc:
        for ename, fl in ENG:
            if ename == "linear" and len(tri) + ns > 4000:
                row.append("-"); continue
            rq = F.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=6, seed=5, flags=fl)
            reqs = []
            for k in range(4):
                r = rq.copy(); r.division_no = k; reqs.append(r)
            sc.render_tiles(reqs)
            best = 1e9
            for _ in range(3):
                _, _, st = sc.render_tiles(reqs)
                best = min(best, st.kernel_ms)
            row.append(f"{st.ray_segments / best / 1e3:9.0f} (e{st.engine})")
    print(f"{str(len(tri)) + ' tris + ' + str(ns) + ' sph':24s} " + " ".join(f"{c:>16s}" for c in row), flush=True)
