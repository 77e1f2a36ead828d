#!/usr/bin/env python3
"""Sphere scenes too large for the LDS tree and below RT_QNODES_MIN_PRIMS (about 1100 ... 4096 spheres): the 64-byte exact nodes or the
32-byte quantised ones (plain and culled)?  Kernel Mrays/s, 1080p, 4 spp, depth 6, four scene families, and the host's pick.
usage (GPU box): python3 tools/qnodes_mid_matrix.py"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F
from small_scene_matrix import field

def clusters(n, g):
    k = g.uniform([-30, 0, -70], [30, 10, -5], (24, 3))
    s = field(n, g, [0, 0, 0], [1, 1, 1], (0.02, 0.15))
    c = k[g.integers(0, 24, n)] + g.normal(0, 0.8, (n, 3))
    s["cx"][1:], s["cy"][1:], s["cz"][1:] = c[1:, 0], c[1:, 1], c[1:, 2]
    return s

ENG = [("L2 exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_NO_CULL_WALK),
       ("L2 exact culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK),
       ("L2 quant", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_CULL_WALK),
       ("L2 quant culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_CULL_WALK), ("default", 0)]
g = np.random.default_rng(17)
print(f"{'scene':16s} " + " ".join(f"{e[0]:>16s}" for e in ENG))
for n in (1200, 1600, 2048, 3000, 4000):
    for name, s in (("field", field(n, g, [-40, -1, -80], [40, 15, -3], (0.12, 0.55))), ("sheet", field(n, g, [-40, 1.0, -90], [40, 1.05, -4], (0.05, 0.2))),
                    ("clusters", clusters(n, g)), ("dense", field(n, g, [-5, -1, -16], [5, 5, -4], (0.2, 0.6)))):
        row = []
        with rt.Scene(0, rt.World(s)) as sc:
            for ename, fl in ENG:
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
        print(f"{name + ' ' + str(n):16s} " + " ".join(f"{c:>16s}" for c in row), flush=True)
