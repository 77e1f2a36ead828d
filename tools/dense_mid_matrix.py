#!/usr/bin/env python3
"""Piles of 256 ... 1000 overlapping spheres, whose tree still fits a CU's LDS: the LDS tree (no distance culling) against the culled
walks through L2 (quantised and exact nodes) and the linear scan; kernel Mrays/s at depth 6, and the engine the host picks.
usage (GPU box): python3 tools/dense_mid_matrix.py"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F
from small_scene_matrix import field

ENG = [("linear", F.RT_FLAG_LINEAR_SCAN), ("LDS tree", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES),
       ("L2 quant culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK),
       ("L2 exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE), ("default", 0)]
g = np.random.default_rng(13)
print(f"{'scene':16s} " + " ".join(f"{e[0]:>16s}" for e in ENG))
import os
SIZES = tuple(int(x) for x in os.environ.get("RT_DM_SIZES", "256,384,512,768,1000").split(","))
for n in SIZES:
    for name, s in (("pile", field(n, g, [-3, -1, -10], [3, 3, -4], (0.3, 0.8))), ("wide pile", field(n, g, [-8, -1, -20], [8, 4, -4], (0.3, 0.8))),
                    ("tight pile", field(n, g, [-2, -1, -8], [2, 2, -4], (0.4, 0.9)))):
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
