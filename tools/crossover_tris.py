#!/usr/bin/env python3
"""Linear scan vs BVH traversal on triangle meshes of growing size (the constant TRAVERSE_MIN_TRIS in rt_api.hip)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
rt.init()
for n in (2, 3, 4, 6, 8, 12, 16):
    tri = scenes.mesh_world(n, n)
    res = {}
    for name, fl in (("linear", _abi.RT_FLAG_LINEAR_SCAN), ("traverse", _abi.RT_FLAG_BVH_TRAVERSE), ("default", 0)):
        rq = _abi.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=4, seed=5, flags=fl)
        reqs = []
        for k in range(4):
            r = rq.copy(); r.division_no = k; reqs.append(r)
        with rt.Scene(0, rt.World(triangles=tri)) as sc:
            sc.render_tiles(reqs)
            _, _, st = sc.render_tiles(reqs)
        res[name] = (st.ray_segments / st.kernel_ms / 1e3, st.engine)
    print(f"{len(tri):6d} triangles: " + "  ".join(f"{k} {v[0]:8.0f} Mrays/s (engine {v[1]})" for k, v in res.items()))
