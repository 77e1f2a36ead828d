#!/usr/bin/env python3
"""Dense scenes below the quantised-node threshold: default engine (LDS-resident tree) vs the culled L2 walk forced on."""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
rt.init()
F = _abi
sys.path.insert(0, "tools")
from cull_matrix import field, clusters   # noqa  (runs its matrix too when imported: guard below)
g = np.random.default_rng(2028)
cases = []
for n in (300, 1000, 2000, 3500):
    cases.append((f"dense {n}", field(n, g, [-4, -1, -14], [4, 4, -3], (0.2, 0.6))))
    cases.append((f"mixed radii {n}", field(n, g, [-24, -1, -48], [24, 10, -3], (0.05, 2.0))))
    cases.append((f"field {n}", field(n, g, [-24, -1, -48], [24, 10, -3], (0.15, 0.6))))
engines = [("default", 0), ("culled L2", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_CULL_WALK),
           ("quant L2", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_CULL_WALK)]
print(f"{'scene':24s} " + " ".join(f"{e[0]:>11s}" for e in engines) + "   culled / default")
for name, sph in cases:
    row = []
    with rt.Scene(0, rt.World(sph)) as sc:
        for ename, fl in engines:
            rq = F.default_request(width=2560, height=1440, divisions=4, spp=4, max_bounces=6, seed=5, flags=fl)
            reqs = []
            for k in range(4):
                r = rq.copy(); r.division_no = k; reqs.append(r)
            sc.render_tiles(reqs)
            _, _, st = sc.render_tiles(reqs)
            row.append((st.ray_segments / st.kernel_ms / 1e3, st.engine))
    print(f"{name:24s} " + " ".join(f"{v:9.0f}/{e}" for v, e in row) + f"   {row[1][0] / row[0][0]:.3f}", flush=True)
