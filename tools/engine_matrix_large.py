import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
from cull_matrix import field, clusters
rt.init()
F = _abi
g = np.random.default_rng(2027)
cases = []
for n in (4096, 16384, 65536):
    cases.append((f"rand field {n} (c5 recipe)", scenes.rand65536(n=n)))
for n in (6000, 40000):
    cases.append((f"field {n}", field(n, g, [-60, -1, -120], [60, 20, -3], (0.1, 0.5))))
    cases.append((f"dense {n}", field(n, g, [-6, -1, -20], [6, 5, -4], (0.2, 0.6))))
    cases.append((f"mixed radii {n}", field(n, g, [-60, -1, -120], [60, 20, -3], (0.02, 2.0))))
engines = [("default", 0), ("quantised", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_CULL_WALK), ("culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_CULL_WALK),
           ("exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE)]
print(f"{'scene':30s} " + " ".join(f"{e[0]:>10s}" for e in engines))
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
    best = max(v for v, _ in row[1:])
    print(f"{name:30s} " + " ".join(f"{v:8.0f}/{e}" for v, e in row) + f"   default / best {row[0][0] / best:.3f}", flush=True)
