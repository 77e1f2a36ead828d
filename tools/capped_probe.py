import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi
from cull_matrix import field
rt.init()
F = _abi
g = np.random.default_rng(2029)
for name, sph in ((f"field 150000", field(150000, g, [-60, -1, -120], [60, 20, -3], (0.1, 0.5))), ("dense 150000", field(150000, g, [-6, -1, -20], [6, 5, -4], (0.2, 0.6))),
                  ("sparse field 300000", field(300000, g, [-300, -1, -600], [300, 60, -3], (0.1, 0.4)))):
    row = []
    with rt.Scene(0, rt.World(sph)) as sc:
        for fl in (0, F.RT_FLAG_NO_CULL_WALK):
            rq = F.default_request(width=2560, height=1440, divisions=4, spp=4, max_bounces=6, seed=5, flags=fl)
            reqs = []
            for k in range(4):
                r = rq.copy(); r.division_no = k; reqs.append(r)
            sc.render_tiles(reqs)
            _, _, st = sc.render_tiles(reqs)
            row.append((st.ray_segments / st.kernel_ms / 1e3, st.engine))
    print(f"{name:24s} default {row[0][0]:8.0f}/{row[0][1]}   plain walk {row[1][0]:8.0f}/{row[1][1]}", flush=True)
