import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
rt.init()
F = _abi
tri = scenes.mesh_world()
for (w, h, spp) in ((1920, 1080, 4), (3840, 2160, 8)):
    with rt.Scene(0, rt.World(np.zeros(0, F.SPHERE_DTYPE), tri)) as sc:
        for name, fl in (("exact (default)", 0), ("quantised", F.RT_FLAG_QUANT_NODES), ("exact, counted", F.RT_FLAG_COUNT_STEPS)):
            rq = F.default_request(width=w, height=h, divisions=4, spp=spp, max_bounces=4, seed=5, flags=fl)
            reqs = []
            for k in range(4):
                r = rq.copy(); r.division_no = k; reqs.append(r)
            sc.render_tiles(reqs)
            _, _, st = sc.render_tiles(reqs)
            print(f"{w}x{h} {spp}spp {name:18s} engine {st.engine}  {st.ray_segments / st.kernel_ms / 1e3:8.0f} Mrays/s  kernel {st.kernel_ms:.2f} ms  steps/seg {st.node_steps / max(st.ray_segments,1):.1f} cand/seg {st.broad_candidates / st.ray_segments:.2f}", flush=True)
