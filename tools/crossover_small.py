"""Linear-scan vs LDS-tree engine on SMALL scenes (Mrays/s, kernel time): c2 itself and the first n spheres of two scene families
(the Cornell room: five huge overlapping wall spheres first; a sparse random field), 1920x1080, 4 spp, depth 4 (c2's request)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
def run(sph, depth=4):
    rq = _abi.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=depth, seed=5)
    reqs = []
    for k in range(4):
        r = rq.copy(); r.division_no = k; reqs.append(r)
    out = []
    with rt.Scene(0, rt.World(sph)) as sc:
        for fl in (32, 16):
            for r in reqs: r.flags = fl
            sc.render_tiles(reqs)
            best = 1e9
            for _ in range(5):
                _, _, st = sc.render_tiles(reqs)
                best = min(best, st.kernel_ms)
            out.append((st.ray_segments / best / 1e3, st.engine))
    return out
room, field = scenes.cornell16(), scenes.rand1024(n=64)
for name, src in (("room", room), ("field", field)):
    for n in (2, 3, 4, 6, 8, 12, 16):
        o = run(src[:n])
        print(f"{name:5s} n={n:3d}  linear {o[0][0]:9.1f} (engine {o[0][1]})  tree {o[1][0]:9.1f} (engine {o[1][1]})  ratio {o[1][0] / o[0][0]:.2f}", flush=True)
