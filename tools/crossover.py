"""Linear-scan vs BVH-traversal engine across scene sizes (Mrays/s), 1920x1080, 4 spp, depth 8."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
for n in (16, 17, 20, 24, 28, 32, 48, 64, 128, 256, 512, 1024, 2048, 4096):
    sph = scenes.cornell16() if n == 16 else scenes.rand1024(n=n)
    rq = _abi.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=8, seed=5)
    reqs = []
    for k in range(4):
        r = rq.copy(); r.division_no = k; reqs.append(r)
    out = []
    with rt.Scene(0, rt.World(sph)) as sc:
        for fl in (32, 16):
            for r in reqs: r.flags = fl
            sc.render_tiles(reqs)
            best = 1e9
            for _ in range(3):
                _, _, st = sc.render_tiles(reqs)
                best = min(best, st.kernel_ms)
            out.append(st.ray_segments / best / 1e3)
    print(f"N={n:5d}  linear {out[0]:9.1f}  traverse {out[1]:9.1f}  Mrays/s   ratio {out[1]/out[0]:.2f}")
