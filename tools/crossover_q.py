"""Exact 64-byte vs quantised 32-byte traversal nodes across scene sizes (Mrays/s), 1920x1080, 4 spp, depth 8.
Two scene families: the c3 field (rand1024 box, n spheres) and the c5 field (rand65536 box)."""
import sys
sys.path.insert(0, ".")
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
for fam, gen in (("c3-field", scenes.rand1024), ("c5-field", scenes.rand65536)):
    for n in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        sph = gen(n=n)
        rq = _abi.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=8, seed=5)
        reqs = []
        for k in range(4):
            r = rq.copy(); r.division_no = k; reqs.append(r)
        out = []
        with rt.Scene(0, rt.World(sph)) as sc:
            for fl in (16 | 64, 16 | 128):
                for r in reqs: r.flags = fl
                sc.render_tiles(reqs)
                best = 1e9
                for _ in range(3):
                    _, _, st = sc.render_tiles(reqs)
                    best = min(best, st.kernel_ms)
                out.append(st.ray_segments / best / 1e3)
        print(f"{fam} N={n:6d}  exact {out[0]:9.1f}  quantised {out[1]:9.1f}  Mrays/s   ratio {out[1]/out[0]:.3f}", flush=True)
