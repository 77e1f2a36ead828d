"""Exact vs quantised nodes on the c3 scene across frame sizes (device output, one launch per frame)."""
import sys
sys.path.insert(0, ".")
import torch
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes
rt.init()
sph, rq0 = scenes.config("c3")
with rt.Scene(0, rt.World(sph)) as sc:
    for (w, h, spp) in ((640, 360, 8), (1280, 720, 8), (1920, 1080, 4), (1920, 1080, 8), (2560, 1440, 8), (3840, 2160, 4), (3840, 2160, 8)):
        out = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
        res = []
        for fl in (64, 128):
            rq = rq0.copy()
            rq.width, rq.height, rq.divisions, rq.division_no, rq.spp, rq.flags = w, h, 1, 0, spp, fl
            best = 1e9
            for _ in range(4):
                sc.render_tiles_device([rq], [out.data_ptr()], w * h * 3, 0)
                torch.cuda.synchronize()
                st = sc.collect()
                best = min(best, st.kernel_ms)
            res.append(st.ray_segments / best / 1e3)
        print(f"{w}x{h} {spp} spp ({w*h*spp/1e6:5.1f} M primary): exact {res[0]:8.1f}  quantised {res[1]:8.1f}  ratio {res[1]/res[0]:.3f}")
