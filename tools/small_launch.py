"""Kernel time of small launches (fixed per-launch cost): c3 scene, 8 spp, growing frame sizes, device output."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
sph, rq0 = scenes.config("c3")
with rt.Scene(0, rt.World(sph)) as sc:
    for (w, h) in ((64, 36), (128, 72), (256, 144), (512, 288), (960, 540), (1920, 1080), (3840, 2160)):
        rq = rq0.copy()
        rq.width, rq.height, rq.divisions, rq.division_no = w, h, 1, 0
        out = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda")
        best = 1e9
        for _ in range(5):
            sc.render_tiles_device([rq], [out.data_ptr()], w * h * 3, 0)
            torch.cuda.synchronize()
            st = sc.collect()
            best = min(best, st.kernel_ms)
        ideal = st.ray_segments / 9.5e9 * 1e3
        print(f"{w}x{h}: {st.ray_segments:>10d} segments  kernel {best:7.3f} ms  at 9.5 G/s {ideal:7.3f} ms  -> {st.ray_segments / best / 1e3:8.1f} Mrays/s")
