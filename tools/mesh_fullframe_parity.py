import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
from oracle import oracle as orc
rt.init()
sph, tri, rq = scenes.config_world("mesh")
for seed in (rq.seed, 1, 2):
    r = rq.copy(); r.divisions = 1; r.division_no = 0; r.seed = seed; r.spp = 8
    ref, _, info = orc.render(r, None, tri, backend=1)
    with rt.Scene(0, rt.World(sph, tri)) as sc:
        rgb, _, st = sc.render_tile(r)
    print("seed", seed, "engine", st.engine, "segments", st.ray_segments, "exact", bool(np.array_equal(rgb, ref)), st.ray_segments == info["ray_segments"])
