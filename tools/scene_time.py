import sys, time
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes
rt.init()
for name in ("c2", "c3", "c5"):
    sph, rq = scenes.config(name)
    w = rt.World(sph)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        sc = rt.Scene(0, w)
        t1 = time.perf_counter()
        sc.close()
        ts.append((t1 - t0) * 1e3)
    print(name, len(sph), "scene create ms:", [round(t, 2) for t in ts])
