#!/usr/bin/env python3
"""Culled walk over the exact nodes (triangles) against the oracle: terrains at several resolutions and scales (the scale moves
|e1||e2| from 0.002 to the bound's limit of 0.25 and beyond, where the host must switch culling off), a dense triangle soup,
mixed scenes; 1280x720, 4 spp.  (Reduced version in the suite: tests/test_gpu_cull_soaks.py.)"""
import sys
sys.path[:0] = [".", "tests"]
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F
from oracle import oracle as orc
from _cull_cases import XCULL, soup_case, terrain_case
rt.init()
bad = 0
cases = [terrain_case(nx, scale) for nx, scale in ((64, 1.0), (64, 0.35), (128, 1.0), (128, 2.0), (224, 1.0), (224, 2.4), (224, 3.3), (32, 0.6))]
g = np.random.default_rng(5)
cases += [soup_case(g, n, ext, edge) for n, ext, edge in ((20000, 12.0, 0.5), (60000, 30.0, 0.3), (8000, 4.0, 0.7))]
for k, (name, sph, tri) in enumerate(cases):
    rq = F.default_request(width=1280, height=720, divisions=1, spp=4, max_bounces=2 + 2 * (k % 3), seed=500 + k)
    ref, _, info = orc.render(rq, sph if len(sph) else None, tri, backend=1)
    for fl in (XCULL, 0):
        r = rq.copy(); r.flags = fl
        with rt.Scene(0, rt.World(sph, tri)) as sc:
            rgb, _, st = sc.render_tile(r)
        ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
        bad += 0 if ok else 1
        print(f"{name:36s} depth {rq.max_bounces} flags {fl:5d} engine {st.engine} segments {st.ray_segments:9d} {st.ray_segments / st.kernel_ms / 1e3:7.0f} Mrays/s exact {ok}", flush=True)
print("FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
