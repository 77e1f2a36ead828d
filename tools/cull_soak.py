#!/usr/bin/env python3
"""Culled walk against the oracle on c5-recipe scenes of several sizes, seeds and depths (1280x720, 4 spp), plus dense fields.
(Reduced version in the suite: tests/test_gpu_cull_soaks.py.)"""
import sys
sys.path[:0] = [".", "tests"]
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F
from oracle import oracle as orc
from _cull_cases import QCULL, sphere_field_cases
rt.init()
bad = 0
for sph, k in sphere_field_cases():
    rq = F.default_request(width=1280, height=720, divisions=1, spp=4, max_bounces=3 + 2 * (k % 4), seed=1000 + k)
    ref, _, info = orc.render(rq, sph, backend=1)
    for fl in (QCULL, 0):
        r = rq.copy(); r.flags = fl
        with rt.Scene(0, rt.World(sph)) as sc:
            rgb, _, st = sc.render_tile(r)
        ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
        bad += 0 if ok else 1
        print(f"n={len(sph):6d} depth {rq.max_bounces} flags {fl:5d} engine {st.engine} segments {st.ray_segments:9d} exact {ok}", flush=True)
print("FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
