#!/usr/bin/env python3
"""Long thin triangles through the culled walk over the exact nodes, 40 scenes against the oracle.
(Eight of them are in the suite: tests/test_gpu_cull_soaks.py.)"""
import sys
sys.path[:0] = [".", "tests"]
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F
from oracle import oracle as orc
from _cull_cases import XCULL, sliver_case
rt.init()
bad = used = 0
for case in range(40):
    t, rq, (n, L, Wd) = sliver_case(case)
    ref, _, info = orc.render(rq, None, t, backend=1)
    r = rq.copy(); r.flags = XCULL
    with rt.Scene(0, rt.World(np.zeros(0, F.SPHERE_DTYPE), t)) as sc:
        rgb, _, st = sc.render_tile(r)
    ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
    used += st.engine == 6
    bad += 0 if ok else 1
    if not ok or case % 8 == 0:
        print(f"case {case}: {n} slivers {L} x {Wd} engine {st.engine} segments {st.ray_segments} exact {ok}", flush=True)
print(f"40 cases, {used} through engine 6:", "FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
