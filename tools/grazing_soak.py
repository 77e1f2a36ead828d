#!/usr/bin/env python3
"""Random variations of test_grazing_rays_over_triangle_floors: layers / walls of small triangles seen at grazing angles (the worst
case of the triangle bound of the culled walk), engine 6 against the oracle.  usage: tools/grazing_soak.py [cases]
(Twelve of them are in the suite: tests/test_gpu_cull_soaks.py.)"""
import sys
sys.path[:0] = [".", "tests"]
import numpy as np
import ray_tracer_s8_amd as rt
from oracle import oracle as orc
from _cull_cases import XCULL, grazing_case
rt.init()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = used = 0
for case in range(n_cases):
    sph, tr, rq = grazing_case(case)
    ref, _, info = orc.render(rq, sph if len(sph) else None, tr, backend=1)
    r = rq.copy(); r.flags = XCULL
    with rt.Scene(0, rt.World(sph, tr)) as sc:
        rgb, _, st = sc.render_tile(r)
    ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
    used += st.engine == 6
    bad += 0 if ok else 1
    if not ok or case % 10 == 0:
        print(f"case {case}: {len(tr)} triangles engine {st.engine} segments {st.ray_segments} exact {ok}", flush=True)
print(f"{n_cases} cases, {used} through the culled walk over the exact nodes:", "FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
