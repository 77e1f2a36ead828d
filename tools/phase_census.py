#!/usr/bin/env python3
"""Phase census: build with -DRT_PROFILE_PHASES, render one c3 frame, print per-wave-iteration counts."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
# an instrumented VARIANT beside the product library (lib/librt_s8_pcensus.so): the product library is never rebuilt in place
os.environ["RT_LIB_VARIANT"] = "pcensus" + os.environ.get("RT_PT_TAG", "")
from ray_tracer_s8_amd import build
if not build.LIB_PATH.exists() or os.environ.get("RT_PT_FLAGS") is not None:
    os.environ["RT_EXTRA_HIPCC_FLAGS"] = "-DRT_PROFILE_PHASES -DRT_DEBUG_HOOKS " + os.environ.get("RT_PT_FLAGS", "")
    build.build(force=True)
    del os.environ["RT_EXTRA_HIPCC_FLAGS"]
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
sph, tri, rq = scenes.config_world(sys.argv[1] if len(sys.argv) > 1 else "c3")
reqs = []
for k in range(rq.divisions):
    r = rq.copy(); r.division_no = k; r.flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0; reqs.append(r)
names = ["main-loop iterations", "unit acquisition body", "camera gen", "UnitDisc loop iters", "broad pass-branch entries",
         "narrow loop iters", "exact hits -> consider", "bvh validation", "shade hit branch", "scatter (UnitSphere)",
         "UnitSphere loop iters", "sky branch", "finish", "path product loop iters", "slot commit (per pixel)"]
with rt.Scene(0, rt.World(sph, tri)) as sc:
    outs, _, st = sc.render_tiles(reqs)
    lib = _abi.load()
    # read raw counters through a tiny HIP memcpy via torch (device pointer is internal) -> use hip runtime
    hip = _abi.hip_runtime()
    # counters pointer is not exported; re-render with stats only: use rt_debug_counters
    buf = (C.c_ulonglong * 32)()
    lib.rt_debug_read_counters.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.rt_debug_read_counters(sc._h, 4 + 8192, 32, buf)
    it = buf[0]
    print(f"segments {st.ray_segments}  wave-iterations {it}  lanes/iter {st.ray_segments / it:.1f}")
    for i, n in enumerate(names):
        print(f"  [{i:2d}] {n:28s} {buf[i]:12d}   per iteration {buf[i] / it:7.3f}")
    lbuf = (C.c_ulonglong * 64)()
    lib.rt_debug_read_counters(sc._h, 4 + 8192 + 32, 64, lbuf)
    lnames = ["unit acquisition", "sampler loop round", "ray generation (common)", "  bounce part", "  camera part",
              "traversal step", "root-test (flush) round", "shade classify", "  hit branch", "  sky branch", "finish path",
              "  path product round", "slot commit (per pixel)", "lanes idling through a round", "out of slots (lanes of the wave)"]
    print("active lanes per execution (of 64):")
    for i, n in enumerate(lnames):
        if lbuf[32 + i]:
            print(f"  {n:28s} executions/iter {lbuf[32 + i] / it:7.3f}   mean active lanes {lbuf[i] / lbuf[32 + i]:5.1f}")
