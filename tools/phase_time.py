#!/usr/bin/env python3
"""Phase clock: build with -DRT_PROFILE_TIME, render one frame, print the share of wave cycles per phase.
usage: tools/phase_time.py [config] [flags]"""
import ctypes as C, os, sys
sys.path.insert(0, ".")
os.environ["RT_EXTRA_HIPCC_FLAGS"] = "-DRT_PROFILE_TIME " + os.environ.get("RT_PT_FLAGS", "")
from ray_tracer_s8_amd import build
build.build(force=True)
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
sph, rq = scenes.config(sys.argv[1] if len(sys.argv) > 1 else "c3")
reqs = []
for k in range(rq.divisions):
    r = rq.copy(); r.division_no = k; r.flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0; reqs.append(r)
names = ["pixel acquisition", "ray generation", "traversal steps (+ inline flushes)", "root tests (flush)",
         "shade + finish + store", "loop top / counter drain", "-", "-"]
with rt.Scene(0, rt.World(sph)) as sc:
    sc.render_tiles(reqs)
    lib = _abi.load()
    lib.rt_debug_read_counters.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    zero = (C.c_ulonglong * 8)()
    outs, _, st = sc.render_tiles(reqs)
    buf = (C.c_ulonglong * 8)()
    lib.rt_debug_read_counters(sc._h, 4 + 8192 + 160, 8, buf)
    tot = sum(buf)
    print(f"segments {st.ray_segments}  kernel {st.kernel_ms:.2f} ms (two frames accumulated in the clock)  wave cycles {tot:.3e}")
    for i, n in enumerate(names):
        if buf[i]:
            print(f"  {n:36s} {100.0 * buf[i] / tot:6.2f} %")
os.environ["RT_EXTRA_HIPCC_FLAGS"] = ""
build.build(force=True)
