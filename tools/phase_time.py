#!/usr/bin/env python3
"""Phase clock: build with -DRT_PROFILE_TIME, render one frame, print the share of wave cycles per phase.
usage: tools/phase_time.py [config] [flags]"""
import ctypes as C, os, sys
sys.path.insert(0, ".")
# an instrumented VARIANT beside the product library (lib/librt_s8_ptime.so): the product library is never rebuilt in place
os.environ["RT_LIB_VARIANT"] = "ptime" + os.environ.get("RT_PT_TAG", "")
from ray_tracer_s8_amd import build
if not build.LIB_PATH.exists() or os.environ.get("RT_PT_FLAGS") is not None:
    os.environ["RT_EXTRA_HIPCC_FLAGS"] = "-DRT_PROFILE_TIME -DRT_DEBUG_HOOKS " + os.environ.get("RT_PT_FLAGS", "")
    build.build(force=True)
    del os.environ["RT_EXTRA_HIPCC_FLAGS"]
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
rt.init()
sph, tri, rq = scenes.config_world(sys.argv[1] if len(sys.argv) > 1 else "c3")
reqs = []
for k in range(rq.divisions):
    r = rq.copy(); r.division_no = k; r.flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0; reqs.append(r)
names = ["unit acquisition: seeding", "ray generation", "traversal steps (+ inline flushes)", "root tests (flush)",
         "shade + deposit", "loop top", "commit (ordered sums, mean / gamma / store)", "unit acquisition: slot assignment"]
hip = _abi.hip_runtime()
nb = (rq.height // rq.divisions) * rq.width * 3
with rt.Scene(0, rt.World(sph, tri)) as sc:
    dbuf = C.c_void_p()
    assert hip.hipMalloc(C.byref(dbuf), C.c_size_t(nb * len(reqs))) == 0
    ptrs = [dbuf.value + i * nb for i in range(len(reqs))]
    sc.render_tiles_device(reqs, ptrs, nb)               # ONE launch per frame (the host-buffer path splits it in two)
    hip.hipDeviceSynchronize()
    sc.collect()
    sc.render_tiles(reqs[:1])
    lib = _abi.load()
    lib.rt_debug_read_counters.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    zero = (C.c_ulonglong * 13)()
    lib.rt_debug_read_counters(sc._h, 4 + 8192 + 160, 13, zero)      # (counters so far: warm-up launches)
    base = list(zero)
    sc.render_tiles_device(reqs, ptrs, nb)
    hip.hipDeviceSynchronize()
    st = sc.collect()
    raw = (C.c_ulonglong * 13)()
    lib.rt_debug_read_counters(sc._h, 4 + 8192 + 160, 13, raw)
    buf = [int(raw[i]) - int(base[i]) for i in range(8)]
    tot = sum(buf)
    print(f"segments {st.ray_segments}  kernel {st.kernel_ms:.2f} ms (one launch)  wave cycles {tot:.3e}")
    for i, n in enumerate(names):
        if buf[i]:
            print(f"  {n:36s} {100.0 * buf[i] / tot:6.2f} %")
    # wave start / end times of THIS launch (s_memtime, one device clock): the sums and the count accumulate, the latest end
    # is the last launch's
    n_w = int(raw[11]) - int(base[11])
    if os.environ.get("RT_PT_DEBUG"):
        print("  raw ", [int(v) for v in raw][8:]); print("  base", [int(v) for v in base][8:])
    if n_w:
        t_last = int(raw[8])
        mean_end = (int(raw[9]) - int(base[9])) / n_w
        mean_start = (int(raw[12]) - int(base[12])) / n_w
        span = t_last - mean_start
        print(f"  {n_w} waves (s_memrealtime, 10 ns ticks): mean start -> latest end {span / 100:.1f} us; the mean wave ends "
              f"{(t_last - mean_end) / 100:.1f} us before the last one = {100.0 * (t_last - mean_end) / span:.1f} % of the span idle at the end")
        # the distribution of the waves' ends (last launch): how far before the last wave the p-th percentile wave ended, when the waves
        # found the queue empty, and how long they took from there to their end (the drain)
        ends = (C.c_ulonglong * min(n_w, 7900))()
        lib.rt_debug_read_counters(sc._h, 4 + 8192 + 256, len(ends), ends)
        M = 0xfffffff
        tl = t_last & M
        vals = [int(v) for v in ends if int(v)]
        e = sorted((tl - (v & M)) & M for v in vals)
        dq = sorted((tl - ((v >> 28) & M)) & M for v in vals if (v >> 28) & M)
        dr = sorted(((v & M) - ((v >> 28) & M)) & M for v in vals if (v >> 28) & M)
        nr = sorted((v >> 56) * 100 for v in vals if (v >> 28) & M)
        fr = (0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99)
        for name, a in (("wave ends before the last end (us)", e[::-1]), ("queue found empty before the last end (us)", dq[::-1]),
                        ("from queue-empty to the wave's end (us)", dr), ("loop rounds from queue-empty to the wave's end", nr)):
            if a:
                print(f"  {name}: " + "  ".join(f"p{int(f * 100)} {a[min(len(a) - 1, int(f * len(a)))] / 100:.0f}" for f in fr) + f"  max {a[-1] / 100:.0f}")
