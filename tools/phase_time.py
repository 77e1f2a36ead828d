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
    PT_BASE, NW = 4 + 16384, 8192
    # a launch to measure: every wave overwrites its own block of 16 words at its end (the warm-up launches above used no more waves)
    zero = (C.c_ulonglong * (16 * NW))()
    sc.render_tiles_device(reqs, ptrs, nb)
    hip.hipDeviceSynchronize()
    st = sc.collect()
    raw = (C.c_ulonglong * (16 * NW))()
    lib.rt_debug_read_counters(sc._h, PT_BASE, 16 * NW, raw)
    blocks = [[int(raw[16 * w + i]) for i in range(12)] for w in range(NW)]
    t_newest = max(b[8] for b in blocks)
    blocks = [(w, b) for w, b in enumerate(blocks) if b[8] and t_newest - b[8] < 100 * 1000 * 1000]      # (ends within a second of the newest)
    buf = [sum(b[i] for _, b in blocks) for i in range(8)]
    tot = sum(buf)
    print(f"segments {st.ray_segments}  kernel {st.kernel_ms:.2f} ms (one launch)  wave cycles {tot:.3e}")
    for i, n in enumerate(names):
        if buf[i]:
            print(f"  {n:36s} {100.0 * buf[i] / tot:6.2f} %")
    n_w = len(blocks)
    if n_w:
        t_last = max(b[8] for _, b in blocks)
        mean_end = sum(b[8] for _, b in blocks) / n_w
        mean_start = sum(b[9] for _, b in blocks) / n_w
        span = t_last - mean_start
        print(f"  {n_w} waves (s_memrealtime, 10 ns ticks): mean start -> latest end {span / 100:.1f} us; the mean wave ends "
              f"{(t_last - mean_end) / 100:.1f} us before the last one = {100.0 * (t_last - mean_end) / span:.1f} % of the span idle at the end")
        # the distribution of the waves' ends: how far before the last wave the p-th percentile wave ended, when the waves found the
        # queue empty, how long they took from there to their end (the drain) and in how many loop rounds
        e = sorted(t_last - b[8] for _, b in blocks)
        dq = sorted(t_last - b[10] for _, b in blocks if b[10])
        dr = sorted(b[8] - b[10] for _, b in blocks if b[10])
        nr = sorted(b[11] * 100 for _, b in blocks if b[10])
        fr = (0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99)
        for name, a in (("wave ends before the last end (us)", e[::-1]), ("queue found empty before the last end (us)", dq[::-1]),
                        ("from queue-empty to the wave's end (us)", dr), ("loop rounds from queue-empty to the wave's end", nr)):
            if a:
                print(f"  {name}: " + "  ".join(f"p{int(f * 100)} {a[min(len(a) - 1, int(f * len(a)))] / 100:.0f}" for f in fr) + f"  max {a[-1] / 100:.0f}")
        if os.environ.get("RT_PT_TRACE"):
            # the drain of every 20th wave, round by round: us since its queue-empty stamp (lanes still holding a unit)
            M = 0xfffffff
            tr = (C.c_ulonglong * (215 * 16))()
            lib.rt_debug_read_counters(sc._h, 4 + 8192 + 4400, len(tr), tr)
            byw = dict(blocks)
            rows = []
            for k in range(215):
                w = k * 20
                if w not in byw or not byw[w][10]:
                    continue
                tq, te = byw[w][10], byw[w][8]
                row = [int(tr[k * 16 + i]) for i in range(16) if int(tr[k * 16 + i])]
                n = min(len(row), byw[w][11])
                rows.append((te, f"  wave {w:4d} (queue empty {(t_last - tq) / 100:.0f} us before the last end): " +
                             "  ".join(f"{(((v & M) - tq) & M) / 100:.0f}({v >> 28})" for v in row[:n]) + f"  end {(te - tq) / 100:.0f}"))
            rows.sort()
            for _, line in rows[-int(os.environ["RT_PT_TRACE"]):]:              # the sampled waves that ended last
                print(line)
