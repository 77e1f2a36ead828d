#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of the HIP tile renderer on BASELINE config c3
(1024 random spheres, 3840x2160, 8 spp, depth 8), strips sharded over N GPUs.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5|mesh] [--strong]

A "step" is one pass of the hot path over one batch of synthetic input: every rank renders the strips it owns of the
job.  Default = weak scaling: N frames of the workload at N GPUs, unit (frame f, strip d) goes to rank (f + d) % N.
--strong = the controller's split that BASELINE configs 4 and 5 name: ONE frame, strip d goes to rank d % N.  No
collective on the data path either way.  The scene is resident in HBM before the timed region; output strips are written
to HBM.  Timing: barrier + synchronize on both sides of exactly K steps, MAX over ranks.  value = ray segments of all
ranks / that time.

Also reported in the same JSON line:
  other_workloads  BASELINE configs c2 / c4 / c5 and the 100 352-triangle mesh through the same device-resident entry point,
                   three timed launches each after the timed region (rank 0, N = 1): value, engine, ms per frame
  frame_path       wall clock of the in-process product path that replaces the controller's dispatch + assembly: c3 and c4
                   frames through a persistent rt_frame_ctx into a host buffer — first frame (pays pin_ms, carries scene_ms)
                   and a later frame of the job (pays neither): wall / pin / scene / kernel / exposed download / host ms
  strong_c4        N > 1 only: ONE c4 frame split by strip over the ranks (BASELINE config 4), max-rank time, beside the
                   weak-scaling value of the line
  frame_path_all_devices   N > 1 only: c4 frames through ONE rt_frame_ctx over all N devices inside rank 0 (static split and
                   strip queue), wall clock: the in-process replacement of the controller's dispatch loop on N GPUs
  roofline         the timed kernel against the FP32 vector peak, from work it EXECUTES: for the traversal engines the
                   slab tests (48 flop per node visited) and root tests (20 flop per leaf reached), both counted by one
                   extra launch of the same frame through the kernel's counting twin, outside the timed region, against
                   the NON-FMA peak (78.65 TFLOP/s: the parity build's slab and root tests are sub / mul / compare); avg
                   launch duration from HIP events recorded by the library on the stream the kernel runs on.  frac <= 1
                   is asserted.  valu_fraction_8d = SURVEY 8(d)'s segments x 20 x N / t / 78.65e12 (> 1 for a traversal
                   engine: not a hardware fraction).  valu_issue / traffic come from committed PMC passes and say so.
  roofline_linear  the same frame through the north-star-shaped kernel (linear scan over the LDS-resident sphere list),
                   own launches in this run: SURVEY 8(d)'s 20*N flop per segment / launch time
  cpu_baseline     the CPU oracle (BVH back-end = the reference's algorithm) timed on this box's host cores on a
                   bounded sample, rank 0 at N=1 only.  A baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector (FMA = 2 flop)
PEAK_FP32_NOFMA_TFLOPS = 78.65     # the same pipe issuing add / sub / mul (1 flop per lane per issue): SURVEY 8(d)'s peak for
                                   # the parity build (-ffp-contract=off: the slab and root tests are sub / mul, not fma)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FLOPS_PER_TEST = 20                # faithful ray-sphere test = 20 flop + 1 sqrt
FLOPS_PER_NODE_STEP = 48           # two child-box slab tests: 2 x (6 sub + 6 mul + 12 min / max / compare)
FLOPS_PER_TRI_TEST = 45            # two-sided Moller-Trumbore (mesh.rs:109-161): 2 cross (9), 4 dot (5), 2 sub (3), 1 div


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5", "mesh", "mesh_ref", "c3_ref"],
                    help="BASELINE configs c2-c5, or `mesh`: a generated 100 352-triangle OBJ through the controller's ingest "
                         "rules at the controller's literal 1920x1080 / 20 strips (not a BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", type=int, default=0, metavar="N", help="NOT the contract's measurement: consecutive steps go to N streams "
                    "with N output buffers, so that a launch starts filling the GPU while the previous one drains its "
                    "tail; per-launch HIP-event times then include queueing and the roofline object is not comparable")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) pass: profiling runs "
                    "then see whole-frame launches only")
    ap.add_argument("--no-linear", action="store_true", help="skip the linear-engine launches behind roofline_linear")
    ap.add_argument("--no-others", action="store_true", help="skip the other_workloads passes (c2 / c4 / c5 / mesh, 3 steps each)")
    ap.add_argument("--no-frame", action="store_true", help="skip the frame_path pass (wall-clock of the frame context)")
    ap.add_argument("--strong", action="store_true", help="strong scaling, the controller's split (BASELINE c4 / c5): ONE "
                    "frame, strip d goes to rank d mod N; value = the frame's ray segments / max-rank time")
    ap.add_argument("--frame", action="store_true", help="print ONLY the frame_path object: wall-clock frames of --workload "
                    "through a persistent rt_frame_ctx (host buffer out), --steps frames after the first")
    ap.add_argument("--frame-queue", action="store_true", help="with --frame: the strip-queue assignment (RT_FLAG_FRAME_QUEUE)")
    ap.add_argument("--frame-static", action="store_true", help="with --frame: strip k -> entry k mod n (RT_FLAG_FRAME_STATIC)")
    ap.add_argument("--frame-devices", default="", help="with --frame: comma-separated device ordinals of the frame context "
                    "(default: the rank's device); a device may be listed more than once")
    ap.add_argument("--cpu-scale", type=int, default=1, help="CPU baseline renders the frame at 1/scale resolution")
    ap.add_argument("--flags", type=int, default=0, help="rt_tile_request.flags (1 = exact scan)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / timing reduce (gloo: rehearsal on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: all ranks use GPU 0 (a one-GPU box cannot run RCCL with 2 ranks)")
    return ap.parse_args()


def effective_cpus() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box
    exposes 256 hardware threads but grants a 16-CPU share per GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(workload: str, scale: int):
    """Time the oracle (test infrastructure, used here only as the reported CPU baseline)."""
    from oracle import oracle as orc
    from ray_tracer_s8_amd import scenes
    sph, tri, rq = scenes.config_world(workload)
    rq.width //= scale
    rq.height //= scale
    rq.divisions = 1
    rq.division_no = 0
    threads = effective_cpus()
    _, _, info = orc.render(rq, sph if len(sph) else None, tri if len(tri) else None, backend=1, nthreads=threads)
    secs = info["render_ms"] / 1e3
    return {
        "value": info["ray_segments"] / secs / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"{workload} scene, same spp/depth/seed, full frame at 1/{scale} resolution "
                   f"({rq.width}x{rq.height}); C++ oracle with the reference's SAH-BVH candidate filter, "
                   f"64-pixel spans over all usable host cores; {info['ray_segments']} ray segments in {secs:.2f} s "
                   f"(+{info['bvh_build_ms']:.1f} ms BVH build); omits the Rust slave's per-ray heap "
                   "allocations, so optimistic for the reference"),
        "cpu_seconds": secs * threads,
    }


ASSIGNMENTS = {0: "static: strip k -> entry k mod n", 1: "snake (no costs measured yet)",
               2: "longest-first by the previous frame's per-strip ray segments", 3: "strip queue"}


def frame_path(rt, _abi, scenes, workload: str, dev_index, frames: int = 3, queue: bool = False, static: bool = False):
    """Wall clock of the in-process product path that replaces the controller's dispatch + assembly (controller
    main.rs:47-75, 109-115): rt_frame_ctx_render into a host frame buffer — kernels, strip downloads, dispatcher
    wake-ups, everything — frame after frame of one job.  Frame 1 carries the one-off costs (page-locking the buffer,
    the world's host preparation + upload); the later frames must not."""
    import numpy as np
    sph, tri, rq = scenes.config_world(workload)
    if queue:
        rq.flags |= _abi.RT_FLAG_FRAME_QUEUE
    if static:
        rq.flags |= _abi.RT_FLAG_FRAME_STATIC
    buf = np.zeros(rq.width * rq.height * 3, np.uint8)            # (zeros: the pages exist before the first frame)
    devices = list(dev_index) if isinstance(dev_index, (list, tuple)) else [dev_index]
    t0 = time.perf_counter()
    fc = rt.FrameContext(devices=devices)
    t1 = time.perf_counter()
    fc.set_world(rt.World(sph, tri))
    t2 = time.perf_counter()
    recs = []
    for i in range(1 + max(frames, 1)):
        ta = time.perf_counter()
        _, fs = fc.render(rq, out=buf)
        tb = time.perf_counter()
        recs.append({"wall_ms_python": (tb - ta) * 1e3, "wall_ms": fs.wall_ms, "pin_ms": fs.pin_ms, "scene_ms": fs.scene_ms,
                     "kernel_ms": fs.kernel_ms, "d2h_exposed_ms": fs.d2h_exposed_ms, "host_ms": fs.host_ms,
                     "pinned": int(fs.pinned), "launches": int(fs.totals.n_launches),
                     "segments": int(fs.totals.ray_segments), "assignment": ASSIGNMENTS.get(int(fs.assignment), "?"),
                     "balance": {"max_over_mean_segments": float(fs.balance_max_over_mean),
                                 "per_entry_segments": [int(v) for v in list(fs.entry_segments)[:min(len(devices), 16)]]}})
    fc.close()
    steady = sorted(recs[1:], key=lambda r: r["wall_ms_python"])[len(recs[1:]) // 2]      # median later frame
    out = {"workload": workload, "n_devices": len(devices), "devices": devices,
           "assignment": steady["assignment"],
           "ctx_create_ms": (t1 - t0) * 1e3, "set_world_ms": (t2 - t1) * 1e3,
           "first_frame": recs[0], "steady_frame": steady, "frames_after_first": len(recs) - 1,
           "mrays_per_s": steady["segments"] / (steady["wall_ms_python"] / 1e3) / 1e6,
           "note": "wall clock around rt_frame_ctx_render (host RGB8 frame out); the first frame page-locks the buffer "
                   "(pin_ms) and is charged the world's preparation + upload (scene_ms); later frames of the job pay neither"}
    return out


def other_workload(rt, scenes, workload: str, dev_index: int, stream: int, torch, steps: int = 3):
    """One more BASELINE config through the same device-resident entry point, `steps` timed launches after one
    warm-up, own scene: value / engine / ms, so that every number the README quotes is in the driver's line."""
    sph, tri, rq0 = scenes.config_world(workload)
    strip_bytes = (rq0.height // rq0.divisions) * rq0.width * 3
    out = torch.empty(rq0.divisions * strip_bytes, dtype=torch.uint8, device="cuda")
    reqs = []
    for d in range(rq0.divisions):
        r = rq0.copy()
        r.division_no = d
        reqs.append(r)
    ptrs = [out.data_ptr() + i * strip_bytes for i in range(len(reqs))]
    with rt.Scene(dev_index, rt.World(sph, tri)) as sc:
        sc.render_tiles_device(reqs, ptrs, strip_bytes, stream)
        torch.cuda.synchronize()
        sc.collect()
        t0 = time.perf_counter()
        for _ in range(steps):
            sc.render_tiles_device(reqs, ptrs, strip_bytes, stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = sc.collect()
        # ... and the same frame as a STREAM of frames: consecutive launches on three HIP streams (own output buffers), so that each
        # launch's tail — waves running half empty while the last units finish — is filled by the next launch's first waves.  Not
        # the contract's single-stream figure; what a caller that renders frame after frame gets from the *_device entry points.
        seg_per_frame = float(st.ray_segments) / steps
        extra = [(torch.empty_like(out), torch.cuda.Stream()) for _ in range(2)]
        lanes = [(stream, ptrs)] + [(sb.cuda_stream, [ob.data_ptr() + i * strip_bytes for i in range(len(reqs))]) for ob, sb in extra]
        n3 = 3 * max(steps, 2)
        for st_i, ps_i in lanes:
            sc.render_tiles_device(reqs, ps_i, strip_bytes, st_i)
        torch.cuda.synchronize()
        sc.collect()
        t2 = time.perf_counter()
        for k in range(n3):
            st_i, ps_i = lanes[k % 3]
            sc.render_tiles_device(reqs, ps_i, strip_bytes, st_i)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        sc.collect()
        del extra
    del out
    return {"value": float(st.ray_segments) / (t1 - t0) / 1e6, "unit": "Mrays/s", "ms_per_frame": (t1 - t0) / steps * 1e3,
            "value_3_streams": seg_per_frame * n3 / (t3 - t2) / 1e6, "ms_per_frame_3_streams": (t3 - t2) / n3 * 1e3,
            "kernel_ms_per_launch": st.kernel_ms / max(st.n_launches, 1), "launches": int(st.n_launches), "steps": steps,
            "engine": int(st.engine), "spheres": len(sph), "triangles": len(tri),
            "frame": f"{rq0.width}x{rq0.height}, {rq0.spp} spp, depth {rq0.max_bounces}, {rq0.divisions} strips in one launch"}


def _build_flags_ok() -> bool:
    """False if librt_s8.so was compiled with RT_EXTRA_HIPCC_FLAGS (an experiment's variant): such a line is not the product's."""
    from ray_tracer_s8_amd import build
    return build.built_with_default_flags()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        # not under torchrun: start the ranks as children and exit with their code
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
               str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    import ray_tracer_s8_amd as rt
    from ray_tracer_s8_amd import scenes
    from ray_tracer_s8_amd import _abi
    from ray_tracer_s8_amd.dispatch import job_shards, strips_for_worker

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")

    n_dev = rt.init()
    assert dev_index < n_dev
    if args.frame:
        if world != 1:
            raise SystemExit("--frame is the in-process path: run it with --gpus 1")
        devs = [int(x) for x in args.frame_devices.split(",") if x != ""] or dev_index
        print(json.dumps({"frame_path": frame_path(rt, _abi, scenes, args.workload, devs, args.steps, args.frame_queue, args.frame_static)}), flush=True)
        return
    sph, tri, rq0 = scenes.config_world(args.workload)
    rq0.flags = args.flags
    if args.strong:
        # the controller's split of ONE frame (controller main.rs:47-75): strip d -> rank d mod N
        n_frames = 1
        if rq0.divisions % world:
            raise SystemExit(f"--strong: {rq0.divisions} strips do not divide over {world} ranks")
        units = [(0, d) for d in strips_for_worker(rq0.divisions, rank, world)]
    else:
        n_frames = world                               # weak scaling: one frame of work per GPU
        units = job_shards(n_frames, rq0.divisions, rank, world)
    strip_bytes = (rq0.height // rq0.divisions) * rq0.width * 3
    out = torch.empty(len(units) * strip_bytes, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream   # kernels and torch.cuda.synchronize share it
    scene = rt.Scene(dev_index, rt.World(sph, tri))    # world resident in HBM before timing

    reqs = []
    for (f, d) in units:
        r = rq0.copy()
        r.division_no = d
        r.seed = rq0.seed + f
        reqs.append(r)

    out_ptrs = [out.data_ptr() + i * strip_bytes for i in range(len(reqs))]
    # all strips owned by this rank go out as ONE batched launch per step: a batch may mix division_no and
    # seed (frames), every other field is the frame's (rt_scene_render_tiles_device)
    batches = [(reqs, out_ptrs)]

    step_no = [0]
    if args.overlap:
        lanes, keep = [(stream, out_ptrs)], []
        for _ in range(max(args.overlap, 2) - 1):
            ob, sb = torch.empty_like(out), torch.cuda.Stream()
            keep.append((ob, sb))
            lanes.append((sb.cuda_stream, [ob.data_ptr() + i * strip_bytes for i in range(len(reqs))]))

    def step():
        if args.overlap:
            st_i, ptrs_i = lanes[step_no[0] % len(lanes)]
            step_no[0] += 1
            scene.render_tiles_device(reqs, ptrs_i, strip_bytes, st_i)
            return
        for rs, ps in batches:
            scene.render_tiles_device(rs, ps, strip_bytes, stream)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    scene.collect()                                    # drop warm-up counters / events
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    st = scene.collect()

    elapsed = t1 - t0
    segs = float(st.ray_segments)
    prim = float(st.primary_rays)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([segs, prim], dtype=torch.float64, device=red_dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        segs, prim = float(c[0].item()), float(c[1].item())

    # PCIe-inclusive rate (never `value`): the same frame through the host-buffer entry point
    # (rt_scene_render_tiles: kernel launches + D2H copies into pageable host memory; above 64 MiB the copies of all
    # but the last quarter of the strips run under a second launch; d2h_ms = the exposed part), N = 1 only
    pcie = None
    if rank == 0 and world == 1 and not args.no_pcie:
        bufs, _, _ = scene.render_tiles(reqs)          # first pass touches the pages of the host buffers
        th0 = time.perf_counter()
        _, _, st_h = scene.render_tiles(reqs, out=bufs)
        th1 = time.perf_counter()
        pcie = {"ms_per_frame_host_buffers": (th1 - th0) * 1e3, "kernel_ms": st_h.kernel_ms, "d2h_ms": st_h.d2h_ms,
                "mrays_per_s": float(st_h.ray_segments) / (th1 - th0) / 1e6,
                "scene_h2d_bytes": int(36 * len(sph) + 56 * len(tri))}

    # ---- work census for the roofline object (rank 0, outside the timed region): one more launch of the same
    # frame through the kernel's counting twin (RT_FLAG_COUNT_STEPS: same image, also counts the BVH nodes visited).
    # The render is deterministic in (scene, seed), so the counts are exactly those of every timed launch.
    census = None
    if rank == 0 and st.engine >= 2:
        creqs = []
        for r in reqs:
            c = r.copy()
            c.flags = r.flags | _abi.RT_FLAG_COUNT_STEPS
            creqs.append(c)
        scene.render_tiles_device(creqs, out_ptrs, strip_bytes, stream)
        torch.cuda.synchronize()
        sc_ = scene.collect()
        census = {"segments": int(sc_.ray_segments), "node_steps": int(sc_.node_steps),
                  "root_tests": int(sc_.broad_candidates), "launch_ms_counting_twin": sc_.kernel_ms}
    # ---- the north-star-shaped kernel beside it: linear scan over the LDS-resident primitive list (flags 32), the engine
    # that performs SURVEY 8(d)'s N exact-or-conservative sphere tests per segment.  Own launches, own HIP-event timing.
    linear = None
    if rank == 0 and world == 1 and st.engine >= 2 and len(sph) <= 2048 and not len(tri) and not args.no_linear and not args.overlap:
        lreqs = []
        for r in reqs:
            c = r.copy()
            c.flags = _abi.RT_FLAG_LINEAR_SCAN
            lreqs.append(c)
        scene.render_tiles_device(lreqs, out_ptrs, strip_bytes, stream)     # warm-up
        torch.cuda.synchronize()
        scene.collect()
        for _ in range(max(1, min(args.steps, 3))):
            scene.render_tiles_device(lreqs, out_ptrs, strip_bytes, stream)
        torch.cuda.synchronize()
        sl = scene.collect()
        l_launch_s = sl.kernel_ms / 1e3 / max(sl.n_launches, 1)
        l_tflops = float(sl.ray_segments) / max(sl.n_launches, 1) * FLOPS_PER_TEST * len(sph) / l_launch_s / 1e12
        linear = {"bound": "valu", "kernel": "rtk::rt_tile_kernel<0, expanded> (linear scan, scene resident in LDS)",
                  "achieved": l_tflops, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                  "frac": l_tflops / PEAK_FP32_VALU_TFLOPS, "avg_launch_ms": l_launch_s * 1e3,
                  "launches": int(sl.n_launches), "mrays_per_s": float(sl.ray_segments) / (sl.kernel_ms / 1e3) / 1e6,
                  "work": f"ray segments/launch x {FLOPS_PER_TEST} flop x {len(sph)} spheres (SURVEY 8(d)): this engine tests "
                          "every sphere against every segment"}
        assert 0.0 < linear["frac"] <= 1.0, linear

    # ---- the other BASELINE configs and the mesh workload, 3 launches each after the timed region (rank 0, N = 1)
    others = None
    if rank == 0 and world == 1 and not args.no_others and args.workload == "c3" and not args.flags and not args.overlap:
        others = {}
        # ("c3_again": the headline frame through this function, for its `value_3_streams` — the timed region above is single-stream)
        for w in ("c2", "c4", "c5", "mesh", "mesh_ref", "c3_ref", "c3_again"):
            try:
                others[w] = other_workload(rt, scenes, w.split("_again")[0], dev_index, stream, torch, steps=2 if w.endswith("_ref") else 3)
            except Exception as e:                   # a failure here must not take the headline line with it
                others[w] = {"error": repr(e)}
    # ---- the in-process frame path (rt_frame_ctx: host frame out, dispatcher threads, pinned buffer), wall clock
    fpath = None
    if rank == 0 and world == 1 and not args.no_frame and args.workload == "c3" and not args.flags and not args.overlap:
        fpath = {}
        for w in ("c3", "c4"):
            try:
                fpath[w] = frame_path(rt, _abi, scenes, w, dev_index, 3)
            except Exception as e:
                fpath[w] = {"error": repr(e)}
        # The strip assignment's balance at EIGHT entries, on the one device this run has (eight dispatcher entries on it): how
        # evenly the ray segments fall is a property of the assignment, not of the hardware.  BASELINE c4 (32 strips) and c5 (16).
        bal = {}
        for w in ("c4", "c5"):
            try:
                st_ = frame_path(rt, _abi, scenes, w, [dev_index] * 8, 2, static=True)
                dy_ = frame_path(rt, _abi, scenes, w, [dev_index] * 8, 2)
                bal[w] = {"static_k_mod_n": st_["steady_frame"]["balance"], "first_frame_snake": dy_["first_frame"]["balance"],
                          "later_frames_by_cost": dy_["steady_frame"]["balance"],
                          "assignments": [st_["steady_frame"]["assignment"], dy_["first_frame"]["assignment"], dy_["steady_frame"]["assignment"]]}
            except Exception as e:
                bal[w] = {"error": repr(e)}
        fpath["balance_8_entries_on_one_device"] = bal
    # ---- N > 1: BASELINE config 4 as the controller splits it — ONE c4 frame, strip d -> rank d mod N, max-rank time —
    # beside the weak-scaling value of the line, so that a scaling run records it at every N
    strong_c4 = None
    if world > 1 and not args.strong and args.workload == "c3" and not args.no_others:
        s4, t4, r4 = scenes.config_world("c4")
        if r4.divisions % world == 0:
            mine = strips_for_worker(r4.divisions, rank, world)
            sb4 = (r4.height // r4.divisions) * r4.width * 3
            out4 = torch.empty(len(mine) * sb4, dtype=torch.uint8, device="cuda")
            rq4 = []
            for d_ in mine:
                r_ = r4.copy()
                r_.division_no = d_
                rq4.append(r_)
            p4 = [out4.data_ptr() + i * sb4 for i in range(len(rq4))]
            with rt.Scene(dev_index, rt.World(s4, t4)) as sc4:
                sc4.render_tiles_device(rq4, p4, sb4, stream)
                torch.cuda.synchronize()
                sc4.collect()
                barrier()
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(3):
                    sc4.render_tiles_device(rq4, p4, sb4, stream)
                torch.cuda.synchronize()
                barrier()
                tb = time.perf_counter()
                st4 = sc4.collect()
            tt = torch.tensor([tb - ta], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cc = torch.tensor([float(st4.ray_segments)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(cc, op=dist.ReduceOp.SUM)
            strong_c4 = {"value": float(cc.item()) / float(tt.item()) / 1e6, "unit": "Mrays/s", "scaling": "strong",
                         "ms_per_frame": float(tt.item()) / 3 * 1e3, "steps": 3, "n_gpus": world,
                         "strips_per_rank": len(mine),
                         "workload": f"c4: ONE {r4.width}x{r4.height} / {r4.spp} spp frame, {r4.divisions} strips, strip d -> rank d mod {world} "
                                     "(controller main.rs:47-75), max-rank time"}
            del out4

    # ---- N > 1: the IN-PROCESS product path over all N devices (one rt_frame_ctx in rank 0, one dispatcher thread per GPU; the
    # other ranks wait at the barrier, their GPUs are idle): c4 frames into a host buffer, wall clock — the controller's
    # dispatch + assembly replaced in one process, timed on as many GPUs as the line covers
    fpath_all = None
    if world > 1 and not args.strong and args.workload == "c3" and not args.no_frame:
        barrier()
        if rank == 0:
            # in a CHILD process (this script with --frame): nothing that goes wrong there — this path has never run on more
            # than one physical device — can take the scaling line with it
            devs = ",".join(str(dev_index if args.share_device else d) for d in range(world))
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
                                "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")}
            fpath_all = {}
            # (three assignments: the default — snake on frame 1, then longest-first by the measured per-strip costs —, the round 1-3
            # split k mod n, and the strip queue; each child at most two minutes, well inside the barrier's collective timeout)
            for name, extra in (("balanced", []), ("static_k_mod_n", ["--frame-static"]), ("strip_queue", ["--frame-queue"])):
                try:
                    pr = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--frame", "--workload", "c4", "--steps", "3",
                                         "--frame-devices", devs] + extra, capture_output=True, text=True, timeout=120, env=env)
                    last = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
                    fpath_all[name] = json.loads(last[-1])["frame_path"] if pr.returncode == 0 and last else \
                        {"error": f"rc {pr.returncode}: {pr.stderr[-400:]}"}
                except Exception as e:
                    fpath_all[name] = {"error": repr(e)}
        barrier()

    if rank == 0:
        n_sph = len(sph)
        launches = max(st.n_launches, 1)
        avg_launch_s = st.kernel_ms / 1e3 / launches
        segs_per_launch = float(st.ray_segments) / launches
        algorithmic_equiv_tflops = segs_per_launch * FLOPS_PER_TEST * n_sph / avg_launch_s / 1e12
        if st.engine >= 2:
            # executed work of the traversal engines: per internal node visited two child-box slab tests
            # (2 x (6 sub + 6 mul + 12 min / max / compare) = 48 flop), per leaf reached one exact root test (20 flop + sqrt)
            scale = segs_per_launch / max(census["segments"], 1)          # launches of the step = census launch (1.0)
            per_root = FLOPS_PER_TRI_TEST if (len(tri) and not len(sph)) else FLOPS_PER_TEST
            flops_per_launch = (census["node_steps"] * FLOPS_PER_NODE_STEP + census["root_tests"] * per_root) * scale
            work = {"node_steps_per_segment": census["node_steps"] / max(census["segments"], 1),
                    "root_tests_per_segment": census["root_tests"] / max(census["segments"], 1),
                    "flops_per_node_step": FLOPS_PER_NODE_STEP, "flops_per_root_test": per_root,
                    "source": "counting twin of the timed kernel, one extra launch of the same frame in this run"}
        else:
            flops_per_launch = segs_per_launch * FLOPS_PER_TEST * n_sph
            work = {"tests_per_segment": n_sph, "flops_per_test": FLOPS_PER_TEST,
                    "source": "SURVEY 8(d): the linear engine tests every sphere against every segment"}
        achieved_tflops = flops_per_launch / avg_launch_s / 1e12
        # the slab and root tests are sub / mul / compare — no fma under -ffp-contract=off — so the pipe's rate for them is
        # one flop per lane per issue: SURVEY 8(d)'s non-FMA peak.  (The linear engines' broad phase IS fma: full peak.)
        peak_tflops = PEAK_FP32_NOFMA_TFLOPS if st.engine >= 2 else PEAK_FP32_VALU_TFLOPS
        hbm_bytes_per_launch = strip_bytes * len(reqs) / max(len(batches), 1) + 36 * n_sph + 56 * len(tri)   # RGB8 out + scene in

        def committed(name):
            fp = ROOT / "profiles" / name
            try:
                d = json.loads(fp.read_text()).get(args.workload)
            except Exception:
                return None
            if isinstance(d, dict):
                d = dict(d)
                d["source"] = f"committed profile profiles/{name} (rocprofv3 --pmc pass of this command, not measured in this run)"
            return d
        traffic_rec = committed("r04_hbm_traffic.json") or committed("r03_hbm_traffic.json")
        issue = committed("r04_valu_issue.json") or committed("r03_valu_issue.json")
        eng_names = ["linear scan, scene resident in LDS", "linear scan, scene streamed through LDS",
                     "per-lane traversal of the reference BVH (exact nodes gathered from L2)",
                     "per-lane traversal of the reference BVH (quantised nodes, exact leaf validation)",
                     "per-lane traversal of the reference BVH (exact nodes resident in LDS)",
                     "per-lane traversal of the reference BVH (quantised nodes, nearer child first, distance culling, exact leaf validation)",
                     "per-lane traversal of the reference BVH (exact nodes gathered from L2, nearer child first, distance culling)",
                     "per-lane traversal of the reference BVH (exact nodes resident in LDS, nearer child first, distance culling)"]
        line = {
            "metric": "Mrays/sec @ 4K/8spp 1024-sphere" if args.workload == "c3" else f"Mrays/sec @ {args.workload}",
            "value": segs / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {n_sph} spheres, {len(tri)} triangles, {rq0.width}x{rq0.height}, {rq0.spp} spp, "
                            f"depth {rq0.max_bounces}, {rq0.divisions} strips/frame, "
                            + (f"ONE frame, strip d -> rank d mod {world} (controller split)" if args.strong else
                               f"{n_frames} frame(s) sharded by strip over {world} GPU(s)"),
                "rays": "ray segments (primary + secondary closest-hit queries)",
                "mprimary_per_s": prim / elapsed / 1e6,
                "segments_per_primary": segs / prim,
                "engine": eng_names[st.engine],
                "flags": args.flags,
                "default_build_flags": _build_flags_ok(),
                "overlapped_steps": bool(args.overlap),
                "pcie_inclusive": pcie,
                "broad_candidates_per_segment": float(st.broad_candidates) / max(float(st.ray_segments), 1.0),
                "exact_fallbacks": int(st.exact_fallbacks),
            },
            "roofline": {
                "bound": "valu",
                "achieved": achieved_tflops,
                "peak": peak_tflops,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / peak_tflops,
                # (rounds 1-2 divided the traversal engines' flops by the FMA peak, 157.3 T: the same achieved figure on that scale,
                # so that lines of different rounds can be compared — round-3 advisor)
                "frac_fma_peak": achieved_tflops / PEAK_FP32_VALU_TFLOPS,
                "peak_note": "non-FMA FP32 vector peak (sub / mul / compare, 1 flop per lane per issue; SURVEY 8(d) parity mode)"
                             if st.engine >= 2 else "FP32 vector peak, FMA = 2 flop (the broad phase is packed fma)",
                # SURVEY 8(d)'s own figure, as named there: ray_segments x 20 x N / t / 78.6e12.  For the traversal engines
                # it exceeds 1 — they do O(log N) tests per segment, not N — so it is NOT a fraction of the hardware; it
                # says how many times faster than a perfect linear scan at that peak the launch was.
                "valu_fraction_8d": algorithmic_equiv_tflops / PEAK_FP32_NOFMA_TFLOPS,
                "traffic": traffic_rec.get("bytes_per_launch") if isinstance(traffic_rec, dict) else None,
                "traffic_source": traffic_rec.get("source") if isinstance(traffic_rec, dict) else None,
                "kernel": "rtk::rt_tile_kernel (" + eng_names[st.engine] + ")",
                "avg_launch_ms": avg_launch_s * 1e3,
                "launches": launches,
                "executed_flops_per_launch": flops_per_launch,
                "work": work,
                "note": "SURVEY 8(d): neither HBM nor MFMA binds this path; the FP32 vector pipe does.  achieved = the flops "
                        "of the slab and root tests the timed engine executes (counted, not modelled) / HIP-event launch "
                        "time; the rest of the kernel's instructions (control, LDS, 64-bit integer RNG, IEEE div/sqrt "
                        "expansions) are not flops and are not counted",
                # NOT a hardware fraction: what a linear scan would have to sustain to match this launch time
                "algorithmic_equiv_tflops": algorithmic_equiv_tflops,
                # committed PMC pass of this workload: VALU issue share of the SIMD cycles, active lanes per instruction
                "valu_issue": issue,
                "hbm": {
                    "algorithmic_bytes_per_launch": hbm_bytes_per_launch,
                    "achieved_GBs": hbm_bytes_per_launch / avg_launch_s / 1e9,
                    "peak_GBs": PEAK_HBM_GBS,
                    "frac": hbm_bytes_per_launch / avg_launch_s / 1e9 / PEAK_HBM_GBS,
                },
            },
            "roofline_linear": linear,
            "other_workloads": others,
            "frame_path": fpath,
            "strong_c4": strong_c4,
            "frame_path_all_devices": fpath_all,
        }
        if st.engine in (2, 3, 5, 6):
            # The L2-gather walks are bound by their node gathers, not by flops: beside the FP32 object, the
            # node records fetched per second against the chip's rate for fully divergent gathers of that record size,
            # measured by tools/ubench/gather_rate.hip (committed: profiles/r02_gather_rate.json).  Lanes that share a line
            # are cheaper than the microbenchmark's, so this fraction is an estimate and is not asserted <= 1.
            try:
                gr = json.loads((ROOT / "profiles" / "r02_gather_rate.json").read_text())
                key, nbytes, ngath = ("64B_4_gathers", 64, 4) if st.engine in (2, 6) else ("32B_2_gathers", 32, 2)
                peak_rec = gr["lane_records_per_cycle_per_cu"][key]["64_lanes"] * gr["cus"] * gr["clock_mhz"] * 1e6
                ach_rec = census["node_steps"] * scale / avg_launch_s
                line["roofline_gather"] = {
                    "bound": "divergent gathers (texture-address / L1 path)", "unit": "G node records/s",
                    "achieved": ach_rec / 1e9, "peak": peak_rec / 1e9, "frac": ach_rec / peak_rec,
                    "record_bytes": nbytes, "gather_instructions_per_record": ngath,
                    "peak_source": "committed microbenchmark profiles/r02_gather_rate.json (tools/ubench/gather_rate.hip: every "
                                   "lane on its own line, 2 MiB table, 5 waves per SIMD), not measured in this run"}
            except Exception as e:                       # the file is part of the repo; a bench line without it still stands
                line["roofline_gather"] = {"error": repr(e)}
        assert 0.0 < line["roofline"]["frac"] <= 1.0, line["roofline"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_scale)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
