"""Worker for tests/test_distributed_gloo.py: the N>1 sharding path on CPU (gloo).

Each rank takes its units from dispatch.job_shards, renders them (the CPU oracle stands in for
the GPU renderer — this is a test), then the results are gathered ONLY for verification: the
data path itself has no collective.  Also exercises bench.py's MAX-time / SUM-segments reduce."""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402
from ray_tracer_s8_amd import dispatch, scenes  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    strong = len(sys.argv) > 1 and sys.argv[1] == "strong"
    sph, rq0 = scenes.config("c2")
    rq0.width, rq0.height, rq0.divisions, rq0.spp = 64, 48, 6, 2
    if strong:
        # bench.py --strong = the controller's split of ONE frame (BASELINE c4 / c5): strip d -> rank d mod N
        n_frames = 1
        units = [(0, d) for d in dispatch.strips_for_worker(rq0.divisions, rank, world)]
    else:
        n_frames = world
        units = dispatch.job_shards(n_frames, rq0.divisions, rank, world)
    t0 = time.perf_counter()
    mine, segs = [], 0
    for f, d in units:
        r = rq0.copy()
        r.division_no, r.seed = d, rq0.seed + f
        rgb, _, info = oracle.render(r, sph, nthreads=1)
        mine.append((f, d, rgb))
        segs += info["ray_segments"]
    elapsed = time.perf_counter() - t0 + 0.01 * rank          # make the ranks' times differ
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(segs)], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    times = [None] * world
    dist.all_gather_object(times, elapsed)
    if rank == 0:
        assert abs(t.item() - max(times)) < 1e-12
        total = 0
        for f in range(n_frames):
            slices = [(d, rgb) for part in gathered for (ff, d, rgb) in part if ff == f]
            img = dispatch.assemble(slices, rq0.width, rq0.height, rq0.divisions)
            whole = rq0.copy()
            whole.divisions, whole.division_no, whole.seed = 1, 0, rq0.seed + f
            ref, _, info = oracle.render(whole, sph, nthreads=1)
            assert np.array_equal(img.reshape(-1), ref), f"frame {f} differs"
            total += info["ray_segments"]
        assert int(c.item()) == total
        owners = sorted(len(p) for p in gathered)
        if strong:
            assert owners == [rq0.divisions // world] * world  # strong scaling: the frame's strips split evenly
            assert sorted(d for part in gathered for (_, d, _) in part) == list(range(rq0.divisions))
        else:
            assert owners == [rq0.divisions] * world          # weak scaling: equal work per rank
        print(f"GLOO_OK world={world} frames={n_frames} segments={total} mode={'strong' if strong else 'weak'}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
