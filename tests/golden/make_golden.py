#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (BVH back-end = reference semantics, and linear).

The reference (Rust) cannot run in this image, so these vectors are outputs of the
line-traceable restatement, cross-checked by oracle/restate_np.py — they pin regressions of
the oracle and give the GPU tests committed bytes to match.  Re-run only when the normative
spec changes:  python tests/golden/make_golden.py
"""
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import oracle  # noqa: E402
from ray_tracer_s8_amd import _abi, scenes  # noqa: E402

from _world_cases import interleave, tie_world  # noqa: E402

HERE = Path(__file__).resolve().parent


def cases():
    sph, rq = scenes.config("c1")                      # BASELINE c1 at full size: 256x256, 1 spp
    yield "c1_single_sphere_256", rq, sph, None, True, None
    sph, rq = scenes.config("c2")
    rq.width, rq.height, rq.divisions, rq.division_no = 96, 54, 3, 1
    yield "c2_cornell_96x54_strip1of3", rq, sph, None, True, None
    sph, rq = scenes.config("c3")
    rq.width, rq.height, rq.divisions, rq.spp = 96, 54, 1, 4
    yield "c3_rand1024_96x54", rq, sph, None, True, None
    sph, tri = scenes.quad_room()
    rq = _abi.default_request(width=80, height=48, divisions=1, spp=4, max_bounces=5, seed=5)
    yield "quad_room_80x48", rq, sph, tri, True, None
    sph = scenes.rand65536(n=9000)
    rq = _abi.default_request(width=64, height=40, divisions=1, spp=2, max_bounces=4, seed=99)
    yield "rand9000_streamed_64x40", rq, sph, None, True, None
    # the order of `world` (ABI v3): interleaved spheres and triangles, identical copies, every hit an exact tie; the arrays
    # and the world_index are stored with the vector (the scene is data of the fixture, not a generator call)
    sph, tri = tie_world(5)
    rq = _abi.default_request(width=96, height=64, divisions=1, spp=3, max_bounces=4, seed=17)
    yield "world_order_ties_96x64", rq, sph, tri, True, interleave(len(sph), len(tri), 41)
    # larger frames: only a checksum is stored
    sph, rq = scenes.config("c2")
    rq.width, rq.height, rq.divisions = 480, 270, 1
    yield "c2_cornell_480x270_sha", rq, sph, None, False, None


def req_fields(rq):
    return {k: getattr(rq, k) for k, _ in rq._fields_}


def main():
    only = sys.argv[1:]                                  # names to (re)generate; default: all
    for name, rq, sph, tri, store, wi in cases():
        if only and name not in only:
            continue
        # reference semantics = BVH candidate filter (backend 1); plain linear scan (backend 0) also pinned
        rgb, f32, info = oracle.render(rq, sph, tri, backend=1, want_f32=True, world_index=wi)
        lin, lin_f, info_l = oracle.render(rq, sph, tri, backend=0, want_f32=True, world_index=wi)
        sha = hashlib.sha256(rgb.tobytes()).hexdigest()
        sha_f = hashlib.sha256(f32.tobytes()).hexdigest()
        out = dict(sha256_rgb_linear=np.array(hashlib.sha256(lin.tobytes()).hexdigest()),
                   sha256_f32_linear=np.array(hashlib.sha256(lin_f.tobytes()).hexdigest()),
                   ray_segments_linear=np.array(info_l["ray_segments"], dtype=np.uint64),
                   request=np.array([tuple(req_fields(rq).values())],
                                    dtype=[(k, "f8" if isinstance(v, float) else "u8") for k, v in req_fields(rq).items()]),
                   sha256_rgb=np.array(sha), sha256_f32=np.array(sha_f), ray_segments=np.array(info["ray_segments"], dtype=np.uint64))
        if store:
            out["rgb"] = rgb
        if wi is not None:
            out["world_index"] = np.asarray(wi, np.uint32)
            out["spheres"] = np.ascontiguousarray(sph).view(np.float32).reshape(len(sph), 9)
            out["triangles"] = np.ascontiguousarray(tri).view(np.float32).reshape(len(tri), 14)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(name, rgb.size, sha[:16], info["ray_segments"])


if __name__ == "__main__":
    main()
