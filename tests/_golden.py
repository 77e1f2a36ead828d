"""Loader for tests/golden/*.npz: rebuilds (request, scene) for each committed vector."""
from pathlib import Path

import numpy as np

from ray_tracer_s8_amd import _abi, scenes

GOLDEN = Path(__file__).resolve().parent / "golden"


def _scene(name):
    if name.startswith("c1"):
        return scenes.single_sphere(), None
    if name.startswith("c2"):
        return scenes.cornell16(), None
    if name.startswith("c3"):
        return scenes.rand1024(), None
    if name.startswith("quad_room"):
        return scenes.quad_room()
    if name.startswith("rand9000"):
        return scenes.rand65536(n=9000), None
    raise KeyError(name)


def load_all():
    out = []
    for f in sorted(GOLDEN.glob("*.npz")):
        z = np.load(f, allow_pickle=False)
        rec = z["request"][0]
        rq = _abi.TileRequest()
        for k in rec.dtype.names:
            v = rec[k]
            setattr(rq, k, float(v) if rec.dtype[k].kind == "f" else int(v))
        wi = None
        if "world_index" in z.files:                     # a vector that carries its own world (arrays + the list's order)
            sph = np.ascontiguousarray(z["spheres"]).reshape(-1).view(_abi.SPHERE_DTYPE)
            tri = np.ascontiguousarray(z["triangles"]).reshape(-1).view(_abi.TRIANGLE_DTYPE)
            wi = z["world_index"].astype(np.uint32)
        else:
            sph, tri = _scene(f.stem)
        out.append(dict(name=f.stem, req=rq, spheres=sph, triangles=tri, world_index=wi,
                        rgb=z["rgb"] if "rgb" in z.files else None,
                        sha256_rgb=str(z["sha256_rgb"]), sha256_f32=str(z["sha256_f32"]),
                        ray_segments=int(z["ray_segments"]),
                        sha256_rgb_linear=str(z["sha256_rgb_linear"]), sha256_f32_linear=str(z["sha256_f32_linear"]),
                        ray_segments_linear=int(z["ray_segments_linear"])))
    return out
