#!/usr/bin/env python3
"""Linear scan or LDS tree for SMALL scenes (2 ... 32 spheres)?  Kernel Mrays/s of both engines across scene families, with the
host's leaf density (sum of the primitive boxes' areas / area of the scene box) beside them.  1080p, 4 spp, depth 4 and 8.
usage (GPU box): python3 tools/small_scene_matrix.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
rt.init()
F = _abi

def field(n, g, lo, hi, rr, ground=True):
    s = np.zeros(n, F.SPHERE_DTYPE)
    c = g.uniform(lo, hi, (n, 3))
    s["cx"], s["cy"], s["cz"], s["radius"] = c[:, 0], c[:, 1], c[:, 2], g.uniform(rr[0], rr[1], n)
    if ground:
        s["cx"][0], s["cy"][0], s["cz"][0], s["radius"][0] = 0.0, -1001.0, -20.0, 1000.0
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        s[ch] = g.uniform(0.1, 0.95, n)
    s["roughness"] = g.choice([0.0, 0.0, 0.4, 1.0], n)
    s["emission"] = np.where(g.uniform(size=n) < 0.1, g.uniform(2, 6, n), 0.0)
    return s

def density(s):
    lo = np.stack([s["cx"] - s["radius"], s["cy"] - s["radius"], s["cz"] - s["radius"]], 1).astype(np.float64)
    hi = np.stack([s["cx"] + s["radius"], s["cy"] + s["radius"], s["cz"] + s["radius"]], 1).astype(np.float64)
    e = hi - lo
    area = (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0]).sum()
    E = hi.max(0) - lo.min(0)
    return area / (E[0] * E[1] + E[1] * E[2] + E[2] * E[0])

def run(sph, depth):
    rq = F.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=depth, seed=5)
    reqs = []
    for k in range(4):
        r = rq.copy(); r.division_no = k; reqs.append(r)
    out = []
    with rt.Scene(0, rt.World(sph)) as sc:
        for fl in (F.RT_FLAG_LINEAR_SCAN, F.RT_FLAG_BVH_TRAVERSE):
            for r in reqs: r.flags = fl
            sc.render_tiles(reqs)
            best = 1e9
            for _ in range(4):
                _, _, st = sc.render_tiles(reqs)
                best = min(best, st.kernel_ms)
            out.append(st.ray_segments / best / 1e3)
        for r in reqs: r.flags = 0
        _, _, st = sc.render_tiles(reqs)
    return out, st.engine

if __name__ == "__main__":
    g = np.random.default_rng(7)
    room = scenes.cornell16()
    print(f"{'scene':18s} {'density':>8s}   depth 4: linear / tree (ratio)      depth 8: linear / tree (ratio)    default engine")
    for n in (2, 4, 8, 12, 16, 24, 32):
        fams = [("field", field(n, g, [-24, -1, -48], [24, 10, -3], (0.15, 0.6))),
                ("field no ground", field(n, g, [-6, -2, -16], [6, 4, -4], (0.3, 0.9), ground=False)),
                ("dense", field(n, g, [-2, -1, -8], [2, 2, -4], (0.4, 0.9))),
                ("sheet", field(n, g, [-8, 1.0, -20], [8, 1.05, -4], (0.1, 0.4)))]
        if n <= 16:
            fams.append(("room", room[:n]))
        for name, s in fams:
            (l4, t4), e = run(s, 4)
            (l8, t8), _ = run(s, 8)
            print(f"{name + ' ' + str(n):18s} {density(s):8.2f}   {l4:8.0f} / {t4:8.0f} ({t4 / l4:.2f})          {l8:8.0f} / {t8:8.0f} ({t8 / l8:.2f})    {e}", flush=True)
