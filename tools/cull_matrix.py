#!/usr/bin/env python3
"""Quantised walk vs culled walk (RT_FLAG_CULL_WALK) on the sphere-scene families the quantised walk is the default for.
2560x1440, 4 spp, depth 6.  usage (GPU box): python3 tools/cull_matrix.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
sys.path.insert(0, "tools")

rt.init()
F = _abi


def field(n, g, lo, hi, rr):
    s = np.zeros(n, F.SPHERE_DTYPE)
    c = g.uniform(lo, hi, (n, 3))
    s["cx"], s["cy"], s["cz"], s["radius"] = c[:, 0], c[:, 1], c[:, 2], g.uniform(rr[0], rr[1], n)
    s["cx"][0], s["cy"][0], s["cz"][0], s["radius"][0] = 0.0, -1001.0, -20.0, 1000.0
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        s[ch] = g.uniform(0.1, 0.95, n)
    s["roughness"] = g.choice([0.0, 0.0, 0.4, 1.0], n)
    s["emission"] = np.where(g.uniform(size=n) < 0.03, g.uniform(2, 6, n), 0.0)
    return s


def clusters(n, g):
    k = g.uniform([-30, 0, -70], [30, 10, -5], (24, 3))
    s = field(n, g, [0, 0, 0], [1, 1, 1], (0.02, 0.15))
    c = k[g.integers(0, 24, n)] + g.normal(0, 0.8, (n, 3))
    s["cx"][1:], s["cy"][1:], s["cz"][1:] = c[1:, 0], c[1:, 1], c[1:, 2]
    return s


def main():
    g = np.random.default_rng(2027)
    cases = []
    for n in (4096, 16384, 65536):
        cases.append((f"rand field {n} (c5 recipe)", scenes.rand65536(n=n)))
    for n in (6000, 40000, 150000):
        cases.append((f"field {n}", field(n, g, [-60, -1, -120], [60, 20, -3], (0.1, 0.5))))
        cases.append((f"dense {n}", field(n, g, [-6, -1, -20], [6, 5, -4], (0.2, 0.6))))
        cases.append((f"clusters {n}", clusters(n, g)))
        cases.append((f"sheet {n}", field(n, g, [-25, 1.0, -60], [25, 1.05, -4], (0.02, 0.1))))
        cases.append((f"mixed radii {n}", field(n, g, [-60, -1, -120], [60, 20, -3], (0.02, 2.0))))
    engines = [("quantised", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES), ("culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_CULL_WALK),
               ("exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE)]
    print(f"{'scene':30s} " + " ".join(f"{e[0]:>10s}" for e in engines) + "   culled / quantised")
    for name, sph in cases:
        row = []
        with rt.Scene(0, rt.World(sph)) as sc:
            for ename, fl in engines:
                rq = F.default_request(width=2560, height=1440, divisions=4, spp=4, max_bounces=6, seed=5, flags=fl)
                reqs = []
                for k in range(4):
                    r = rq.copy(); r.division_no = k; reqs.append(r)
                sc.render_tiles(reqs)
                _, _, st = sc.render_tiles(reqs)
                row.append((st.ray_segments / st.kernel_ms / 1e3, st.engine))
        print(f"{name:30s} " + " ".join(f"{v:8.0f}/{e}" for v, e in row) + f"   {row[1][0] / row[0][0]:.3f}", flush=True)


if __name__ == "__main__":
    main()
