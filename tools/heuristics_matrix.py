#!/usr/bin/env python3
"""Engine choice across scene FAMILIES (not only the rand-field generators the constants in rt_api.hip were tuned on):
Mrays/s of every engine that can take the scene, and which one the host heuristics pick.  1080p, 4 spp, depth 6.
usage (GPU box): python3 tools/heuristics_matrix.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes

rt.init()


def field(n, g, lo, hi, rr, ground=True):
    s = np.zeros(n, _abi.SPHERE_DTYPE)
    c = g.uniform(lo, hi, (n, 3))
    s["cx"], s["cy"], s["cz"], s["radius"] = c[:, 0], c[:, 1], c[:, 2], g.uniform(rr[0], rr[1], n)
    if ground:
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


g = np.random.default_rng(2026)
cases = [("cornell16", scenes.cornell16(), None)]
for n in (48, 200, 1000):
    cases.append((f"field {n}", field(n, g, [-24, -1, -48], [24, 10, -3], (0.15, 0.6)), None))
    cases.append((f"dense {n}", field(n, g, [-4, -1, -14], [4, 4, -3], (0.2, 0.6)), None))
    cases.append((f"sheet {n}", field(n, g, [-25, 1.0, -60], [25, 1.05, -4], (0.05, 0.2)), None))
    cases.append((f"clusters {n}", clusters(n, g), None))
for n in (3000, 12000, 40000):
    cases.append((f"field {n}", field(n, g, [-60, -1, -120], [60, 20, -3], (0.1, 0.5)), None))
    cases.append((f"dense {n}", field(n, g, [-6, -1, -20], [6, 5, -4], (0.2, 0.6)), None))
    cases.append((f"clusters {n}", clusters(n, g), None))
for nx in (8, 20, 60, 224):
    cases.append((f"mesh {2 * nx * nx} tris", None, scenes.mesh_world(nx, nx)))

F = _abi
engines = [("linear", F.RT_FLAG_LINEAR_SCAN), ("L2 exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE),
           ("L2 quant", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES), ("LDS tree", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES),
           ("default", 0)]
names = {0: "linear", 1: "linear(streamed)", 2: "L2 exact", 3: "L2 quant", 4: "LDS tree", 5: "L2 quant culled", 6: "L2 exact culled", 7: "LDS tree culled"}
print(f"{'scene':22s} " + " ".join(f"{e[0]:>10s}" for e in engines) + "   default picks / best")
for name, sph, tri in cases:
    world = rt.World(sph if sph is not None else np.zeros(0, F.SPHERE_DTYPE), tri if tri is not None else np.zeros(0, F.TRIANGLE_DTYPE))
    n = len(world.spheres) + len(world.triangles)
    row, picked, best = [], None, (0.0, "")
    with rt.Scene(0, world) as sc:
        for ename, fl in engines:
            if ename == "linear" and n > 6000:
                row.append("-")
                continue
            rq = F.default_request(width=1920, height=1080, divisions=4, spp=4, max_bounces=6, seed=5, flags=fl)
            reqs = []
            for k in range(4):
                r = rq.copy(); r.division_no = k; reqs.append(r)
            sc.render_tiles(reqs)
            _, _, st = sc.render_tiles(reqs)
            v = st.ray_segments / st.kernel_ms / 1e3
            if ename == "default":
                picked = names[st.engine]
            elif (ename != "LDS tree" or st.engine in (4, 7)) and v > best[0]:
                best = (v, names[st.engine])
            row.append(f"{v:.0f}" + ("" if ename != "LDS tree" or st.engine in (4, 7) else "*"))
    print(f"{name:22s} " + " ".join(f"{c:>10s}" for c in row) + f"   {picked} / {best[1]}")
print("(* = the tree does not fit LDS: the L2-gather engine ran)")
