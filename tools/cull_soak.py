#!/usr/bin/env python3
"""Culled walk against the oracle on c5-recipe scenes of several sizes, seeds and depths (1280x720, 4 spp), plus dense fields."""
import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes
from oracle import oracle as orc
rt.init()
F = _abi
bad = 0
cases = [(scenes.rand65536(n=n, seed=0x5EED1000 + k), k) for k, n in enumerate((65536, 30000, 12000, 65536, 120000, 8000))]
g = np.random.default_rng(99)
for k, n in enumerate((9000, 20000)):
    s = scenes.rand65536(n=n, seed=77 + k)
    s["cx"] *= 0.08; s["cy"] *= 0.2; s["cz"] = -3 + (s["cz"] + 3) * 0.1            # squeezed: dense overlap
    cases.append((s, 10 + k))
for sph, k in cases:
    rq = F.default_request(width=1280, height=720, divisions=1, spp=4, max_bounces=3 + 2 * (k % 4), seed=1000 + k)
    ref, _, info = orc.render(rq, sph, backend=1)
    for fl in (F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_CULL_WALK, 0):
        r = rq.copy(); r.flags = fl
        with rt.Scene(0, rt.World(sph)) as sc:
            rgb, _, st = sc.render_tile(r)
        ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
        bad += 0 if ok else 1
        print(f"n={len(sph):6d} depth {rq.max_bounces} flags {fl:5d} engine {st.engine} segments {st.ray_segments:9d} exact {ok}", flush=True)
print("FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
