#!/usr/bin/env python3
"""Linear scan or LDS tree for PILES of overlapping spheres above the 32-sphere threshold?  Kernel Mrays/s of both engines for 32 ... 256
spheres in a small volume (two box sizes) and for small spheres in the Cornell room, depth 4 and 8, and the engine the host picks.
usage (GPU box): python3 tools/dense_matrix.py"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import numpy as np
from small_scene_matrix import field, run, density
from ray_tracer_s8_amd import scenes

g = np.random.default_rng(11)
print(f"{'scene':18s}   depth 4: linear / tree (ratio)      depth 8: linear / tree (ratio)    default engine")
for n in (32, 48, 64, 96, 128, 192, 256):
    for name, s in (("dense", field(n, g, [-2, -1, -8], [2, 2, -4], (0.4, 0.9))),
                    ("dense wide", field(n, g, [-4, -1, -12], [4, 3, -4], (0.4, 0.9))),
                    ("room+", np.concatenate([scenes.cornell16()[:6], field(n - 6, g, [-1.4, -1.8, -5.5], [1.4, 0.5, -2.5], (0.1, 0.3), ground=False)]))):
        (l4, t4), e = run(s, 4)
        (l8, t8), _ = run(s, 8)
        print(f"{name + ' ' + str(n):18s}   {l4:8.0f} / {t4:8.0f} ({t4 / l4:.2f})          {l8:8.0f} / {t8:8.0f} ({t8 / l8:.2f})    {e}", flush=True)
