import sys
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi
from oracle import oracle as orc
rt.init()
F = _abi
bad = used = 0
for case in range(40):
    g = np.random.default_rng(7000 + case)
    n = int(g.choice([2000, 8000]))
    L, Wd = float(g.choice([2.0, 6.0, 15.0])), float(g.choice([1e-4, 1e-3, 1e-2]))
    t = np.zeros(n, F.TRIANGLE_DTYPE)
    a = g.uniform([-10, -3, -40], [10, 6, -3], (n, 3))
    dirv = g.normal(size=(n, 3)); dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
    side = g.normal(size=(n, 3)); side -= (side * dirv).sum(1, keepdims=True) * dirv; side /= np.linalg.norm(side, axis=1, keepdims=True)
    t["a"], t["b"], t["c"] = a, a + dirv * L * g.uniform(0.3, 1.0, (n, 1)), a + side * Wd
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        t[ch] = g.uniform(0.2, 0.9, n)
    t["roughness"] = g.choice([0.0, 1.0], n)
    t["emission"] = np.where(g.uniform(size=n) < 0.05, 3.0, 0.0)
    rq = F.default_request(width=160, height=90, divisions=1, spp=2, max_bounces=4, seed=case, t_max=500.0)
    ref, _, info = orc.render(rq, None, t, backend=1)
    r = rq.copy(); r.flags = F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK
    with rt.Scene(0, rt.World(np.zeros(0, F.SPHERE_DTYPE), t)) as sc:
        rgb, _, st = sc.render_tile(r)
    ok = bool(np.array_equal(rgb, ref)) and st.ray_segments == info["ray_segments"]
    used += st.engine == 6
    bad += 0 if ok else 1
    if not ok or case % 8 == 0:
        print(f"case {case}: {n} slivers {L} x {Wd} engine {st.engine} segments {st.ray_segments} exact {ok}", flush=True)
print(f"40 cases, {used} through engine 6:", "FAILED" if bad else "all exact")
