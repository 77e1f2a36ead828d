"""The host's engine rules (rt_api.hip launch_batch: which of the seven closest-hit engines renders a scene) on scenes they were NOT
tuned on.  Round-3 verdict: the density / size constants were fitted to the generators of tools/*_matrix.py and validated on the same
generators.  All engines give the same bits (the parity tests), so a wrong rule costs time, never correctness; this test prices it:
every applicable engine is forced on 26 scenes from other generators and seeds (tests/_rule_scenes.py), five timed launches each,
and the default must be within TOL of the best.  The table is printed (pytest -s) and quoted in DESIGN.md 4.9."""
import numpy as np
import pytest

import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F

from _rule_scenes import cases

pytestmark = pytest.mark.gpu
TOL = 0.20            # (24 of the 26 scenes are within 10 %; the two that are not are named in DESIGN.md 4.9)

ENGINES = [("scan", F.RT_FLAG_LINEAR_SCAN, (0, 1)),
           ("LDS tree", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_CULL_WALK, (4,)),
           ("L2 exact", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_NO_CULL_WALK, (2,)),
           ("L2 exact culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_EXACT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK, (6,)),
           ("L2 quant", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_NO_CULL_WALK, (3,)),
           ("L2 quant culled", F.RT_FLAG_BVH_TRAVERSE | F.RT_FLAG_QUANT_NODES | F.RT_FLAG_NO_LDS_TREE | F.RT_FLAG_CULL_WALK, (5,))]


def _time(sc, flags, n_prims):
    if flags & F.RT_FLAG_LINEAR_SCAN and n_prims > 4096:
        return None, None                                         # (an O(N) scan of a large scene: minutes; never a candidate)
    rq = F.default_request(width=1280, height=720, divisions=4, spp=4, max_bounces=6, seed=77, flags=flags)
    reqs = []
    for k in range(4):
        r = rq.copy()
        r.division_no = k
        reqs.append(r)
    sc.render_tiles(reqs)
    best, st = 1e9, None
    for _ in range(5):
        _, _, st = sc.render_tiles(reqs)
        best = min(best, st.kernel_ms)
    return st.ray_segments / best / 1e3, int(st.engine)


def test_default_engine_is_near_the_best_on_unseen_scenes(ndev):
    rows, worst = [], (0.0, None)
    for name, sph, tri in cases():
        n = len(sph) + len(tri)
        with rt.Scene(0, rt.World(sph, tri)) as sc:
            dflt, deng = _time(sc, 0, n)
            got = {}
            for ename, fl, want in ENGINES:
                v, e = _time(sc, fl, n)
                if v is not None and e in want:                   # (a forced engine the scene cannot take falls back: not that engine's time)
                    got[ename] = v
        best = max(list(got.values()) + [dflt])
        loss = 1.0 - dflt / best
        rows.append((name, n, dflt, deng, got, loss))
        if loss > worst[0]:
            worst = (loss, name)
    names = [e[0] for e in ENGINES]
    print("\\n%-20s %7s %12s  " % ("scene", "prims", "default") + " ".join("%16s" % n for n in names) + "   loss")
    for name, n, dflt, deng, got, loss in rows:
        print("%-20s %7d %8.0f (e%d)  " % (name, n, dflt, deng) + " ".join(("%16.0f" % got[k]) if k in got else "%16s" % "-" for k in names)
              + "  %5.1f %%" % (100 * loss))
    print("Mrays/s, kernel time of the best of five launches (1280x720, 4 spp, depth 6); loss = 1 - default / best engine")
    n10 = sum(1 for r in rows if r[5] <= 0.10)
    print(f"{n10} of {len(rows)} scenes within 10 % of the best engine; worst: {100 * worst[0]:.1f} % on '{worst[1]}'")
    assert worst[0] <= TOL, f"default engine {100 * worst[0]:.1f} % behind the best on '{worst[1]}'"
    assert n10 >= len(rows) - 3
