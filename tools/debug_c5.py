"""GPU vs oracle (BVH and linear) on a c5 strip: locate differing pixels and decide who is right."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes
from oracle import oracle
rt.init()
sph, rq = scenes.config("c5")
r = rq.copy(); r.division_no = 9
with rt.Scene(0, rt.World(sph)) as sc:
    a, af, st = sc.render_tile(r, want_f32=True)
    r.flags = 1
    ax, _, stx = sc.render_tile(r)
    r.flags = 0
print("gpu filter vs gpu exact-scan bytes differ:", int((a != ax).sum()), st.ray_segments, stx.ray_segments, "fallbacks", st.exact_fallbacks)
t = time.time(); ref, reff, info = oracle.render(r, sph, backend=1, want_f32=True); print("bvh oracle", time.time() - t)
hs = r.height // r.divisions
d = (a != ref).reshape(hs, r.width, 3).any(axis=2)
ys, xs = np.nonzero(d)
print("pixels differing vs BVH oracle:", len(ys), list(zip(ys.tolist(), xs.tolist()))[:20], "segs", st.ray_segments, info["ray_segments"])
rows = sorted(set(ys.tolist()))[:3]
for yl in rows:
    rr = r.copy(); rr.divisions = r.height; rr.division_no = hs * r.division_no + yl
    t = time.time(); lin, _, _ = oracle.render(rr, sph, backend=0); print("linear oracle row", yl, time.time() - t)
    g = a.reshape(hs, r.width, 3)[yl].reshape(-1); b = ref.reshape(hs, r.width, 3)[yl].reshape(-1)
    print("  row", yl, "gpu!=linear:", int((g != lin).sum()), " bvh!=linear:", int((b != lin).sum()))
