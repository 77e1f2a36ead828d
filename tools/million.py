"""One-off: a million spheres through the default engine against the oracle (small frame)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import scenes, _abi
from oracle import oracle as orc
rt.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sph = scenes._rand_field(n, 0x5EED2000, ((-300, 300), (-1, 60), (-600, -3)), (0.1, 0.4), None) if False else None
g = np.random.default_rng(3)
sph = np.zeros(n, _abi.SPHERE_DTYPE)
sph["cx"], sph["cy"], sph["cz"] = g.uniform(-300, 300, n), g.uniform(-1, 60, n), g.uniform(-600, -3, n)
sph["radius"] = g.uniform(0.1, 0.4, n)
sph["cx"][0], sph["cy"][0], sph["cz"][0], sph["radius"][0] = 0, -1001, -20, 1000
for c in ("albedo_r", "albedo_g", "albedo_b"):
    sph[c] = g.uniform(0.1, 0.9, n)
sph["roughness"] = g.choice([0.0, 1.0], n)
rq = _abi.default_request(width=192, height=108, divisions=1, spp=2, max_bounces=6, seed=5)
t0 = time.perf_counter()
sc = rt.Scene(0, rt.World(sph))
t1 = time.perf_counter()
rgb, _, st = sc.render_tile(rq)
t2 = time.perf_counter()
print(f"n={n}: scene create {1e3*(t1-t0):.1f} ms, render {1e3*(t2-t1):.1f} ms (kernel {st.kernel_ms:.2f} ms), engine {st.engine}, "
      f"{st.ray_segments} segments")
ref, _, info = orc.render(rq, sph, backend=1)
print("bit-exact vs oracle:", bool(np.array_equal(rgb, ref)), st.ray_segments == info["ray_segments"], f"oracle bvh build {info['bvh_build_ms']:.0f} ms")
sc.close()
