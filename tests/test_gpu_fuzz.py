"""Differential fuzz: random small scenes, image shapes, knobs and engine flags, HIP path vs oracle, bit for bit.
Deterministic (fixed generator seeds) so a failure is reproducible by its case number."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi

ENGINE_FLAGS = [0, _abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_FULL_CHAIN, _abi.RT_FLAG_LINEAR_SCAN | _abi.RT_FLAG_OC_BROAD_PHASE,
                _abi.RT_FLAG_NO_BVH_CULL, _abi.RT_FLAG_EXACT_SCAN, _abi.RT_FLAG_LINEAR_SCAN | _abi.RT_FLAG_FULL_CHAIN,
                # small trees walk an LDS-resident copy by default; keep the L2-gather walks of the same nodes covered
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_FULL_CHAIN | _abi.RT_FLAG_NO_LDS_TREE,
                # the culled walk (nearer child first, distance culling) forced onto small scenes (sphere-only ones take it)
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK | _abi.RT_FLAG_FULL_CHAIN,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_NO_CULL_WALK,
                # the culled walk over the exact nodes (scenes with triangles take it)
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE | _abi.RT_FLAG_CULL_WALK,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE | _abi.RT_FLAG_CULL_WALK | _abi.RT_FLAG_FULL_CHAIN,
                # the culled walk of the LDS-resident tree (round 4: engine 7; trees that fit LDS, spheres and triangles)
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_CULL_WALK,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_CULL_WALK | _abi.RT_FLAG_FULL_CHAIN,
                _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_CULL_WALK]


def _random_case(i):
    g = np.random.default_rng(1000 + i)
    n_sph = int(g.choice([0, 1, 2, 7, 33, 200, 700, 2500]))
    n_tri = int(g.choice([0, 0, 0, 1, 12, 90]))
    scale = float(g.choice([1.0, 1.0, 8.0, 0.2]))
    sph = np.zeros(n_sph, _abi.SPHERE_DTYPE)
    if n_sph:
        sph["cx"] = g.uniform(-6, 6, n_sph) * scale
        sph["cy"] = g.uniform(-3, 4, n_sph) * scale
        sph["cz"] = g.uniform(-14, -1.5, n_sph) * scale
        sph["radius"] = g.uniform(0.05, 0.9, n_sph) * scale * g.choice([1.0, 1.0, 3.0])
        for c in ("albedo_r", "albedo_g", "albedo_b"):
            sph[c] = g.uniform(0.05, 1.0, n_sph)
        sph["roughness"] = g.choice([0.0, 0.0, 0.5, 1.0], n_sph)
        sph["emission"] = np.where(g.uniform(size=n_sph) < 0.1, g.uniform(1, 6, n_sph), 0.0)
    tri = np.zeros(n_tri, _abi.TRIANGLE_DTYPE)
    for t in range(n_tri):
        c = np.array([g.uniform(-5, 5), g.uniform(-2, 3), g.uniform(-12, -2)]) * scale
        tri["a"][t], tri["b"][t], tri["c"][t] = c, c + g.uniform(-1.5, 1.5, 3) * scale, c + g.uniform(-1.5, 1.5, 3) * scale
        tri["albedo_r"][t], tri["albedo_g"][t], tri["albedo_b"][t] = g.uniform(0.1, 1.0, 3)
        tri["roughness"][t] = g.choice([0.0, 0.3, 1.0])
        tri["emission"][t] = 4.0 if g.uniform() < 0.1 else 0.0
    w, h = int(g.integers(1, 140)), int(g.integers(1, 90))
    div = int(g.integers(1, max(2, min(h, 6) + 1)))
    rq = _abi.default_request(width=w, height=h, divisions=div, division_no=int(g.integers(0, div)),
                              spp=int(g.integers(1, 6)), max_bounces=int(g.choice([0, 1, 3, 10, 25])),
                              aperture=float(g.choice([0.0, 0.1, 0.5])), focus_distance=float(g.choice([1.0, 4.0])),
                              fov=float(g.choice([0.6, 1.5707964, 2.2])), focal_length=float(g.choice([1.0, 2.0])),
                              t_min=float(g.choice([0.001, 0.05])), t_max=float(g.choice([1000.0, 12.0 * scale])),
                              seed=int(g.integers(0, 2**63)))
    if i % 5 == 3:
        # round 4 (sample units): every fifth case at 8 samples per pixel and more — one pixel per slot, counts that are not powers
        # of two, slots of up to 100 units; own generator: the other cases stay what they were
        g2 = np.random.default_rng(88000 + i)
        rq.spp = int(g2.choice([8, 9, 13, 16, 31, 64, 100]))
        rq.width, rq.height = max(1, rq.width // 2), max(rq.divisions, rq.height // 2)
    flags = int(ENGINE_FLAGS[i % len(ENGINE_FLAGS)])
    return sph, tri, rq, flags


def _world_order(i, sph, tri):
    """Every third case (round 3): the world is given in an order of its own — a random world_index — and a few primitives
    are made exact copies of others (different albedo), so that hits at exactly equal distance exist and the order decides them.
    Own generator: the scenes of the other cases stay what they were."""
    if i % 3 != 1 or len(sph) + len(tri) < 2:
        return sph, tri, None
    g = np.random.default_rng(77000 + i)
    sph, tri = sph.copy(), tri.copy()
    for arr, geo in ((sph, ("cx", "cy", "cz", "radius")), (tri, ("a", "b", "c"))):
        for _ in range(min(4, len(arr) // 2)):
            j, k = g.integers(0, len(arr), 2)
            for f in geo:
                arr[f][k] = arr[f][j]
    return sph, tri, g.permutation(len(sph) + len(tri)).astype(np.uint32)


N_CASES = int(os.environ.get("RT_FUZZ_CASES", "66"))      # RT_FUZZ_CASES=1000 for a long soak


@pytest.mark.parametrize("i", range(N_CASES))
def test_fuzz_case(ndev, oracle, i):
    sph, tri, rq, flags = _random_case(i)
    if rq.height // rq.divisions == 0:
        pytest.skip("zero-row strip")
    sph, tri, wi = _world_order(i, sph, tri)
    backend = 0 if (flags & _abi.RT_FLAG_NO_BVH_CULL) else 1
    ref, ref_f, info = oracle.render(rq, sph if len(sph) else None, tri if len(tri) else None, backend=backend,
                                     want_f32=True, world_index=wi)
    r = rq.copy()
    r.flags = flags
    with rt.Scene(0, rt.World(sph, tri, wi)) as sc:
        rgb, f32, st = sc.render_tile(r, want_f32=True)
    assert np.array_equal(rgb, ref), f"case {i}: {int((rgb != ref).sum())} bytes differ (flags {flags})"
    assert np.array_equal(f32.view(np.uint32), ref_f.view(np.uint32)), f"case {i}"
    assert st.ray_segments == info["ray_segments"], f"case {i}"


def _big_case(i):
    """Large scenes for the traversal engines: thousands of spheres in different distributions (uniform fields, tight
    clusters, sheets, radius ratios up to 1:2000 with a ground sphere), medium frames."""
    g = np.random.default_rng(5000 + i)
    n = int(g.choice([3000, 6000, 12000, 30000]))
    kind = i % 4
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    if kind == 0:                                   # uniform field
        c = g.uniform([-40, -1, -90], [40, 15, -3], (n, 3))
        r = g.uniform(0.1, 0.5, n)
    elif kind == 1:                                 # tight clusters
        k = g.uniform([-30, 0, -70], [30, 10, -5], (24, 3))
        c = k[g.integers(0, 24, n)] + g.normal(0, 0.8, (n, 3))
        r = g.uniform(0.02, 0.15, n)
    elif kind == 2:                                 # a thin sheet (small extent along y)
        c = g.uniform([-25, 1.0, -60], [25, 1.05, -4], (n, 3))
        r = g.uniform(0.05, 0.2, n)
    else:                                           # dense overlap (leaf density >> 2): exact-node kernel by the heuristic
        c = g.uniform([-6, -1, -20], [6, 5, -4], (n, 3))
        r = g.uniform(0.2, 0.6, n)
    sph["cx"], sph["cy"], sph["cz"], sph["radius"] = c[:, 0], c[:, 1], c[:, 2], r
    sph["cx"][0], sph["cy"][0], sph["cz"][0], sph["radius"][0] = 0.0, -1001.0, -20.0, 1000.0   # ground
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        sph[ch] = g.uniform(0.1, 0.95, n)
    sph["roughness"] = g.choice([0.0, 0.0, 0.4, 1.0], n)
    sph["emission"] = np.where(g.uniform(size=n) < 0.03, g.uniform(2, 6, n), 0.0)
    rq = _abi.default_request(width=int(g.choice([160, 256])), height=int(g.choice([90, 144])), divisions=1, spp=2,
                              max_bounces=int(g.choice([4, 8])), seed=int(g.integers(0, 2**63)))
    flags = [0, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES,
             _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_NO_CULL_WALK][(i // 4 + i) % 4]
    return sph, rq, flags


N_BIG = int(os.environ.get("RT_FUZZ_BIG", "6"))           # RT_FUZZ_BIG=60 for a long soak


@pytest.mark.parametrize("i", range(N_BIG))
def test_fuzz_big_scene(ndev, oracle, i):
    sph, rq, flags = _big_case(i)
    ref, ref_f, info = oracle.render(rq, sph, backend=1, want_f32=True)
    r = rq.copy()
    r.flags = flags
    with rt.Scene(0, rt.World(sph)) as sc:
        rgb, f32, st = sc.render_tile(r, want_f32=True)
    assert st.engine in (2, 3, 5)
    _BIG_ENGINES.append(st.engine)
    assert np.array_equal(rgb, ref), f"big case {i}: {int((rgb != ref).sum())} bytes differ (flags {flags}, engine {st.engine})"
    assert np.array_equal(f32.view(np.uint32), ref_f.view(np.uint32)), f"big case {i}"
    assert st.ray_segments == info["ray_segments"], f"big case {i}"


_BIG_ENGINES = []


def test_big_cases_covered_plain_and_culled_walks(ndev):
    if len(_BIG_ENGINES) < 6:
        pytest.skip("big cases did not run")
    assert 3 in _BIG_ENGINES and 5 in _BIG_ENGINES, _BIG_ENGINES


N_MIXED = int(os.environ.get("RT_FUZZ_MIXED", "8"))


@pytest.mark.parametrize("i", range(N_MIXED))
def test_fuzz_big_mixed_scene(ndev, oracle, i):
    """Thousands of spheres AND triangles in one scene: more spheres than triangles (the quantised walk validates triangle
    leaves through their box chain) or more triangles (exact nodes, root tests compacted over mixed primitives)."""
    g = np.random.default_rng(9000 + i)
    ns, nt = [(5000, 3000), (2000, 6000), (9000, 500), (300, 4000)][i % 4]
    sph = np.zeros(ns, _abi.SPHERE_DTYPE)
    c = g.uniform([-30, -1, -70], [30, 12, -3], (ns, 3))
    sph["cx"], sph["cy"], sph["cz"], sph["radius"] = c[:, 0], c[:, 1], c[:, 2], g.uniform(0.1, 0.6, ns)
    sph["cx"][0], sph["cy"][0], sph["cz"][0], sph["radius"][0] = 0.0, -501.0, -20.0, 500.0
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        sph[ch] = g.uniform(0.1, 0.95, ns)
    sph["roughness"] = g.choice([0.0, 0.0, 0.4, 1.0], ns)
    sph["emission"] = np.where(g.uniform(size=ns) < 0.03, g.uniform(2, 6, ns), 0.0)
    tri = np.zeros(nt, _abi.TRIANGLE_DTYPE)
    a = g.uniform([-30, -1, -70], [30, 12, -3], (nt, 3))
    tri["a"], tri["b"], tri["c"] = a, a + g.uniform(-1.2, 1.2, (nt, 3)), a + g.uniform(-1.2, 1.2, (nt, 3))
    for ch in ("albedo_r", "albedo_g", "albedo_b"):
        tri[ch] = g.uniform(0.1, 0.95, nt)
    tri["roughness"] = g.choice([0.0, 0.3, 1.0], nt)
    tri["emission"] = np.where(g.uniform(size=nt) < 0.02, 4.0, 0.0)
    rq = _abi.default_request(width=192, height=108, divisions=1, spp=2, max_bounces=int(g.choice([3, 6])), seed=int(g.integers(0, 2**63)))
    flags = [0, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_CULL_WALK,
             _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_CULL_WALK][(i // 4) % 4]
    # odd cases: the spheres and triangles interleaved at random in `world` (round 3), a hundred of each made exact copies
    wi = None
    if i % 2:
        gw = np.random.default_rng(88000 + i)
        for arr, geo in ((sph, ("cx", "cy", "cz", "radius")), (tri, ("a", "b", "c"))):
            j, k = gw.integers(1, len(arr), 100), gw.integers(1, len(arr), 100)
            for f in geo:
                arr[f][k] = arr[f][j]
        wi = gw.permutation(ns + nt).astype(np.uint32)
    ref, ref_f, info = oracle.render(rq, sph, tri, backend=1, want_f32=True, world_index=wi)
    r = rq.copy()
    r.flags = flags
    with rt.Scene(0, rt.World(sph, tri, wi)) as sc:
        rgb, f32, st = sc.render_tile(r, want_f32=True)
    assert st.engine in (2, 3, 6)
    assert np.array_equal(rgb, ref), f"mixed case {i}: {int((rgb != ref).sum())} bytes differ (flags {flags}, engine {st.engine})"
    assert np.array_equal(f32.view(np.uint32), ref_f.view(np.uint32)), f"mixed case {i}"
    assert st.ray_segments == info["ray_segments"], f"mixed case {i}"
