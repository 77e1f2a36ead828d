"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle
on identical scene + seed.  Bar: bit-exact RGB8, bit-exact f32 framebuffer, equal ray-segment
counts (integer), i.e. mean |delta| = 0 <= the 1e-5 tolerance BASELINE.json states."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi, scenes

TOL_MEAN_ABS = 1e-5      # BASELINE.json north_star: mean per-channel |delta| <= 1e-5 vs CPU


def _small(name, w, h, spp=None, div=1):
    sph, rq = scenes.config(name)
    rq.width, rq.height, rq.divisions = w, h, div
    if spp:
        rq.spp = spp
    return sph, rq


def _compare(oracle, rq, sph, tri=None, flags=0):
    """flags clear: reference semantics (BVH-culled candidates) vs the oracle's BVH back-end;
    RT_FLAG_NO_BVH_CULL: plain linear scan vs the oracle's linear back-end."""
    rq = rq.copy()
    backend = 0 if (flags & rt.RT_FLAG_NO_BVH_CULL) else 1
    ref_rgb, ref_f, info = oracle.render(rq, sph, tri, backend=backend, want_f32=True)
    rq.flags = flags
    with rt.Scene(0, rt.World(sph, tri)) as sc:
        rgb, f, st = sc.render_tile(rq, want_f32=True)
    mean_abs = float(np.abs(f.astype(np.float64) - ref_f.astype(np.float64)).mean())
    assert mean_abs <= TOL_MEAN_ABS, mean_abs
    nbad = int((rgb != ref_rgb).sum())
    assert nbad == 0, f"{nbad} RGB8 bytes differ, mean|d|={mean_abs}"
    assert np.array_equal(f.view(np.uint32), ref_f.view(np.uint32)), "f32 framebuffer not bit-identical"
    assert st.ray_segments == info["ray_segments"]
    assert st.primary_rays == (rq.height // rq.divisions) * rq.width * rq.spp
    return st


from _golden import load_all as _load_golden
import hashlib

_GOLDEN = _load_golden()


@pytest.mark.parametrize("case", _GOLDEN, ids=[c["name"] for c in _GOLDEN])
def test_gpu_matches_committed_golden(ndev, case):
    """HIP path vs the committed fixture bytes (no oracle call involved)."""
    with rt.Scene(0, rt.World(case["spheres"], case["triangles"], case["world_index"])) as sc:
        rl = case["req"].copy()
        rl.flags = rt.RT_FLAG_NO_BVH_CULL
        lin, lin_f, st_l = sc.render_tile(rl, want_f32=True)
        rgb, f32, st = sc.render_tile(case["req"], want_f32=True)
    assert hashlib.sha256(lin.tobytes()).hexdigest() == case["sha256_rgb_linear"]
    assert hashlib.sha256(lin_f.tobytes()).hexdigest() == case["sha256_f32_linear"]
    assert st_l.ray_segments == case["ray_segments_linear"]
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == case["sha256_rgb"]
    assert hashlib.sha256(f32.tobytes()).hexdigest() == case["sha256_f32"]
    assert st.ray_segments == case["ray_segments"]
    if case["rgb"] is not None:
        assert np.array_equal(rgb, case["rgb"])
        assert rgb.std() > 1.0


def test_c1_single_sphere(ndev, oracle):
    sph, rq = scenes.config("c1")
    _compare(oracle, rq, sph)


def test_c2_cornell_small(ndev, oracle):
    sph, rq = _small("c2", 480, 270)
    st = _compare(oracle, rq, sph)
    assert st.exact_fallbacks == 0


def test_c3_rand1024_small(ndev, oracle):
    sph, rq = _small("c3", 240, 136, spp=4)
    _compare(oracle, rq, sph)


def test_exact_scan_flag_equals_filter(ndev, oracle):
    sph, rq = _small("c3", 240, 136, spp=2)
    a = _compare(oracle, rq, sph, flags=0)
    b = _compare(oracle, rq, sph, flags=rt.RT_FLAG_EXACT_SCAN)
    assert a.ray_segments == b.ray_segments
    assert b.broad_candidates == 0


@pytest.mark.parametrize("flags", [2, 3])
def test_linear_scan_semantics_flag(ndev, oracle, flags):
    # RT_FLAG_NO_BVH_CULL (with and without the broad phase) == the oracle's linear back-end
    sph, rq = _small("c3", 240, 136, spp=2)
    _compare(oracle, rq, sph, flags=flags)
    sph, tri = scenes.quad_room()
    rq = _abi.default_request(width=120, height=72, divisions=1, spp=2, max_bounces=5, seed=5)
    _compare(oracle, rq, sph, tri, flags=flags)


def test_expanded_and_oc_broad_phase_agree(ndev, oracle):
    # c3 qualifies for the 8-op expanded broad phase; RT_FLAG_OC_BROAD_PHASE forces the 11-op form.
    # Both are conservative, so images and segment counts are identical; the expanded form's wider
    # margin can only add candidates.
    sph, rq = _small("c3", 320, 180, spp=4)
    a = _compare(oracle, rq, sph, flags=0)
    b = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_OC_BROAD_PHASE)
    assert a.ray_segments == b.ray_segments
    assert a.broad_candidates >= b.broad_candidates > 0
    # a scene far from its own centroid scale (large |c|^2 against r^2) must still be exact
    far = sph.copy()
    far["cx"] += 900.0
    far["cz"] -= 700.0
    rq2 = rq.copy()
    rq2.width, rq2.height = 96, 54
    _compare(oracle, rq2, far, flags=0)


def test_far_false_hits_are_culled_like_the_reference(ndev, oracle):
    """A distant small sphere: the reference's roots (sphere.rs:45) report a hit for rays that
    miss the sphere's AABB; its BVH drops them.  Default flags must follow the BVH back-end."""
    sph = scenes.rand65536(n=6000)
    rq = _abi.default_request(width=256, height=144, divisions=1, spp=4, max_bounces=6, seed=1234)
    _compare(oracle, rq, sph, flags=0)
    _compare(oracle, rq, sph, flags=_abi.RT_FLAG_FULL_CHAIN)
    _compare(oracle, rq, sph, flags=rt.RT_FLAG_NO_BVH_CULL)


def test_ragged_sizes_and_strips(ndev, oracle):
    # W, Hs not multiples of the 16x16 tile; height % divisions != 0 (slave renders floor(H/div) rows)
    sph, rq = _small("c2", 203, 131, spp=2, div=3)
    for k in range(3):
        rq.division_no = k
        _compare(oracle, rq, sph)


def test_empty_world_is_sky(ndev, oracle):
    rq = _abi.default_request(width=64, height=48, divisions=1, spp=2, max_bounces=3, seed=7)
    _compare(oracle, rq, None)


def test_depth_zero_bounces(ndev, oracle):
    sph, rq = _small("c2", 160, 90, spp=2)
    rq.max_bounces = 0
    _compare(oracle, rq, sph)


def test_streamed_scene_larger_than_lds(ndev, oracle):
    sph = scenes.rand65536(n=9000)          # > 4096: streamed through LDS chunks
    rq = _abi.default_request(width=96, height=64, divisions=1, spp=2, max_bounces=4, seed=99)
    _compare(oracle, rq, sph)


def test_triangles_mixed_scene(ndev, oracle):
    sph, tri = scenes.quad_room()
    rq = _abi.default_request(width=160, height=96, divisions=1, spp=4, max_bounces=5, seed=5)
    _compare(oracle, rq, sph, tri)


@pytest.mark.parametrize("flags", [0, 2])
def test_triangle_mesh(ndev, oracle, flags):
    # 385 triangles + 3 spheres; BVH semantics use the own-leaf AABB test as the triangle broad phase
    sph, tri = scenes.tri_terrain()
    rq = _abi.default_request(width=192, height=108, divisions=2, division_no=1, spp=4, max_bounces=6, seed=21)
    _compare(oracle, rq, sph, tri, flags=flags)
    rq.division_no = 0
    _compare(oracle, rq, None, tri, flags=flags)            # triangles only


@pytest.mark.parametrize("scene", ["c2", "c3", "mesh", "single", "streamed"])
def test_bvh_traversal_engine(ndev, oracle, scene):
    """RT_FLAG_BVH_TRAVERSE: per-lane traversal of the reference BVH == the oracle's BVH back-end, and
    == the linear-scan engine (RT_FLAG_LINEAR_SCAN), bit for bit."""
    tri = None
    if scene == "c2":
        sph, rq = _small("c2", 240, 136)
    elif scene == "c3":
        sph, rq = _small("c3", 240, 136, spp=4)
    elif scene == "mesh":
        sph, tri = scenes.tri_terrain()
        rq = _abi.default_request(width=160, height=90, divisions=1, spp=4, max_bounces=6, seed=21)
    elif scene == "single":
        sph, rq = _small("c1", 64, 64)
    else:
        sph = scenes.rand65536(n=9000)
        rq = _abi.default_request(width=128, height=80, divisions=1, spp=2, max_bounces=5, seed=99)
    a = _compare(oracle, rq, sph, tri, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_NO_CULL_WALK)
    # the same nodes nearer child first with distance culling (sphere scenes; a mesh keeps the plain walk)
    k = _compare(oracle, rq, sph, tri, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK)
    c = _compare(oracle, rq, sph, tri, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE)
    b = _compare(oracle, rq, sph, tri, flags=_abi.RT_FLAG_LINEAR_SCAN)
    # the exact nodes walked from an LDS-resident copy (the default for trees that fit; a single-leaf tree has no nodes)
    e = _compare(oracle, rq, sph, tri, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES)
    assert a.ray_segments == b.ray_segments == c.ray_segments == e.ray_segments
    assert c.engine == 2 and b.engine in (0, 1)
    assert e.engine == (4 if scene in ("c2", "c3", "mesh") else 2)
    assert e.broad_candidates == c.broad_candidates                    # same tree, same leaves reached
    assert a.broad_candidates >= c.broad_candidates                    # rounded boxes can only admit more leaves
    assert k.ray_segments == a.ray_segments and k.broad_candidates <= a.broad_candidates    # culling only skips
    if scene != "single":
        assert a.engine == 3 and k.engine == (3 if scene == "mesh" else 5)


def test_strips_equal_whole_frame(ndev):
    # size-independent property: per-pixel RNG streams => stitched strips == one-strip frame
    sph, rq = _small("c2", 320, 180, spp=2, div=1)
    with rt.Scene(0, rt.World(sph)) as sc:
        whole, _, _ = sc.render_tile(rq)
        rq.divisions = 6
        parts = []
        for k in range(6):
            rq.division_no = k
            parts.append(sc.render_tile(rq)[0])
    assert np.array_equal(np.concatenate(parts), whole)


def test_render_frame_dispatcher(ndev):
    sph, rq = _small("c2", 320, 180, spp=2, div=6)
    img, st = rt.render_frame_native(rt.World(sph), rq)
    rq1 = rq.copy()
    rq1.divisions = 1
    with rt.Scene(0, rt.World(sph)) as sc:
        whole, _, st1 = sc.render_tile(rq1)
    assert np.array_equal(img.reshape(-1), whole)
    assert st.ray_segments == st1.ray_segments
    assert st.n_launches == 1          # a small frame: one launch, the download after it


def test_large_frame_splits_its_batch_to_hide_the_download(ndev):
    """Above 64 MiB of pixels per call the last quarter of the strips is a second launch that runs under the first one's D2H
    copies; the frame is the same as the one the per-strip queue makes."""
    sph, rq = _small("c2", 5760, 4320, spp=1, div=8)
    rq.max_bounces = 2
    a, st_a = rt.render_frame_native(rt.World(sph), rq)
    rq_q = rq.copy()
    rq_q.flags = _abi.RT_FLAG_FRAME_QUEUE
    b, st_b = rt.render_frame_native(rt.World(sph), rq_q)
    assert st_a.n_launches == 2 and st_b.n_launches == 8
    assert np.array_equal(a, b) and st_a.ray_segments == st_b.ray_segments


def test_batched_strips_one_launch(ndev, oracle):
    # strips of one frame with different seeds / division_no in one launch == strip-by-strip oracle
    sph, rq = _small("c3", 200, 120, spp=2, div=5)
    reqs = []
    for k in (4, 0, 2):
        r = rq.copy(); r.division_no = k; r.seed = rq.seed + k
        reqs.append(r)
    with rt.Scene(0, rt.World(sph)) as sc:
        outs, outf, st = sc.render_tiles(reqs, want_f32=True)
    assert st.n_launches == 1
    segs = 0
    for r, o, f in zip(reqs, outs, outf):
        ref, ref_f, info = oracle.render(r, sph, backend=1, want_f32=True)
        assert np.array_equal(o, ref)
        assert np.array_equal(f.view(np.uint32), ref_f.view(np.uint32))
        segs += info["ray_segments"]
    assert st.ray_segments == segs


def test_error_paths(ndev):
    lib = _abi.load()
    sph, rq = _small("c1", 64, 64)
    out = np.zeros(64 * 64 * 3, np.uint8)
    st = _abi.TileStats()
    bad = rq.copy(); bad.division_no = 1
    assert lib.rt_render_tile(0, C.byref(bad), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(out), out.size, None, C.byref(st)) == _abi.RT_ERR_BAD_ARG
    assert lib.rt_render_tile(0, C.byref(rq), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(out), 10, None, C.byref(st)) == _abi.RT_ERR_BUFFER_TOO_SMALL
    assert lib.rt_render_tile(99, C.byref(rq), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(out), out.size, None, C.byref(st)) == _abi.RT_ERR_BAD_DEVICE
    bad = rq.copy(); bad.max_bounces = 1000
    assert lib.rt_render_tile(0, C.byref(bad), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(out), out.size, None, C.byref(st)) == _abi.RT_ERR_LIMIT
    fr = rq.copy(); fr.height = 65; fr.divisions = 2
    big = np.zeros(65 * 64 * 3, np.uint8)
    assert lib.rt_render_frame(None, 0, C.byref(fr), _abi.ptr(sph), 1, None, 0, None, _abi.ptr(big), big.size, C.byref(st)) == _abi.RT_ERR_FRAME_SIZE
    assert lib.rt_last_error().decode() != ""


def test_determinism_and_seed(ndev):
    sph, rq = _small("c3", 128, 72, spp=2)
    with rt.Scene(0, rt.World(sph)) as sc:
        a = sc.render_tile(rq)[0]
        b = sc.render_tile(rq)[0]
        rq.seed += 1
        c = sc.render_tile(rq)[0]
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c)


def test_http_slave_shim_on_gpu(ndev, oracle):
    """POST RenderInfo JSON to the GPU slave shim, collect ImageSlice JSON at a fake master, assemble."""
    import urllib.request
    from _fakes import FakeMaster
    from ray_tracer_s8_amd import dispatch, wire
    from ray_tracer_s8_amd.interface import RenderInfo, RenderMeta, RenderSettings, World
    from ray_tracer_s8_amd.slave_shim import REPLY, SlaveService
    master = FakeMaster()
    svc = SlaveService(device=0, master_url=f"http://127.0.0.1:{master.port}/result", host="127.0.0.1", port=0,
                       settings=RenderSettings(spp=4, max_bounces=4), fixed_seed=1234).start()
    try:
        meta = RenderMeta(height=90, width=160, divisions=5)
        world = World(scenes.cornell16())
        for k in range(5):
            body = wire.encode_render_info(RenderInfo(world, meta, k)).encode()
            req = urllib.request.Request(f"http://127.0.0.1:{svc.port}/", data=body,
                                         headers={"Content-Type": "application/json"}, method="POST")
            with urllib.request.urlopen(req, timeout=60) as r:
                assert r.read() == REPLY
        svc.wait_idle()
        slices = [wire.decode_image_slice(b) for _, _, b in master.got]
        frame = dispatch.assemble([(s.division_no, s.image) for s in slices], 160, 90, 5)
        rq = RenderInfo(world, meta, 0, RenderSettings(spp=4, max_bounces=4, seed=1234)).request()
        rq.divisions = 1
        ref, _, _ = oracle.render(rq, world.spheres, backend=1)
        assert np.array_equal(frame.reshape(-1), ref)
    finally:
        svc.stop()
        master.stop()


@pytest.mark.parametrize("engine", [_abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE])
def test_non_default_knobs(ndev, oracle, engine):
    """Every knob the reference hard-codes, moved off its literal (camera, t window, spp, depth, 64-bit seed)."""
    sph, _ = scenes.config("c3")
    rq = _abi.default_request(width=150, height=110, divisions=5, division_no=3, spp=3, max_bounces=13,
                              aperture=0.0, focus_distance=3.5, fov=1.1, focal_length=1.5, t_min=0.01, t_max=40.0,
                              seed=0xFEDCBA9876543210, flags=engine)
    _compare(oracle, rq, sph, flags=engine)
    rq2 = rq.copy()
    rq2.aperture, rq2.fov, rq2.t_max = 0.4, 2.4, 1e30
    _compare(oracle, rq2, sph, flags=engine)


def test_reference_literals_small_frame(ndev, oracle):
    # the slave's own settings: 100 spp, 10 bounces, 20 divisions (main.rs:39,51; controller main.rs:33-39)
    sph = scenes.cornell16()
    rq = _abi.default_request(width=96, height=60, division_no=7, seed=3)      # 20 strips of 3 rows
    assert (rq.spp, rq.max_bounces, rq.divisions) == (100, 10, 20)
    _compare(oracle, rq, sph)


def test_max_bounces_limit_and_mirror_box(ndev, oracle):
    # mirror walls keep paths alive to the depth limit: path stack of 63 entries, right-to-left product
    s = scenes.cornell16().copy()
    s["roughness"][:5] = 1.0
    s["albedo_r"][:5] = s["albedo_g"][:5] = s["albedo_b"][:5] = 0.98
    s["emission"][5] = 0.0
    back = s[:1].copy()                                   # sixth wall behind the camera closes the box
    back["cx"], back["cy"], back["cz"] = 0.0, 0.0, 104.0
    s = np.concatenate([s, back])
    rq = _abi.default_request(width=64, height=40, divisions=1, spp=2, max_bounces=_abi.RT_MAX_BOUNCES, seed=9)
    st = _compare(oracle, rq, s, flags=_abi.RT_FLAG_LINEAR_SCAN)
    assert st.ray_segments > 64 * 40 * 2 * 40          # nothing escapes: paths run towards the depth limit
    _compare(oracle, rq, s, flags=_abi.RT_FLAG_BVH_TRAVERSE)


def test_more_strips_than_one_launch_holds(ndev, oracle):
    sph, rq = _small("c2", 64, 130, spp=1, div=130)          # 130 one-row strips -> 3 launches of <= 64 strips
    reqs = []
    for k in range(130):
        r = rq.copy(); r.division_no = k
        reqs.append(r)
    with rt.Scene(0, rt.World(sph)) as sc:
        outs, _, st = sc.render_tiles(reqs)
    assert st.n_launches == 3
    whole = rq.copy(); whole.divisions = 1
    ref, _, info = oracle.render(whole, sph, backend=1)
    assert np.array_equal(np.concatenate(outs), ref) and st.ray_segments == info["ray_segments"]


@pytest.mark.parametrize("flags", [_abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE, _abi.RT_FLAG_NO_BVH_CULL])
def test_exact_distance_ties_follow_the_reference_order(ndev, oracle, flags):
    """Coincident spheres with different albedos: every hit is an exact distance tie.  The reference keeps the
    first minimum of the BVH traversal output (DFS leaf order; its build halves index lists when all centroids
    coincide), the linear semantics the lowest index."""
    base = scenes.cornell16()
    dup = np.repeat(base[6:9], 5, axis=0).copy()            # three small spheres, five copies each
    rng = np.random.default_rng(3)
    dup["albedo_r"], dup["albedo_g"], dup["albedo_b"] = rng.uniform(0.1, 0.9, (3, len(dup))).astype(np.float32)
    sph = np.concatenate([base, dup])
    rq = _abi.default_request(width=160, height=90, divisions=1, spp=4, max_bounces=4, seed=17)
    _compare(oracle, rq, sph, flags=flags)


def test_plain_c_client(ndev, tmp_path):
    """examples/render_frame.c through the C-ABI only (no Python binding in the loop)."""
    import shutil, subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "render_frame"
    lib = _abi.lib_path().parent
    r = subprocess.run([shutil.which("gcc"), "-std=c99", "-O2", f"-I{root / 'include'}", str(root / "examples" / "render_frame.c"),
                        f"-L{lib}", "-lrt_s8", f"-Wl,-rpath,{lib}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe), str(tmp_path / "f.ppm")], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "C_CLIENT_OK" in run.stdout, run.stdout + run.stderr
    ppm = (tmp_path / "f.ppm").read_bytes()
    assert ppm.startswith(b"P6\n256 160\n255\n") and len(ppm) == 15 + 256 * 160 * 3


def _skewer_scene(n=240):
    """n overlapping spheres threaded on the view axis: a central ray crosses every one of them."""
    z = -3.0 - 0.25 * np.arange(n, dtype=np.float32)
    s = np.zeros(n, _abi.SPHERE_DTYPE)
    s["cz"], s["radius"] = z, 0.2
    s["cx"] = 0.002 * np.sin(np.arange(n)).astype(np.float32)
    s["albedo_r"], s["albedo_g"], s["albedo_b"] = 0.7, 0.6, 0.5
    s["roughness"] = (np.arange(n) % 3 == 0).astype(np.float32)
    return s


def test_candidate_list_overflow_paths(ndev, oracle):
    """More candidates than the per-lane lists hold: the linear engine must fall back to the exact scan of the
    chunk (exact_fallbacks > 0), the traversal engine must flush its leaf list in mid-walk."""
    sph = _skewer_scene()
    rq = _abi.default_request(width=64, height=64, divisions=1, spp=2, max_bounces=3, aperture=0.0, fov=0.05, seed=4)
    st = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_LINEAR_SCAN)
    assert st.exact_fallbacks > 0
    st = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_LINEAR_SCAN | _abi.RT_FLAG_OC_BROAD_PHASE)
    assert st.exact_fallbacks > 0
    st = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_NO_CULL_WALK)
    assert st.broad_candidates > 20 * st.ray_segments // 4          # dozens of leaves per primary ray
    st = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_CULL_WALK)
    assert st.engine == 7 and st.broad_candidates < 20 * st.ray_segments // 4    # the culled LDS tree skips most of the skewer
    _compare(oracle, rq, sph, flags=_abi.RT_FLAG_NO_BVH_CULL)


def test_two_host_threads_on_one_device(ndev):
    """rt_render_frame with devices = [0, 0]: two dispatcher threads, each with its own scene, share GPU 0."""
    sph, rq = _small("c2", 320, 180, spp=2, div=6)
    a, st_a = rt.render_frame_native(rt.World(sph), rq, devices=[0, 0])
    b, st_b = rt.render_frame_native(rt.World(sph), rq, devices=[0])
    assert np.array_equal(a, b) and st_a.ray_segments == st_b.ray_segments
    assert st_a.n_launches == 2 and st_b.n_launches == 1     # [0, 0]: three strips each; [0]: six strips in one launch


def test_frame_dispatcher_strip_queue_and_pinning(ndev, oracle):
    """rt_render_frame's assignments give the same frame: static strip k -> device k mod n in one batch (default),
    and RT_FLAG_FRAME_QUEUE — strips pulled bottom-up from a host-atomic queue, two launches in flight per device —
    with one and with three dispatcher threads on GPU 0, with and without the page-locked frame buffer."""
    sph, rq = _small("c3", 256, 144, spp=3, div=12)
    ref, st_ref = rt.render_frame_native(rt.World(sph), rq, devices=[0])
    one = rq.copy()
    one.divisions = 1
    want, _, info = oracle.render(one, sph, backend=1)
    assert np.array_equal(ref.reshape(-1), want)
    Q, NP = _abi.RT_FLAG_FRAME_QUEUE, _abi.RT_FLAG_FRAME_NO_PIN
    for fl in (Q, Q | NP, NP):
        r = rq.copy()
        r.flags = fl
        for devs in ([0], [0, 0, 0]):
            img, st = rt.render_frame_native(rt.World(sph), r, devices=devs)
            assert np.array_equal(img, ref), (fl, devs)
            assert st.ray_segments == st_ref.ray_segments == info["ray_segments"]
            if fl & Q:
                assert st.n_launches == 12                   # one launch per strip


def test_frame_context_reuses_everything_across_frames(ndev, oracle):
    """rt_frame_ctx: the job's state is made once.  Frame 1 pays the page-locking and carries the world's preparation in
    scene_ms; frames 2.. of the job (same world, same buffer) register, upload and spawn nothing: pin_ms == scene_ms == 0,
    the buffer stays page-locked.  Static and strip-queue frames, seeds, a second world and a second buffer in one
    context; every frame equals the oracle's."""
    sph, rq = _small("c3", 256, 144, spp=3, div=12)
    one = rq.copy()
    one.divisions = 1
    buf = np.zeros(256 * 144 * 3, np.uint8)
    with rt.FrameContext(devices=[0, 0], world=rt.World(sph)) as fc:
        img, fs = fc.render(rq, out=buf)
        want, _, info = oracle.render(one, sph, backend=1)
        assert np.array_equal(img.reshape(-1), want) and fs.totals.ray_segments == info["ray_segments"]
        assert fs.scene_ms > 0 and fs.pinned == 1 and fs.pin_ms > 0 and fs.n_devices == 2
        assert abs(fs.wall_ms - (fs.pin_ms + fs.kernel_ms + fs.d2h_exposed_ms + fs.host_ms)) < 1e-3
        for seed, fl in ((1, 0), (2, _abi.RT_FLAG_FRAME_QUEUE), (3, 0)):
            r = rq.copy()
            r.seed, r.flags = seed, fl
            img, fs = fc.render(r, out=buf)
            o = one.copy()
            o.seed = seed
            want, _, _ = oracle.render(o, sph, backend=1)
            assert np.array_equal(img.reshape(-1), want), (seed, fl)
            assert fs.pin_ms == 0.0 and fs.scene_ms == 0.0 and fs.pinned == 1
        # another buffer: registered in its turn (the first is released); NO_PIN: nothing registered
        img2, fs2 = fc.render(rq)
        assert fs2.pin_ms > 0 and fs2.pinned == 1 and np.array_equal(img2.reshape(-1), oracle.render(one, sph, backend=1)[0])
        img2 = img2.copy()
        r = rq.copy()
        r.flags = _abi.RT_FLAG_FRAME_NO_PIN
        img3, fs3 = fc.render(r, out=buf)
        assert fs3.pinned == 0 and fs3.pin_ms == 0.0 and np.array_equal(img3, img2)
        # a second world in the same context
        sph2 = scenes.cornell16()
        fc.set_world(rt.World(sph2))
        img4, fs4 = fc.render(rq, out=buf)
        want4, _, _ = oracle.render(one, sph2, backend=1)
        assert np.array_equal(img4.reshape(-1), want4) and fs4.scene_ms > 0
        fc.release_buffer()


def test_frame_context_balances_entries_by_measured_strip_cost(ndev, oracle):
    """rt_frame_ctx's strip assignment (rt_assign.h) at eight entries on the one device: frame 1 of a job goes out in snake order,
    the kernels count every strip's ray segments, frame 2 is assigned longest-first by them; RT_FLAG_FRAME_STATIC keeps k % n.
    Same bytes whatever the assignment; the per-entry segment counts add up to the frame's; the measured balance improves
    on the static split's for a frame whose bottom rows cost more than its top rows (c3's scene)."""
    sph, rq = _small("c3", 320, 256, spp=4, div=32)
    one = rq.copy()
    one.divisions = 1
    want, _, info = oracle.render(one, sph, backend=1)
    with rt.FrameContext(devices=[0] * 8, world=rt.World(sph)) as fc:
        seen = {}
        for name, fl in (("static", _abi.RT_FLAG_FRAME_STATIC), ("first", 0), ("second", 0), ("third", 0), ("static2", _abi.RT_FLAG_FRAME_STATIC)):
            r = rq.copy()
            r.flags = fl
            img, fs = fc.render(r)
            assert np.array_equal(img.reshape(-1), want), name
            per = list(fs.entry_segments)[:8]
            assert sum(per) == fs.totals.ray_segments == info["ray_segments"]
            assert abs(fs.balance_max_over_mean - max(per) * 8 / sum(per)) < 1e-5
            seen[name] = (int(fs.assignment), fs.balance_max_over_mean)
        # (the static frame measured ITS strips — the request's 32; the balanced assignments cut the frame into their own, at least six
        # per entry, so the first default frame is a snake frame and measures those)
        assert seen["static"][0] == 0 and seen["static2"][0] == 0
        assert seen["first"][0] == 1 and seen["second"][0] == 2 and seen["third"][0] == 2
        # (tiny strips here — 8 rows, 5 tiles — and four of them per entry: bench.py reports the balance of c4 / c5 at full size)
        assert seen["second"][1] < seen["static"][1] and seen["second"][1] < 1.05, seen
        # another world: costs are forgotten, the next frame is a snake frame again
        sph2 = scenes.cornell16()
        fc.set_world(rt.World(sph2))
        img, fs = fc.render(rq)
        assert fs.assignment == 1 and np.array_equal(img.reshape(-1), oracle.render(one, sph2, backend=1)[0])
        img, fs = fc.render(rq)
        assert fs.assignment == 2
        # another frame geometry: measured again as well
        r = rq.copy()
        r.spp = 2
        img, fs = fc.render(r)
        assert fs.assignment == 1


def test_frame_context_outlives_the_buffers_it_was_given(ndev, oracle):
    """Round-3 advisor: the context keeps the frame buffer page-locked after the call and recognises it by ADDRESS; a numpy
    array that was dropped and whose address the next allocation takes again must not inherit the stale registration.  The
    wrapper keeps the registered array alive and releases the registration before another array takes its place."""
    sph, rq = _small("c3", 192, 96, spp=2, div=4)
    one = rq.copy()
    one.divisions = 1
    n = 192 * 96 * 3
    with rt.FrameContext(devices=[0], world=rt.World(sph)) as fc:
        for seed in range(6):
            r = rq.copy()
            r.seed = seed
            tmp = np.empty(n, np.uint8)                       # a temporary: dropped at the end of the iteration
            img, fs = fc.render(r, out=tmp)
            assert fs.pinned == 1 and fs.pin_ms > 0           # a NEW registration every time: never a stale one by address
            o = one.copy()
            o.seed = seed
            assert np.array_equal(img.reshape(-1), oracle.render(o, sph, backend=1)[0])
            assert fc._pinned is tmp
            del img, tmp
        own, fs = fc.render(rq)                               # then the context's own buffer, and a change of frame size
        assert fs.pinned == 1
        big = rq.copy()
        big.width, big.height = 256, 128
        img, fs = fc.render(big)
        ob = big.copy()
        ob.divisions = 1
        assert fs.pinned == 1 and fs.pin_ms > 0 and np.array_equal(img.reshape(-1), oracle.render(ob, sph, backend=1)[0])


@pytest.mark.parametrize("cull", [0, _abi.RT_FLAG_CULL_WALK])
def test_capped_stack_launches_on_two_streams(ndev, oracle, cull):
    """The capped-stack walk (stack entries beyond a few LDS slots live in one per-scene HBM area) enqueued on two
    streams at once: the library chains such launches, so frames rendered 'concurrently' are still exact.  Both the plain
    and the culled walk have a capped-stack variant."""
    with _abi.debug_library():               # (the knobs are hooks of the TEST library: the product library exports none)
        rt.init()
        hip = _abi.hip_runtime()             # the runtime the library is bound to (never a second one: profiles/README.md)
        prev = [_abi.debug_set("RT_FORCE_CAPPED", 1), _abi.debug_set("RT_STACK_LDS", 3)]
        try:
            _capped_two_streams(hip, oracle, cull)
        finally:
            _abi.debug_set("RT_FORCE_CAPPED", prev[0])
            _abi.debug_set("RT_STACK_LDS", prev[1])


def _capped_two_streams(hip, oracle, cull):
    sph = scenes.rand65536(n=9000)
    rq = _abi.default_request(width=128, height=80, divisions=1, spp=2, max_bounces=5, seed=99,
                              flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | cull)
    nb = 128 * 80 * 3
    with rt.Scene(0, rt.World(sph)) as sc:                   # (the scene makes device 0 current for this thread)
        streams, outs = [], []
        for _ in range(2):
            st_ = C.c_void_p()
            assert hip.hipStreamCreate(C.byref(st_)) == 0
            streams.append(st_)
        for _ in range(6):
            d = C.c_void_p()
            assert hip.hipMalloc(C.byref(d), C.c_size_t(nb)) == 0
            outs.append(d)
        reqs = []
        for i, o in enumerate(outs):
            r = rq.copy()
            r.seed = 99 + (i % 3)
            reqs.append(r)
            sc.render_tiles_device([r], [o.value], nb, streams[i % 2].value)
        assert hip.hipDeviceSynchronize() == 0
        st = sc.collect()
        got = []
        for o in outs:
            h = np.zeros(nb, np.uint8)
            assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), o, C.c_size_t(nb), 2) == 0      # hipMemcpyDeviceToHost
            got.append(h)
            hip.hipFree(o)
        for s_ in streams:
            hip.hipStreamDestroy(s_)
    assert st.engine == (5 if cull else 3) and st.n_launches == 6
    for r, h in zip(reqs, got):
        want, _, _ = oracle.render(r, sph, backend=1)
        assert np.array_equal(h, want)


def test_frame_over_two_devices(ndev, oracle):
    """rt_render_frame and a persistent rt_frame_ctx over devices [0, 1] (and the strip queue over them): the same frame as
    one device, frame after frame.  Needs a box with two GPUs; the one-GPU boxes of rounds 1-3 skip it."""
    if ndev < 2:
        pytest.skip("needs two GPUs")
    sph, rq = _small("c3", 256, 144, spp=3, div=12)
    ref, st_ref = rt.render_frame_native(rt.World(sph), rq, devices=[0])
    img, st = rt.render_frame_native(rt.World(sph), rq, devices=[0, 1])
    assert np.array_equal(img, ref) and st.ray_segments == st_ref.ray_segments
    q = rq.copy()
    q.flags = _abi.RT_FLAG_FRAME_QUEUE
    img, st = rt.render_frame_native(rt.World(sph), q, devices=[1, 0])
    assert np.array_equal(img, ref) and st.ray_segments == st_ref.ray_segments
    with rt.FrameContext(devices=[1, 0], world=rt.World(sph)) as fc:     # (device 0 is not the first entry: the
        buf = np.zeros(256 * 144 * 3, np.uint8)                          #  registration must not assume it)
        for r in (rq, q, rq):
            img, fs = fc.render(r, out=buf)
            assert np.array_equal(img, ref) and fs.totals.ray_segments == st_ref.ray_segments and fs.n_devices == 2
        assert fs.pin_ms == 0.0 and fs.scene_ms == 0.0


def test_shutdown_is_refused_while_a_scene_is_alive(ndev):
    """rt_shutdown used to delete the device contexts under live scenes; now it is a no-op with a message until the last
    scene is destroyed."""
    lib = _abi.load()
    lib.rt_last_error.restype = C.c_char_p
    sph, rq = _small("c2", 64, 36, spp=1, div=1)
    with rt.Scene(0, rt.World(sph)) as sc:
        lib.rt_shutdown()
        assert b"refused" in lib.rt_last_error()
        a, _, _ = sc.render_tiles([rq])                      # the scene and its context still work
    with rt.Scene(0, rt.World(sph)) as sc2:
        b, _, _ = sc2.render_tile(rq)
    assert np.array_equal(a[0], b)


def test_python_controller_and_slave_mirror(ndev, oracle):
    from ray_tracer_s8_amd.interface import Controller, RenderMeta, RenderSettings, World
    sph = scenes.cornell16()
    ctl = Controller(devices=[0])
    try:
        meta = RenderMeta(height=60, width=96, divisions=4)
        img = ctl.render_frame(World(sph), meta, RenderSettings(spp=3, max_bounces=4, seed=8))
    finally:
        ctl.close()
    rq = _abi.default_request(width=96, height=60, divisions=1, spp=3, max_bounces=4, seed=8)
    ref, _, _ = oracle.render(rq, sph, backend=1)
    assert np.array_equal(img.reshape(-1), ref)


def test_controller_shim_with_gpu_slaves(ndev, oracle):
    """/upload -> in-process GPU slaves -> /poll -> JPEG: 'the controller dispatches tiles to GPUs'."""
    import io, time, urllib.request, uuid
    PIL = pytest.importorskip("PIL.Image")
    from ray_tracer_s8_amd import obj
    from ray_tracer_s8_amd.controller_shim import ControllerService
    from ray_tracer_s8_amd.interface import RenderSettings
    from test_obj import MTL, OBJ
    ctl = ControllerService(devices=[0], host="127.0.0.1", port=0, width=96, height=64, divisions=4,
                            settings=RenderSettings(spp=8, max_bounces=4, seed=5)).start()
    post = lambda path, data: urllib.request.urlopen(
        urllib.request.Request(f"http://127.0.0.1:{ctl.port}{path}", data=data, method="POST"), timeout=60).read()
    try:
        job = post(f"/upload/{len(OBJ)}/", OBJ + MTL).decode()
        out = b""
        for _ in range(400):
            out = post("/poll", job.encode())
            if out[:2] == b"\xff\xd8":
                break
            time.sleep(0.02)
        img = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB")).astype(np.float64)
        rq = _abi.default_request(width=96, height=64, divisions=1, spp=8, max_bounces=4, seed=5)
        ref, _, _ = oracle.render(rq, None, obj.build_world(OBJ + MTL, len(OBJ)), backend=1)
        assert np.abs(img - ref.reshape(64, 96, 3)).mean() < 6.0
    finally:
        ctl.stop()


OBJ2 = b"""v -2 -1 -4
v 2 -1 -4
v 0 2 -4
v -2 -1 -2
v 2 -1 -2
usemtl a
f 1 2 3
f 4 5 1
f 5 2 1
"""
MTL2 = b"""newmtl a
Kd 0.2 0.7 0.9
Ns 100
"""


def test_controller_shim_two_overlapping_jobs(ndev, oracle):
    """Two uploads with DIFFERENT worlds before the first poll (the reference controller keeps a list of concurrent
    jobs): both jobs' dispatcher threads go through the same in-process GPU slaves.  Without the per-slave lock one
    job destroyed the scene the other was rendering from."""
    import io, time, urllib.request
    PIL = pytest.importorskip("PIL.Image")
    from ray_tracer_s8_amd import obj
    from ray_tracer_s8_amd.controller_shim import ControllerService
    from ray_tracer_s8_amd.interface import RenderSettings
    from test_obj import MTL, OBJ
    ctl = ControllerService(devices=[0, 0], host="127.0.0.1", port=0, width=96, height=64, divisions=8,
                            settings=RenderSettings(spp=8, max_bounces=4, seed=5)).start()
    post = lambda path, data: urllib.request.urlopen(
        urllib.request.Request(f"http://127.0.0.1:{ctl.port}{path}", data=data, method="POST"), timeout=60).read()
    try:
        bodies = [(OBJ, MTL), (OBJ2, MTL2), (OBJ, MTL), (OBJ2, MTL2)]
        jobs = [post(f"/upload/{len(o)}/", o + m).decode() for o, m in bodies]        # all four before any poll
        for job, (o, m) in zip(jobs, bodies):
            out = b""
            for _ in range(600):
                out = post("/poll", job.encode())
                if out[:2] == b"\xff\xd8":
                    break
                time.sleep(0.02)
            assert out[:2] == b"\xff\xd8", out[:60]
            img = np.asarray(PIL.open(io.BytesIO(out)).convert("RGB")).astype(np.float64)
            rq = _abi.default_request(width=96, height=64, divisions=1, spp=8, max_bounces=4, seed=5)
            ref, _, _ = oracle.render(rq, None, obj.build_world(o + m, len(o)), backend=1)
            assert np.abs(img - ref.reshape(64, 96, 3)).mean() < 6.0
    finally:
        ctl.stop()


@pytest.mark.parametrize("flags", [_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_NO_CULL_WALK,
                                   _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK,
                                   _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES])
def test_quantised_walk_fallback_lanes(ndev, oracle, flags):
    """Rays the grid form cannot carry walk the exact nodes inside the quantised kernel: (a) a tiny cluster seen from
    far outside its 16-bit grid (|origin| > 2^18 grid units: every camera ray), (b) axis-parallel directions (aperture 0,
    a sphere row straight ahead: +-0 direction components after the mirror bounce off flat-facing surfaces)."""
    g = np.random.default_rng(77)
    n = 600
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    sph["cx"] = g.uniform(-0.01, 0.01, n)
    sph["cy"] = g.uniform(-0.01, 0.01, n)
    sph["cz"] = -3.0 + g.uniform(-0.01, 0.01, n)
    sph["radius"] = g.uniform(0.0004, 0.0012, n)
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        sph[c] = g.uniform(0.2, 0.9, n)
    sph["roughness"] = g.choice([0.0, 1.0], n)
    rq = _abi.default_request(width=96, height=96, divisions=1, spp=8, max_bounces=4, seed=5)
    rq.fov = 0.02                                           # the cluster fills the frame
    st = _compare(oracle, rq, sph, flags=flags)
    assert st.ray_segments > rq.width * rq.height * rq.spp     # some paths did bounce
    assert st.engine == (5 if flags & _abi.RT_FLAG_CULL_WALK else 3 if flags & _abi.RT_FLAG_QUANT_NODES else st.engine if st.engine in (4, 7) else -1)
    # (b) mirror spheres on the optical axis, pinhole camera: reflected rays with exact zero components
    row = np.zeros(3, _abi.SPHERE_DTYPE)
    row["cz"] = [-4.0, -9.0, -2000.0]
    row["radius"] = [1.0, 2.0, 1900.0]
    row["albedo_r"] = row["albedo_g"] = row["albedo_b"] = 0.9
    row["roughness"] = 1.0
    rq2 = _abi.default_request(width=65, height=65, divisions=1, spp=4, max_bounces=6, seed=9)
    rq2.aperture = 0.0
    # (the three large spheres make the 16-bit grid too coarse for the cluster: the host falls back to the exact nodes here)
    _compare(oracle, rq2, np.concatenate([row, sph]), flags=flags)


@pytest.mark.parametrize("flags", [0, _abi.RT_FLAG_EXACT_NODES, _abi.RT_FLAG_LINEAR_SCAN])
def test_more_than_65536_primitives(ndev, oracle, flags):
    """Primitive indices beyond 16 bits: the path stack switches to 32-bit entries (KParams::path32), leaf references and
    candidate lists carry 17+ bits; spheres and a few triangles (indices >= n_spheres) in one scene."""
    sph = scenes.rand65536(n=70000)
    g = np.random.default_rng(11)
    tri = np.zeros(40, _abi.TRIANGLE_DTYPE)
    for t in range(40):
        c = np.array([g.uniform(-20, 20), g.uniform(0, 8), g.uniform(-40, -6)])
        tri["a"][t], tri["b"][t], tri["c"][t] = c, c + g.uniform(-2, 2, 3), c + g.uniform(-2, 2, 3)
        tri["albedo_r"][t], tri["albedo_g"][t], tri["albedo_b"][t] = g.uniform(0.2, 0.9, 3)
        tri["roughness"][t] = g.choice([0.0, 1.0])
    rq = _abi.default_request(width=96, height=54, divisions=1, spp=2, max_bounces=6, seed=77)
    st = _compare(oracle, rq, sph, tri, flags=flags)
    assert st.engine == {0: 3, _abi.RT_FLAG_EXACT_NODES: 2, _abi.RT_FLAG_LINEAR_SCAN: 1}[flags]


def test_concurrent_host_threads_share_device_and_scene(ndev):
    """C-ABI threading contract: calls on one device (own scenes or one shared scene) serialise and stay correct.
    Four threads render different strips at once, each through its own scene and through one shared scene."""
    import threading
    sph, rq = _small("c3", 256, 144, spp=2, div=8)
    with rt.Scene(0, rt.World(sph)) as ref_sc:
        want = []
        for k in range(8):
            r = rq.copy(); r.division_no = k
            want.append(ref_sc.render_tile(r)[0].copy())
    shared = rt.Scene(0, rt.World(sph))
    errors = []

    def work(tid):
        try:
            own = rt.Scene(0, rt.World(sph))
            for it in range(6):
                k = (tid * 3 + it) % 8
                r = rq.copy(); r.division_no = k
                a = own.render_tile(r)[0]
                b = shared.render_tile(r)[0]
                outs, _, _ = shared.render_tiles([r, r])
                if not (np.array_equal(a, want[k]) and np.array_equal(b, want[k]) and np.array_equal(outs[1], want[k])):
                    errors.append((tid, it, k))
            own.close()
        except Exception as e:          # pragma: no cover
            errors.append((tid, repr(e)))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    shared.close()
    assert not errors, errors


@pytest.mark.parametrize("kind", ["many_big", "camera_inside", "odd_radii", "equal_spheres"])
def test_culled_walk_odd_scenes(ndev, oracle, kind):
    """Scenes that stress the culled walk's side conditions (all forced through it, and through the plain quantised walk):
    more large spheres than its 'big' list holds (the slack radius becomes large: little is culled, nothing may be lost),
    a camera inside a pile of overlapping spheres (box entry distances of zero), zero / negative / tiny radii, and many
    coincident equal spheres (exact distance ties between candidates that arrive out of depth-first order)."""
    g = np.random.default_rng({"many_big": 1, "camera_inside": 2, "odd_radii": 3, "equal_spheres": 4}[kind])
    n = 3000
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.uniform(-30, 30, n), g.uniform(-2, 12, n), g.uniform(-70, -4, n)
    sph["radius"] = g.uniform(0.1, 0.5, n)
    if kind == "many_big":
        big = g.choice(n, 40, replace=False)
        sph["radius"][big] = g.uniform(8.0, 30.0, 40)
        sph["cy"][big] = -40.0
    elif kind == "camera_inside":
        sph["cx"], sph["cy"], sph["cz"] = g.uniform(-3, 3, n), g.uniform(-3, 3, n), g.uniform(-8, 2, n)
        sph["radius"] = g.uniform(0.3, 1.2, n)
    elif kind == "odd_radii":
        sph["radius"][:300] = 0.0
        sph["radius"][300:600] = -g.uniform(0.1, 0.5, 300)
        sph["radius"][600:900] = 1e-6
    else:
        sph["cx"][:1500], sph["cy"][:1500], sph["cz"][:1500] = sph["cx"][1500:], sph["cy"][1500:], sph["cz"][1500:]
        sph["radius"][:1500] = sph["radius"][1500:]
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        sph[c] = g.uniform(0.2, 0.9, n)
    sph["roughness"] = g.choice([0.0, 0.5, 1.0], n)
    sph["emission"] = np.where(g.uniform(size=n) < 0.02, 4.0, 0.0)
    rq = _abi.default_request(width=160, height=90, divisions=1, spp=3, max_bounces=6, seed=31)
    base = _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES
    a = _compare(oracle, rq, sph, flags=base | _abi.RT_FLAG_CULL_WALK)
    b = _compare(oracle, rq, sph, flags=base | _abi.RT_FLAG_NO_CULL_WALK)
    assert a.ray_segments == b.ray_segments
    if a.engine in (3, 5):                               # (a grid too coarse for the scene falls back to the exact nodes)
        assert a.engine == 5 and b.engine == 3 and a.broad_candidates <= b.broad_candidates
    # the other engines on the same scene, and the LDS-resident tree on a part of it.  (A sphere of negative radius has an
    # AABB with lo > hi, which the reference's sign-selected slab test rejects: the exact-node L2 walk once entered it.)
    c = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE)
    d = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_LINEAR_SCAN)
    part = sph[:900] if kind != "odd_radii" else np.concatenate([sph[200:500], sph[900:1400]])
    e = _compare(oracle, rq, part, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_CULL_WALK)
    e7 = _compare(oracle, rq, part, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_CULL_WALK)
    f = _compare(oracle, rq, part, flags=0)
    assert c.engine == 2 and d.engine in (0, 1) and e.engine == 4 and f.engine in (4, 5, 7)
    # (the culled LDS-resident tree wherever its bound is valid: not with an inverted box or an unbounded slack radius)
    assert e7.engine in (4, 7) and e7.ray_segments == e.ray_segments and (e7.engine == 4 or e7.broad_candidates <= e.broad_candidates)


_EXTREME = {
    "tiny_fov": dict(fov=1e-3), "wide_fov": dict(fov=3.1), "huge_aperture": dict(aperture=8.0, focus_distance=12.0),
    "zero_focus": dict(focus_distance=0.0), "negative_focal": dict(focal_length=-1.0), "t_min_zero": dict(t_min=0.0),
    "t_min_negative": dict(t_min=-5.0), "t_window_empty": dict(t_min=10.0, t_max=5.0), "t_max_tiny": dict(t_max=0.01),
    "seed_all_ones": dict(seed=0xFFFFFFFFFFFFFFFF), "seed_zero": dict(seed=0), "one_row_strips": dict(divisions=54, division_no=53),
    "single_sample": dict(spp=1, max_bounces=1), "zero_fov": dict(fov=0.0), "pi_fov": dict(fov=3.1415927), "negative_aperture": dict(aperture=-0.5),
    "huge_focus": dict(focus_distance=1e30, aperture=0.3), "zero_focal": dict(focal_length=0.0),
}


@pytest.mark.parametrize("case", sorted(_EXTREME))
def test_extreme_knobs_and_materials(ndev, oracle, case):
    """Request knobs at and beyond the edges of what a camera means (the reference does not validate them either), and
    materials outside [0, 1] (albedo > 1, negative emission, roughness 2 and -1): same bits as the oracle in the linear engine,
    the LDS-resident tree, the exact and the quantised L2 walks."""
    g = np.random.default_rng(17)
    n = 300
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.uniform(-8, 8, n), g.uniform(-3, 5, n), g.uniform(-20, -2, n)
    sph["radius"] = g.uniform(0.2, 0.9, n)
    sph["cx"][0], sph["cy"][0], sph["cz"][0], sph["radius"][0] = 0.0, -203.0, -10.0, 200.0
    sph["albedo_r"], sph["albedo_g"], sph["albedo_b"] = g.uniform(0.0, 1.6, n), g.uniform(0.0, 1.0, n), g.uniform(0.0, 1.2, n)
    sph["roughness"] = g.choice([0.0, 0.3, 1.0, 2.0, -1.0], n)
    sph["emission"] = np.where(g.uniform(size=n) < 0.1, g.choice([3.0, -2.0, 0.0, 1e-30], n), 0.0)
    kw = dict(width=120, height=54, divisions=2, division_no=1, spp=2, max_bounces=5, seed=5)
    kw.update(_EXTREME[case])
    rq = _abi.default_request(**kw)
    for flags in (_abi.RT_FLAG_LINEAR_SCAN, 0, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES):
        _compare(oracle, rq, sph, flags=flags)


def test_wide_image_and_far_camera_scene(ndev, oracle):
    """A strip wider than 65 535 pixels (pixel columns beyond 16 bits, tiles_x > 1024), and a scene far from the origin with a
    large common offset (the cancellation regime of the expanded broad phase and of the grid form of the quantised walk)."""
    sph = scenes.cornell16()
    rq = _abi.default_request(width=70001, height=2, divisions=2, division_no=1, spp=1, max_bounces=2, seed=77)
    _compare(oracle, rq, sph)
    g = np.random.default_rng(23)
    n = 5000
    far = np.zeros(n, _abi.SPHERE_DTYPE)
    far["cx"], far["cy"], far["cz"] = g.uniform(-30, 30, n), g.uniform(-20, 20, n), -5000.0 + g.uniform(-60, -4, n)   # (the camera sits at the origin, looking down -z)
    far["radius"] = g.uniform(0.1, 0.6, n)
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        far[c] = g.uniform(0.2, 0.9, n)
    far["roughness"] = g.choice([0.0, 1.0], n)
    rq2 = _abi.default_request(width=96, height=64, divisions=1, spp=2, max_bounces=4, seed=8, fov=0.02, t_max=1e5)
    for flags in (0, _abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES):
        st = _compare(oracle, rq2, far, flags=flags)
    assert st.ray_segments > rq2.width * rq2.height * rq2.spp          # the cluster is in view: some paths bounce


def test_degenerate_and_axis_aligned_triangles(ndev, oracle):
    """Triangles the reference does not reject: zero area (two equal vertices, three collinear ones), boxes of zero extent
    along an axis (floor / wall quads) seen by rays that run inside their plane (pinhole camera at the plane's height: the
    0 * inf slabs of ray.rs:174-194), one huge and many tiny triangles, with a few spheres; every engine."""
    g = np.random.default_rng(41)
    tris = []

    def tri(a, b, c, alb=(0.7, 0.6, 0.5), rough=0.0, emis=0.0):
        tris.append((a, b, c, alb[0], alb[1], alb[2], rough, emis))

    tri((-6, 0, -2), (6, 0, -2), (6, 0, -14))                       # floor quad at the camera's height y = 0: rays with d.y = 0 lie in it
    tri((-6, 0, -2), (6, 0, -14), (-6, 0, -14))
    tri((-6, -3, -14), (6, -3, -14), (6, 5, -14), (0.3, 0.5, 0.8), 1.0)   # back wall z = -14 (mirror)
    tri((-6, -3, -14), (6, 5, -14), (-6, 5, -14), (0.3, 0.5, 0.8), 1.0)
    tri((0, -3, -2), (0, 5, -2), (0, 5, -14), (0.8, 0.3, 0.3))      # wall in the plane x = 0: contains the camera's optical axis
    tri((1, 1, -5), (1, 1, -5), (2, 2, -6))                         # two equal vertices
    tri((-2, 1, -5), (-1, 2, -6), (0, 3, -7))                       # collinear
    tri((3, 3, -9), (3, 3, -9), (3, 3, -9))                         # a point
    tri((-500, -4, 100), (500, -4, 100), (0, -4, -900), (0.5, 0.5, 0.5))  # huge
    for _ in range(60):                                             # tiny ones
        c = np.array([g.uniform(-4, 4), g.uniform(-2, 3), g.uniform(-12, -3)])
        tri(tuple(c), tuple(c + g.uniform(-1e-3, 1e-3, 3)), tuple(c + g.uniform(-1e-3, 1e-3, 3)), (0.9, 0.9, 0.2), 0.0, 5.0)
    for _ in range(40):
        c = np.array([g.uniform(-4, 4), g.uniform(-2, 3), g.uniform(-12, -3)])
        tri(tuple(c), tuple(c + g.uniform(-1, 1, 3)), tuple(c + g.uniform(-1, 1, 3)), tuple(g.uniform(0.2, 0.9, 3)), float(g.choice([0.0, 1.0])))
    tr = np.array(tris, dtype=_abi.TRIANGLE_DTYPE)
    sph = np.zeros(5, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"], sph["radius"] = [-3, 3, 0, -1, 2], [1, 1, 2, -1, -2], [-6, -8, -10, -4, -5], [0.8, 0.6, 1.0, 0.4, 0.5]
    sph["albedo_r"] = sph["albedo_g"] = sph["albedo_b"] = 0.8
    sph["roughness"] = [0.0, 1.0, 0.5, 1.0, 0.0]
    # odd image sizes put a pixel column / row on the optical axis; aperture 0 keeps those rays exactly axis-parallel
    rq = _abi.default_request(width=97, height=65, divisions=1, spp=3, max_bounces=6, seed=13, aperture=0.0)
    for flags in (0, _abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_FULL_CHAIN,
                  _abi.RT_FLAG_NO_BVH_CULL):
        _compare(oracle, rq, sph, tr, flags=flags)
    _compare(oracle, rq, None, tr, flags=0)                          # triangles only


@pytest.mark.parametrize("n", [700, 5000])
def test_hundreds_of_candidates_per_ray(ndev, oracle, n):
    """A pile of large overlapping spheres around the optical axis: every ray enters dozens to hundreds of leaf boxes, so the leaf lists
    of the walks overflow all the time — the branch-free steps' stalled lanes, the flushes between blocks, the compacted root
    tests with every lane's list full, the culled walk's early flushes.  n = 700 fits the LDS-resident tree, 5000 does not."""
    g = np.random.default_rng(n)
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.normal(0, 1.5, n), g.normal(0, 1.5, n), -12.0 + g.normal(0, 3.0, n)
    sph["radius"] = g.uniform(1.0, 4.0, n)
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        sph[c] = g.uniform(0.3, 0.9, n)
    sph["roughness"] = g.choice([0.0, 1.0], n)
    rq = _abi.default_request(width=80, height=48, divisions=1, spp=2, max_bounces=4, seed=3)
    engines = {}
    for flags in (0, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_NO_CULL_WALK,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK, _abi.RT_FLAG_LINEAR_SCAN):
        st = _compare(oracle, rq, sph, flags=flags)
        engines[flags] = st.engine
        if flags and not (flags & _abi.RT_FLAG_LINEAR_SCAN) and not (flags & _abi.RT_FLAG_CULL_WALK):
            assert st.broad_candidates > 20 * st.ray_segments, st.broad_candidates / st.ray_segments
    assert engines[0] in (2, 3, 4, 5, 7)         # (whatever the host picks for such a pile: deep tree, large slack radius)


def test_thousands_of_identical_spheres(ndev, oracle):
    """2 000 copies of one mirror sphere, 2 000 of a diffuse one and a ground: every hit is an exact distance tie between
    thousands of candidates (the first leaf in depth-first order wins, shapes/mod.rs:177-182), the centroid bounds of the
    build collapse (halving fallback, bvh_impl.rs), the tree is as deep as it gets for its size."""
    sph = np.zeros(4001, _abi.SPHERE_DTYPE)
    sph["cx"][:2000], sph["cy"][:2000], sph["cz"][:2000], sph["radius"][:2000] = -1.5, 0.5, -6.0, 1.5
    sph["cx"][2000:4000], sph["cy"][2000:4000], sph["cz"][2000:4000], sph["radius"][2000:4000] = 1.8, 0.2, -5.0, 1.2
    sph["cx"][4000], sph["cy"][4000], sph["cz"][4000], sph["radius"][4000] = 0.0, -101.0, -6.0, 100.0
    g = np.random.default_rng(5)
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        sph[c] = g.uniform(0.2, 0.95, 4001)                  # the copies differ in colour: the winner of a tie shows
    sph["roughness"][:2000] = 1.0
    rq = _abi.default_request(width=64, height=40, divisions=1, spp=2, max_bounces=3, seed=1)
    for flags in (0, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_NO_CULL_WALK,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK, _abi.RT_FLAG_LINEAR_SCAN):
        _compare(oracle, rq, sph, flags=flags)


def test_tall_image_deep_paths_many_samples(ndev, oracle):
    """A strip near the bottom of an image 70 000 rows tall (global rows beyond 16 bits); the 62-bounce limit on the default
    engines of a 1 024-sphere scene (the path stack then does not fit beside an LDS-resident tree: the host must fall back);
    500 samples per pixel (long per-pixel RNG streams and sums)."""
    sph, _ = scenes.config("c3")
    rq = _abi.default_request(width=5, height=70000, divisions=700, division_no=698, spp=2, max_bounces=3, seed=2)
    _compare(oracle, rq, sph)
    rq2 = _abi.default_request(width=48, height=30, divisions=1, spp=2, max_bounces=_abi.RT_MAX_BOUNCES, seed=4)
    st = _compare(oracle, rq2, sph)
    assert st.engine in (2, 3, 4)
    _compare(oracle, rq2, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES)
    rq3 = _abi.default_request(width=12, height=8, divisions=1, spp=500, max_bounces=6, seed=6)
    _compare(oracle, rq3, sph)
    _compare(oracle, rq3, scenes.cornell16())


@pytest.mark.parametrize("spp,w,h", [(4096, 7, 3), (1000, 9, 5), (257, 70, 3), (33, 130, 7), (17, 67, 5)])
def test_sample_counts_up_to_the_limit(ndev, oracle, spp, w, h):
    """Round 4: the sample units at the counts the host rules change at — 17 (every tile in quarters), 33 (sixteenths), 257 (8 pixel
    slots per wave instead of 16), 1 000, and RT_MAX_SPP = 4 096 (4 slots of 4 097 records) — on frames whose width is no multiple of
    the part width, through the linear scan, the LDS-resident tree (plain and culled) and the L2-gather walks."""
    assert spp <= _abi.RT_MAX_SPP
    rq = _abi.default_request(width=w, height=h, divisions=1, spp=spp, max_bounces=6, seed=1234 + spp)
    st = _compare(oracle, rq, scenes.cornell16(), flags=_abi.RT_FLAG_LINEAR_SCAN)
    assert st.engine == 0
    sph, _ = scenes.config("c3")
    if spp <= 1000:                                   # (the oracle's share of the run time: a 1 024-sphere scene at 4 096 spp is left out)
        st = _compare(oracle, rq, sph)
        assert st.engine in (4, 7)
        _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_CULL_WALK)
        _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE)
        _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES | _abi.RT_FLAG_CULL_WALK)


def test_subnormal_values(ndev, oracle):
    """binary32 subnormals in the scene (radii, coordinates, albedo, emission) and therefore in intermediate results: the kernels
    must not flush them (the reference's SSE arithmetic does not)."""
    g = np.random.default_rng(61)
    n = 120
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.uniform(-4, 4, n), g.uniform(-2, 3, n), g.uniform(-12, -2, n)
    sph["radius"] = g.uniform(0.2, 0.9, n)
    sph["cx"][:10] = g.uniform(-1, 1, 10) * 1e-40                # subnormal coordinates
    sph["radius"][10:20] = 1e-41                                # subnormal radii
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        sph[c] = g.uniform(0.2, 0.9, n)
    sph["albedo_r"][20:50] = 3e-39                              # products with subnormal results
    sph["albedo_g"][20:50] = 1e-20
    sph["emission"][50:80] = 1e-40
    sph["emission"][80:90] = 2e-38
    sph["roughness"] = g.choice([0.0, 1.0, 1e-42], n)
    rq = _abi.default_request(width=96, height=60, divisions=1, spp=4, max_bounces=6, seed=19)
    for flags in (0, _abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES):
        _compare(oracle, rq, sph, flags=flags)


def test_grazing_rays_over_triangle_floors(ndev, oracle):
    """The regime the triangle bound of the culled walk has to survive: rays almost parallel to layers of small floor triangles a
    hair below the camera, so that the reference's determinant is a few times its 1e-5 threshold and its roots are off by
    per cent; edges of 0.49 put |e1||e2| next to the bound's limit of 0.25.  Culled and plain walks over the exact nodes, the
    quantised walk, the linear engine."""
    g = np.random.default_rng(7)
    tris = []
    for layer, y in enumerate((-1e-3, -2.5e-3, -6e-3, -2e-2)):
        alb = [(0.9, 0.2, 0.2), (0.2, 0.9, 0.2), (0.2, 0.2, 0.9), (0.8, 0.8, 0.2)][layer]
        for ix in range(-8, 8):
            for iz in range(2, 60):
                if g.uniform() < 0.35:
                    continue                                              # holes: the layers below show through
                e = 0.49                                                  # |e1||e2| = 0.24: just inside the bound's K <= 0.25
                x0, z0 = e * ix, -e * iz
                jy = y * (1.0 + 0.2 * g.uniform())
                tris.append(((x0, jy, z0), (x0 + e, jy, z0), (x0, jy, z0 - e), *alb, float(g.choice([0.0, 1.0])), 0.0))
                tris.append(((x0 + e, jy, z0 - e), (x0, jy, z0 - e), (x0 + e, jy, z0), *alb, 0.0, 0.0))
    tr = np.array(tris, dtype=_abi.TRIANGLE_DTYPE)
    sph = np.zeros(30, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.uniform(-3, 3, 30), g.uniform(0.0, 0.6, 30), g.uniform(-28, -3, 30)
    sph["radius"] = g.uniform(0.1, 0.4, 30)
    sph["albedo_r"] = sph["albedo_g"] = sph["albedo_b"] = 0.7
    sph["roughness"] = g.choice([0.0, 1.0], 30)
    # a narrow vertical field of view around the horizon: every pixel row is a grazing angle of 1e-5 ... 1e-2
    rq = _abi.default_request(width=96, height=81, divisions=1, spp=3, max_bounces=4, seed=29, aperture=0.0, fov=0.02, t_max=200.0)
    engines = []
    for flags in (_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE | _abi.RT_FLAG_CULL_WALK,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_LDS_TREE | _abi.RT_FLAG_NO_CULL_WALK,
                  0, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES, _abi.RT_FLAG_LINEAR_SCAN):
        st = _compare(oracle, rq, sph, tr, flags=flags)
        engines.append(st.engine)
    assert engines[0] == 6 and engines[1] == 2
    assert st.ray_segments > rq.width * rq.height * rq.spp            # the floors are hit


def test_sqrt_rn_is_the_ieee_square_root_on_every_f32(ndev):
    """sqrt_rn (rt_kernel.hip.h: v_rsq_f32 + one coupled Newton step, the compiler's sequence for the operands outside its
    domain) against __builtin_sqrtf on the device, all 2^32 bit patterns — negative, NaN, zero, subnormal and infinite ones
    included — in both of its forms.  The oracle's sqrtf is the CPU's IEEE one; the committed golden vectors tie the two together."""
    lib = _abi.load_debug()                  # (a hook of the test library, compiled from the same kernel header)
    lib.rt_debug_sqrt_selftest.restype = C.c_int
    lib.rt_debug_sqrt_selftest.argtypes = [C.c_int, C.c_uint32, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    total = 0
    for from_ in range(0, 1 << 32, 1 << 30):
        bad = C.c_ulonglong(12345)
        assert lib.rt_debug_sqrt_selftest(0, from_, 1 << 30, C.byref(bad)) == 0
        total += bad.value
    assert total == 0


@pytest.mark.parametrize("n", [33, 65, 256, 700, 1024, 1090, 1160])
def test_lds_tree_around_its_fit_boundary(ndev, oracle, n):
    """The LDS-resident tree at sizes from just above the linear-scan rule to past what a CU's LDS holds: biased node references,
    the NaN field behind the tree that a leaf's address falls into (the highest primitive index reads its last 19 dwords), leaf
    lists of 12 ... 16 slots as the tree leaves room, partial rounds of root tests.  Whatever engine the host picks at a size, the
    frame is the oracle's; up to the headline scene's size it is engine 4, and the L2 walk of the same scene gives the same bits."""
    sph = scenes.rand1024(seed=0x1D5 + n, n=n)
    rq = _abi.default_request(width=128, height=72, divisions=1, spp=3, max_bounces=7, seed=1000 + n)
    st = _compare(oracle, rq, sph, flags=0)
    if n <= 1024:
        assert st.engine == 4
    else:
        assert st.engine in (2, 3, 4, 5)
    st2 = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_NO_LDS_TREE)
    assert st2.engine != 4 and st2.ray_segments == st.ray_segments


def test_lds_tree_near_axis_rays_and_long_lists(ndev, oracle):
    """Rays almost parallel to the z axis (a tiny field of view: huge inverse-direction components) through a pile of overlapping
    spheres — leaf lists that fill up between blocks, so that the capacity flush and the partial rounds of root tests both run — in
    the LDS-tree kernel."""
    g = np.random.default_rng(77)
    n = 400
    sph = np.zeros(n, dtype=_abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.uniform(-1.5, 1.5, n), g.uniform(-1.5, 1.5, n), g.uniform(-9.0, -6.0, n)
    sph["radius"] = g.uniform(0.6, 1.4, n)                                  # a pile: every ray meets dozens of leaf boxes
    sph["albedo_r"], sph["albedo_g"], sph["albedo_b"] = g.uniform(0.2, 0.9, n), g.uniform(0.2, 0.9, n), g.uniform(0.2, 0.9, n)
    sph["roughness"] = np.where(g.uniform(size=n) < 0.5, 1.0, 0.0)          # mirrors keep directions axis-parallel after a bounce
    sph["emission"] = np.where(g.uniform(size=n) < 0.05, 3.0, 0.0)
    # a camera looking down -z with no aperture and a tiny field of view
    rq = _abi.default_request(width=65, height=65, divisions=1, spp=2, max_bounces=6, aperture=0.0, fov=0.02, seed=5)
    st = _compare(oracle, rq, sph, flags=_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_CULL_WALK)
    assert st.engine == 4
    assert st.broad_candidates > 8 * st.ray_segments // 4                   # long lists: several candidates per segment
    st7 = _compare(oracle, rq, sph, flags=0)                                # (by default a pile is walked nearer child first, culled)
    assert st7.engine == 7 and st7.broad_candidates < st.broad_candidates


@pytest.mark.parametrize("scale", [1e12, 1e19])
def test_lds_tree_with_astronomic_coordinates(ndev, oracle, scale):
    """Scenes whose coordinates are large enough for the slab arithmetic to overflow (products of 1e19-sized values: inf, and
    inf - inf = NaN inside the box tests) through the LDS-tree engines, plain and culled, with an unbounded t window: the
    branch-free step keeps no clamp on its stack pointer and relies on node DONE's all-of-space box (round-3 advisor); lanes whose
    origin or inverse direction is not finite take the clamped step.  Same bits as the oracle."""
    g = np.random.default_rng(123)
    n = 300
    sph = np.zeros(n, _abi.SPHERE_DTYPE)
    sph["cx"], sph["cy"], sph["cz"] = g.uniform(-6, 6, n) * scale, g.uniform(-3, 4, n) * scale, g.uniform(-14, -2, n) * scale
    sph["radius"] = g.uniform(0.2, 0.9, n) * scale
    for c in ("albedo_r", "albedo_g", "albedo_b"):
        sph[c] = g.uniform(0.2, 0.9, n)
    sph["roughness"] = g.choice([0.0, 1.0], n)
    rq = _abi.default_request(width=96, height=64, divisions=1, spp=2, max_bounces=6, seed=9, t_max=float("inf"))
    for flags in (_abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_NO_CULL_WALK,
                  _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_EXACT_NODES | _abi.RT_FLAG_CULL_WALK, 0):
        st = _compare(oracle, rq, sph, flags=flags)
        assert st.engine in (4, 7) or flags == 0
