"""The order of `RenderInfo.world` across the C-ABI (include/rt_tile.h "the world's order", ABI v3).

The reference's world is ONE list mixing spheres and triangles (lib.rs:11, shapes/mod.rs:23-27) and its order is
observable: BVH::build numbers the shapes by position (bvh_impl.rs:421-427), which fixes the halves of the
split_at(len / 2) fallback (:277-291), the depth-first order of the leaves, and so the winner among hits at exactly equal
distance (min_by keeps the first, shapes/mod.rs:177-182).  The ABI carries two typed arrays plus `world_index`.

CPU tests: the oracle's world_index against the independent statement "pass the list already in that order"; the
product's host builder (rt_bvh.h through the g++ harness) against the oracle's tree over the reordered boxes, on
coincident centroids; the JSON codec.  GPU tests: every engine against the oracle on interleaved worlds with exact ties."""
import ctypes as C

import numpy as np
import pytest

from ray_tracer_s8_amd import _abi, wire
from ray_tracer_s8_amd.interface import World
from test_host_bvh import host, _p  # noqa: F401  (fixture)
from _world_cases import interleave, tie_world


def small_request(**kw):
    d = dict(width=96, height=64, divisions=1, spp=3, max_bounces=4, seed=11)
    d.update(kw)
    return _abi.default_request(**d)


@pytest.mark.parametrize("backend", [0, 1])
def test_oracle_world_index_equals_reordered_arrays(oracle, backend):
    """For a one-type world, `world_index` must mean exactly: render the array sorted into world order.  (Independent of
    the mechanism: the second render passes no world_index at all.)  And the order must be observable on this scene."""
    sph, _ = tie_world(1, with_tris=False)
    rq = small_request()
    wi = interleave(len(sph), 0, 5)
    order = np.argsort(wi)                                          # order[w] = the sphere at world position w
    a, af, ia = oracle.render(rq, sph, backend=backend, world_index=wi, want_f32=True)
    b, bf, ib = oracle.render(rq, sph[order], backend=backend, want_f32=True)
    assert np.array_equal(a, b) and np.array_equal(af.view(np.uint32), bf.view(np.uint32)) and ia["ray_segments"] == ib["ray_segments"]
    plain, _, _ = oracle.render(rq, sph, backend=backend)
    assert not np.array_equal(a, plain), "the scene does not make the world's order visible"
    ident, _, _ = oracle.render(rq, sph, backend=backend, world_index=np.arange(len(sph), dtype=np.uint32))
    assert np.array_equal(ident, plain)


def test_oracle_rejects_a_bad_world_index(oracle):
    sph, tri = tie_world(2)
    with pytest.raises(ValueError):
        oracle.render(small_request(), sph, tri, world_index=np.zeros(len(sph) + len(tri), np.uint32))
    with pytest.raises(ValueError):
        oracle.render(small_request(), sph, tri, world_index=np.arange(3, dtype=np.uint32))


def test_product_builder_follows_the_world_order(host, oracle):
    """rt_bvh.h build(prim, order) against the oracle's bvh_build over the boxes laid out in world order: the same
    candidates in the same order for every ray, on sets with coincident centroids (the split_at(len / 2) fallback,
    bvh_impl.rs:277-291, is where the order of the list decides the tree)."""
    g = np.random.default_rng(9)
    for n, groups in ((12, 3), (64, 8), (300, 30), (7, 7)):
        c = np.repeat(g.uniform(-6, 6, (groups, 3)), n // groups, axis=0)[:n]
        r = np.repeat(g.uniform(0.2, 0.9, (groups, 1)), n // groups, axis=0)[:n]
        r = r * (1.0 + 0.25 * (np.arange(n)[:, None] % 3))              # same centroid, different extents
        b = np.concatenate([c - r, c + r], axis=1).astype(np.float32)
        n = len(b)
        wi = g.permutation(n).astype(np.uint32)
        order = np.argsort(wi).astype(np.uint32)
        differs = 0
        for k in range(30):
            o = g.uniform(-10, 10, 3).astype(np.float32)
            d = (c[k % n] + g.normal(size=3) * 0.1 - o).astype(np.float32)
            out = np.zeros(n + 1, np.uint32)
            nn, dp = C.c_uint32(0), C.c_uint32(0)
            m = host.host_bvh_traverse_ordered(_p(b), C.c_uint32(n), _p(order), _p(o), _p(d), _p(out), C.c_uint32(n + 1),
                                               C.byref(nn), C.byref(dp))
            got = out[:m].tolist()
            ref_pos, nn_ref = oracle.bvh_traverse_boxes(b[order], o, d)       # positions in the world list
            assert got == [int(order[w]) for w in ref_pos], (n, k)
            assert nn.value == nn_ref == 2 * n - 1
            plain = np.zeros(n + 1, np.uint32)
            mp = host.host_bvh_traverse(_p(b), C.c_uint32(n), _p(o), _p(d), _p(plain), C.c_uint32(n + 1), None, None)
            differs += got != plain[:mp].tolist()
        assert differs > 0 or n < 8


def test_json_world_keeps_its_order():
    """The slave's JSON `world` is the list itself: decoding records where each entry stood, encoding writes it back."""
    sph, tri = tie_world(3)
    wi = interleave(len(sph), len(tri), 8)
    w = World(sph, tri, wi)
    objs = wire.world_to_json_obj(w)
    tags = [next(iter(o)) for o in objs]
    assert tags.count("Sphere") == len(sph) and tags.count("Triangle") == len(tri)
    assert any(a != b for a, b in zip(tags, sorted(tags)))                 # really interleaved
    for i in range(len(sph)):                                              # every primitive stands at its position
        assert objs[int(wi[i])]["Sphere"]["p_albedo_at"]["r"] == float(sph[i]["albedo_r"])
    back = wire.world_from_json_obj(objs)
    # decoding numbers the arrays in list order, so the same LIST comes back (arrays and index may be re-sorted)
    assert wire.world_to_json_obj(back) == objs
    assert wire.world_to_json_text(w) == wire.world_to_json_text(back)
    # spheres-then-triangles lists need no index
    assert wire.world_from_json_obj(wire.world_to_json_obj(World(sph, tri))).world_index is None
    info = wire.decode_render_info(wire.encode_render_info(wire.RenderInfo(w, wire.RenderMeta(64, 96, 4), 1)))
    assert wire.world_to_json_obj(info.world) == objs


# ------------------------------------------------------------------------------------------------ GPU
ENGINES = [0, _abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_BVH_TRAVERSE, _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_NO_LDS_TREE,
           _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_NO_LDS_TREE | _abi.RT_FLAG_CULL_WALK,
           _abi.RT_FLAG_BVH_TRAVERSE | _abi.RT_FLAG_QUANT_NODES, _abi.RT_FLAG_NO_BVH_CULL,
           _abi.RT_FLAG_NO_BVH_CULL | _abi.RT_FLAG_EXACT_SCAN, _abi.RT_FLAG_FULL_CHAIN]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_every_engine_follows_the_world_order(ndev, oracle, seed):
    """Interleaved sphere / triangle worlds in which every hit is an exact tie between identical copies: RGB8, f32 bits and
    segment counts equal the oracle's under the same world_index in every engine and in both semantics, differ from the
    spheres-then-triangles order, and a one-type world equals its reordered arrays."""
    import ray_tracer_s8_amd as rt
    sph, tri = tie_world(seed)
    wi = interleave(len(sph), len(tri), 20 + seed)
    rq = small_request(seed=seed)
    seen = {}
    with rt.Scene(0, World(sph, tri, wi)) as sc, rt.Scene(0, World(sph, tri)) as plain_sc:
        for fl in ENGINES:
            r = rq.copy()
            r.flags = fl
            got, gf, st = sc.render_tile(r, want_f32=True)
            backend = 0 if fl & _abi.RT_FLAG_NO_BVH_CULL else 1
            want, wf, info = oracle.render(r, sph, tri, backend=backend, want_f32=True, world_index=wi)
            assert np.array_equal(got, want), (seed, fl, int((got != want).sum()))
            assert np.array_equal(gf.view(np.uint32), wf.view(np.uint32)) and st.ray_segments == info["ray_segments"]
            seen[fl] = got
        p, _, _ = plain_sc.render_tile(rq)
        assert not np.array_equal(p, seen[0]), "the world's order is not visible in this scene"
    order = np.argsort(wi[:len(sph)])
    with rt.Scene(0, World(sph, world_index=np.argsort(order).astype(np.uint32))) as a, rt.Scene(0, World(sph[order])) as b:
        for fl in (0, _abi.RT_FLAG_LINEAR_SCAN, _abi.RT_FLAG_NO_BVH_CULL):
            r = rq.copy()
            r.flags = fl
            assert np.array_equal(a.render_tile(r)[0], b.render_tile(r)[0]), fl


@pytest.mark.gpu
def test_world_order_through_frame_paths_and_wire(ndev, oracle):
    """world_index through rt_render_tile, rt_render_frame, a frame context and the JSON-decoded world of a slave."""
    import ray_tracer_s8_amd as rt
    from ray_tracer_s8_amd.interface import RenderInfo, RenderMeta, RenderSettings, Slave
    sph, tri = tie_world(4)
    wi = interleave(len(sph), len(tri), 31)
    w = World(sph, tri, wi)
    rq = small_request(divisions=4, seed=3)
    one = rq.copy()
    one.divisions = 1
    want, _, _ = oracle.render(one, sph, tri, backend=1, world_index=wi)
    img, _ = rt.render_frame_native(w, rq, devices=[0])
    assert np.array_equal(img.reshape(-1), want)
    with rt.FrameContext(devices=[0, 0], world=w) as fc:
        img2, _ = fc.render(rq)
        assert np.array_equal(img2.reshape(-1), want)
    lib = _abi.load()
    out = np.zeros(rq.width * rq.height * 3 // 4, np.uint8)
    r2 = rq.copy()
    r2.division_no = 2
    assert lib.rt_render_tile(0, C.byref(r2), _abi.ptr(sph), len(sph), _abi.ptr(tri), len(tri), _abi.ptr(wi), _abi.ptr(out),
                              out.size, None, None) == 0
    assert np.array_equal(out, want.reshape(4, -1)[2])
    # a slave fed the JSON text of the interleaved list
    info = wire.decode_render_info(wire.encode_render_info(RenderInfo(w, RenderMeta(rq.height, rq.width, 4), 1)),
                                   RenderSettings(spp=rq.spp, max_bounces=rq.max_bounces, seed=rq.seed))
    sl = Slave(0)
    try:
        s = sl.render(info)
    finally:
        sl.close()
    assert np.array_equal(s.image, want.reshape(4, -1)[1])
