"""The C++ oracle against the independent numpy.float32 restatement (oracle/restate_np.py),
bit for bit, on tiny images.  Two restatements written separately from the reference source
agreeing is the strongest pin available without a Rust toolchain ("parity unpinned" otherwise)."""
import numpy as np
import pytest

from oracle import restate_np
from ray_tracer_s8_amd import _abi, scenes


def _both(oracle, rq, sph, tri=None, world_index=None):
    rgb, f32, info = oracle.render(rq, sph, tri, backend=0, want_f32=True, world_index=world_index)
    hs = rq.height // rq.divisions
    rgb2, f2, segs2 = restate_np.render(rq, sph, tri, world_index=world_index)
    assert np.array_equal(rgb.reshape(hs, rq.width, 3), rgb2)
    assert np.array_equal(f32.reshape(hs, rq.width, 3).view(np.uint32), f2.view(np.uint32))
    assert info["ray_segments"] == segs2


def test_cross_cornell(oracle):
    sph, rq = scenes.config("c2")
    rq.width, rq.height, rq.divisions, rq.division_no, rq.spp = 24, 14, 2, 1, 2
    _both(oracle, rq, sph)


def test_cross_rand_spheres(oracle):
    sph = scenes.rand1024(n=48)
    rq = _abi.default_request(width=20, height=12, divisions=1, spp=2, max_bounces=6, seed=11)
    _both(oracle, rq, sph)


def test_cross_triangles(oracle):
    sph, tri = scenes.quad_room()
    rq = _abi.default_request(width=20, height=12, divisions=1, spp=2, max_bounces=5, seed=5)
    _both(oracle, rq, sph, tri)


def test_cross_world_order(oracle):
    """An interleaved world of identical copies (every hit an exact tie: the order of `world` alone picks the winner): the
    two restatements agree under the same world_index, and differ from the spheres-then-triangles order."""
    from _world_cases import interleave, tie_world
    sph, tri = tie_world(7, n_groups=3, dup=3)
    wi = interleave(len(sph), len(tri), 3)
    rq = _abi.default_request(width=18, height=12, divisions=1, spp=2, max_bounces=3, seed=2)
    _both(oracle, rq, sph, tri, wi)
    a, _, _ = oracle.render(rq, sph, tri, backend=0, world_index=wi)
    b, _, _ = oracle.render(rq, sph, tri, backend=0)
    assert not np.array_equal(a, b)


def test_cross_rng_stream(oracle):
    r = restate_np.SmallRng.seed_from_u64(restate_np.pixel_seed(42, 1234))
    st = oracle.seed_from_u64(oracle.pixel_seed(42, 1234))
    assert [int(x) for x in st] == r.s
    assert [r.next_u64() for _ in range(8)] == [int(x) for x in oracle.xoshiro_from_state(st, 8)]
