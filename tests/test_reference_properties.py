"""The vendored bvh crate's own PROPERTY tests for code on this path, re-run here as seeded numpy-driven cases
(no proptest: fixed seeds, 10 000 cases per property) against BOTH the oracle's helpers (oracle/rt_oracle.cpp) and the
product's host BVH helpers (csrc/rt_bvh.h through the g++ harness tests/host/bvh_host.cpp).

Sources (ray-tracer-slave/local-dependencies/bvh/src/):
  aabb.rs:671-849   ten AABB properties (empty / default contain nothing, centre inside, join of two point sets, points
                    relative to centre and size, surface area >= 0 and = 6 s^2 for a cube, volume >= 0 and = |sx sy sz|,
                    with_bounds(aabb[0], aabb[1]) is the same box)
  ray.rs:377-512    a ray aimed at a box's centre intersects it, and turned round it does not unless its origin is
                    inside — for the optimised `intersects_aabb` (the one BVH::traverse uses, ray.rs:174-194) and for the
                    `naive` and `branchless` variants (dead code in the product, restated here in numpy.float32 so that
                    the three are run side by side as in the crate); a ray aimed at a point of a triangle hits it.
  testbase.rs:26-42 the value ranges: "small" +-10e10, "large" +-10e30.  The crate's unconstrained `TupleVec` draws any
                    finite f32; here: log-uniform magnitudes up to 1e37 with random signs (box extents then stay finite,
                    which the crate's properties silently need).

What is restated in this file rather than called: `contains` (aabb.rs:149-156), `approx_contains_eps` (:178-185), `volume`
(:547-550: the product of `size()`), float_eq's `rmax` relative comparison, the two dead slab variants (ray.rs:218-244,
263-285).  Everything else — empty, grow, join, size, center, surface_area, Ray::new, intersects_aabb — is the code under
test.  Parity stays "partial": these pin the candidate filter's box and ray arithmetic, not the render arithmetic."""
import ctypes as C

import numpy as np
import pytest

from test_reference_doc_vectors import host, _p, _host_kat  # noqa: F401

N = 10_000
EPS = np.float32(0.00001)                                  # bvh::EPSILON, lib.rs:80
f32 = np.float32


def _any(g, n):           # stand-in for the crate's unconstrained TupleVec
    mag = np.exp(g.uniform(np.log(1e-30), np.log(1e37), (n, 3)))
    v = (mag * g.choice([-1.0, 1.0], (n, 3))).astype(np.float32)
    v[g.random((n, 3)) < 0.02] = 0.0
    return v


def _small(g, n):         # tuplevec_small_strategy, testbase.rs:26-32
    return g.uniform(-10e10, 10e10, (n, 3)).astype(np.float32)


def _large(g, n):         # tuplevec_large_strategy, testbase.rs:36-42
    return g.uniform(-10e30, 10e30, (n, 3)).astype(np.float32)


def contains(box, p):     # aabb.rs:149-156
    return bool(np.all(p >= box[:3]) and np.all(p <= box[3:]))


def approx_contains_eps(box, p, eps):      # aabb.rs:178-185, f32 arithmetic
    with np.errstate(over="ignore", invalid="ignore"):
        return bool(np.all((p - box[:3]) > -eps) and np.all((p - box[3:]) < eps))


def float_eq_rmax(a, b, tol):              # float_eq: |a - b| <= tol * max(|a|, |b|); equal values (also both inf) pass
    a, b = f32(a), f32(b)
    if a == b:
        return True
    with np.errstate(over="ignore", invalid="ignore"):
        return bool(abs(a - b) <= f32(tol) * max(abs(a), abs(b)))


class Impl:
    """empty / grow / join / size / center / surface_area of one implementation, boxes as 6 floats (min xyz, max xyz)."""

    def __init__(self, name, kat, empty):
        self.name, self._kat, self._empty = name, kat, empty

    def empty(self):
        return self._empty()

    def grow(self, box, p):
        return self._kat(box, box, p)["grow"].copy()

    def join(self, a, b):
        return self._kat(a, b, (0, 0, 0))["join"].copy()

    def stats(self, box):
        return self._kat(box, box, (0, 0, 0))

    def span(self, *pts):                     # AABB::empty().grow(p1).grow(p2)...
        b = self.empty()
        for p in pts:
            b = self.grow(b, p)
        return b


@pytest.fixture(scope="module")
def impls(host, oracle):
    def host_empty():
        e = np.zeros(6, np.float32)
        host.host_empty_box(_p(e))
        return e
    return [Impl("oracle", lambda a, b, p: oracle.aabb_kat(a, b, p), oracle.aabb_empty),
            Impl("rt_bvh.h", lambda a, b, p: _host_kat(host, a, b, p), host_empty)]


def test_empty_and_default_contain_nothing(impls):
    # aabb.rs:684-708 (Default = empty(), :587-591)
    g = np.random.default_rng(1)
    pts = np.concatenate([_any(g, N), _small(g, 100), np.zeros((1, 3), np.float32)])
    for im in impls:
        e = im.empty()
        assert im.stats(e)["is_empty"]
        assert not any(contains(e, p) for p in pts), im.name


def test_aabb_contains_center(impls):
    # aabb.rs:710-723: empty().grow(p1).join_bounded(p2) contains its centre (join_bounded(p) = join(p.aabb()), :440-442)
    g = np.random.default_rng(2)
    a, b = _any(g, N), _any(g, N)
    for im in impls:
        for p1, p2 in zip(a, b):
            box = im.join(im.grow(im.empty(), p1), np.concatenate([p2, p2]))
            assert contains(box, im.stats(box)["center"]), (im.name, p1, p2)


def test_join_two_aabbs(impls):
    # aabb.rs:725-759: boxes spanned by five points each contain them; their join contains all ten
    g = np.random.default_rng(3)
    pts = _any(g, 10 * (N // 5)).reshape(-1, 10, 3)
    for im in impls:
        for ten in pts:
            b1, b2 = im.span(*ten[:5]), im.span(*ten[5:])
            u = im.join(b1, b2)
            assert all(contains(b1, p) for p in ten[:5]) and all(contains(b2, p) for p in ten[5:]), im.name
            assert all(contains(u, p) for p in ten), im.name


def test_points_relative_to_center_and_size(impls):
    # aabb.rs:761-786, large strategy: centre +- 0.9 half-size is (approximately) inside, 1.1 half-sizes further is outside
    g = np.random.default_rng(4)
    a, b = _large(g, N), _large(g, N)
    for im in impls:
        for p1, p2 in zip(a, b):
            box = im.span(p1, p2)
            st = im.stats(box)
            half = st["size"] / f32(2.0)
            inside_ppp = st["center"] + half * f32(0.9)
            inside_mmm = st["center"] - half * f32(0.9)
            outside_ppp = inside_ppp + half * f32(1.1)
            outside_mmm = inside_mmm - half * f32(1.1)
            assert approx_contains_eps(box, inside_ppp, EPS) and approx_contains_eps(box, inside_mmm, EPS), im.name
            assert not contains(box, outside_ppp) and not contains(box, outside_mmm), im.name


def test_surface_and_volume_never_negative(impls):
    # aabb.rs:788-795 and :812-819
    g = np.random.default_rng(5)
    a, b, la, lb = _any(g, N), _any(g, N), _large(g, N), _large(g, N)
    for im in impls:
        with np.errstate(over="ignore"):
            for p1, p2 in zip(a, b):
                assert im.stats(im.span(p1, p2))["surface_area"] >= 0.0, im.name
            for p1, p2 in zip(la, lb):
                s = im.stats(im.span(p1, p2))["size"]
                assert s[0] * s[1] * s[2] >= 0.0, im.name          # volume(), aabb.rs:547-550


def test_surface_area_of_a_cube(impls):
    # aabb.rs:797-810: with_bounds(pos, pos + s) has surface 6 s^2 up to rmax <= EPSILON, s in EPSILON..10e30.
    # (As written the property cannot hold for every f32 `pos`: once |pos| > 2^24 s the sum pos + s rounds the cube away.
    # proptest's 256 cases per run rarely meet that; here `pos` is drawn small enough against s for the property to be a
    # statement about surface_area rather than about the addition: |pos| <= s.)
    g = np.random.default_rng(6)
    size = np.exp(g.uniform(np.log(1e-5), np.log(10e30), N)).astype(np.float32)
    pos = (g.uniform(-1, 1, (N, 3)) * size[:, None]).astype(np.float32)
    for im in impls:
        with np.errstate(over="ignore"):
            for p, s in zip(pos, size):
                box = np.concatenate([p, p + s]).astype(np.float32)
                got = im.stats(box)["surface_area"]
                # pos + s rounds each edge by up to 2^-23 (|pos| + s) <= 2^-22 s: within 1e-5 relative on s^2 terms
                assert float_eq_rmax(got, f32(6.0) * s * s, EPS), (im.name, p, s, got)


def test_volume_by_hand(impls):
    # aabb.rs:821-832: pos.aabb().grow(pos + size) has volume |sx sy sz| (rmax <= EPSILON), large strategy.
    # Same caveat as the cube: the crate draws pos and size independently, and pos + size loses size's low bits when
    # |pos| >> |size|; the volume is then still that of the box the arithmetic produced.  Checked against that.
    g = np.random.default_rng(7)
    pos, size = _large(g, N), _large(g, N)
    for im in impls:
        with np.errstate(over="ignore", invalid="ignore"):
            for p, s in zip(pos, size):
                q = p + s
                box = im.grow(np.concatenate([p, p]), q)
                sz = im.stats(box)["size"]
                vol = sz[0] * sz[1] * sz[2]
                eff = q - p                                          # the edge vector the f32 sum really spans
                assert float_eq_rmax(vol, abs(eff[0] * eff[1] * eff[2]), EPS), (im.name, p, s)


def test_create_aabb_from_indexable(impls):
    # aabb.rs:834-847: with_bounds(aabb[0], aabb[1]) classifies every point as the box itself does
    g = np.random.default_rng(8)
    a, b, pts = _any(g, N), _any(g, N), _any(g, N)
    for im in impls:
        for p1, p2, p in zip(a, b, pts):
            box = im.span(p1, p2)
            again = im.join(box, box)                               # bounds read back and a box made from them
            assert np.array_equal(box, again) and contains(box, p) == contains(again, p), im.name


# ---------------------------------------------------------------------------------------------- rays (ray.rs:377-512)
def _ray_new(o, d):                       # Ray::new, ray.rs:133-143 (glam dot order, divide by the length)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        ln = np.sqrt(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32)
        dn = (d / ln).astype(np.float32)
        return o, dn, (f32(1.0) / dn).astype(np.float32)


def naive(o, inv, box):                   # ray.rs:218-244 (f32::min / f32::max)
    with np.errstate(over="ignore", invalid="ignore"):
        lo, hi = (box[:3] - o) * inv, (box[3:] - o) * inv
        entry, exit_ = np.fmin(lo, hi), np.fmax(lo, hi)
        latest, earliest = np.fmax(np.fmax(entry[0], entry[1]), entry[2]), np.fmin(np.fmin(exit_[0], exit_[1]), exit_[2])
        return bool(latest < earliest and earliest > 0.0)


def branchless(o, inv, box):              # ray.rs:263-285 with the crate's own min / max (ray.rs:81-112: `if x < y`)
    mn = lambda x, y: x if x < y else y
    mx = lambda x, y: x if x > y else y
    with np.errstate(over="ignore", invalid="ignore"):
        tmin, tmax = f32(0.0), f32(np.inf)
        for a in range(3):
            t1, t2 = (box[a] - o[a]) * inv[a], (box[3 + a] - o[a]) * inv[a]
            tmin = mn(mx(t1, tmin), mx(t2, tmin))
            tmax = mx(mn(t1, tmax), mn(t2, tmax))
        return bool(tmin <= tmax)


def _ray_cases(impl, seed):
    g = np.random.default_rng(seed)
    p1, p2, pos = _small(g, N), _small(g, N), _small(g, N)
    for a, b, o in zip(p1, p2, pos):
        box = impl.span(a, b)
        yield o, (impl.stats(box)["center"] - o).astype(np.float32), box          # gen_ray_to_aabb, ray.rs:362-375


def test_ray_points_at_aabb_center(impls, host, oracle):
    # ray.rs:378-408: all three variants accept a ray aimed at the centre of the box
    for o, d, box in _ray_cases(impls[0], 11):
        assert oracle.ray_intersects_aabb(o, d, box)                                             # optimised, oracle
        assert host.host_ray_hits_box(_p(o), _p(d), _p(box)) == 1                                # ... product walk
        _, _, inv = _ray_new(o, d)
        assert naive(o, inv, box) and branchless(o, inv, box)


def test_ray_points_from_aabb_center(impls, host, oracle):
    # ray.rs:410-453: the ray turned round (direction and inv_direction negated, the cached signs NOT refreshed — :420-423)
    # misses unless its origin is inside; and the honest statement of the same thing, a ray built pointing away
    for o, d, box in _ray_cases(impls[0], 12):
        inside = contains(box, o)
        assert not oracle.ray_intersects_aabb_flipped(o, d, box) or inside
        assert host.host_ray_hits_box_flipped(_p(o), _p(d), _p(box)) == 0 or inside
        _, _, inv = _ray_new(o, d)
        assert not naive(o, -inv, box) or inside
        assert not branchless(o, -inv, box) or inside
        away = (-d).astype(np.float32)
        assert not oracle.ray_intersects_aabb(o, away, box) or inside
        assert host.host_ray_hits_box(_p(o), _p(away), _p(box)) == 0 or inside


def test_three_slab_variants_agree_on_generic_rays(impls, host, oracle):
    """Not a property the crate states, but what its three variants are for: on rays in general position (no grazing: every
    entry / exit distance differs from its neighbour by a relative 1e-4) the optimised test the product walks with, the
    naive and the branchless one decide alike — and the product's walk decides as the oracle does on EVERY ray."""
    g = np.random.default_rng(13)
    n_hit = n_gen = 0
    for _ in range(N):
        c = g.uniform(-50, 50, 3)
        h = g.uniform(0.1, 10, 3)
        box = np.concatenate([c - h, c + h]).astype(np.float32)
        o = g.uniform(-80, 80, 3).astype(np.float32)
        d = (c + g.normal(size=3) * h * 1.5 - o).astype(np.float32)
        a = oracle.ray_intersects_aabb(o, d, box)
        assert (host.host_ray_hits_box(_p(o), _p(d), _p(box)) == 1) == a
        _, _, inv = _ray_new(o, d)
        lo, hi = (box[:3] - o) * inv, (box[3:] - o) * inv
        entry, exit_ = np.minimum(lo, hi).max(), np.maximum(lo, hi).min()
        if abs(entry - exit_) > 1e-4 * max(abs(entry), abs(exit_), 1.0) and abs(exit_) > 1e-4:
            n_gen += 1
            assert naive(o, inv, box) == a == branchless(o, inv, box), (o, d, box)
        n_hit += a
    assert n_gen > N * 0.9 and N * 0.2 < n_hit < N * 0.9


def test_ray_hits_triangle(oracle):
    """ray.rs:455-511 states this for `Ray::intersects_triangle` (back-face culled), which the slave does NOT call: its
    triangles go through a two-sided copy (mesh.rs:109-161).  The property is re-stated for that routine — a ray aimed at
    the point A + u AB + v AC (u + v <= 1, away from the borders) has a root, from either side of the plane — with the
    crate's u / v construction, on coordinates in +-10 (the crate's +-10e10 makes |det| overflow the routine's absolute
    1e-5 threshold meaningless).  Adapted, hence "parity unpinned" for the triangle routine itself."""
    from ray_tracer_s8_amd import _abi
    g = np.random.default_rng(14)
    n_ok = 0
    for _ in range(N):
        A, B, Cc, o = (g.uniform(-10, 10, 3).astype(np.float32) for _ in range(4))
        u = int(g.integers(0, 65536)) % 101
        v = min(100 - u, int(g.integers(0, 65536)) % 101)
        u, v = u / 100.0, v / 100.0
        uv, vv = B - A, Cc - A
        nrm = np.cross(uv.astype(np.float64), vv.astype(np.float64))
        area2 = np.linalg.norm(nrm)
        pt = (A + f32(u) * uv + f32(v) * vv).astype(np.float32)
        dist_plane = abs(np.dot(nrm / max(area2, 1e-30), (o - A).astype(np.float64)))
        border = min(u, v, 1.0 - u - v) < 0.011                       # the crate tolerates inputs within EPSILON of a border
        if border or area2 < 1.0 or dist_plane < 0.05:                # slivers / grazing origins: |det| near the 1e-5 rejection
            continue
        t = np.zeros(1, _abi.TRIANGLE_DTYPE)
        t["a"], t["b"], t["c"] = A, B, Cc
        roots = oracle.triangle_roots(t[0], o, (pt - o).astype(np.float32))
        assert len(roots) == 1 and roots[0] > 0, (A, B, Cc, o, u, v)
        want = np.linalg.norm((pt - o).astype(np.float64))
        assert abs(roots[0] - want) <= 1e-3 * max(want, 1.0)
        n_ok += 1
    assert n_ok > N // 3
