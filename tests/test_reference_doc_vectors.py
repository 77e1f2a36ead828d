"""The concrete vectors the reference itself holds for this path, beyond the 21-box traversal fixture: the doc-tests of
the vendored bvh crate (ray-tracer-slave/local-dependencies/bvh/src/aabb.rs, ray.rs, axis.rs).  Each is checked against
BOTH the oracle's AABB / ray helpers (oracle/rt_oracle.cpp) and the product's host BVH helpers (csrc/rt_bvh.h, through
the g++ harness tests/host/bvh_host.cpp).  The expected values below are the literals of the reference's asserts."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "host" / "bvh_host.cpp"
OUT = ROOT / "tests" / "host" / "_build" / "libbvh_host.so"


@pytest.fixture(scope="module")
def host():
    OUT.parent.mkdir(exist_ok=True)
    hdr = ROOT / "ray_tracer_s8_amd" / "csrc" / "rt_bvh.h"
    if not OUT.exists() or OUT.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", f"-I{hdr.parent}",
                        "-o", str(OUT), str(SRC)], check=True)
    return C.CDLL(str(OUT))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _host_kat(lib, a6, b6, pt):
    out = np.zeros(21, np.float32)
    lib.host_aabb_kat(_p(np.asarray(a6, np.float32)), _p(np.asarray(b6, np.float32)), _p(np.asarray(pt, np.float32)), _p(out))
    return {"size": out[0:3], "center": out[3:6], "surface_area": float(out[6]), "largest_axis": int(out[7]),
            "is_empty": bool(out[8]), "join": out[9:15], "grow": out[15:21]}


def _both(host, oracle, a6, b6=None, pt=(0, 0, 0)):
    b6 = a6 if b6 is None else b6
    return _host_kat(host, a6, b6, pt), oracle.aabb_kat(a6, b6, pt)


def _contains(box6, p):
    return all(box6[i] <= p[i] <= box6[3 + i] for i in range(3))      # AABB::contains, aabb.rs:146-153


def test_aabb_size_doc_vector(host, oracle):
    # aabb.rs:450-453: with_bounds((-1,-1,-1), (1,1,1)).size() == (2, 2, 2)
    for r in _both(host, oracle, [-1, -1, -1, 1, 1, 1]):
        assert r["size"].tolist() == [2.0, 2.0, 2.0]


def test_aabb_center_doc_vector(host, oracle):
    # aabb.rs:468-474: min 41, max 43 -> center (42, 42, 42)
    for r in _both(host, oracle, [41, 41, 41, 43, 43, 43]):
        assert r["center"].tolist() == [42.0, 42.0, 42.0]


def test_aabb_surface_area_doc_vector(host, oracle):
    # aabb.rs:514-520: min 41, max 43 -> surface_area 24 (the SAH cost's only input, bvh_impl.rs:332-341)
    for r in _both(host, oracle, [41, 41, 41, 43, 43, 43]):
        assert r["surface_area"] == 24.0


def test_aabb_largest_axis_doc_vector(host, oracle):
    # aabb.rs:559-565: (-100,0,0)..(100,0,0) -> Axis::X (the split axis rule, bvh_impl.rs:273)
    for r in _both(host, oracle, [-100, 0, 0, 100, 0, 0]):
        assert r["largest_axis"] == 0
    # the rule's other arms (aabb.rs:570-580): y wins over z only when strictly larger, z otherwise (also on ties)
    for box, want in (([0, -5, 0, 1, 5, 2], 1), ([0, 0, -7, 1, 2, 7], 2), ([0, 0, 0, 3, 3, 3], 2), ([0, 0, 0, 3, 3, 1], 1)):
        for r in _both(host, oracle, box):
            assert r["largest_axis"] == want


def test_aabb_join_doc_vector(host, oracle):
    # aabb.rs:243-263: [-101,0,0]-[-100,1,1] joined with [100,0,0]-[101,1,1] contains a point of each and (0, .5, .5)
    a, b = [-101, 0, 0, -100, 1, 1], [100, 0, 0, 101, 1, 1]
    for r in _both(host, oracle, a, b):
        j = r["join"].tolist()
        assert j == [-101.0, 0.0, 0.0, 101.0, 1.0, 1.0]
        assert _contains(j, (-100.5, 0.5, 0.5)) and _contains(j, (100.5, 0.5, 0.5)) and _contains(j, (0.0, 0.5, 0.5))
    assert not _contains(a, (0.0, 0.5, 0.5)) and not _contains(b, (0.0, 0.5, 0.5))


def test_aabb_empty_and_grow_doc_vectors(host, oracle):
    # aabb.rs:103-129: empty() = (+inf, -inf) contains no point;  aabb.rs:337-351: empty().grow(p) contains p only
    e = oracle.aabb_empty()
    assert np.all(np.isposinf(e[:3])) and np.all(np.isneginf(e[3:]))
    assert oracle.aabb_kat(e, e, (0, 0, 0))["is_empty"]
    he = np.zeros(6, np.float32)
    host.host_empty_box(_p(he))                              # the builder's empty_box() is the crate's AABB::empty()
    assert he.tolist() == e.tolist() and _host_kat(host, he, he, (0, 0, 0))["is_empty"]
    for pt, inside, outside in (((0, 0, 0), (0, 0, 0), (1, 1, 1)), ((1, 1, 1), (1, 1, 1), (2, 2, 2))):
        g = oracle.aabb_kat(e, e, pt)["grow"].tolist()
        assert g == [*map(float, pt), *map(float, pt)]
        assert _contains(g, inside) and not _contains(g, outside)
        out = np.zeros(6, np.float32)
        host.host_empty_grow(_p(np.asarray(pt, np.float32)), _p(out))
        assert out.tolist() == g


def test_with_bounds_doc_vector(host, oracle):
    # aabb.rs:88-91: with_bounds keeps min / max as given
    for r in _both(host, oracle, [-1, -1, -1, 1, 1, 1], pt=(0, 0, 0)):
        assert r["grow"].tolist() == [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0] and not r["is_empty"]


def test_axis_doc_vectors(oracle):
    # axis.rs:15-20: position[Axis::Y] *= 4 on [1, .5, 42] gives 2;  axis.rs:28-33: position[Axis::X] = 1000
    pos = [1.0, 0.5, 42.0]
    assert oracle.axis_get(pos, 1) * 4.0 == 2.0
    assert [oracle.axis_get([1000.0, 2.0, 3.0], a) for a in (0, 1, 2)] == [1000.0, 2.0, 3.0]


def test_ray_intersects_aabb_doc_vector(host, oracle):
    # ray.rs:158-168: the ray (0,0,0) -> (1,0,0) intersects the box (99.9,-1,-1)-(100.1,1,1)
    o, d, box = (0.0, 0.0, 0.0), (1.0, 0.0, 0.0), [99.9, -1.0, -1.0, 100.1, 1.0, 1.0]
    assert oracle.ray_intersects_aabb(o, d, box)
    ob, db, bb = (np.asarray(v, np.float32) for v in (o, d, box))
    assert host.host_ray_hits_box(_p(ob), _p(db), _p(bb)) == 1
    # and the same box is missed from behind and from the side (sanity of the slab signs)
    for o2, d2 in (((200.0, 0.0, 0.0), (1.0, 0.0, 0.0)), ((0.0, 5.0, 0.0), (1.0, 0.0, 0.0))):
        assert not oracle.ray_intersects_aabb(o2, d2, box)
        o2b, d2b = np.asarray(o2, np.float32), np.asarray(d2, np.float32)
        assert host.host_ray_hits_box(_p(o2b), _p(d2b), _p(bb)) == 0
