"""CPU tests of the product's host-side BVH builder (ray_tracer_s8_amd/csrc/rt_bvh.h, the tree the HIP kernels walk):
same candidates, in the same order, as the oracle's independent build + recursive traverse — including the
reference's own 21-box fixture (testbase.rs:92-99,127-166) — and the invariants of the quantised twin."""
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
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", f"-I{hdr.parent}", "-o", str(OUT),
                        str(SRC)], check=True)
    return C.CDLL(str(OUT))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _traverse(lib, boxes, o, d):
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    out = np.zeros(len(b) + 1, np.uint32)
    nn, dp = C.c_uint32(0), C.c_uint32(0)
    n = lib.host_bvh_traverse(_p(b), C.c_uint32(len(b)), _p(np.asarray(o, np.float32)), _p(np.asarray(d, np.float32)),
                              _p(out), C.c_uint32(len(out)), C.byref(nn), C.byref(dp))
    return out[:n].tolist(), nn.value, dp.value


def _check(lib, boxes):
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    out = np.zeros(3, np.uint32)
    return lib.host_bvh_check(_p(b), C.c_uint32(len(b)), _p(out)), out.tolist()


def _unit_boxes():
    return np.array([[x - 0.5, -0.5, -0.5, x + 0.5, 0.5, 0.5] for x in range(-10, 11)], np.float32)


def test_reference_fixture_candidates(host, oracle):
    b = _unit_boxes()
    for o, d, want in (((-1000, 0, 0), (1, 0, 0), set(range(21))), ((0, -1000, 0), (0, 1, 0), {10}),
                       ((6, 0.5, 0), (-2, -1, 0), {14, 15, 16})):
        got, nn, _ = _traverse(host, b, o, d)
        ref, nn_ref = oracle.bvh_traverse_boxes(b, o, d)
        assert got == ref and set(got) == want and nn == nn_ref == 41


def _sphere_boxes(g, n, spread, rmax):
    c = g.uniform(-spread, spread, (n, 3)).astype(np.float32)
    r = g.uniform(0.05, rmax, (n, 1)).astype(np.float32)
    return np.concatenate([c - r, c + r], axis=1)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 50, 400, 3000])
def test_same_candidates_as_the_oracle_tree(host, oracle, n):
    g = np.random.default_rng(100 + n)
    b = _sphere_boxes(g, n, 10.0, 1.5 if n < 100 else 0.6)
    rays = []
    for k in range(40):
        o = g.uniform(-14, 14, 3)
        aim = 0.5 * (b[k % n, :3] + b[k % n, 3:]) + g.normal(size=3) * (0.0 if k % 2 else 0.5)
        rays.append((o, aim - o if k % 4 else g.normal(size=3)))
    # axis-parallel rays and origins exactly on a box plane: the 0 * inf = NaN slabs of ray.rs:174-194
    rays += [((float(b[0, 0]), 0.0, -20.0), (0, 0, 1)), ((0.0, float(b[n // 2, 4]), 0.0), (1, 0, 0)),
             ((-20, 0.25, 0.25), (1, 0, 0)), ((3, 30, -2), (0, -1, 0)), ((0, 0, 0), (0, 0, -1))]
    total = 0
    for o, d in rays:
        got, nn, dp = _traverse(host, b, o, d)
        ref, nn_ref = oracle.bvh_traverse_boxes(b, o, d)
        assert got == ref, (n, o, d)
        assert nn == nn_ref == 2 * n - 1
        total += len(got)
    assert total > 0


@pytest.mark.parametrize("n", [1, 2, 5, 64, 1024, 20000])
def test_flat_tree_and_quantised_twin_invariants(host, n):
    g = np.random.default_rng(7 * n + 1)
    b = _sphere_boxes(g, n, 50.0, 0.5)
    rc, (n_int, grid_ok, depth) = _check(host, b)
    assert rc == 0, f"host_bvh_check code {rc}"
    assert n_int == n - 1 and (grid_ok == 1 or n == 1)
    assert depth >= int(np.ceil(np.log2(n))) if n > 1 else depth == 0


def test_degenerate_inputs(host):
    assert _check(host, np.zeros((0, 6), np.float32))[0] == 0            # empty world: no tree
    same = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (9, 1))     # identical boxes: centroid bounds collapse
    rc, (n_int, _, _) = _check(host, same)
    assert rc == 0 and n_int == 8
    flat = _sphere_boxes(np.random.default_rng(3), 30, 5.0, 0.3)
    flat[:, 1] = 0.0
    flat[:, 4] = 0.0                                                       # zero extent along y: grid not usable or exact
    assert _check(host, flat)[0] == 0
