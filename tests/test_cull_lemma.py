"""The inequality the culled walks rest on (engines 5 and 6: nearer child first, subtrees entered beyond
cull_bound(best) / cull_bound_tri(best) skipped), tested ITSELF — not through images.

tests/host/cull_host.cpp evaluates, for single (ray, primitive, box) triples, with the product's bound functions
(csrc/rt_cull.h) and the reference's f32 root arithmetic:  every primitive that the reference would test and whose root is
accepted enters its own box no later than bound(its compared distance).  >= 10^7 seeded triples: generic, tangent rays,
far-and-small spheres (the reference's false-root domain), origins at |o| ~ 10^3, rays that start on or inside spheres;
triangles with K = |e1||e2| up to and at the 0.25 limit, determinants 1..30 times the reference's 10^-5 rejection
threshold (grazing), slivers, far origins.  The host constants behind `big` lists and K <= 0.25 (rt_api.hip
build_host_scene) cite this test."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "host" / "cull_host.cpp"
OUT = ROOT / "tests" / "host" / "_build" / "libcull_host.so"
HDR = ROOT / "ray_tracer_s8_amd" / "csrc" / "rt_cull.h"


@pytest.fixture(scope="module")
def lib():
    OUT.parent.mkdir(exist_ok=True)
    if not OUT.exists() or OUT.stat().st_mtime < max(SRC.stat().st_mtime, HDR.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", f"-I{HDR.parent}",
                        "-o", str(OUT), str(SRC)], check=True)
    l = C.CDLL(str(OUT))
    l.cull_check_spheres.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_void_p]
    l.cull_check_triangles.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    l.cull_bound_value.restype = C.c_float
    l.cull_bound_value.argtypes = [C.c_float] * 5
    l.cull_bound_tri_value.restype = C.c_float
    l.cull_bound_tri_value.argtypes = [C.c_float] * 8
    return l


def _tally(out):
    return {"cases": int(out[0]), "roots": int(out[1]), "candidates": int(out[2]), "violations": int(out[3]),
            "min_slack": float(out[4]), "worst": out[5:].tolist()}


SPHERE_MODES = {0: "generic", 1: "tangent rays", 2: "far and small (false roots)", 3: "origins at 10^3", 4: "origin on / inside"}
TRI_MODES = {0: "generic", 1: "K -> 0.25", 2: "grazing, |det| -> 1e-5", 3: "origins at 10^3", 4: "slivers from afar", 5: "K -> 0.25 at grazing angles"}
PER_MODE = 1_200_000          # x 5 modes x (spheres at two slack factors + triangles) = 1.8e7 triples


@pytest.mark.parametrize("mode", sorted(SPHERE_MODES))
@pytest.mark.parametrize("slack", [1.0, 3.0])
def test_no_sphere_root_beats_cull_bound(lib, mode, slack):
    """r_slack = the sphere's own radius (the tightest the host can ever pass: r_slack is the LARGEST radius among the
    spheres outside the `big` list) and a looser one."""
    out = np.zeros(21, np.float64)
    lib.cull_check_spheres(1234 + mode, PER_MODE, mode, slack, out.ctypes.data_as(C.c_void_p))
    t = _tally(out)
    assert t["cases"] == PER_MODE and t["violations"] == 0, (SPHERE_MODES[mode], t)
    assert t["candidates"] > PER_MODE // 50, (SPHERE_MODES[mode], t)          # the mode really produces accepted roots
    assert t["min_slack"] > 0.0, t


@pytest.mark.parametrize("mode", sorted(TRI_MODES))
def test_no_triangle_root_beats_cull_bound_tri(lib, mode):
    out = np.zeros(21, np.float64)
    lib.cull_check_triangles(4321 + mode, PER_MODE, mode, out.ctypes.data_as(C.c_void_p))
    t = _tally(out)
    assert t["cases"] == PER_MODE and t["violations"] == 0, (TRI_MODES[mode], t)
    assert t["candidates"] > PER_MODE // 100, (TRI_MODES[mode], t)
    assert t["min_slack"] > 0.0, t


def test_bounds_are_monotone_in_the_running_best(lib):
    """The walk applies the bound to whatever the running closest distance is at the time: the argument above needs
    bound(b) >= bound(D) for b >= D (f32, as evaluated), and bound(D) >= D."""
    g = np.random.default_rng(5)
    for _ in range(20000):
        d0 = float(np.float32(np.exp(g.uniform(np.log(1e-3), np.log(1e3)))))
        d1 = float(np.nextafter(np.float32(d0), np.float32(np.inf))) if g.random() < 0.5 else d0 * float(g.uniform(1.0, 3.0))
        o = g.uniform(-1500, 1500, 3)
        rs = float(g.uniform(0, 5))
        a, b = lib.cull_bound_value(d0, *o, rs), lib.cull_bound_value(d1, *o, rs)
        assert b >= a >= np.float32(d0)
        k, dg, es, e = float(g.uniform(0, 0.25)), float(g.uniform(0, 2)), float(g.uniform(0, 2)), float(g.uniform(0, 1.5))
        a, b = lib.cull_bound_tri_value(d0, *o, k, dg, es, e), lib.cull_bound_tri_value(d1, *o, k, dg, es, e)
        assert b >= a >= np.float32(d0)


def test_total_is_at_least_ten_million_triples():
    assert PER_MODE * (len(SPHERE_MODES) * 2 + len(TRI_MODES)) >= 10_000_000 and len(TRI_MODES) == 6
