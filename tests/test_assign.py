"""The frame context's strip assignment (csrc/rt_assign.h, the product's function through a g++ harness): which device entry
renders which strip.  Reference: the controller fires one request per strip and Docker's DNS spreads them over the slaves
(C/main.rs:47-75); a frame is done when the slowest slave is (C/main.rs:106-115)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "host" / "assign_host.cpp"
OUT = ROOT / "tests" / "host" / "_build" / "libassign_host.so"
HDR = ROOT / "ray_tracer_s8_amd" / "csrc" / "rt_assign.h"
STATIC, SNAKE, BY_COST = 0, 1, 2


@pytest.fixture(scope="module")
def lib():
    OUT.parent.mkdir(exist_ok=True)
    if not OUT.exists() or OUT.stat().st_mtime < max(SRC.stat().st_mtime, HDR.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", f"-I{HDR.parent}", "-o", str(OUT), str(SRC)], check=True)
    l = C.CDLL(str(OUT))
    l.assign_strips.restype = C.c_double
    l.assign_strips.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    return l


def assign(lib, div, n, cost, mode):
    owner = np.zeros(div, np.uint32)
    c = None if cost is None else np.ascontiguousarray(cost, np.float64)
    mom = lib.assign_strips(div, n, None if c is None else c.ctypes.data_as(C.c_void_p), mode, owner.ctypes.data_as(C.c_void_p))
    return owner, mom


def loads(owner, cost, n):
    return np.bincount(owner, weights=cost, minlength=n)


def test_static_is_the_reference_round_robin(lib):
    owner, _ = assign(lib, 20, 3, None, STATIC)
    assert owner.tolist() == [k % 3 for k in range(20)]


def test_one_entry_takes_everything(lib):
    for mode in (STATIC, SNAKE, BY_COST):
        owner, _ = assign(lib, 7, 1, np.arange(7.0), mode)
        assert not owner.any()


@pytest.mark.parametrize("div,n", [(32, 8), (16, 8), (20, 3), (8, 8), (64, 6)])
def test_snake_evens_out_a_linear_profile(lib, div, n):
    """Cost linear in the strip's position (sky at the top, twice as expensive at the bottom: DESIGN.md 6): the snake's
    entries carry equal sums whenever every entry has an even number of strips; k % n leaves the last one on top."""
    cost = 1.0 + (np.arange(div) + 0.5) / div
    owner, mom = assign(lib, div, n, cost, SNAKE)
    cnt = np.bincount(owner, minlength=n)
    assert cnt.max() - cnt.min() <= 1                                # one strip of each row of n
    _, mom_static = assign(lib, div, n, cost, STATIC)
    assert mom <= mom_static + 1e-12
    if div % (2 * n) == 0:
        assert abs(mom - 1.0) < 1e-12 and mom_static > 1.03
    # each row of n strips goes to n different entries
    for r in range(div // n):
        assert sorted(owner[r * n:(r + 1) * n].tolist()) == list(range(n))


def test_longest_first_by_measured_cost(lib):
    rng = np.random.default_rng(5)
    for div, n in [(32, 8), (16, 8), (20, 3), (64, 8), (40, 7)]:
        for trial in range(50):
            kind = trial % 3
            if kind == 0:
                cost = 1.0 + (np.arange(div) + 0.5) / div + 0.05 * rng.standard_normal(div)     # the usual frame
            elif kind == 1:
                cost = rng.lognormal(0.0, 0.6, div)                                              # anything
            else:
                cost = np.where(np.arange(div) < div // 5, 5.0, 1.0) * (1 + 0.02 * rng.random(div))   # a few heavy strips on top
            cost = np.abs(cost)
            owner, mom = assign(lib, div, n, cost, BY_COST)
            ld = loads(owner, cost, n)
            assert abs(mom - ld.max() / ld.mean()) < 1e-9
            # Graham: longest-first is within 4/3 - 1/(3n) of the optimum, and the optimum is at least max(mean, largest strip)
            lower = max(ld.mean(), cost.max())
            assert ld.max() <= (4.0 / 3.0 - 1.0 / (3.0 * n)) * lower * (1 + 1e-12) + 1e-12 or ld.max() <= lower * 1.34
            # never worse than the plain round robin on the same costs
            _, mom_static = assign(lib, div, n, cost, STATIC)
            assert mom <= mom_static + 1e-9
            if kind == 0 and div >= 4 * n and div % n == 0:
                assert mom < 1.02, (div, n, mom)                     # the verdict's bar for c4 / c5 at 8 entries


def test_longest_first_is_deterministic_and_breaks_ties_low(lib):
    cost = np.ones(12)
    a, _ = assign(lib, 12, 4, cost, BY_COST)
    b, _ = assign(lib, 12, 4, cost, BY_COST)
    assert a.tolist() == b.tolist() == [0, 1, 2, 3] * 3              # equal costs: strip order, lowest entry first
    owner, _ = assign(lib, 5, 2, [9, 1, 1, 1, 1], BY_COST)
    assert owner[0] == 0 and (owner[1:] == 1).all()                   # the heavy strip alone


def test_by_cost_without_costs_falls_back_to_round_robin(lib):
    owner, _ = assign(lib, 9, 4, None, BY_COST)
    assert owner.tolist() == [k % 4 for k in range(9)]
