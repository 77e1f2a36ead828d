"""Reduced, deterministic versions of the culled-walk soaks (tools/cull_soak.py, tricull_soak.py, grazing_soak.py,
sliver_soak.py), a few seconds each, so that the adversarial evidence for engines 5 and 6 — the DEFAULT engines of c5 and of
every dense mesh — is part of what the driver runs: RGB8 and segment counts against the oracle (reference semantics), with the
culled walk forced and with the host's own choice.  The inequality itself is tested on the CPU (tests/test_cull_lemma.py)."""
import numpy as np
import pytest

import ray_tracer_s8_amd as rt
from ray_tracer_s8_amd import _abi as F
from _cull_cases import QCULL, XCULL, grazing_case, sliver_case, soup_case, sphere_field_cases, terrain_case

pytestmark = pytest.mark.gpu


def _exact(oracle, rq, sph, tri, flags_list):
    ref, _, info = oracle.render(rq, sph if sph is not None and len(sph) else None, tri if tri is not None and len(tri) else None, backend=1)
    engines = []
    with rt.Scene(0, rt.World(sph if sph is not None else np.zeros(0, F.SPHERE_DTYPE),
                              tri if tri is not None else np.zeros(0, F.TRIANGLE_DTYPE))) as sc:
        for fl in flags_list:
            r = rq.copy()
            r.flags = fl
            rgb, _, st = sc.render_tile(r)
            assert np.array_equal(rgb, ref), (fl, int((rgb != ref).sum()))
            assert st.ray_segments == info["ray_segments"], fl
            engines.append(st.engine)
    return engines


def test_culled_quantised_walk_on_sphere_fields(ndev, oracle):
    """c5-recipe fields (30 000, 12 000, 8 000 spheres) and a squeezed, densely overlapping one; depths 3..9."""
    used = 0
    for sph, k in sphere_field_cases(sizes=(30000, 12000, 8000), squeezed=(9000,)):
        rq = F.default_request(width=640, height=360, divisions=1, spp=3, max_bounces=3 + 2 * (k % 4), seed=1000 + k)
        eng = _exact(oracle, rq, sph, None, (QCULL, 0))
        used += eng[0] == 5
    assert used == 4


def test_culled_exact_walk_on_terrains_and_soups(ndev, oracle):
    """Terrains whose scale puts |e1||e2| at 0.002, at the limit and beyond it (culling must switch itself off there and the
    image must still be exact), and a dense triangle soup with spheres, one of them huge (the `big` list)."""
    # |e1||e2| at most: 0.047, 0.114, 0.222 (culled walk), 0.455 and 0.42 (thousands of triangles beyond the limit of 0.25: the
    # host must fall back to the plain walk), soups with edges up to 0.3 (culled) and 0.7 (beyond)
    cases = [terrain_case(64, 0.35), terrain_case(128, 1.0), terrain_case(224, 2.4), terrain_case(128, 2.0), terrain_case(224, 3.3),
             soup_case(np.random.default_rng(5), 8000, 4.0, 0.3), soup_case(np.random.default_rng(6), 8000, 4.0, 0.7)]
    engines = []
    for k, (name, sph, tri) in enumerate(cases):
        rq = F.default_request(width=480, height=270, divisions=1, spp=2, max_bounces=2 + 2 * (k % 3), seed=500 + k)
        engines.append(_exact(oracle, rq, sph, tri, (XCULL, 0))[0])
    assert engines == [6, 6, 6, 2, 2, 6, 2]              # culled where the bound admits the triangles, plain walk beyond


def test_grazing_angle_scenes(ndev, oracle):
    """Twelve random variations of test_grazing_rays_over_triangle_floors (floors and side walls, tilted, edges 0.05..0.49)."""
    used = 0
    for case in range(12):
        sph, tr, rq = grazing_case(case)
        used += _exact(oracle, rq, sph, tr, (XCULL,))[0] == 6
    assert used >= 10


def test_long_thin_triangles(ndev, oracle):
    """Eight sliver scenes (thousands of triangles 2..15 long and 1e-4..1e-2 wide)."""
    for case in range(8):
        t, rq, _ = sliver_case(case)
        _exact(oracle, rq, None, t, (XCULL, 0))
