"""The oracle against the committed golden vectors (tests/golden, made by make_golden.py)."""
import hashlib

import numpy as np
import pytest

from _golden import load_all

CASES = load_all()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_golden(oracle, case):
    lin, lin_f, info_l = oracle.render(case["req"], case["spheres"], case["triangles"], backend=0, want_f32=True,
                                       world_index=case["world_index"])
    assert hashlib.sha256(lin.tobytes()).hexdigest() == case["sha256_rgb_linear"]
    assert hashlib.sha256(lin_f.tobytes()).hexdigest() == case["sha256_f32_linear"]
    assert info_l["ray_segments"] == case["ray_segments_linear"]
    rgb, f32, info = oracle.render(case["req"], case["spheres"], case["triangles"], backend=1, want_f32=True,
                                   world_index=case["world_index"])
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == case["sha256_rgb"]
    assert hashlib.sha256(f32.tobytes()).hexdigest() == case["sha256_f32"]
    assert info["ray_segments"] == case["ray_segments"]
    if case["rgb"] is not None:
        assert np.array_equal(rgb, case["rgb"])


def test_golden_set_is_present():
    assert len(CASES) >= 7
    assert any(c["triangles"] is not None for c in CASES)
    assert any(c["world_index"] is not None for c in CASES)          # the order of `world` is pinned by a committed vector too
