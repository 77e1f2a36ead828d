"""SURVEY 8(d) "statistical sanity vs the reference's distribution": the deterministic RNG stream this build defines
(one xoshiro256++ stream per (pixel, sample)) against the reference's own structure.

The reference seeds one SmallRng per ROW from entropy and consumes it pixel after pixel, sample after sample
(S/main.rs:69-77).  No bit-level parity with an entropy-seeded run exists; what can be checked is that the replacement
stream renders images with the same DISTRIBUTION.  The oracle (and only the oracle) renders in three modes:

    RNG_ROW     the reference's literal structure, a deterministic seed per row standing in for from_entropy()
    RNG_PIXEL   one stream per pixel spanning its samples — the definition of rounds 1-3
    RNG_SAMPLE  one stream per (pixel, sample) — the normative definition since round 4, what the HIP kernel implements

Over N_SEEDS job seeds, per mode:  (i) the per-channel image mean (linear radiance, i.e. the f32 framebuffer squared) of
every render — two modes must agree within a z-score of Z_MAX on the difference of their means over the seeds;
(ii) per pixel and channel, mean and variance over the seeds — the per-pixel z-scores between two modes must look like
N(0, 1) draws (mean within Z_MAX standard errors of 0, spread within [0.9, 1.1], no |z| above Z_PIXEL_MAX), and the
per-pixel log variance ratio must average to 0 within Z_MAX of ITS standard error (estimated over the pixels).
A negative control shows the statistic has the power to see a wrong estimator: 1 spp against 4 spp has the same means and
four times the per-pixel variance, and the variance statistic must flag it far beyond the bound.
All thresholds are deterministic outcomes (fixed seeds); the bounds are stated here once.
"""
import numpy as np
import pytest

from ray_tracer_s8_amd import scenes
from ray_tracer_s8_amd._abi import default_request

N_SEEDS = 96
Z_MAX = 4.0            # |z| bound for every mean-type statistic (two-sided normal tail 6e-5 per statistic)
Z_PIXEL_MAX = 6.5      # largest per-pixel |z| tolerated among ~1.6e4 pixel-channels (heavy-tailed light paths included)


def _renders(oracle, sph, rq, mode, backend):
    out = []
    for j in range(N_SEEDS):
        r = rq.copy()
        r.seed = 0x51A7 + 7919 * j
        _, f32, _ = oracle.render(r, sph, None, backend=backend, want_f32=True, rng_mode=mode)
        out.append((f32.astype(np.float64) ** 2).reshape(-1, 3))          # linear pixel means (main.rs:78-80 before the sqrt)
    return np.stack(out)                                                  # [seed, pixel, channel]


def _image_mean_z(a, b):
    ma, mb = a.mean(axis=1), b.mean(axis=1)                               # [seed, channel]
    se = np.sqrt(ma.var(axis=0, ddof=1) / len(ma) + mb.var(axis=0, ddof=1) / len(mb))
    return (ma.mean(axis=0) - mb.mean(axis=0)) / se


def _pixel_stats(a, b):
    n = a.shape[0]
    va, vb = a.var(axis=0, ddof=1), b.var(axis=0, ddof=1)
    ok = (va > 0) & (vb > 0)
    z = (a.mean(axis=0) - b.mean(axis=0))[ok] / np.sqrt((va[ok] + vb[ok]) / n)
    lr = np.log(va[ok] / vb[ok])
    z_lr = lr.mean() / (lr.std(ddof=1) / np.sqrt(lr.size))
    return z, z_lr, float(ok.mean())


SCENES = {
    # BASELINE config 2 (Cornell-16) at its own spp / depth, small frame; linear back-end
    "c2": lambda: (scenes.cornell16(), default_request(width=64, height=36, divisions=1, spp=4, max_bounces=4), 0),
    # BASELINE config 3's scene, 8 spp / depth 8, through the reference's BVH candidate filter
    "c3": lambda: (scenes.rand1024(), default_request(width=48, height=27, divisions=1, spp=8, max_bounces=8), 1),
}


@pytest.fixture(scope="module", params=sorted(SCENES))
def triple(request, oracle):
    sph, rq, backend = SCENES[request.param]()
    return {m: _renders(oracle, sph, rq, m, backend) for m in (oracle.RNG_ROW, oracle.RNG_PIXEL, oracle.RNG_SAMPLE)}


@pytest.mark.parametrize("pair", ["row-sample", "row-pixel", "pixel-sample"])
def test_streams_agree_in_distribution(oracle, triple, pair):
    modes = {"row": oracle.RNG_ROW, "pixel": oracle.RNG_PIXEL, "sample": oracle.RNG_SAMPLE}
    a, b = (triple[modes[k]] for k in pair.split("-"))
    assert not np.array_equal(a, b)                                       # different streams, different images
    zi = _image_mean_z(a, b)
    assert np.all(np.abs(zi) < Z_MAX), f"per-channel image means differ: z = {zi}"
    z, z_lr, frac = _pixel_stats(a, b)
    assert frac > 0.95                                                    # nearly every pixel-channel varies with the seed
    assert abs(z.mean()) < Z_MAX / np.sqrt(z.size), f"per-pixel means biased: mean z = {z.mean():.4f} over {z.size}"
    assert 0.9 < z.std() < 1.1, f"per-pixel z spread {z.std():.3f}"
    assert np.abs(z).max() < Z_PIXEL_MAX, f"largest per-pixel |z| = {np.abs(z).max():.2f}"
    assert abs(z_lr) < Z_MAX, f"per-pixel variances differ: z of the mean log ratio = {z_lr:.2f}"


def test_variance_statistic_has_power(oracle):
    """Negative control: same scene at 1 spp and at 4 spp — equal means, 4 x the per-pixel variance."""
    sph, rq, backend = SCENES["c2"]()
    a = _renders(oracle, sph, rq, oracle.RNG_SAMPLE, backend)
    r1 = rq.copy()
    r1.spp = 1
    b = _renders(oracle, sph, r1, oracle.RNG_SAMPLE, backend)
    assert np.all(np.abs(_image_mean_z(a, b)) < Z_MAX)                    # the estimator is unbiased at any spp
    _, z_lr, _ = _pixel_stats(a, b)
    assert z_lr < -10 * Z_MAX, f"variance statistic blind: z = {z_lr:.1f}"


def test_sample_streams_are_disjoint_blocks_of_the_job_sequence(oracle):
    """The normative seeding: stream (p, s) = SplitMix64 outputs 4 i + 1 .. 4 i + 4 of the job seed, i = p * S + s."""
    phi = 0x9E3779B97F4A7C15
    m64 = (1 << 64) - 1
    for seed, p, spp, s in [(0, 0, 1, 0), (0, 0, 8, 1), (0x5EED0400, 3840 * 100 + 17, 8, 7), (m64, 123456789, 100, 99)]:
        i = p * spp + s
        assert oracle.sample_seed(seed, p, spp, s) == (seed + 4 * phi * i) & m64
        st = oracle.seed_from_u64(oracle.sample_seed(seed, p, spp, s))
        # the same four words, read off the job's SplitMix64 sequence directly (seed_from_u64(seed) yields outputs 1 .. 4)
        direct = oracle.seed_from_u64((seed + 4 * phi * i) & m64)
        assert list(st) == list(direct)
        if i:                                                             # ... and the block before it ends where this one starts
            prev = oracle.seed_from_u64(oracle.sample_seed(seed, 0, 1, i - 1))
            assert len(set(map(int, prev)) & set(map(int, st))) == 0
