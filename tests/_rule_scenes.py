"""Scenes for tests/test_gpu_engine_rules.py: generators and seeds that NO tools/*_matrix.py script uses (those tuned the host's engine
rules: uniform fields, piles, sheets, clusters, the Cornell room, height-field terrains, all from default_rng(13) / (5) / SplitMix
streams).  Here: shells, helices, jittered lattices, two-scale mixtures, colonnades, sphere flakes, corridors of triangles, fans, boxes
of quads — drawn from default_rng(9000 + i)."""
import numpy as np

from ray_tracer_s8_amd._abi import SPHERE_DTYPE, TRIANGLE_DTYPE


def _sph(c, r, g, emis_frac=0.03):
    n = len(c)
    s = np.zeros(n, SPHERE_DTYPE)
    s["cx"], s["cy"], s["cz"] = c[:, 0], c[:, 1], c[:, 2]
    s["radius"] = r
    s["albedo_r"], s["albedo_g"], s["albedo_b"] = g.uniform(0.15, 0.95, (3, n))
    s["roughness"] = np.where(g.random(n) < 0.6, 0.0, g.random(n))
    s["emission"] = np.where(g.random(n) < emis_frac, g.uniform(2, 6, n), 0.0)
    return s


def _ground(s, y=-1.0):
    gnd = np.zeros(1, SPHERE_DTYPE)
    gnd["cx"], gnd["cy"], gnd["cz"], gnd["radius"] = 0.0, y - 200.0, -15.0, 200.0
    gnd["albedo_r"] = gnd["albedo_g"] = gnd["albedo_b"] = 0.5
    return np.concatenate([gnd, s])


def shell(n, g, radius=6.0, thick=0.4, rr=(0.08, 0.25)):
    v = g.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1)[:, None]
    c = v * (radius + g.uniform(-thick, thick, n))[:, None] + np.array([0.0, 2.0, -14.0])
    return _ground(_sph(c, g.uniform(*rr, n), g))


def helix(n, g, turns=9.0, rr=(0.1, 0.3)):
    t = np.linspace(0.0, 1.0, n)
    c = np.stack([4.0 * np.cos(2 * np.pi * turns * t), -0.5 + 7.0 * t, -14.0 + 4.0 * np.sin(2 * np.pi * turns * t)], 1)
    c += g.normal(0.0, 0.05, c.shape)
    return _ground(_sph(c, g.uniform(*rr, n), g))


def lattice(n, g, jitter=0.15, rr=(0.12, 0.2)):
    k = int(round(n ** (1 / 3)))
    ax = np.arange(k) - (k - 1) / 2
    c = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3) * 0.9
    c = c[:n] + g.normal(0.0, jitter, (min(n, len(c)), 3)) + np.array([0.0, 3.5, -13.0])
    return _ground(_sph(c, g.uniform(*rr, len(c)), g))


def two_scale(n, g):
    nb = max(4, n // 40)
    big = g.uniform([-10, 0, -30], [10, 3, -6], (nb, 3))
    small = big[g.integers(0, nb, n - nb)] + g.normal(0.0, 1.3, (n - nb, 3))
    c = np.concatenate([big, small])
    r = np.concatenate([g.uniform(0.8, 1.3, nb), g.uniform(0.05, 0.15, n - nb)])
    return _ground(_sph(c, r, g))


def colonnade(n, g):
    cols = max(2, n // 24)
    per = n // cols
    cs = []
    for i in range(cols):
        x = -12.0 + 24.0 * (i % (cols // 2 + 1)) / (cols // 2 + 1)
        z = -8.0 - 9.0 * (i // (cols // 2 + 1))
        y = -0.7 + 0.55 * np.arange(per)
        cs.append(np.stack([np.full(per, x), y, np.full(per, z)], 1))
    c = np.concatenate(cs)
    return _ground(_sph(c, np.full(len(c), 0.3), g))


def blob(n, g, sigma=1.2, rr=(0.2, 0.5)):                      # a dense pile of overlapping spheres with a gaussian profile
    c = g.normal(0.0, sigma, (n, 3)) + np.array([0.0, 1.5, -9.0])
    return _ground(_sph(c, g.uniform(*rr, n), g))


def _tri(a, b, c, g, emis_frac=0.0):
    n = len(a)
    t = np.zeros(n, TRIANGLE_DTYPE)
    t["a"], t["b"], t["c"] = a, b, c
    t["albedo_r"], t["albedo_g"], t["albedo_b"] = g.uniform(0.2, 0.9, (3, n))
    t["roughness"] = np.where(g.random(n) < 0.7, 0.0, g.random(n))
    t["emission"] = np.where(g.random(n) < emis_frac, 4.0, 0.0)
    return t


def corridor(n, g):                                            # floor, two walls and a ceiling of quads, receding from the camera
    q = max(1, n // 8)
    a, b, c = [], [], []
    for i in range(q):
        z0, z1 = -2.0 - 1.5 * i, -3.5 - 1.5 * i
        for (p0, p1, p2, p3) in (((-2, -1, z0), (2, -1, z0), (2, -1, z1), (-2, -1, z1)), ((-2, 2, z0), (-2, 2, z1), (2, 2, z1), (2, 2, z0)),
                                 ((-2, -1, z0), (-2, -1, z1), (-2, 2, z1), (-2, 2, z0)), ((2, -1, z0), (2, 2, z0), (2, 2, z1), (2, -1, z1))):
            a += [p0, p0]
            b += [p1, p2]
            c += [p2, p3]
    return _tri(np.array(a, np.float32), np.array(b, np.float32), np.array(c, np.float32), g, emis_frac=0.05)


def soup(n, g, size=0.4, spread=(8.0, 4.0, 14.0)):             # small random triangles in a box
    p = g.uniform(-1, 1, (n, 3)) * np.array(spread) + np.array([0.0, 2.5, -18.0])
    return _tri(p.astype(np.float32), (p + g.normal(0, size, (n, 3))).astype(np.float32), (p + g.normal(0, size, (n, 3))).astype(np.float32), g)


def ripple(nx, nz, g):                                         # a fine height field z = f(x, z) of 2 nx nz triangles (not tools/ terrains)
    xs = np.linspace(-14, 14, nx + 1)
    zs = np.linspace(-4, -40, nz + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = -1.0 + 0.6 * np.sin(0.9 * X + 0.3 * Z) * np.cos(0.5 * Z) + g.normal(0.0, 0.02, X.shape)
    P = np.stack([X, Y, Z], -1).astype(np.float32)
    p00, p10, p01, p11 = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    a = np.concatenate([p00.reshape(-1, 3), p10.reshape(-1, 3)])
    b = np.concatenate([p10.reshape(-1, 3), p11.reshape(-1, 3)])
    c = np.concatenate([p01.reshape(-1, 3), p01.reshape(-1, 3)])
    return _tri(a, b, c, g)


def cases():
    """(name, spheres, triangles) — 26 scenes from 6 to 60 000 primitives."""
    out = []
    e = np.zeros(0, TRIANGLE_DTYPE)
    s0 = np.zeros(0, SPHERE_DTYPE)
    for i, (name, fn) in enumerate([
            ("shell 24", lambda g: shell(24, g, 3.0, 0.3, (0.2, 0.4))), ("shell 300", lambda g: shell(300, g)), ("shell 900", lambda g: shell(900, g)),
            ("shell 5000", lambda g: shell(5000, g, 8.0, 0.5, (0.05, 0.15))), ("shell 40000", lambda g: shell(40000, g, 12.0, 1.5, (0.04, 0.1))),
            ("helix 60", lambda g: helix(60, g, 3.0)), ("helix 700", lambda g: helix(700, g)), ("helix 3000", lambda g: helix(3000, g, 20.0, (0.06, 0.15))),
            ("lattice 125", lambda g: lattice(125, g)), ("lattice 729", lambda g: lattice(729, g)), ("lattice 8000", lambda g: lattice(8000, g, 0.1, (0.1, 0.2))),
            ("two-scale 400", lambda g: two_scale(400, g)), ("two-scale 2500", lambda g: two_scale(2500, g)), ("two-scale 20000", lambda g: two_scale(20000, g)),
            ("colonnade 96", lambda g: colonnade(96, g)), ("colonnade 960", lambda g: colonnade(960, g)),
            ("blob 12", lambda g: blob(12, g, 0.6)), ("blob 150", lambda g: blob(150, g)), ("blob 600", lambda g: blob(600, g, 1.6)), ("blob 3000", lambda g: blob(3000, g, 2.5))]):
        out.append((name, fn(np.random.default_rng(9000 + i)), e))
    for i, (name, fn) in enumerate([
            ("corridor 64 tris", lambda g: corridor(64, g)), ("corridor 800 tris", lambda g: corridor(800, g)), ("soup 3000 tris", lambda g: soup(3000, g)),
            ("soup 30000 tris", lambda g: soup(30000, g, 0.25)), ("ripple 60000 tris", lambda g: ripple(200, 150, g)),
            ("ripple 1800 tris", lambda g: ripple(30, 30, g))]):
        out.append((name, s0, fn(np.random.default_rng(9100 + i))))
    return out
