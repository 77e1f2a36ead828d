"""Deterministic synthetic scenes for the BASELINE configs.

The reference ships no scene (the controller only ingests OBJ uploads), so the benchmark
scenes are defined here.  Generator: a SplitMix64 stream, next_f32 = (u64 >> 40) * 2^-24,
values rounded to binary32 once.  Camera = the reference literals for every config.
"""
from __future__ import annotations

import numpy as np

from ._abi import SPHERE_DTYPE, TRIANGLE_DTYPE, TileRequest, default_request

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def f(self) -> float:
        return (self.next_u64() >> 40) * (1.0 / (1 << 24))

    def u(self, lo: float, hi: float) -> float:
        return lo + (hi - lo) * self.f()


def _sph(c, r, albedo, rough=0.0, emis=0.0):
    return (c[0], c[1], c[2], r, albedo[0], albedo[1], albedo[2], rough, emis)


def single_sphere() -> np.ndarray:
    """c1: one diffuse sphere in front of the camera."""
    return np.array([_sph((0.0, 0.0, -3.0), 1.0, (0.8, 0.3, 0.3))], dtype=SPHERE_DTYPE)


def cornell16(seed: int = 0xC0A11E16) -> np.ndarray:
    """c2: 5 wall spheres (r = 100), 1 light, 10 small spheres."""
    g = SplitMix64(seed)
    grey = (0.73, 0.73, 0.73)
    s = [
        _sph((-102.0, 0.0, -4.0), 100.0, (0.75, 0.15, 0.15)),
        _sph((102.0, 0.0, -4.0), 100.0, (0.15, 0.75, 0.15)),
        _sph((0.0, -102.0, -4.0), 100.0, grey),
        _sph((0.0, 102.0, -4.0), 100.0, grey),
        _sph((0.0, 0.0, -108.0), 100.0, grey),
        _sph((0.0, 1.6, -4.0), 0.5, (1.0, 1.0, 1.0), 0.0, 8.0),
    ]
    rough = [0, 0, 0, 0, 0, 0, 0.3, 0.7, 1, 1]
    for i in range(10):
        r = g.u(0.25, 0.45)
        x = g.u(-1.4, 1.4)
        z = g.u(-5.5, -2.5)
        alb = (g.u(0.2, 0.9), g.u(0.2, 0.9), g.u(0.2, 0.9))
        s.append(_sph((x, -2.0 + r, z), r, alb, rough[i]))
    return np.array(s, dtype=SPHERE_DTYPE)


def _rand_field(n_total: int, seed: int, box, rr, ground):
    g = SplitMix64(seed)
    s = [ground]
    (x0, x1), (y0, y1), (z0, z1) = box
    while len(s) < n_total:
        x, y, z = g.u(x0, x1), g.u(y0, y1), g.u(z0, z1)
        r = g.u(*rr)
        alb = (g.u(0.1, 0.95), g.u(0.1, 0.95), g.u(0.1, 0.95))
        pr = g.f()
        rv = g.f()
        pe = g.f()
        ev = g.u(2.0, 6.0)
        if (x * x + y * y + z * z) ** 0.5 - r < 0.5:   # camera clearance (lens radius 0.05)
            continue
        rough = 0.0 if pr < 0.6 else (rv if pr < 0.85 else 1.0)
        emis = ev if pe < 0.02 else 0.0
        s.append(_sph((x, y, z), r, alb, rough, emis))
    return np.array(s, dtype=SPHERE_DTYPE)


def rand1024(seed: int = 0x5EED0400, n: int = 1024) -> np.ndarray:
    """c3/c4: ground sphere + n-1 random spheres; fits LDS (16 KiB of geometry)."""
    ground = _sph((0.0, -101.0, -20.0), 100.0, (0.5, 0.5, 0.5))
    return _rand_field(n, seed, ((-24, 24), (-1, 10), (-48, -3)), (0.15, 0.6), ground)


def rand65536(seed: int = 0x5EED1000, n: int = 65536) -> np.ndarray:
    """c5: scene larger than LDS (1 MiB of geometry), streamed through LDS chunks."""
    ground = _sph((0.0, -101.0, -20.0), 100.0, (0.5, 0.5, 0.5))
    return _rand_field(n, seed, ((-96, 96), (-1, 40), (-192, -3)), (0.1, 0.4), ground)


def quad_room(seed: int = 0x7121A9) -> tuple[np.ndarray, np.ndarray]:
    """Small mixed scene (spheres + triangles) for the triangle path (reference mesh.rs)."""
    g = SplitMix64(seed)
    tris = []

    def tri(a, b, c, alb, rough=0.0, emis=0.0):
        tris.append((a, b, c, alb[0], alb[1], alb[2], rough, emis))

    # floor quad y = -1 and a back wall z = -6, a tilted emissive panel
    tri((-4, -1, -1), (4, -1, -1), (4, -1, -7), (0.6, 0.6, 0.6))
    tri((-4, -1, -1), (4, -1, -7), (-4, -1, -7), (0.6, 0.6, 0.6))
    tri((-4, -1, -6), (4, -1, -6), (4, 3, -6), (0.3, 0.5, 0.8), 0.5)
    tri((-4, -1, -6), (4, 3, -6), (-4, 3, -6), (0.3, 0.5, 0.8), 0.5)
    tri((-1, 2.5, -3), (1, 2.5, -3), (0, 2.9, -5), (1.0, 1.0, 1.0), 0.0, 6.0)
    for _ in range(11):
        c = (g.u(-3, 3), g.u(-0.8, 2.0), g.u(-5.5, -2.0))
        e1 = (g.u(-0.8, 0.8), g.u(-0.8, 0.8), g.u(-0.8, 0.8))
        e2 = (g.u(-0.8, 0.8), g.u(-0.8, 0.8), g.u(-0.8, 0.8))
        a = c
        b = tuple(c[i] + e1[i] for i in range(3))
        cc = tuple(c[i] + e2[i] for i in range(3))
        tri(a, b, cc, (g.u(0.2, 0.9), g.u(0.2, 0.9), g.u(0.2, 0.9)), g.f() if g.f() < 0.5 else 0.0)
    sph = [
        _sph((-1.2, -0.5, -3.5), 0.5, (0.8, 0.3, 0.3)),
        _sph((1.1, -0.6, -3.0), 0.4, (0.9, 0.9, 0.9), 1.0),
        _sph((0.0, -0.7, -2.4), 0.3, (0.3, 0.8, 0.3), 0.3),
    ]
    return np.array(sph, dtype=SPHERE_DTYPE), np.array(tris, dtype=TRIANGLE_DTYPE)


def tri_terrain(nx: int = 16, nz: int = 12, seed: int = 0x7E44A1) -> tuple[np.ndarray, np.ndarray]:
    """A height-field mesh of 2*nx*nz triangles (what an OBJ upload looks like to the slave) + 3 spheres."""
    g = SplitMix64(seed)
    hs = [[g.u(-1.6, -0.6) for _ in range(nx + 1)] for _ in range(nz + 1)]
    X = lambda i: -6.0 + 12.0 * i / nx
    Z = lambda k: -2.0 - 10.0 * k / nz
    tris = []
    for k in range(nz):
        for i in range(nx):
            p00, p10 = (X(i), hs[k][i], Z(k)), (X(i + 1), hs[k][i + 1], Z(k))
            p01, p11 = (X(i), hs[k + 1][i], Z(k + 1)), (X(i + 1), hs[k + 1][i + 1], Z(k + 1))
            alb = (g.u(0.3, 0.9), g.u(0.3, 0.9), g.u(0.3, 0.9))
            rough = 0.0 if g.f() < 0.7 else g.f()
            tris.append((p00, p10, p11, alb[0], alb[1], alb[2], rough, 0.0))
            tris.append((p00, p11, p01, alb[0], alb[1], alb[2], rough, 0.0))
    tris.append(((-2, 3, -5), (2, 3, -5), (0, 3.5, -8), 1.0, 1.0, 1.0, 0.0, 5.0))      # light
    sph = [_sph((-1.5, 0.2, -4.0), 0.6, (0.9, 0.9, 0.9), 1.0), _sph((1.2, 0.0, -5.0), 0.5, (0.8, 0.3, 0.3)),
           _sph((0.0, -0.2, -3.0), 0.3, (0.3, 0.8, 0.4), 0.4)]
    return np.array(sph, dtype=SPHERE_DTYPE), np.array(tris, dtype=TRIANGLE_DTYPE)


def terrain_obj(nx: int = 224, nz: int = 224, seed: int = 0x0B1E5) -> tuple[bytes, bytes]:
    """(OBJ text, MTL text) of a rolling height field of 2*nx*nz triangles in four materials (224 x 224: 100 352
    triangles) — the shape of input the shipped controller produces: OBJ upload -> tobj -> `Object::Triangle` list
    (controller obj.rs:10-53).  Deterministic; vertices are written with 6 decimals, i.e. what a client would send."""
    import math
    g = SplitMix64(seed)
    ph = [g.u(0.0, 6.28) for _ in range(6)]

    def h(x: float, z: float) -> float:
        return (-1.4 + 0.45 * math.sin(0.9 * x + ph[0]) * math.cos(0.7 * z + ph[1]) + 0.18 * math.sin(2.3 * x + 1.7 * z + ph[2])
                + 0.06 * math.sin(7.1 * x + ph[3]) * math.sin(6.3 * z + ph[4]))

    lines = ["# generated height field", "mtllib terrain.mtl"]
    for k in range(nz + 1):
        z = -1.5 - 22.0 * k / nz
        for i in range(nx + 1):
            x = -14.0 + 28.0 * i / nx
            lines.append(f"v {x:.6f} {h(x, z):.6f} {z:.6f}")
    names = ["grass", "rock", "wet", "snow"]
    cur = None
    for k in range(nz):
        for i in range(nx):
            m = names[(int(g.f() * 2.0) + (i // 28) + (k // 28)) % 4]
            if m != cur:
                lines.append(f"usemtl {m}")
                cur = m
            a = k * (nx + 1) + i + 1
            b, c, d = a + 1, a + nx + 1, a + nx + 2
            lines.append(f"f {a} {b} {d}")
            lines.append(f"f {a} {d} {c}")
    mtl = ("newmtl grass\nKd 0.25 0.55 0.2\nNs 0\nnewmtl rock\nKd 0.5 0.45 0.4\nNs 120\n"
           "newmtl wet\nKd 0.7 0.75 0.8\nNs 900\nnewmtl snow\nKd 0.92 0.92 0.95\nNs 30\n")
    return ("\n".join(lines) + "\n").encode(), mtl.encode()


def mesh_world(nx: int = 224, nz: int = 224) -> np.ndarray:
    """The triangle list the controller would build from terrain_obj() (through obj.build_world)."""
    from . import obj
    o, m = terrain_obj(nx, nz)
    return obj.build_world(o + m, len(o))


# ---- BASELINE.json configs -------------------------------------------------------------
def config(name: str) -> tuple[np.ndarray, TileRequest]:
    """(spheres, request template) for c1..c5.  `divisions` is chosen so H % div == 0."""
    if name == "c1":
        return single_sphere(), default_request(width=256, height=256, divisions=1, spp=1, max_bounces=10, seed=1)
    if name == "c2":
        return cornell16(), default_request(width=1920, height=1080, divisions=20, spp=4, max_bounces=4,
                                            seed=0xC0A11E16)
    if name == "c3":
        return rand1024(), default_request(width=3840, height=2160, divisions=8, spp=8, max_bounces=8,
                                           seed=0x5EED0400)
    if name == "c4":
        return rand1024(), default_request(width=7680, height=4320, divisions=32, spp=16, max_bounces=8,
                                           seed=0x5EED0400)
    if name == "c5":
        return rand65536(), default_request(width=3840, height=2160, divisions=16, spp=8, max_bounces=8,
                                            seed=0x5EED1000)
    if name == "mesh":
        # not a BASELINE config: the only primitive the shipped controller emits, at the controller's literal frame
        # (1920x1080, 20 strips); spheres = none, triangles = mesh_world()
        return np.zeros(0, SPHERE_DTYPE), default_request(width=1920, height=1080, divisions=20, spp=4, max_bounces=4,
                                                          seed=0x0B1E5)
    if name == "c3_ref":
        # not a BASELINE config either: c3's 1024 spheres at the reference's literal settings (as mesh_ref below)
        return rand1024(), default_request(seed=0x5EED0400)
    if name == "mesh_ref":
        # the same mesh at the reference's LITERAL settings: 1920x1080, 20 strips (controller main.rs:33-39), 100 samples per pixel,
        # 10 bounces (slave main.rs:39, 51) — what a job of the shipped controller + slave actually renders
        return np.zeros(0, SPHERE_DTYPE), default_request(seed=0x0B1E5)
    raise KeyError(name)


def config_world(name: str):
    """(spheres, triangles, request template): config() plus the triangle list of the mesh workloads."""
    sph, rq = config(name)
    tri = mesh_world() if name in ("mesh", "mesh_ref") else np.zeros(0, TRIANGLE_DTYPE)
    return sph, tri, rq
