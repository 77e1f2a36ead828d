"""ctypes wrapper around oracle/_build/librt_oracle.so.

*** TEST INFRASTRUCTURE ONLY. ***  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg — never by ray_tracer_s8_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "librt_oracle.so"

_lib = None


def build(force: bool = False) -> Path:
    src = HERE / "rt_oracle.cpp"
    hdr = HERE.parent / "include" / "rt_tile.h"
    stale = (not LIB.exists()) or any(p.stat().st_mtime > LIB.stat().st_mtime for p in (src, hdr, HERE / "Makefile"))
    if force or stale:
        p = subprocess.run(["make", "-C", str(HERE), "-B"], capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{p.stdout}\n{p.stderr}")
    return LIB


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB))
        _lib.rt_oracle_render.restype = C.c_int
        _lib.rt_oracle_hardware_threads.restype = C.c_int
        _lib.rt_oracle_pixel_seed.restype = C.c_uint64
        _lib.rt_oracle_pixel_seed.argtypes = [C.c_uint64, C.c_uint64]
        _lib.rt_oracle_sample_seed.restype = C.c_uint64
        _lib.rt_oracle_sample_seed.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        _lib.rt_oracle_find_roots_quadratic.restype = C.c_int
        _lib.rt_oracle_find_roots_quadratic.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32)


def hardware_threads() -> int:
    return load().rt_oracle_hardware_threads()


RNG_SAMPLE, RNG_PIXEL, RNG_ROW = 0, 1, 2


def render(req, spheres, triangles=None, backend: int = 0, nthreads: int = 0, want_f32: bool = False, world_index=None,
           rng_mode: int = RNG_SAMPLE):
    """Render one strip.  `req` is any ctypes struct with rt_tile_request layout.
    rng_mode: RNG_SAMPLE = the normative stream per (pixel, sample); RNG_PIXEL (one stream per pixel, rounds
    1-3) and RNG_ROW (the reference's structure: one stream per row, S/main.rs:69-77) exist for the distribution test only.
    world_index: position of every sphere, then of every triangle, in the reference's `world: Vec<Object>` (None: spheres
    then triangles).  Returns (rgb uint8 [Hs*W*3], f32 or None, info dict)."""
    lib = load()
    sph = np.ascontiguousarray(spheres) if spheres is not None else np.zeros(0, np.uint8)
    tri = np.ascontiguousarray(triangles) if triangles is not None else np.zeros(0, np.uint8)
    ns = 0 if spheres is None else len(spheres)
    nt = 0 if triangles is None else len(triangles)
    n = (req.height // req.divisions) * req.width * 3
    out = np.zeros(n, np.uint8)
    outf = np.zeros(n, np.float32) if want_f32 else None
    segs = C.c_uint64(0)
    ms = C.c_double(0)
    bms = C.c_double(0)
    wi = None if world_index is None else np.ascontiguousarray(world_index, dtype=np.uint32)
    if wi is not None and wi.size != ns + nt:
        raise ValueError("world_index: one entry per primitive")
    rc = lib.rt_oracle_render(C.byref(req), _p(sph), C.c_uint32(ns), _p(tri), C.c_uint32(nt), C.c_int(backend),
                              C.c_int(nthreads), _p(out), _p(outf) if want_f32 else None, C.byref(segs),
                              C.byref(ms), C.byref(bms), _p(wi) if wi is not None else None, C.c_int(rng_mode))
    if rc != 0:
        raise ValueError(f"rt_oracle_render: bad arguments ({rc})")
    return out, outf, {"ray_segments": segs.value, "render_ms": ms.value, "bvh_build_ms": bms.value}


def xoshiro_from_state(state4, n: int):
    st = np.asarray(state4, dtype=np.uint64)
    out = np.zeros(n, np.uint64)
    load().rt_oracle_xoshiro_from_state(_p(st), _p(out), C.c_int(n))
    return out


def seed_from_u64(seed: int):
    st = np.zeros(4, np.uint64)
    load().rt_oracle_seed_from_u64(C.c_uint64(seed), _p(st))
    return st


def pixel_seed(job_seed: int, pix: int) -> int:
    return load().rt_oracle_pixel_seed(job_seed, pix)


def sample_seed(job_seed: int, pix: int, spp: int, s: int) -> int:
    return load().rt_oracle_sample_seed(job_seed, pix, spp, s)


def draw(state4: np.ndarray, kind: int):
    """kind 0 gen_range(0..1), 1 Uniform(-1,1), 2 UnitDisc, 3 UnitSphere; state advanced in place."""
    out = np.zeros(3, np.float32)
    load().rt_oracle_draw(_p(state4), C.c_int(kind), _p(out))
    return out[: (1, 1, 2, 3)[kind]].copy()


def find_roots_quadratic(a2, a1, a0):
    out = np.zeros(2, np.float32)
    n = load().rt_oracle_find_roots_quadratic(a2, a1, a0, _p(out))
    return [float(x) for x in out[:n]]


def sphere_roots(sphere_rec, origin, direction):
    s = np.ascontiguousarray(sphere_rec)
    out = np.zeros(2, np.float32)
    n = load().rt_oracle_sphere_roots(_p(s), _p(_f3(origin)), _p(_f3(direction)), _p(out))
    return [float(x) for x in out[:n]]


def triangle_roots(tri_rec, origin, direction):
    t = np.ascontiguousarray(tri_rec)
    out = np.zeros(2, np.float32)
    n = load().rt_oracle_triangle_roots(_p(t), _p(_f3(origin)), _p(_f3(direction)), _p(out))
    return [float(x) for x in out[:n]]


def intersect(spheres, triangles, origin, direction, t_min=0.001, t_max=1000.0, backend=0):
    sph = np.ascontiguousarray(spheres) if spheres is not None else None
    tri = np.ascontiguousarray(triangles) if triangles is not None else None
    out = np.zeros(11, np.float32)
    idx = C.c_uint32(0)
    hit = load().rt_oracle_intersect(_p(sph), C.c_uint32(0 if sph is None else len(sph)), _p(tri),
                                     C.c_uint32(0 if tri is None else len(tri)), C.c_float(t_min), C.c_float(t_max),
                                     C.c_int(backend), _p(_f3(origin)), _p(_f3(direction)), _p(out), C.byref(idx))
    if not hit:
        return None
    return {"index": idx.value, "point": out[0:3].copy(), "normal": out[3:6].copy(), "albedo": out[6:9].copy(),
            "roughness": float(out[9]), "emission": float(out[10])}


def ray_color(spheres, triangles, origin, direction, depth, state4, t_min=0.001, t_max=1000.0):
    sph = np.ascontiguousarray(spheres) if spheres is not None else None
    tri = np.ascontiguousarray(triangles) if triangles is not None else None
    out = np.zeros(3, np.float32)
    segs = C.c_uint64(0)
    load().rt_oracle_ray_color(_p(sph), C.c_uint32(0 if sph is None else len(sph)), _p(tri),
                               C.c_uint32(0 if tri is None else len(tri)), C.c_float(t_min), C.c_float(t_max),
                               _p(_f3(origin)), _p(_f3(direction)), C.c_uint32(depth), _p(state4), _p(out),
                               C.byref(segs))
    return out, segs.value


def sky(direction):
    out = np.zeros(3, np.float32)
    load().rt_oracle_sky(_p(_f3(direction)), _p(out))
    return out


def quantise(rgb):
    out = np.zeros(3, np.uint8)
    load().rt_oracle_quantise(_p(_f3(rgb)), _p(out))
    return out


def camera_ray(req, x: int, y_cam: int, state4: np.ndarray):
    out = np.zeros(6, np.float32)
    load().rt_oracle_camera_ray(C.byref(req), C.c_uint32(x), C.c_uint32(y_cam), _p(state4), _p(out))
    return out[:3].copy(), out[3:].copy()


def camera_consts(req):
    out = np.zeros(9, np.float32)
    load().rt_oracle_camera_consts(C.byref(req), _p(out))
    return out[0:3].copy(), out[3:6].copy(), out[6:9].copy()


def bvh_traverse_boxes(boxes, origin, direction):
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    out = np.zeros(len(b) + 1, np.uint32)
    nn = C.c_uint32(0)
    n = load().rt_oracle_bvh_traverse_boxes(_p(b), C.c_uint32(len(b)), _p(_f3(origin)), _p(_f3(direction)), _p(out),
                                            C.c_uint32(len(out)), C.byref(nn))
    return out[:n].tolist(), nn.value


def aabb_kat(a6, b6, point):
    """size, center, surface_area, largest_axis, is_empty, join(a, b), grow(a, point) of the oracle's AABB helpers."""
    out = np.zeros(21, np.float32)
    load().rt_oracle_aabb_kat(_p(np.ascontiguousarray(a6, np.float32)), _p(np.ascontiguousarray(b6, np.float32)),
                              _p(_f3(point)), _p(out))
    return {"size": out[0:3], "center": out[3:6], "surface_area": float(out[6]), "largest_axis": int(out[7]),
            "is_empty": bool(out[8]), "join": out[9:15], "grow": out[15:21]}


def aabb_empty():
    out = np.zeros(6, np.float32)
    load().rt_oracle_aabb_empty(_p(out))
    return out


def axis_get(v, axis: int) -> float:
    lib = load()
    lib.rt_oracle_axis_get.restype = C.c_float
    return float(lib.rt_oracle_axis_get(_p(_f3(v)), C.c_int(axis)))


def ray_intersects_aabb(origin, direction, box6) -> bool:
    return bool(load().rt_oracle_ray_intersects_aabb(_p(_f3(origin)), _p(_f3(direction)),
                                                     _p(np.ascontiguousarray(box6, np.float32))))


def ray_intersects_aabb_flipped(origin, direction, box6) -> bool:
    """Ray::new(origin, direction), then direction and inv_direction negated with the cached signs kept (B/ray.rs:420-423)."""
    return bool(load().rt_oracle_ray_intersects_aabb_flipped(_p(_f3(origin)), _p(_f3(direction)),
                                                             _p(np.ascontiguousarray(box6, np.float32))))
