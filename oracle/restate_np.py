"""Second, independent restatement of the reference render path in pure Python + numpy.float32.

*** TEST INFRASTRUCTURE ONLY. ***  Written directly from the reference sources (not from
rt_oracle.cpp) so that the two restatements check each other: a transcription slip in one
shows up as a pixel difference in tests/test_oracle_cross.py.  Pure-Python loops: use only
for tiny images.  Linear intersect only (no BVH).

Every arithmetic value is a numpy.float32 scalar, so each +,-,*,/ and sqrt is one correctly
rounded IEEE binary32 operation, as in the Rust build of the reference (x86-64 SSE2, no FMA).

Citations: S = ray-tracer-slave/src, B = ray-tracer-slave/local-dependencies/bvh/src.
"""
from __future__ import annotations

import numpy as np

F = np.float32
M64 = (1 << 64) - 1
ZERO, ONE, TWO, HALF = F(0), F(1), F(2), F(0.5)


# ------------------------------------------------------------------ rand 0.8.5 SmallRng
class SmallRng:
    """xoshiro256++; seed_from_u64 = 4 SplitMix64 outputs; next_u32 = next_u64 >> 32."""

    def __init__(self, state):
        self.s = list(state)

    @staticmethod
    def _mix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    @classmethod
    def seed_from_u64(cls, state):
        out = []
        for _ in range(4):
            state = (state + 0x9E3779B97F4A7C15) & M64
            out.append(cls._mix(state))
        return cls(out)

    def next_u64(self):
        s = self.s
        rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & M64
        result = (rotl((s[0] + s[3]) & M64, 23) + s[0]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        return result

    def next_u32(self):
        return self.next_u64() >> 32

    def value0_1(self):
        bits = np.uint32((self.next_u32() >> 9) | 0x3F800000)
        return bits.view(np.float32) - ONE

    def gen_range_01(self):                     # rng.gen_range(0f32..1f32)
        return self.value0_1() * ONE + ZERO

    def uniform_m1_1(self):                     # Uniform::new(-1., 1.).sample
        return self.value0_1() * TWO + F(-1)


def pixel_seed(job_seed, pixel_index):          # rounds 1-3 (one stream per pixel); kept for the cross-check of the helper
    return SmallRng._mix((job_seed + (pixel_index + 1) * 0x9E3779B97F4A7C15) & M64)


def sample_seed(job_seed, pixel_index, spp, sample):   # DESIGN.md "RNG": one stream per (pixel, sample)
    return (job_seed + 4 * 0x9E3779B97F4A7C15 * (pixel_index * spp + sample)) & M64


def unit_disc(rng):                             # rand_distr::UnitDisc
    while True:
        x1, x2 = rng.uniform_m1_1(), rng.uniform_m1_1()
        if x1 * x1 + x2 * x2 <= ONE:
            return x1, x2


def unit_sphere(rng):                           # rand_distr::UnitSphere
    while True:
        x1, x2 = rng.uniform_m1_1(), rng.uniform_m1_1()
        s = x1 * x1 + x2 * x2
        if s >= ONE:
            continue
        f = TWO * np.sqrt(ONE - s)
        return (x1 * f, x2 * f, ONE - TWO * s)


# ------------------------------------------------------------------ glam Vec3A as tuples
def add(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def scale(s, a): return (s * a[0], s * a[1], s * a[2])
def divs(a, s): return (a[0] / s, a[1] / s, a[2] / s)
def dot(a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
def length(a): return np.sqrt(dot(a, a))
def cross(a, b): return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def try_normalize(a):
    with np.errstate(divide="ignore", invalid="ignore"):
        rcp = ONE / length(a)
    if np.isfinite(rcp) and rcp > ZERO:
        return scale(rcp, a)
    return None


def normalize_or_zero(a):
    r = try_normalize(a)
    return r if r is not None else (ZERO, ZERO, ZERO)


def ray_new(origin, direction):                 # B/ray.rs:133-143: direction.normalize() = v / |v|
    with np.errstate(divide="ignore", invalid="ignore"):
        return origin, divs(direction, length(direction))


# ------------------------------------------------------------------ roots 0.0.8
def find_roots_quadratic(a2, a1, a0):
    if a2 == ZERO:
        if a1 == ZERO:
            return [ZERO] if a0 == ZERO else []
        return [-a0 / a1]
    disc = a1 * a1 - F(4) * a2 * a0
    if disc < ZERO:
        return []
    a2x2 = TWO * a2
    if disc == ZERO:
        return [-a1 / a2x2]
    sq = np.sqrt(disc)
    if a1 < ZERO:
        same, diff = -a1 + sq, -a1 - sq
    else:
        same, diff = -a1 - sq, -a1 + sq
    if abs(same) > abs(a2x2):
        a0x2 = TWO * a0
        if abs(diff) > abs(a2x2):
            x1, x2 = a0x2 / same, a0x2 / diff
        else:
            x1, x2 = a0x2 / same, same / a2x2
    else:
        x1, x2 = diff / a2x2, same / a2x2
    return [x1, x2] if x1 < x2 else [x2, x1]


# ------------------------------------------------------------------ shapes
def sphere_roots(s, o, d):                      # S/shapes/sphere.rs:42-47
    c = (F(s["cx"]), F(s["cy"]), F(s["cz"]))
    r = F(s["radius"])
    oc = sub(o, c)
    b = dot(scale(TWO, d), oc)
    ln = length(oc)
    cc = ln * ln - r * r
    return find_roots_quadratic(ONE, b, cc)


def triangle_roots(t, o, d):                    # S/shapes/mesh.rs:109-161
    eps = F(0.00001)
    A = tuple(F(x) for x in t["a"])
    B = tuple(F(x) for x in t["b"])
    Cc = tuple(F(x) for x in t["c"])
    a_to_b, a_to_c = sub(B, A), sub(Cc, A)
    u_vec = cross(d, a_to_c)
    det = dot(a_to_b, u_vec)
    if det < eps and det > -eps:
        return []
    inv_det = ONE / det
    a_to_origin = sub(o, A)
    u = dot(a_to_origin, u_vec) * inv_det
    if not (ZERO <= u <= ONE):
        return []
    v_vec = cross(a_to_origin, a_to_b)
    v = dot(d, v_vec) * inv_det
    if v < ZERO or u + v > ONE:
        return []
    dist = dot(a_to_c, v_vec) * inv_det
    return [dist] if dist > eps else []


def pick_t(roots, t_min, t_max):                # S/shapes/mod.rs:106-127
    inr = lambda x: (x >= t_min) and (x < t_max)
    if len(roots) == 0:
        return None
    if len(roots) == 1:
        return roots[0] if inr(roots[0]) else None
    x, y = roots
    xi, yi = inr(x), inr(y)
    if xi and yi:
        return x if x < y else y
    if xi:
        return x
    if yi:
        return y
    return None


class World:
    def __init__(self, spheres, triangles, t_min=0.001, t_max=1000.0, world_index=None):
        self.spheres = [] if spheres is None else list(spheres)
        self.triangles = [] if triangles is None else list(triangles)
        # `world: Vec<Object>` (S/lib.rs:11) is ONE list; world_index gives the position of every sphere, then of every
        # triangle, in it (include/rt_tile.h "the world's order"); None: the spheres, then the triangles
        objs = [("s", ob) for ob in self.spheres] + [("t", ob) for ob in self.triangles]
        if world_index is None:
            self.objects = objs
        else:
            self.objects = [None] * len(objs)
            for i, w in enumerate(world_index):
                self.objects[int(w)] = objs[i]
        self.t_min, self.t_max = F(t_min), F(t_max)
        self.segments = 0

    def intersect(self, o, d):                  # S/shapes/mod.rs:158-191
        best = None
        if True:
            for kind, ob in self.objects:           # front to back through the list
                roots = sphere_roots(ob, o, d) if kind == "s" else triangle_roots(ob, o, d)
                t = pick_t(roots, self.t_min, self.t_max)
                if t is None:
                    continue
                p = add(o, scale(t, d))         # Ray::at
                dist = length(sub(p, o))
                if best is None or best[0] > dist:   # min_by: first minimum wins
                    best = (dist, p, kind, ob)
        if best is None:
            return None
        _, p, kind, ob = best
        if kind == "s":
            n = normalize_or_zero(sub(p, (F(ob["cx"]), F(ob["cy"]), F(ob["cz"]))))
        else:
            A = tuple(F(x) for x in ob["a"])
            B = tuple(F(x) for x in ob["b"])
            Cc = tuple(F(x) for x in ob["c"])
            n = normalize_or_zero(cross(sub(A, B), sub(A, Cc)))
        alb = (F(ob["albedo_r"]), F(ob["albedo_g"]), F(ob["albedo_b"]))
        return p, n, alb, F(ob["roughness"]), F(ob["emission"])


def ray_color(world, o, d, depth, rng):         # S/main.rs:108-146
    if depth == 0:
        return (ZERO, ZERO, ZERO)
    world.segments += 1
    hit = world.intersect(o, d)
    if hit is not None:
        p, n, alb, rough, emis = hit
        if emis > ZERO:
            return (alb[0] * emis, alb[1] * emis, alb[2] * emis)
        diffuse = add(unit_sphere(rng), n)
        glossy = sub(d, scale(TWO * dot(d, n), n))
        scatter = add(diffuse, scale(rough, sub(glossy, diffuse)))
        nd = try_normalize(scatter)
        if nd is None:
            nd = n
        o2, d2 = ray_new(p, nd)
        c = ray_color(world, o2, d2, depth - 1, rng)
        return (alb[0] * c[0], alb[1] * c[1], alb[2] * c[2])
    t = normalize_or_zero(d)[1] * HALF + ONE
    omt = ONE - t
    return (ONE * t + F(0.3) * omt, ONE * t + F(0.3) * omt, ONE * t + F(0.8) * omt)


# ------------------------------------------------------------------ camera (S/camera.rs)
class Camera:
    def __init__(self, origin, aspect_ratio, aperture, focus_distance, fov, focal_length, image_height):
        vh = TWO * np.tan(F(fov) / TWO)
        vw = F(aspect_ratio) * vh
        self.origin = origin
        self.horizontal = (vw, ZERO, ZERO)
        self.vertical = (ZERO, vh, ZERO)
        self.aspect_ratio, self.image_height = F(aspect_ratio), F(image_height)
        self.aperture, self.focus_distance = F(aperture), F(focus_distance)
        self.llc = sub(sub(sub(origin, divs(self.horizontal, TWO)), divs(self.vertical, TWO)),
                       (ZERO, ZERO, F(focal_length)))

    def get_ray(self, x, y, rng):
        lens_radius = self.aperture / TWO
        a, b = unit_disc(rng)
        offset = (a * lens_radius, b * lens_radius, ZERO)
        u = (F(x) + rng.gen_range_01()) / (self.aspect_ratio * self.image_height - ONE)
        v = (F(y) + rng.gen_range_01()) / (self.image_height - ONE)
        dirv = sub(add(add(self.llc, scale(u, self.horizontal)), scale(v, self.vertical)), self.origin)
        fo, fd = ray_new(self.origin, normalize_or_zero(dirv))
        focal_point = add(fo, scale(self.focus_distance, fd))
        final_origin = add(self.origin, offset)
        return ray_new(final_origin, normalize_or_zero(sub(focal_point, final_origin)))


def as_u8(c):                                   # S/color.rs:13-19, Rust saturating `as u8`
    v = c * F(255.999)
    if np.isnan(v) or v <= 0:
        return 0
    if v >= 255:
        return 255
    return int(v)


def render(req, spheres, triangles=None, world_index=None):
    """Tile loop S/main.rs:37-83 for strip req.division_no.  Returns (rgb uint8 [Hs,W,3], f32, segments)."""
    W, H = req.width, req.height
    hs = H // req.divisions
    cam = Camera((ZERO, ZERO, ZERO), F(W) / F(H), req.aperture, req.focus_distance, req.fov, req.focal_length, F(H))
    world = World(spheres, triangles, req.t_min, req.t_max, world_index)
    rgb = np.zeros((hs, W, 3), np.uint8)
    f32 = np.zeros((hs, W, 3), np.float32)
    n = F(req.spp)
    for yl in range(hs):
        yg = hs * req.division_no + yl
        yc = H - yg - 1
        for x in range(W):
            pr, pg, pb = ZERO, ZERO, ZERO
            for s in range(req.spp):
                rng = SmallRng.seed_from_u64(sample_seed(req.seed, yg * W + x, req.spp, s))
                o, d = cam.get_ray(x, yc, rng)
                c = ray_color(world, o, d, req.max_bounces + 1, rng)
                pr, pg, pb = pr + c[0], pg + c[1], pb + c[2]
            pix = (np.sqrt(pr / n), np.sqrt(pg / n), np.sqrt(pb / n))
            f32[yl, x] = pix
            rgb[yl, x] = [as_u8(pix[0]), as_u8(pix[1]), as_u8(pix[2])]
    return rgb, f32, world.segments
