// Test harness (CPU only, built by tests/test_cull_lemma.py with g++ -ffp-contract=off): the inequality the culled walks
// rest on, evaluated for single (ray, primitive, box) triples.
//
// The culled walks (rt_kernel.hip.h, ISECT 7 / 8 / 9) skip every subtree whose box the ray ENTERS beyond
// t_far = cull_bound(best) (spheres) or cull_bound_tri(best) (triangles), `best` being the compared distance |P - o| of
// the running closest hit.  That is exact iff no primitive X below such a box could still have beaten or tied `best`:
//
//     for every ray and every primitive X that the reference would test (the ray passes X's own AABB, hence every
//     ancestor's) and whose reference root test returns a root x in [t_min, t_max) with compared distance D_X:
//              entry(ray, AABB_X)  <=  bound(D_X)                                                    (*)
//
// (bound is monotone in its first argument and ancestors' boxes are entered no later than X's own, so (*) for the leaf box
// and best = D_X covers every box on X's path and every running best >= D_X.)  Here (*) is evaluated with the PRODUCT's
// bound functions (csrc/rt_cull.h), the reference's f32 root arithmetic restated below (sphere.rs:42-47 +
// roots::find_roots_quadratic + shapes/mod.rs:106-129; mesh.rs:109-161), and the box entry both as the kernels compute it
// in f32 (slabs_finite) and in double.  A violation is a triple where X is a candidate with an accepted root and its box
// entry exceeds the bound.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "rt_cull.h"

namespace {

struct V3 {
    float x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }     // glam dot3 order
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) {                                                       // glam normalize: divide (Ray::new)
    const float l = length(a);
    return {a.x / l, a.y / l, a.z / l};
}
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// splitmix64 stream
struct Rng {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    double u() { return (double)(next() >> 11) * 0x1p-53; }                       // [0, 1)
    double range(double a, double b) { return a + (b - a) * u(); }
    double logrange(double a, double b) { return std::exp(range(std::log(a), std::log(b))); }
    V3 unit() {
        for (;;) {
            const double x = range(-1, 1), y = range(-1, 1), z = range(-1, 1), l = x * x + y * y + z * z;
            if (l > 1e-4 && l <= 1.0) {
                const double r = 1.0 / std::sqrt(l);
                return {(float)(x * r), (float)(y * r), (float)(z * r)};
            }
        }
    }
};

// ---- the reference's sphere root (restated; the oracle holds the same restatement, oracle/rt_oracle.cpp)
bool find_roots_quadratic_1(float a1, float a0, float* r0, float* r1, int* n) {   // a2 = 1
    const float disc = a1 * a1 - 4.0f * a0;
    if (disc < 0.0f) {
        *n = 0;
        return true;
    }
    if (disc == 0.0f) {
        *n = 1;
        *r0 = -a1 / 2.0f;
        return true;
    }
    if (!(disc > 0.0f)) {                    // NaN
        *n = 0;
        return true;
    }
    const float sq = sqrtf(disc);
    float same, diff;
    if (a1 < 0.0f) {
        same = -a1 + sq;
        diff = -a1 - sq;
    } else {
        same = -a1 - sq;
        diff = -a1 + sq;
    }
    float x1, x2;
    if (fabsf(same) > 2.0f) {
        const float a0x2 = 2.0f * a0;
        if (fabsf(diff) > 2.0f) {
            x1 = a0x2 / same;
            x2 = a0x2 / diff;
        } else {
            x1 = a0x2 / same;
            x2 = same / 2.0f;
        }
    } else {
        x1 = diff / 2.0f;
        x2 = same / 2.0f;
    }
    *n = 2;
    if (x1 < x2) {
        *r0 = x1;
        *r1 = x2;
    } else {
        *r0 = x2;
        *r1 = x1;
    }
    return true;
}
bool select_t(int n, float x, float y, float t_min, float t_max, float* t) {      // shapes/mod.rs:106-129
    auto in = [&](float v) { return v >= t_min && v < t_max; };
    if (n == 0) return false;
    if (n == 1) {
        if (in(x)) {
            *t = x;
            return true;
        }
        return false;
    }
    const bool xi = in(x), yi = in(y);
    if (xi && yi) {
        *t = x < y ? x : y;
        return true;
    }
    if (xi) {
        *t = x;
        return true;
    }
    if (yi) {
        *t = y;
        return true;
    }
    return false;
}
bool ref_sphere(V3 o, V3 d, V3 c, float r, float t_min, float t_max, float* t) {  // sphere.rs:42-47
    const V3 oc = o - c;
    const float b = dot(2.0f * d, oc);
    const float len = length(oc);
    const float cc = len * len - r * r;
    float x = 0, y = 0;
    int n = 0;
    find_roots_quadratic_1(b, cc, &x, &y, &n);
    return select_t(n, x, y, t_min, t_max, t);
}
bool ref_triangle(V3 o, V3 d, V3 A, V3 B, V3 C, float t_min, float t_max, float* t) {   // mesh.rs:109-161
    const float EPSILON = 0.00001f;
    const V3 a_to_b = B - A, a_to_c = C - A;
    const V3 u_vec = cross(d, a_to_c);
    const float det = dot(a_to_b, u_vec);
    if (det < EPSILON && det > -EPSILON) return false;
    const float inv_det = 1.0f / det;
    const V3 a_to_origin = o - A;
    const float u = dot(a_to_origin, u_vec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    const V3 v_vec = cross(a_to_origin, a_to_b);
    const float v = dot(d, v_vec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    const float dist = dot(a_to_c, v_vec) * inv_det;
    if (!(dist > EPSILON)) return false;
    *t = dist;
    return dist >= t_min && dist < t_max;
}
// the compared distance: P = ray.at(t), |P - origin| (shapes/mod.rs:128, 177-182)
float compared_distance(V3 o, V3 d, float t) {
    const V3 p = o + t * d;
    return length(p - o);
}

// ---- box entry.  f32 exactly as the exact-node culled walk evaluates it (rt_kernel.hip.h slabs_finite, finite inverse
// direction): returns false when the kernel's own test `le <= lmax` fails (X is then no candidate for that ray).
bool entry_f32(V3 o, V3 d, const float lo[3], const float hi[3], float* le) {
    const float ox[3] = {o.x, o.y, o.z}, dx[3] = {d.x, d.y, d.z};
    float mn = -INFINITY, mx = INFINITY;
    for (int a = 0; a < 3; a++) {
        const float inv = 1.0f / dx[a];
        if (!(fabsf(inv) < INFINITY)) return false;          // +-0 component: the kernels walk the literal test, no culling claims
        const float t0 = (lo[a] - ox[a]) * inv, t1 = (hi[a] - ox[a]) * inv;
        mn = fmaxf(mn, fminf(t0, t1));
        mx = fminf(mx, fmaxf(t0, t1));
    }
    *le = fmaxf(mn, 0.0f);
    return *le <= mx;
}
bool entry_f64(V3 o, V3 d, const float lo[3], const float hi[3], double* le) {
    const double ox[3] = {o.x, o.y, o.z}, dx[3] = {d.x, d.y, d.z};
    double mn = -INFINITY, mx = INFINITY;
    for (int a = 0; a < 3; a++) {
        if (dx[a] == 0.0) {
            if (ox[a] < lo[a] || ox[a] > hi[a]) return false;
            continue;
        }
        const double t0 = ((double)lo[a] - ox[a]) / dx[a], t1 = ((double)hi[a] - ox[a]) / dx[a];
        mn = std::fmax(mn, std::fmin(t0, t1));
        mx = std::fmin(mx, std::fmax(t0, t1));
    }
    *le = std::fmax(mn, 0.0);
    return *le <= mx;
}

struct Tally {
    uint64_t cases = 0, roots = 0, candidates = 0, violations = 0;
    double min_slack = INFINITY;          // smallest (bound - entry) / max(bound, 1e-30) over the candidates with a root
    double worst[16] = {0};
    void judge(bool cand32, float le32, bool cand64, double le64, float bound, const double* rec, int nrec) {
        if (!(cand32 || cand64)) return;
        candidates++;
        double ent = -INFINITY;
        if (cand32) ent = std::fmax(ent, (double)le32);
        if (cand64) ent = std::fmax(ent, le64);
        const double slack = ((double)bound - ent) / std::fmax((double)bound, 1e-30);
        if (!(ent <= (double)bound)) violations++;
        if (slack < min_slack || !(slack == slack)) {
            min_slack = slack;
            for (int i = 0; i < nrec && i < 16; i++) worst[i] = rec[i];
        }
    }
    void out(double* o) const {
        o[0] = (double)cases;
        o[1] = (double)roots;
        o[2] = (double)candidates;
        o[3] = (double)violations;
        o[4] = min_slack;
        for (int i = 0; i < 16; i++) o[5 + i] = worst[i];
    }
};

}  // namespace

extern "C" {

// Spheres.  mode: 0 generic (aimed inside 1.3 r of the centre), 1 tangent rays (offset r (1 +- 10^-2..-7)), 2 far and
// small (|oc| 10^2..10^3, r 0.01..0.3: the false-root domain), 3 origins at |o| ~ 10^3, 4 origin ON another sphere's
// surface and inside spheres (bounce rays, far root).  slack_factor: r_slack = slack_factor * |r| (>= 1: the scene's
// r_slack is the LARGEST radius among the non-big spheres).  out[21] receives the tally.
void cull_check_spheres(uint64_t seed, uint64_t n, int mode, double slack_factor, double* out) {
    Rng g{seed * 0x100000001b3ull + (uint64_t)mode};
    Tally ty;
    const float t_min = 0.001f, t_max = 1000.0f;
    for (uint64_t i = 0; i < n; i++) {
        ty.cases++;
        double r, dist;
        V3 o;
        const double oscale = mode == 3 ? 1000.0 : 50.0;
        o = {(float)g.range(-oscale, oscale), (float)g.range(-oscale, oscale), (float)g.range(-oscale, oscale)};
        if (mode == 3) {                                   // somewhere on the |o| ~ 10^3 shell
            const V3 u = g.unit();
            const float m = (float)g.range(800.0, 1500.0);
            o = {u.x * m, u.y * m, u.z * m};
        }
        switch (mode) {
            case 2: r = g.logrange(0.01, 0.3); dist = g.logrange(100.0, 995.0); break;
            case 3: r = g.logrange(0.01, 3.0); dist = g.logrange(0.05, 900.0); break;
            case 4: r = g.logrange(0.05, 5.0); dist = g.range(0.0, 1.0) * r * (g.u() < 0.5 ? 1.0 : 1.0 + 1e-4 * g.range(-1, 1)); break;
            default: r = g.logrange(0.01, 5.0); dist = g.logrange(0.01, 990.0); break;
        }
        const V3 dirc = g.unit();                            // towards the centre
        const V3 c = {(float)(o.x + dirc.x * dist), (float)(o.y + dirc.y * dist), (float)(o.z + dirc.z * dist)};
        // aim point: centre + offset in the plane normal to dirc
        V3 side = cross(dirc, g.unit());
        side = normalize(side);
        double off;
        if (mode == 1) off = r * (1.0 + (g.u() < 0.5 ? -1.0 : 1.0) * std::pow(10.0, -g.range(2.0, 7.0)));
        else if (mode == 2) off = r * g.range(0.0, 1.6);
        else off = r * g.range(0.0, 1.3);
        V3 aim = {(float)(c.x + side.x * off), (float)(c.y + side.y * off), (float)(c.z + side.z * off)};
        if (mode == 4) aim = o + g.unit();                   // any direction from inside / from the surface
        const V3 d = normalize(aim - o);                     // Ray::new
        if (!(d.x == d.x) || !(fabsf(d.x) + fabsf(d.y) + fabsf(d.z) > 0.5f)) continue;
        const float rf = (float)r;
        float t;
        if (!ref_sphere(o, d, c, rf, t_min, t_max, &t)) continue;
        ty.roots++;
        const float D = compared_distance(o, d, t);
        const float lo[3] = {c.x - rf, c.y - rf, c.z - rf}, hi[3] = {c.x + rf, c.y + rf, c.z + rf};   // Sphere::aabb, sphere.rs:65-72
        float le32 = 0;
        double le64 = 0;
        const bool c32 = entry_f32(o, d, lo, hi, &le32), c64 = entry_f64(o, d, lo, hi, &le64);
        const float bound = rtk::cull_bound(D, o.x, o.y, o.z, (float)(slack_factor * r));
        const double rec[] = {o.x, o.y, o.z, d.x, d.y, d.z, c.x, c.y, c.z, rf, t, D, le32, le64, bound, (double)mode};
        ty.judge(c32, le32, c64, le64, bound, rec, 16);
    }
    ty.out(out);
}

// Triangles.  mode: 0 generic (K log-uniform up to 0.25), 1 K -> 0.25 (|e1| = |e2| = 0.5 (1 - 10^-1..-6)), 2 grazing
// (|det| between 1 and 30 times the reference's 10^-5 threshold), 3 origins at |o| ~ 10^3, 4 slivers (edges at 10^-3..-1
// rad) seen from far, 5 = 1 and 2 together (the largest triangles the bound admits, at grazing angles: roots off by per cent).
// The bound's scene maxima are this triangle's own values, formed as the host forms them
// (rt_api.hip build_host_scene: in double, times 1.0001, rounded to f32).
void cull_check_triangles(uint64_t seed, uint64_t n, int mode, double* out) {
    Rng g{seed * 0x100000001b3ull + 0x7419ull + (uint64_t)mode};
    Tally ty;
    const float t_min = 0.001f, t_max = 1000.0f;
    for (uint64_t i = 0; i < n; i++) {
        ty.cases++;
        // edge lengths with K = l1 l2 <= 0.25
        double K = (mode == 1 || mode == 5) ? 0.25 * std::pow(1.0 - std::pow(10.0, -g.range(1.0, 6.0)), 2.0) : g.logrange(1e-4, 0.25);
        double ratio = (mode == 1 || mode == 5) ? 1.0 : g.logrange(0.2, 5.0);
        double l1 = std::sqrt(K * ratio), l2 = std::sqrt(K / ratio);
        const V3 u1 = g.unit();
        V3 u2 = g.unit();
        if (mode == 4) {                                    // sliver: second edge almost along the first
            const double ang = g.logrange(1e-3, 1e-1);
            const V3 perp = normalize(cross(u1, g.unit()));
            u2 = {(float)(std::cos(ang) * u1.x + std::sin(ang) * perp.x), (float)(std::cos(ang) * u1.y + std::sin(ang) * perp.y),
                  (float)(std::cos(ang) * u1.z + std::sin(ang) * perp.z)};
        }
        const double ascale = mode == 3 ? 1000.0 : 40.0;
        V3 A = {(float)g.range(-ascale, ascale), (float)g.range(-ascale, ascale), (float)g.range(-ascale, ascale)};
        const V3 B = {(float)(A.x + l1 * u1.x), (float)(A.y + l1 * u1.y), (float)(A.z + l1 * u1.z)};
        const V3 C = {(float)(A.x + l2 * u2.x), (float)(A.y + l2 * u2.y), (float)(A.z + l2 * u2.z)};
        // the rounded edges as the host and the reference see them
        double e1 = 0, e2 = 0, e3 = 0, d2 = 0;
        const float va[3] = {A.x, A.y, A.z}, vb[3] = {B.x, B.y, B.z}, vc[3] = {C.x, C.y, C.z};
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            const double ab = (double)vb[a] - va[a], ac = (double)vc[a] - va[a], bc = (double)vc[a] - vb[a];
            e1 += ab * ab;
            e2 += ac * ac;
            e3 += bc * bc;
            lo[a] = fminf(fminf(va[a], vb[a]), vc[a]);       // Triangle::aabb, mesh.rs:46-96
            hi[a] = fmaxf(fmaxf(va[a], vb[a]), vc[a]);
            const double ext = (double)hi[a] - lo[a];
            d2 += ext * ext;
        }
        e1 = std::sqrt(e1);
        e2 = std::sqrt(e2);
        e3 = std::sqrt(e3);
        const float kk = (float)(e1 * e2 * 1.0001), dg = (float)(std::sqrt(d2) * 1.0001), es = (float)((e1 + e2) * 1.0001),
                    em = (float)(std::fmax(e1, std::fmax(e2, e3)) * 1.0001);
        if (kk > 0.25f) continue;                            // such a triangle is in the `big` list: no claim
        // a point of the triangle's plane in or near the triangle, and an origin
        double bu = g.range(-0.05, 1.05), bv = g.range(-0.05, 1.05);
        if (bu + bv > 1.0 && g.u() < 0.9) {
            bu = 1.0 - bu;
            bv = 1.0 - bv;
        }
        const V3 P = {(float)(A.x + bu * (B.x - A.x) + bv * (C.x - A.x)), (float)(A.y + bu * (B.y - A.y) + bv * (C.y - A.y)),
                      (float)(A.z + bu * (B.z - A.z) + bv * (C.z - A.z))};
        V3 nrm = cross(B - A, C - A);
        const float nl = length(nrm);
        if (!(nl > 0.0f)) continue;
        nrm = {nrm.x / nl, nrm.y / nl, nrm.z / nl};
        V3 dir = g.unit();
        if (mode == 2 || mode == 5) {
            // grazing: |det| = |d . (e1 x e2)| = |d . n| * |e1 x e2| a few times the 1e-5 threshold
            const double target = 1e-5 * g.logrange(1.0, 30.0) / (double)nl;
            V3 inpl = cross(nrm, g.unit());
            inpl = normalize(inpl);
            const double sg = g.u() < 0.5 ? -1.0 : 1.0;
            dir = {(float)(inpl.x + sg * target * nrm.x), (float)(inpl.y + sg * target * nrm.y), (float)(inpl.z + sg * target * nrm.z)};
        }
        const double far = mode == 3 ? g.logrange(1.0, 990.0) : mode == 4 ? g.logrange(10.0, 990.0) : g.logrange(0.002, 990.0);
        const V3 o = {(float)(P.x - dir.x * far), (float)(P.y - dir.y * far), (float)(P.z - dir.z * far)};
        const V3 d = normalize(P - o);
        if (!(d.x == d.x)) continue;
        float t;
        if (!ref_triangle(o, d, A, B, C, t_min, t_max, &t)) continue;
        ty.roots++;
        const float D = compared_distance(o, d, t);
        float le32 = 0;
        double le64 = 0;
        const bool c32 = entry_f32(o, d, lo, hi, &le32), c64 = entry_f64(o, d, lo, hi, &le64);
        const float bound = rtk::cull_bound_tri(D, o.x, o.y, o.z, kk, dg, es, em);
        const double rec[] = {o.x, o.y, o.z, d.x, d.y, d.z, A.x, A.y, A.z, kk, t, D, le32, le64, bound, (double)mode};
        ty.judge(c32, le32, c64, le64, bound, rec, 16);
    }
    ty.out(out);
}

// the bounds themselves (the product's functions), for the monotonicity check
float cull_bound_value(float best, float ox, float oy, float oz, float r_slack) { return rtk::cull_bound(best, ox, oy, oz, r_slack); }
float cull_bound_tri_value(float best, float ox, float oy, float oz, float k, float diag, float es, float e) {
    return rtk::cull_bound_tri(best, ox, oy, oz, k, diag, es, e);
}

}  // extern "C"
