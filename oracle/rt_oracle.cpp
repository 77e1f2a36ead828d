// rt_oracle.cpp — CPU restatement of the ray-tracer-s8 slave render path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing in the product path (ray_tracer_s8_amd/,
// the C-ABI library, bench.py's GPU leg) may link, import or call this file.  Only
// tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, as the
// checker / the timed CPU baseline.
//
// What it restates (reference = /root/reference, paths relative to it):
//   S = ray-tracer-slave/src, B = ray-tracer-slave/local-dependencies/bvh/src
//   tile loop            S/main.rs:37-83
//   ray_color            S/main.rs:108-146
//   Camera::new/get_ray  S/camera.rs:19-47, 109-129
//   intersect            S/shapes/mod.rs:98-192   (T_MIN/T_MAX :12-13)
//   Sphere               S/shapes/sphere.rs:41-72
//   Triangle             S/shapes/mesh.rs:46-96, 108-165
//   Color                S/color.rs:12-105
//   Ray::new/at/intersects_aabb   B/ray.rs:81-112, 133-194
//   AABB                 B/aabb.rs:96-98,124-129,268-282,357-372,458-484,525-528,570-580
//   BVH build/traverse   B/bvh/bvh_impl.rs:229-364, 373-398, 421-442; B/utils.rs:19-58
//
// PARITY UNPINNED: the reference is Rust and no Rust toolchain exists
// in this image, so the reference itself cannot be run; and the arithmetic of four
// un-vendored crates is restated from their published sources, not verified offline:
//   glam 0.23.0   (Vec3A, SSE2 path: dot = (x*x'+y*y')+z*z', normalize = v / sqrt(dot),
//                  normalize_or_zero / try_normalize = v * (1/sqrt(dot)) if finite & > 0)
//   roots 0.0.8   (find_roots_quadratic, the "do not use the smallest divisor" form)
//   rand 0.8.5    (SmallRng = xoshiro256++, seed_from_u64 = SplitMix64 fill,
//                  next_u32 = next_u64 >> 32, gen_range(0f32..1f32) = 23-bit mantissa - 1)
//   rand_distr 0.4.3 (UnitDisc rejection, UnitSphere Marsaglia, Uniform(-1,1) = v*2 + -1)
// The only vectors the reference's own tests hold for this path are the bvh crate's
// 21-box traversal fixture (B/testbase.rs:92-166), which pins the BVH back-end here,
// and rand's public xoshiro256++ reference vector.  Everything else is pinned by
// first-principles KATs derived by hand from the cited lines (tests/test_oracle_kat.py)
// and by an independent numpy restatement (oracle/restate_np.py).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
// x86-64 SSE2 scalar float semantics, no FMA contraction: the same arithmetic rustc emits.
//
// Two intersect back-ends behind one switch (rt_oracle_render `backend`):
//   0 = linear scan over all primitives in index order (what the GPU kernel computes)
//   1 = SAH BVH candidate filter as in the reference (used as the timed CPU baseline
//       so the CPU number has the reference's O(log N) asymptotics)
// The reference's `SmallRng::from_entropy()` per row (S/main.rs:69) is replaced by the
// deterministic stream per (pixel, sample) defined in DESIGN.md "RNG" (same in the HIP kernel):
// rt_oracle_render's rng_mode 0.  Two more modes exist HERE ONLY, for the statistical comparison of
// tests/test_rng_distribution.py: 1 = one stream per pixel spanning its samples (the definition of
// rounds 1-3), 2 = the reference's own structure — one stream per ROW, consumed pixel after pixel,
// sample after sample (S/main.rs:69-77) — seeded deterministically per row instead of from entropy.

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../include/rt_tile.h"

namespace {

// ---------------------------------------------------------------- glam::Vec3A (restated)
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glam sse2 dot3: x*x' then +y*y' then +z*z'
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline float length_recip(V3 a) { return 1.0f / std::sqrt(dot(a, a)); }
// glam normalize(): divide by the length (asserts finite only in debug builds)
inline V3 normalize(V3 a) { return a / length(a); }
inline bool try_normalize(V3 a, V3* out) {
    float rcp = length_recip(a);
    if (std::isfinite(rcp) && rcp > 0.0f) {
        *out = a * rcp;
        return true;
    }
    return false;
}
inline V3 normalize_or_zero(V3 a) {
    V3 r;
    return try_normalize(a, &r) ? r : v3(0.f, 0.f, 0.f);
}
// glam sse2 cross: (a.zxy*b - a*b.zxy).zxy  ==  (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float axis_of(V3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }

// ---------------------------------------------------------------- rand 0.8.5 SmallRng
struct Rng {
    uint64_t s[4];
};
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
// xoshiro256++ (rand_xoshiro / rand::rngs::SmallRng on 64-bit targets)
inline uint64_t next_u64(Rng& r) {
    uint64_t result = rotl64(r.s[0] + r.s[3], 23) + r.s[0];
    uint64_t t = r.s[1] << 17;
    r.s[2] ^= r.s[0];
    r.s[3] ^= r.s[1];
    r.s[1] ^= r.s[2];
    r.s[0] ^= r.s[3];
    r.s[2] ^= t;
    r.s[3] = rotl64(r.s[3], 45);
    return result;
}
inline uint32_t next_u32(Rng& r) { return (uint32_t)(next_u64(r) >> 32); }
inline uint64_t splitmix_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
const uint64_t PHI = 0x9e3779b97f4a7c15ull;
// SeedableRng::seed_from_u64 for xoshiro256++: four SplitMix64 outputs
inline Rng seed_from_u64(uint64_t state) {
    Rng r;
    for (int i = 0; i < 4; i++) {
        state += PHI;
        r.s[i] = splitmix_mix(state);
    }
    return r;
}
// DESIGN.md "RNG" (normative since round 4): sample s of pixel p of a job with S samples per pixel draws from
//   SmallRng::seed_from_u64(job_seed + 4 * PHI * (p * S + s))      (wrapping u64 arithmetic)
// i.e. the SplitMix64 sequence of the job seed is cut into consecutive blocks of four outputs and block p * S + s
// is that sample's xoshiro256++ state (seed_from_u64 fills the state with the next four SplitMix64 outputs).
inline uint64_t sample_seed(uint64_t job_seed, uint64_t pixel_index, uint64_t spp, uint64_t sample) {
    return job_seed + 4ull * PHI * (pixel_index * spp + sample);
}
// rng_mode 1 (rounds 1-3, comparison only): pixel p uses SmallRng::seed_from_u64(h) with
// h = the (p+1)-th SplitMix64 output of a generator seeded with the job seed; the stream spans the pixel's samples.
inline uint64_t pixel_seed(uint64_t job_seed, uint64_t pixel_index) {
    return splitmix_mix(job_seed + (pixel_index + 1) * PHI);
}
// rng_mode 2 (comparison only): the reference's one stream per row, seeded from (job seed, global row) instead of entropy
inline uint64_t row_seed(uint64_t job_seed, uint64_t global_row) {
    return splitmix_mix(~job_seed + (global_row + 1) * PHI);
}
// [1,2) float from the top 23 bits, minus 1  (rand UniformFloat / Standard mantissa trick)
inline float u01_from_u32(uint32_t v) {
    uint32_t bits = (v >> 9) | 0x3f800000u;
    float f;
    std::memcpy(&f, &bits, 4);
    return f - 1.0f;
}
// rng.gen_range(0f32..1f32): sample_single, scale = 1, low = 0; res < high always holds
inline float gen_range_01(Rng& r) { return u01_from_u32(next_u32(r)) * 1.0f + 0.0f; }
// Uniform::new(-1f32, 1f32).sample: value0_1 * scale(2) + low(-1)
inline float uniform_m1_1(Rng& r) { return u01_from_u32(next_u32(r)) * 2.0f + -1.0f; }
// rand_distr::UnitDisc
inline void unit_disc(Rng& r, float* a, float* b) {
    float x1, x2;
    for (;;) {
        x1 = uniform_m1_1(r);
        x2 = uniform_m1_1(r);
        if (x1 * x1 + x2 * x2 <= 1.0f) break;
    }
    *a = x1;
    *b = x2;
}
// rand_distr::UnitSphere (Marsaglia 1972)
inline V3 unit_sphere(Rng& r) {
    for (;;) {
        float x1 = uniform_m1_1(r);
        float x2 = uniform_m1_1(r);
        float sum = x1 * x1 + x2 * x2;
        if (sum >= 1.0f) continue;
        float factor = 2.0f * std::sqrt(1.0f - sum);
        return v3(x1 * factor, x2 * factor, 1.0f - 2.0f * sum);
    }
}

// ---------------------------------------------------------------- roots 0.0.8
struct Roots {
    int n;  // 0, 1, 2
    float x[2];
};
inline Roots find_roots_linear(float a1, float a0) {
    if (a1 == 0.0f) {
        // roots: a0 == 0 -> One([0]) else No
        if (a0 == 0.0f) return Roots{1, {0.0f, 0.0f}};
        return Roots{0, {0.f, 0.f}};
    }
    return Roots{1, {-a0 / a1, 0.f}};
}
inline Roots find_roots_quadratic(float a2, float a1, float a0) {
    if (a2 == 0.0f) return find_roots_linear(a1, a0);
    const float _2 = 2.0f, _4 = 4.0f;
    float discriminant = a1 * a1 - _4 * a2 * a0;
    if (discriminant < 0.0f) return Roots{0, {0.f, 0.f}};
    float a2x2 = _2 * a2;
    if (discriminant == 0.0f) return Roots{1, {-a1 / a2x2, 0.f}};
    float sq = std::sqrt(discriminant);
    float same_sign, diff_sign;
    if (a1 < 0.0f) {
        same_sign = -a1 + sq;
        diff_sign = -a1 - sq;
    } else {
        same_sign = -a1 - sq;
        diff_sign = -a1 + sq;
    }
    float x1, x2;
    if (std::fabs(same_sign) > std::fabs(a2x2)) {
        float a0x2 = _2 * a0;
        if (std::fabs(diff_sign) > std::fabs(a2x2)) {
            x1 = a0x2 / same_sign;
            x2 = a0x2 / diff_sign;
        } else {
            x1 = a0x2 / same_sign;
            x2 = same_sign / a2x2;
        }
    } else {
        x1 = diff_sign / a2x2;
        x2 = same_sign / a2x2;
    }
    if (x1 < x2) return Roots{2, {x1, x2}};
    return Roots{2, {x2, x1}};
}

// ---------------------------------------------------------------- bvh::ray::Ray
struct Ray {
    V3 origin, direction, inv_direction;
    int sign_x, sign_y, sign_z;
};
inline Ray ray_new(V3 origin, V3 direction) {  // B/ray.rs:133-143
    V3 d = normalize(direction);
    Ray r;
    r.origin = origin;
    r.direction = d;
    r.inv_direction = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.sign_x = d.x < 0.0f;
    r.sign_y = d.y < 0.0f;
    r.sign_z = d.z < 0.0f;
    return r;
}
inline V3 ray_at(const Ray& r, float t) { return r.origin + t * r.direction; }  // B/ray.rs:147-149

// ---------------------------------------------------------------- bvh::aabb::AABB
struct AABB {
    V3 mn, mx;
};
const float INF = std::numeric_limits<float>::infinity();
inline AABB aabb_empty() { return AABB{v3(INF, INF, INF), v3(-INF, -INF, -INF)}; }
// Rust f32::min / f32::max (IEEE minNum/maxNum) == fminf / fmaxf
inline AABB aabb_join(const AABB& a, const AABB& b) {
    return AABB{v3(fminf(a.mn.x, b.mn.x), fminf(a.mn.y, b.mn.y), fminf(a.mn.z, b.mn.z)),
                v3(fmaxf(a.mx.x, b.mx.x), fmaxf(a.mx.y, b.mx.y), fmaxf(a.mx.z, b.mx.z))};
}
inline AABB aabb_grow(const AABB& a, V3 p) {
    return AABB{v3(fminf(a.mn.x, p.x), fminf(a.mn.y, p.y), fminf(a.mn.z, p.z)),
                v3(fmaxf(a.mx.x, p.x), fmaxf(a.mx.y, p.y), fmaxf(a.mx.z, p.z))};
}
inline V3 aabb_size(const AABB& a) { return a.mx - a.mn; }
inline V3 aabb_center(const AABB& a) { return a.mn + (aabb_size(a) / 2.0f); }
inline bool aabb_is_empty(const AABB& a) { return a.mn.x > a.mx.x || a.mn.y > a.mx.y || a.mn.z > a.mx.z; }
inline float aabb_surface_area(const AABB& a) {
    V3 s = aabb_size(a);
    return 2.0f * (s.x * s.y + s.x * s.z + s.y * s.z);
}
inline int aabb_largest_axis(const AABB& a) {
    V3 s = aabb_size(a);
    if (s.x > s.y && s.x > s.z) return 0;
    if (s.y > s.z) return 1;
    return 2;
}
// B/ray.rs:81-112 custom min/max (x86 minss/maxss semantics)
inline float rmin(float x, float y) { return x < y ? x : y; }
inline float rmax(float x, float y) { return x > y ? x : y; }
inline bool intersects_aabb(const Ray& r, const AABB& b) {  // B/ray.rs:174-194
    const V3* bb[2] = {&b.mn, &b.mx};
    float ray_min = (bb[r.sign_x]->x - r.origin.x) * r.inv_direction.x;
    float ray_max = (bb[1 - r.sign_x]->x - r.origin.x) * r.inv_direction.x;
    float y_min = (bb[r.sign_y]->y - r.origin.y) * r.inv_direction.y;
    float y_max = (bb[1 - r.sign_y]->y - r.origin.y) * r.inv_direction.y;
    ray_min = rmax(ray_min, y_min);
    ray_max = rmin(ray_max, y_max);
    float z_min = (bb[r.sign_z]->z - r.origin.z) * r.inv_direction.z;
    float z_max = (bb[1 - r.sign_z]->z - r.origin.z) * r.inv_direction.z;
    ray_min = rmax(ray_min, z_min);
    ray_max = rmin(ray_max, z_max);
    return rmax(ray_min, 0.0f) <= ray_max;
}

// ---------------------------------------------------------------- bvh::bvh::BVH
struct BVHNode {
    bool leaf;
    uint32_t shape_index;
    uint32_t child_l, child_r;
    AABB child_l_aabb, child_r_aabb;
};
const float BVH_EPSILON = 0.00001f;  // B/lib.rs:80

struct BVH {
    std::vector<BVHNode> nodes;
};

uint32_t bvh_build_node(const std::vector<AABB>& shapes, const std::vector<uint32_t>& indices,
                        std::vector<BVHNode>& nodes) {
    // B/bvh/bvh_impl.rs:247-251
    AABB aabb_bounds = aabb_empty(), centroid_bounds = aabb_empty();
    for (uint32_t idx : indices) {
        const AABB& sa = shapes[idx];
        V3 center = aabb_center(sa);
        aabb_bounds = aabb_join(aabb_bounds, sa);
        centroid_bounds = aabb_grow(centroid_bounds, center);
    }
    if (indices.size() == 1) {  // :254-265
        uint32_t node_index = (uint32_t)nodes.size();
        BVHNode n{};
        n.leaf = true;
        n.shape_index = indices[0];
        nodes.push_back(n);
        return node_index;
    }
    uint32_t node_index = (uint32_t)nodes.size();
    nodes.push_back(BVHNode{});  // dummy (:269-270)
    int split_axis = aabb_largest_axis(centroid_bounds);
    float split_axis_size = axis_of(centroid_bounds.mx, split_axis) - axis_of(centroid_bounds.mn, split_axis);
    uint32_t cl, cr;
    AABB cl_aabb, cr_aabb;
    if (split_axis_size < BVH_EPSILON) {  // :277-291
        size_t half = indices.size() / 2;
        std::vector<uint32_t> li(indices.begin(), indices.begin() + half);
        std::vector<uint32_t> ri(indices.begin() + half, indices.end());
        cl_aabb = aabb_empty();
        for (uint32_t i : li) cl_aabb = aabb_join(cl_aabb, shapes[i]);
        cr_aabb = aabb_empty();
        for (uint32_t i : ri) cr_aabb = aabb_join(cr_aabb, shapes[i]);
        cl = bvh_build_node(shapes, li, nodes);
        cr = bvh_build_node(shapes, ri, nodes);
    } else {  // :293-349
        const int NUM_BUCKETS = 6;
        struct Bucket {
            size_t size;
            AABB aabb;
        } buckets[NUM_BUCKETS];
        std::vector<uint32_t> assign[NUM_BUCKETS];
        for (auto& b : buckets) {
            b.size = 0;
            b.aabb = aabb_empty();
        }
        for (uint32_t idx : indices) {
            const AABB& sa = shapes[idx];
            V3 c = aabb_center(sa);
            float rel = (axis_of(c, split_axis) - axis_of(centroid_bounds.mn, split_axis)) / split_axis_size;
            float fb = rel * ((float)NUM_BUCKETS - 0.01f);
            // Rust `as usize`: truncate toward zero, saturate, NaN -> 0
            size_t bn;
            if (!(fb == fb) || fb <= 0.0f)
                bn = 0;
            else if (fb >= 1.8e19f)
                bn = (size_t)-1;
            else
                bn = (size_t)fb;
            if (bn >= (size_t)NUM_BUCKETS) bn = NUM_BUCKETS - 1;  // Rust would panic (index OOB); unreachable for finite input
            buckets[bn].size += 1;
            buckets[bn].aabb = aabb_join(buckets[bn].aabb, sa);
            assign[bn].push_back(idx);
        }
        int min_bucket = 0;
        float min_cost = INF;
        cl_aabb = aabb_empty();
        cr_aabb = aabb_empty();
        for (int i = 0; i < NUM_BUCKETS - 1; i++) {
            Bucket l{0, aabb_empty()}, r{0, aabb_empty()};
            for (int k = 0; k <= i; k++) {
                l.size += buckets[k].size;
                l.aabb = aabb_join(l.aabb, buckets[k].aabb);
            }
            for (int k = i + 1; k < NUM_BUCKETS; k++) {
                r.size += buckets[k].size;
                r.aabb = aabb_join(r.aabb, buckets[k].aabb);
            }
            float cost = ((float)l.size * aabb_surface_area(l.aabb) + (float)r.size * aabb_surface_area(r.aabb)) /
                         aabb_surface_area(aabb_bounds);
            if (cost < min_cost) {
                min_bucket = i;
                min_cost = cost;
                cl_aabb = l.aabb;
                cr_aabb = r.aabb;
            }
        }
        std::vector<uint32_t> li, ri;
        for (int k = 0; k <= min_bucket; k++) li.insert(li.end(), assign[k].begin(), assign[k].end());
        for (int k = min_bucket + 1; k < NUM_BUCKETS; k++) ri.insert(ri.end(), assign[k].begin(), assign[k].end());
        if (li.empty() || ri.empty()) {
            // the reference would recurse without bound / assert here; cannot happen for finite
            // centroids because bucket 0 and bucket 5 are both populated.  Split in half instead.
            size_t half = indices.size() / 2;
            li.assign(indices.begin(), indices.begin() + half);
            ri.assign(indices.begin() + half, indices.end());
            cl_aabb = aabb_empty();
            for (uint32_t i : li) cl_aabb = aabb_join(cl_aabb, shapes[i]);
            cr_aabb = aabb_empty();
            for (uint32_t i : ri) cr_aabb = aabb_join(cr_aabb, shapes[i]);
        }
        cl = bvh_build_node(shapes, li, nodes);
        cr = bvh_build_node(shapes, ri, nodes);
    }
    BVHNode& n = nodes[node_index];
    n.leaf = false;
    n.child_l = cl;
    n.child_r = cr;
    n.child_l_aabb = cl_aabb;
    n.child_r_aabb = cr_aabb;
    return node_index;
}

BVH bvh_build(const std::vector<AABB>& shapes) {  // B/bvh/bvh_impl.rs:421-427
    BVH b;
    if (shapes.empty()) return b;  // reference recurses without bound; policy: empty BVH
    std::vector<uint32_t> indices(shapes.size());
    for (size_t i = 0; i < shapes.size(); i++) indices[i] = (uint32_t)i;
    b.nodes.reserve(shapes.size() * 2);
    bvh_build_node(shapes, indices, b.nodes);
    return b;
}

void bvh_traverse_recursive(const std::vector<BVHNode>& nodes, uint32_t ni, const Ray& ray,
                            std::vector<uint32_t>& out) {  // :373-398
    const BVHNode& n = nodes[ni];
    if (n.leaf) {
        out.push_back(n.shape_index);
        return;
    }
    if (intersects_aabb(ray, n.child_l_aabb)) bvh_traverse_recursive(nodes, n.child_l, ray, out);
    if (intersects_aabb(ray, n.child_r_aabb)) bvh_traverse_recursive(nodes, n.child_r, ray, out);
}

// ---------------------------------------------------------------- Color (S/color.rs)
struct Color {
    float r, g, b;
};
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color operator*(Color a, float s) { return {a.r * s, a.g * s, a.b * s}; }
inline Color operator*(float s, Color a) { return a * s; }  // S/color.rs:88-94: rhs * self
inline Color blend(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
// Rust `as u8` from f32: truncate toward zero, saturate to [0,255], NaN -> 0
inline uint8_t f32_as_u8(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
inline void color_as_slice(Color c, uint8_t* p) {  // S/color.rs:13-19
    p[0] = f32_as_u8(c.r * 255.999f);
    p[1] = f32_as_u8(c.g * 255.999f);
    p[2] = f32_as_u8(c.b * 255.999f);
}

// ---------------------------------------------------------------- scene
struct Scene {
    std::vector<rt_sphere> spheres;
    std::vector<rt_triangle> tris;
    // `world: Vec<Object>` (S/lib.rs:11) as the ABI carries it: two typed arrays plus, for every position of the list, the
    // object that stands there (objects are numbered spheres first, then triangles).  Identity when no world_index came.
    std::vector<uint32_t> order;
    float t_min, t_max;
    BVH bvh;
    bool use_bvh;
    size_t count() const { return spheres.size() + tris.size(); }
};
inline V3 sph_center(const rt_sphere& s) { return v3(s.cx, s.cy, s.cz); }
inline V3 arr3(const float* p) { return v3(p[0], p[1], p[2]); }

inline Roots sphere_get_roots(const rt_sphere& s, const Ray& ray) {  // S/shapes/sphere.rs:42-47
    float a = 1.0f;
    V3 oc = ray.origin - sph_center(s);
    float b = dot(2.0f * ray.direction, oc);
    float len = length(oc);
    float c = len * len - s.radius * s.radius;  // .length().powi(2) - radius.powi(2)
    return find_roots_quadratic(a, b, c);
}
inline Roots triangle_get_roots(const rt_triangle& t, const Ray& ray) {  // S/shapes/mesh.rs:109-161
    const float EPSILON = 0.00001f;
    V3 A = arr3(t.a), Bv = arr3(t.b), C = arr3(t.c);
    V3 a_to_b = Bv - A;
    V3 a_to_c = C - A;
    V3 u_vec = cross(ray.direction, a_to_c);
    float det = dot(a_to_b, u_vec);
    if (det < EPSILON && det > -EPSILON) return Roots{0, {0.f, 0.f}};
    float inv_det = 1.0f / det;
    V3 a_to_origin = ray.origin - A;
    float u = dot(a_to_origin, u_vec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return Roots{0, {0.f, 0.f}};  // !(0.0..=1.0).contains(&u)
    V3 v_vec = cross(a_to_origin, a_to_b);
    float v = dot(ray.direction, v_vec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return Roots{0, {0.f, 0.f}};
    float dist = dot(a_to_c, v_vec) * inv_det;
    if (dist > EPSILON) return Roots{1, {dist, 0.f}};
    return Roots{0, {0.f, 0.f}};
}
// S/shapes/mod.rs:106-129; returns false for None
inline bool select_t(const Scene& sc, const Roots& r, float* t) {
    auto in_range = [&](float x) { return x >= sc.t_min && x < sc.t_max; };  // (T_MIN..T_MAX).contains
    if (r.n == 0) return false;
    if (r.n == 1) {
        if (in_range(r.x[0])) {
            *t = r.x[0];
            return true;
        }
        return false;
    }
    bool xin = in_range(r.x[0]), yin = in_range(r.x[1]);
    if (xin && yin) {
        *t = r.x[0] < r.x[1] ? r.x[0] : r.x[1];
        return true;
    }
    if (xin) {
        *t = r.x[0];
        return true;
    }
    if (yin) {
        *t = r.x[1];
        return true;
    }
    return false;
}
inline bool object_hit_point(const Scene& sc, uint32_t idx, const Ray& ray, V3* p) {
    Roots r = idx < sc.spheres.size() ? sphere_get_roots(sc.spheres[idx], ray)
                                      : triangle_get_roots(sc.tris[idx - sc.spheres.size()], ray);
    float t;
    if (!select_t(sc, r, &t)) return false;
    *p = ray_at(ray, t);
    return true;
}

struct IntersectionTable {  // S/shapes/mod.rs:15-21
    V3 point, normal;
    Color albedo;
    float roughness, emission;
    uint32_t index;
};

// WorldRefList::intersect (S/shapes/mod.rs:158-191) over `cands` (or all objects in index
// order when cands == nullptr).  min_by keeps the FIRST minimum; partial_cmp(None) -> Less.
bool intersect(const Scene& sc, const Ray& ray, const std::vector<uint32_t>* cands, IntersectionTable* out) {
    bool have = false;
    uint32_t best = 0;
    V3 best_p{};
    float best_d = 0.f;
    size_t n = cands ? cands->size() : sc.count();
    for (size_t k = 0; k < n; k++) {
        // both lists hold POSITIONS in `world` (BVH::build numbers the shapes by position, bvh_impl.rs:421-427; the
        // linear back-end walks the list front to back)
        uint32_t idx = sc.order[cands ? (*cands)[k] : (uint32_t)k];
        V3 p;
        if (!object_hit_point(sc, idx, ray, &p)) continue;
        float d = length(p - ray.origin);
        if (!have) {
            have = true;
            best = idx;
            best_p = p;
            best_d = d;
        } else {
            // cmp::min_by(v1 = running, v2 = new): Greater -> v2, else v1.
            // partial_cmp gives Greater iff best_d > d; unordered -> Less -> keep running.
            if (best_d > d) {
                best = idx;
                best_p = p;
                best_d = d;
            }
        }
    }
    if (!have) return false;
    out->index = best;
    out->point = best_p;
    if (best < sc.spheres.size()) {
        const rt_sphere& s = sc.spheres[best];
        out->emission = s.emission;
        out->normal = normalize_or_zero(best_p - sph_center(s));  // sphere.rs:49-51
        out->albedo = Color{s.albedo_r, s.albedo_g, s.albedo_b};
        out->roughness = s.roughness;
    } else {
        const rt_triangle& t = sc.tris[best - sc.spheres.size()];
        out->emission = t.emission;
        out->normal = normalize_or_zero(cross(arr3(t.a) - arr3(t.b), arr3(t.a) - arr3(t.c)));  // mesh.rs:163-165
        out->albedo = Color{t.albedo_r, t.albedo_g, t.albedo_b};
        out->roughness = t.roughness;
    }
    return true;
}

inline AABB sphere_aabb(const rt_sphere& s) {  // sphere.rs:65-72
    V3 h = v3(s.radius, s.radius, s.radius);
    return AABB{sph_center(s) - h, sph_center(s) + h};
}
// std::cmp::min_by(v1, v2, partial_cmp.unwrap_or(Equal)): v1 unless v1 > v2
inline float pmin(float v1, float v2) { return v1 > v2 ? v2 : v1; }
// std::cmp::max_by(v1, v2, ...): v2 unless v1 > v2
inline float pmax(float v1, float v2) { return v1 > v2 ? v1 : v2; }
inline AABB triangle_aabb(const rt_triangle& t) {  // mesh.rs:46-96
    V3 a = arr3(t.a), b = arr3(t.b), c = arr3(t.c);
    V3 mn = v3(pmin(pmin(a.x, c.x), b.x), pmin(pmin(a.y, c.y), b.y), pmin(pmin(a.z, c.z), b.z));
    V3 mx = v3(pmax(pmax(a.x, c.x), b.x), pmax(pmax(a.y, c.y), b.y), pmax(pmax(a.z, c.z), b.z));
    return AABB{mn, mx};
}

// ---------------------------------------------------------------- Camera (S/camera.rs)
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical;
    float aspect_ratio, image_height, aperture, focus_distance;
};
Camera camera_new(V3 origin, float aspect_ratio, float aperture, float focus_distance, float field_of_view,
                  float focal_length, float image_height) {  // :19-47
    float vh = 2.0f * std::tan(field_of_view / 2.0f);
    float vw = aspect_ratio * vh;
    Camera c;
    c.origin = origin;
    c.horizontal = v3(vw, 0.f, 0.f);
    c.vertical = v3(0.f, vh, 0.f);
    c.aspect_ratio = aspect_ratio;
    c.image_height = image_height;
    c.aperture = aperture;
    c.focus_distance = focus_distance;
    c.lower_left_corner = origin - c.horizontal / 2.0f - c.vertical / 2.0f - v3(0.f, 0.f, focal_length);
    return c;
}
Ray camera_get_ray(const Camera& c, uint32_t x, uint32_t y, Rng& rng) {  // :109-129
    float lens_radius = c.aperture / 2.0f;
    float a, b;
    unit_disc(rng, &a, &b);
    V3 offset = v3(a * lens_radius, b * lens_radius, 0.0f);
    float u = ((float)x + gen_range_01(rng)) / (c.aspect_ratio * c.image_height - 1.0f);
    float v = ((float)y + gen_range_01(rng)) / (c.image_height - 1.0f);
    Ray fr = ray_new(c.origin,
                     normalize_or_zero(c.lower_left_corner + u * c.horizontal + v * c.vertical - c.origin));
    V3 focal_point = ray_at(fr, c.focus_distance);
    V3 final_ray_origin = c.origin + offset;
    return ray_new(final_ray_origin, normalize_or_zero(focal_point - final_ray_origin));
}

// ---------------------------------------------------------------- ray_color (S/main.rs:108-146)
struct Ctx {
    const Scene* sc;
    uint64_t segments;
};
inline Color sky(V3 direction) {  // :135-144
    float t = normalize_or_zero(direction).y * 0.5f + 1.0f;
    return t * Color{1.0f, 1.0f, 1.0f} + (1.0f - t) * Color{0.3f, 0.3f, 0.8f};
}
Color ray_color(Ctx& cx, const Ray& ray, uint32_t depth, Rng& rng) {
    if (depth == 0) return Color{0.f, 0.f, 0.f};
    cx.segments++;
    const Scene& sc = *cx.sc;
    IntersectionTable tb;
    bool hit;
    if (sc.use_bvh) {
        // the reference allocates a fresh Vec per call (bvh_impl.rs:436); a per-thread pool indexed by the
        // recursion depth gives the same candidate list without the allocator traffic
        static thread_local std::vector<std::vector<uint32_t>> pool;
        if (pool.size() <= depth) pool.resize(depth + 1);
        std::vector<uint32_t>& cands = pool[depth];
        cands.clear();
        if (!sc.bvh.nodes.empty()) bvh_traverse_recursive(sc.bvh.nodes, 0, ray, cands);
        hit = intersect(sc, ray, &cands, &tb);
    } else {
        hit = intersect(sc, ray, nullptr, &tb);
    }
    if (hit) {
        if (tb.emission > 0.0f) return tb.emission * tb.albedo;
        V3 diffuse_dir = unit_sphere(rng) + tb.normal;
        V3 glossy_dir = ray.direction - 2.0f * dot(ray.direction, tb.normal) * tb.normal;
        V3 scatter_direction = diffuse_dir + tb.roughness * (glossy_dir - diffuse_dir);
        V3 nd;
        if (!try_normalize(scatter_direction, &nd)) nd = tb.normal;
        Ray next = ray_new(tb.point, nd);
        return blend(tb.albedo, ray_color(cx, next, depth - 1, rng));
    }
    return sky(ray.direction);
}

// ---------------------------------------------------------------- tile loop (S/main.rs:37-83)
struct Job {
    Scene sc;
    Camera cam;
    rt_tile_request req;
    uint32_t hs;
    int rng_mode;   // 0: one stream per (pixel, sample) — normative; 1: per pixel; 2: per row (the reference's structure)
};

// pixels [x_begin, x_end) of strip row yl
void render_span(const Job& job, uint32_t yl, uint32_t x_begin, uint32_t x_end, uint8_t* out_rgb, float* out_f32,
                 uint64_t* segs) {
    const rt_tile_request& rq = job.req;
    Ctx cx{&job.sc, 0};
    const uint32_t W = rq.width, H = rq.height;
    uint32_t yg = job.hs * rq.division_no + yl;  // :66-68
    Rng rng = seed_from_u64(row_seed(rq.seed, yg));   // :69 (rng_mode 2: the caller hands over whole rows)
    for (uint32_t x = x_begin; x < x_end; x++) {
        uint32_t yc = H - yg - 1;  // :71
        if (job.rng_mode == 1) rng = seed_from_u64(pixel_seed(rq.seed, (uint64_t)yg * W + x));
        Color pix{0.f, 0.f, 0.f};
        for (uint32_t s = 0; s < rq.spp; s++) {
            if (job.rng_mode == 0) rng = seed_from_u64(sample_seed(rq.seed, (uint64_t)yg * W + x, rq.spp, s));
            Ray r = camera_get_ray(job.cam, x, yc, rng);
            pix = pix + ray_color(cx, r, rq.max_bounces + 1, rng);
        }
        float n = (float)rq.spp;
        pix.r = std::sqrt(pix.r / n);
        pix.g = std::sqrt(pix.g / n);
        pix.b = std::sqrt(pix.b / n);
        size_t o = ((size_t)yl * W + x) * 3;
        color_as_slice(pix, out_rgb + o);
        if (out_f32) {
            out_f32[o] = pix.r;
            out_f32[o + 1] = pix.g;
            out_f32[o + 2] = pix.b;
        }
    }
    *segs = cx.segments;
}

Scene make_scene(const rt_tile_request* rq, const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt,
                 int backend, const uint32_t* world_index = nullptr) {
    Scene sc;
    sc.spheres.assign(sp, sp + ns);
    sc.tris.assign(tr, tr + nt);
    sc.order.resize((size_t)ns + nt);
    for (uint32_t i = 0; i < ns + nt; i++) sc.order[world_index ? world_index[i] : i] = i;   // (a permutation: checked by the caller)
    sc.t_min = rq->t_min;
    sc.t_max = rq->t_max;
    sc.use_bvh = backend == 1;
    return sc;
}
void build_scene_bvh(Scene& sc) {
    std::vector<AABB> boxes;
    boxes.reserve(sc.count());
    for (uint32_t obj : sc.order)               // shapes in `world` order (S/main.rs:60: BVH::build(&mut req.world))
        boxes.push_back(obj < sc.spheres.size() ? sphere_aabb(sc.spheres[obj]) : triangle_aabb(sc.tris[obj - sc.spheres.size()]));
    sc.bvh = bvh_build(boxes);
}
Camera make_camera(const rt_tile_request* rq) {  // S/main.rs:42-50
    return camera_new(v3(0.f, 0.f, 0.f), (float)rq->width / (float)rq->height, rq->aperture, rq->focus_distance,
                      rq->fov, rq->focal_length, (float)rq->height);
}

}  // namespace

// =====================================================================================
// C interface (ctypes) — test/baseline use only
// =====================================================================================
extern "C" {

// Render one strip.  backend: 0 linear, 1 bvh.  nthreads <= 0: hardware_concurrency.
// Rows are distributed dynamically over threads (rayon-like); results do not depend on
// nthreads because every sample (rng_mode 1: pixel, 2: row) owns its RNG stream.
// rng_mode: 0 = the normative stream per (pixel, sample); 1, 2: see the file header (statistics only).
// out_f32 may be NULL.  *bvh_build_ms (may be NULL) receives the BVH build time.
// world_index (may be NULL): include/rt_tile.h "the world's order".
__attribute__((visibility("default"))) int rt_oracle_render(const rt_tile_request* rq, const rt_sphere* sp,
                                                            uint32_t ns, const rt_triangle* tr, uint32_t nt,
                                                            int backend, int nthreads, uint8_t* out_rgb,
                                                            float* out_f32, uint64_t* ray_segments,
                                                            double* render_ms, double* bvh_build_ms,
                                                            const uint32_t* world_index, int rng_mode) {
    if (!rq || !out_rgb || rq->width == 0 || rq->height == 0 || rq->divisions == 0 ||
        rq->division_no >= rq->divisions || rq->spp == 0 || rng_mode < 0 || rng_mode > 2)
        return -1;
    if (world_index) {
        std::vector<char> seen((size_t)ns + nt, 0);
        for (uint32_t i = 0; i < ns + nt; i++) {
            if (world_index[i] >= ns + nt || seen[world_index[i]]) return -2;
            seen[world_index[i]] = 1;
        }
    }
    Job job;
    job.sc = make_scene(rq, sp, ns, tr, nt, backend, world_index);
    auto t0 = std::chrono::steady_clock::now();
    if (job.sc.use_bvh) build_scene_bvh(job.sc);
    auto t1 = std::chrono::steady_clock::now();
    if (bvh_build_ms) *bvh_build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    job.cam = make_camera(rq);
    job.req = *rq;
    job.hs = rq->height / rq->divisions;
    job.rng_mode = rng_mode;
    int nt_ = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    if (nt_ < 1) nt_ = 1;
    // work unit = 64 consecutive pixels of a row (the reference's rayon unit is a whole row,
    // main.rs:62-65; a finer unit only changes scheduling, every pixel owns its RNG stream)
    // (rng_mode 2: the stream runs along the row, so the unit is the reference's: a whole row)
    const uint32_t SPAN = rng_mode == 2 ? rq->width : 64;
    const uint32_t spans_per_row = (rq->width + SPAN - 1) / SPAN;
    const uint64_t n_units = (uint64_t)spans_per_row * job.hs;
    if ((uint64_t)nt_ > n_units) nt_ = n_units ? (int)n_units : 1;
    std::atomic<uint64_t> next_unit{0};
    std::vector<uint64_t> segs(nt_, 0);
    auto workfn = [&](int tid) {
        uint64_t total = 0;
        for (;;) {
            uint64_t u = next_unit.fetch_add(1);
            if (u >= n_units) break;
            uint32_t row = (uint32_t)(u / spans_per_row), sx = (uint32_t)(u % spans_per_row) * SPAN;
            uint32_t ex = sx + SPAN < rq->width ? sx + SPAN : rq->width;
            uint64_t s = 0;
            render_span(job, row, sx, ex, out_rgb, out_f32, &s);
            total += s;
        }
        segs[tid] = total;
    };
    auto t2 = std::chrono::steady_clock::now();
    if (nt_ == 1) {
        workfn(0);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt_; i++) th.emplace_back(workfn, i);
        for (auto& t : th) t.join();
    }
    auto t3 = std::chrono::steady_clock::now();
    if (render_ms) *render_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    uint64_t tot = 0;
    for (auto s : segs) tot += s;
    if (ray_segments) *ray_segments = tot;
    return 0;
}

__attribute__((visibility("default"))) int rt_oracle_hardware_threads(void) {
    return (int)std::thread::hardware_concurrency();
}

// ---- KAT probes ---------------------------------------------------------------------
__attribute__((visibility("default"))) void rt_oracle_xoshiro_from_state(const uint64_t* state4, uint64_t* out,
                                                                         int n) {
    Rng r;
    for (int i = 0; i < 4; i++) r.s[i] = state4[i];
    for (int i = 0; i < n; i++) out[i] = next_u64(r);
}
__attribute__((visibility("default"))) void rt_oracle_seed_from_u64(uint64_t seed, uint64_t* state4) {
    Rng r = seed_from_u64(seed);
    for (int i = 0; i < 4; i++) state4[i] = r.s[i];
}
__attribute__((visibility("default"))) uint64_t rt_oracle_pixel_seed(uint64_t job_seed, uint64_t pix) {
    return pixel_seed(job_seed, pix);
}
__attribute__((visibility("default"))) uint64_t rt_oracle_sample_seed(uint64_t job_seed, uint64_t pix, uint64_t spp, uint64_t s) {
    return sample_seed(job_seed, pix, spp, s);
}
// draws: kind 0 = gen_range(0..1), 1 = Uniform(-1,1), 2 = UnitDisc (2 floats), 3 = UnitSphere (3 floats)
__attribute__((visibility("default"))) void rt_oracle_draw(uint64_t* state4, int kind, float* out) {
    Rng r;
    for (int i = 0; i < 4; i++) r.s[i] = state4[i];
    if (kind == 0)
        out[0] = gen_range_01(r);
    else if (kind == 1)
        out[0] = uniform_m1_1(r);
    else if (kind == 2)
        unit_disc(r, &out[0], &out[1]);
    else {
        V3 v = unit_sphere(r);
        out[0] = v.x;
        out[1] = v.y;
        out[2] = v.z;
    }
    for (int i = 0; i < 4; i++) state4[i] = r.s[i];
}
// returns number of roots, roots ascending in out[0..1]
__attribute__((visibility("default"))) int rt_oracle_find_roots_quadratic(float a2, float a1, float a0, float* out) {
    Roots r = find_roots_quadratic(a2, a1, a0);
    out[0] = r.x[0];
    out[1] = r.x[1];
    return r.n;
}
// Ray::new(origin, dir) then Sphere::get_roots; returns n roots
__attribute__((visibility("default"))) int rt_oracle_sphere_roots(const rt_sphere* s, const float* origin,
                                                                  const float* dir, float* out) {
    Ray ray = ray_new(arr3(origin), arr3(dir));
    Roots r = sphere_get_roots(*s, ray);
    out[0] = r.x[0];
    out[1] = r.x[1];
    return r.n;
}
__attribute__((visibility("default"))) int rt_oracle_triangle_roots(const rt_triangle* t, const float* origin,
                                                                    const float* dir, float* out) {
    Ray ray = ray_new(arr3(origin), arr3(dir));
    Roots r = triangle_get_roots(*t, ray);
    out[0] = r.x[0];
    out[1] = r.x[1];
    return r.n;
}
// closest hit over the world (linear or bvh).  out9 = point(3) normal(3) albedo... no:
// out = [px,py,pz, nx,ny,nz, ar,ag,ab, roughness, emission]; *index = object index.
__attribute__((visibility("default"))) int rt_oracle_intersect(const rt_sphere* sp, uint32_t ns,
                                                               const rt_triangle* tr, uint32_t nt, float t_min,
                                                               float t_max, int backend, const float* origin,
                                                               const float* dir, float* out, uint32_t* index) {
    rt_tile_request rq{};
    rq.t_min = t_min;
    rq.t_max = t_max;
    Scene sc = make_scene(&rq, sp, ns, tr, nt, backend);
    if (sc.use_bvh) build_scene_bvh(sc);
    Ray ray = ray_new(arr3(origin), arr3(dir));
    IntersectionTable tb;
    bool hit;
    if (sc.use_bvh) {
        std::vector<uint32_t> cands;
        if (!sc.bvh.nodes.empty()) bvh_traverse_recursive(sc.bvh.nodes, 0, ray, cands);
        hit = intersect(sc, ray, &cands, &tb);
    } else
        hit = intersect(sc, ray, nullptr, &tb);
    if (!hit) return 0;
    float v[11] = {tb.point.x,  tb.point.y,  tb.point.z,  tb.normal.x,  tb.normal.y, tb.normal.z,
                   tb.albedo.r, tb.albedo.g, tb.albedo.b, tb.roughness, tb.emission};
    std::memcpy(out, v, sizeof v);
    *index = tb.index;
    return 1;
}
// ray_color for one ray with an explicit RNG state (advanced in place)
__attribute__((visibility("default"))) void rt_oracle_ray_color(const rt_sphere* sp, uint32_t ns,
                                                                const rt_triangle* tr, uint32_t nt, float t_min,
                                                                float t_max, const float* origin, const float* dir,
                                                                uint32_t depth, uint64_t* state4, float* out_rgb,
                                                                uint64_t* segments) {
    rt_tile_request rq{};
    rq.t_min = t_min;
    rq.t_max = t_max;
    Scene sc = make_scene(&rq, sp, ns, tr, nt, 0);
    Ctx cx{&sc, 0};
    Rng r;
    for (int i = 0; i < 4; i++) r.s[i] = state4[i];
    Ray ray = ray_new(arr3(origin), arr3(dir));
    Color c = ray_color(cx, ray, depth, r);
    out_rgb[0] = c.r;
    out_rgb[1] = c.g;
    out_rgb[2] = c.b;
    for (int i = 0; i < 4; i++) state4[i] = r.s[i];
    if (segments) *segments = cx.segments;
}
__attribute__((visibility("default"))) void rt_oracle_sky(const float* dir, float* out_rgb) {
    Color c = sky(arr3(dir));
    out_rgb[0] = c.r;
    out_rgb[1] = c.g;
    out_rgb[2] = c.b;
}
__attribute__((visibility("default"))) void rt_oracle_quantise(const float* rgb, uint8_t* out) {
    color_as_slice(Color{rgb[0], rgb[1], rgb[2]}, out);
}
// camera ray for pixel (x, y_cam) with explicit RNG state; out = origin(3) dir(3)
__attribute__((visibility("default"))) void rt_oracle_camera_ray(const rt_tile_request* rq, uint32_t x,
                                                                 uint32_t y_cam, uint64_t* state4, float* out) {
    Camera cam = make_camera(rq);
    Rng r;
    for (int i = 0; i < 4; i++) r.s[i] = state4[i];
    Ray ray = camera_get_ray(cam, x, y_cam, r);
    out[0] = ray.origin.x;
    out[1] = ray.origin.y;
    out[2] = ray.origin.z;
    out[3] = ray.direction.x;
    out[4] = ray.direction.y;
    out[5] = ray.direction.z;
    for (int i = 0; i < 4; i++) state4[i] = r.s[i];
}
// camera constants: out = llc(3) horizontal(3) vertical(3)
__attribute__((visibility("default"))) void rt_oracle_camera_consts(const rt_tile_request* rq, float* out) {
    Camera c = make_camera(rq);
    float v[9] = {c.lower_left_corner.x, c.lower_left_corner.y, c.lower_left_corner.z, c.horizontal.x, c.horizontal.y,
                  c.horizontal.z,        c.vertical.x,          c.vertical.y,          c.vertical.z};
    std::memcpy(out, v, sizeof v);
}
// AABB helpers on raw boxes (6 floats: min xyz, max xyz), for the reference's own doc-test vectors (B/aabb.rs:453,474,
// 520,565; B/axis.rs:20,33).  out[0..2] size, [3..5] center, [6] surface_area, [7] largest_axis, [8] is_empty,
// [9..14] join(a, b), [15..20] grow(a, point)
__attribute__((visibility("default"))) void rt_oracle_aabb_kat(const float* a6, const float* b6, const float* pt3,
                                                               float* out) {
    const AABB a{v3(a6[0], a6[1], a6[2]), v3(a6[3], a6[4], a6[5])};
    const AABB b{v3(b6[0], b6[1], b6[2]), v3(b6[3], b6[4], b6[5])};
    const V3 sz = aabb_size(a), c = aabb_center(a);
    out[0] = sz.x; out[1] = sz.y; out[2] = sz.z;
    out[3] = c.x; out[4] = c.y; out[5] = c.z;
    out[6] = aabb_surface_area(a);
    out[7] = (float)aabb_largest_axis(a);
    out[8] = aabb_is_empty(a) ? 1.0f : 0.0f;
    const AABB j = aabb_join(a, b), g = aabb_grow(a, arr3(pt3));
    const float jj[12] = {j.mn.x, j.mn.y, j.mn.z, j.mx.x, j.mx.y, j.mx.z, g.mn.x, g.mn.y, g.mn.z, g.mx.x, g.mx.y, g.mx.z};
    std::memcpy(out + 9, jj, sizeof jj);
}
// AABB::empty() (B/aabb.rs:124-129) as 6 floats
__attribute__((visibility("default"))) void rt_oracle_aabb_empty(float* out6) {
    const AABB e = aabb_empty();
    const float v[6] = {e.mn.x, e.mn.y, e.mn.z, e.mx.x, e.mx.y, e.mx.z};
    std::memcpy(out6, v, sizeof v);
}
// component of a vector by Axis (B/axis.rs:36-46: X = 0, Y = 1, Z = 2)
__attribute__((visibility("default"))) float rt_oracle_axis_get(const float* v3p, int axis) { return axis_of(arr3(v3p), axis); }
// Ray::new(origin, direction).intersects_aabb(box) (B/ray.rs:133-143, 174-194)
__attribute__((visibility("default"))) int rt_oracle_ray_intersects_aabb(const float* origin, const float* dir,
                                                                         const float* box6) {
    const Ray r = ray_new(arr3(origin), arr3(dir));
    return intersects_aabb(r, AABB{v3(box6[0], box6[1], box6[2]), v3(box6[3], box6[4], box6[5])}) ? 1 : 0;
}
// The same with the ray turned round the way the crate's own property tests do it (B/ray.rs:420-423):
// `ray.direction = -ray.direction; ray.inv_direction = -ray.inv_direction;` — the cached signs are NOT refreshed.
__attribute__((visibility("default"))) int rt_oracle_ray_intersects_aabb_flipped(const float* origin, const float* dir,
                                                                                 const float* box6) {
    Ray r = ray_new(arr3(origin), arr3(dir));
    r.direction = v3(-r.direction.x, -r.direction.y, -r.direction.z);
    r.inv_direction = v3(-r.inv_direction.x, -r.inv_direction.y, -r.inv_direction.z);
    return intersects_aabb(r, AABB{v3(box6[0], box6[1], box6[2]), v3(box6[3], box6[4], box6[5])}) ? 1 : 0;
}
// BVH over raw AABBs (n boxes, 6 floats each: min xyz, max xyz); traverse one ray; returns
// number of candidate shape indices written to out_idx (DFS leaf order), capacity cap.
__attribute__((visibility("default"))) int rt_oracle_bvh_traverse_boxes(const float* boxes, uint32_t n,
                                                                        const float* origin, const float* dir,
                                                                        uint32_t* out_idx, uint32_t cap,
                                                                        uint32_t* n_nodes) {
    std::vector<AABB> bs(n);
    for (uint32_t i = 0; i < n; i++)
        bs[i] = AABB{v3(boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]),
                     v3(boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5])};
    BVH b = bvh_build(bs);
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    Ray ray = ray_new(arr3(origin), arr3(dir));
    std::vector<uint32_t> c;
    if (!b.nodes.empty()) bvh_traverse_recursive(b.nodes, 0, ray, c);
    uint32_t m = c.size() < cap ? (uint32_t)c.size() : cap;
    for (uint32_t i = 0; i < m; i++) out_idx[i] = c[i];
    return (int)c.size();
}

}  // extern "C"
