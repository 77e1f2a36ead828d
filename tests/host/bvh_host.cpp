// Test harness (CPU only, built by tests/test_host_bvh.py with g++): the PRODUCT's host-side BVH builder
// (ray_tracer_s8_amd/csrc/rt_bvh.h — the tree the HIP kernels walk) behind a few C entry points, so that the CPU
// suite can compare it with the oracle's independent build + recursive traverse, and check the quantised twin.
// The walk below restates bvh::Ray::intersects_aabb (ray.rs:174-194) and BVH::traverse (bvh_impl.rs:373-398)
// over rtbvh::TravNode exactly as the exact-node kernel does.
#include <cmath>
#include <cstdint>
#include <vector>

#include "rt_bvh.h"

namespace {
inline float rmin(float x, float y) { return x < y ? x : y; }      // ray.rs:81-112
inline float rmax(float x, float y) { return x > y ? x : y; }
struct HRay {
    float o[3], inv[3];
    bool s[3];
};
bool hits(const HRay& r, const float* lo, const float* hi) {
    float ray_min = ((r.s[0] ? hi[0] : lo[0]) - r.o[0]) * r.inv[0];
    float ray_max = ((r.s[0] ? lo[0] : hi[0]) - r.o[0]) * r.inv[0];
    const float y_min = ((r.s[1] ? hi[1] : lo[1]) - r.o[1]) * r.inv[1];
    const float y_max = ((r.s[1] ? lo[1] : hi[1]) - r.o[1]) * r.inv[1];
    ray_min = rmax(ray_min, y_min);
    ray_max = rmin(ray_max, y_max);
    const float z_min = ((r.s[2] ? hi[2] : lo[2]) - r.o[2]) * r.inv[2];
    const float z_max = ((r.s[2] ? lo[2] : hi[2]) - r.o[2]) * r.inv[2];
    ray_min = rmax(ray_min, z_min);
    ray_max = rmin(ray_max, z_max);
    return rmax(ray_min, 0.0f) <= ray_max;
}
std::vector<rtbvh::Box> to_boxes(const float* b, uint32_t n) {
    std::vector<rtbvh::Box> v(n);
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            v[i].lo[a] = b[6 * i + a];
            v[i].hi[a] = b[6 * i + 3 + a];
        }
    return v;
}
}  // namespace

extern "C" {

// candidates of one ray in the product tree, depth-first order; returns their number, *n_nodes = nodes built.
// order (may be NULL): order[w] = the primitive at position w of the reference's `world` list (rt_bvh.h build()).
int host_bvh_traverse_ordered(const float* boxes, uint32_t n, const uint32_t* order, const float* origin, const float* dir,
                              uint32_t* out, uint32_t cap, uint32_t* n_nodes, uint32_t* depth) {
    const rtbvh::FlatBVH t = rtbvh::build(to_boxes(boxes, n), order);
    if (n_nodes) *n_nodes = (uint32_t)t.nodes.size();
    if (depth) *depth = t.depth;
    if (n == 0) return 0;
    // Ray::new (ray.rs:133-143): direction.normalize() = divide by the length, glam dot order
    const float len = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    HRay r;
    for (int a = 0; a < 3; a++) {
        const float d = dir[a] / len;
        r.o[a] = origin[a];
        r.inv[a] = 1.0f / d;
        r.s[a] = d < 0.0f;
    }
    std::vector<uint32_t> stack, c;
    uint32_t ref = t.root_ref;
    for (;;) {
        if (ref & rtbvh::LEAF_BIT) {
            c.push_back(ref & ~rtbvh::LEAF_BIT);
            if (stack.empty()) break;
            ref = stack.back();
            stack.pop_back();
            continue;
        }
        const rtbvh::TravNode& nd = t.trav[ref];
        const bool hl = hits(r, nd.l_lo, nd.l_hi), hr = hits(r, nd.r_lo, nd.r_hi);
        if (hl) {
            if (hr) stack.push_back(nd.right);
            ref = nd.left;
        } else if (hr) {
            ref = nd.right;
        } else if (stack.empty()) {
            break;
        } else {
            ref = stack.back();
            stack.pop_back();
        }
    }
    for (uint32_t i = 0; i < c.size() && i < cap; i++) out[i] = c[i];
    return (int)c.size();
}

int host_bvh_traverse(const float* boxes, uint32_t n, const float* origin, const float* dir, uint32_t* out,
                      uint32_t cap, uint32_t* n_nodes, uint32_t* depth) {
    return host_bvh_traverse_ordered(boxes, n, nullptr, origin, dir, out, cap, n_nodes, depth);
}

// Structure checks of the flat tree and its quantised twin.  Returns 0 when every check holds, else a code:
// 1 node count, 2 leaf_of not the depth-first rank, 3 parent links / own boxes, 4 quantised topology, 5 a quantised
// box with less than one grid unit of outward slack, 6 more than four (upper plane: five) units (needlessly loose), 7 depth.
// out[0] = internal nodes, out[1] = grid ok (0/1), out[2] = depth.
int host_bvh_check(const float* boxes, uint32_t n, uint32_t* out) {
    const std::vector<rtbvh::Box> prim = to_boxes(boxes, n);
    const rtbvh::FlatBVH t = rtbvh::build(prim);
    out[0] = (uint32_t)t.trav.size();
    out[1] = t.grid.ok ? 1u : 0u;
    out[2] = t.depth;
    if (n == 0) return t.nodes.empty() ? 0 : 1;
    if (t.nodes.size() != 2 * (size_t)n - 1 || t.trav.size() != (size_t)n - 1 || t.leaf_of.size() != n) return 1;
    // leaf_of[prim] = index of the leaf node; nodes are numbered in creation (= depth-first, left first) order, so the
    // leaves met by a full left-first walk must come with increasing node index
    {
        std::vector<uint32_t> stack, order;
        uint32_t ref = t.root_ref, depth = 0, maxdepth = 0;
        std::vector<uint32_t> dstack;
        for (;;) {
            if (ref & rtbvh::LEAF_BIT) {
                order.push_back(ref & ~rtbvh::LEAF_BIT);
                if (depth > maxdepth) maxdepth = depth;
                if (stack.empty()) break;
                ref = stack.back();
                depth = dstack.back();
                stack.pop_back();
                dstack.pop_back();
                continue;
            }
            stack.push_back(t.trav[ref].right);
            dstack.push_back(depth + 1);
            ref = t.trav[ref].left;
            depth++;
        }
        if (order.size() != n) return 2;
        std::vector<char> seen(n, 0);
        for (size_t i = 0; i < order.size(); i++) {
            if (order[i] >= n || seen[order[i]]) return 2;
            seen[order[i]] = 1;
            if (i && t.leaf_of[order[i]] <= t.leaf_of[order[i - 1]]) return 2;
        }
        if (maxdepth != t.depth) return 7;
    }
    // a leaf's FlatNode carries the primitive's own box and a valid parent (a single-leaf tree has neither: the
    // reference returns its only shape without any box test, bvh_impl.rs:394-396)
    for (uint32_t i = 0; i < n && n > 1; i++) {
        const rtbvh::FlatNode& f = t.nodes[t.leaf_of[i]];
        for (int a = 0; a < 3; a++)
            if (f.lo[a] != prim[i].lo[a] || f.hi[a] != prim[i].hi[a]) return 3;
        if (n > 1 && f.parent >= t.nodes.size()) return 3;
    }
    if (!t.grid.ok) return 0;
    if (t.travq.size() != t.trav.size()) return 4;
    for (size_t k = 0; k < t.trav.size(); k++) {
        const rtbvh::TravNode& e = t.trav[k];
        const rtbvh::QNode& q = t.travq[k];
        if (q.left != e.left || q.right != e.right) return 4;
        const float* elo[2] = {e.l_lo, e.r_lo};
        const float* ehi[2] = {e.l_hi, e.r_hi};
        const uint16_t* qc[2] = {q.l_c, q.r_c};
        const uint16_t* qh[2] = {q.l_h, q.r_h};
        for (int c = 0; c < 2; c++)
            for (int a = 0; a < 3; a++) {
                // centre / half-extent form: lo = c - h, hi = c + h (the upper plane may sit one more unit out)
                if (qh[c][a] > qc[c][a]) return 5;
                const double st = t.grid.step[a], base = t.grid.base[a];
                const double lo = base + st * ((double)qc[c][a] - (double)qh[c][a]), hi = base + st * ((double)qc[c][a] + (double)qh[c][a]);
                if (!(lo <= (double)elo[c][a] - st) || !(hi >= (double)ehi[c][a] + st)) return 5;
                if (!(lo >= (double)elo[c][a] - 4.0 * st) || !(hi <= (double)ehi[c][a] + 5.0 * st)) return 6;
            }
    }
    return 0;
}

// The product's AABB helpers (rt_bvh.h) on raw boxes, for the reference's own doc-test vectors (aabb.rs:453,474,520,565;
// axis.rs:20,33).  out[0..2] size (as the builder forms it: hi - lo), [3..5] center, [6] surface_area, [7] largest_axis,
// [8] is_empty(a), [9..14] join(a, b), [15..20] join(a, point box) = grow
void host_aabb_kat(const float* a6, const float* b6, const float* pt3, float* out) {
    rtbvh::Box a, b, pt;
    for (int i = 0; i < 3; i++) {
        a.lo[i] = a6[i]; a.hi[i] = a6[3 + i];
        b.lo[i] = b6[i]; b.hi[i] = b6[3 + i];
        pt.lo[i] = pt.hi[i] = pt3[i];
    }
    float c[3];
    rtbvh::center(a, c);
    for (int i = 0; i < 3; i++) {
        out[i] = a.hi[i] - a.lo[i];
        out[3 + i] = c[i];
    }
    out[6] = rtbvh::surface_area(a);
    out[7] = (float)rtbvh::largest_axis(a);
    out[8] = (a.lo[0] > a.hi[0] || a.lo[1] > a.hi[1] || a.lo[2] > a.hi[2]) ? 1.0f : 0.0f;     // aabb.rs:501-503
    const rtbvh::Box j = rtbvh::join(a, b), g = rtbvh::join(a, pt);
    for (int i = 0; i < 3; i++) {
        out[9 + i] = j.lo[i]; out[12 + i] = j.hi[i];
        out[15 + i] = g.lo[i]; out[18 + i] = g.hi[i];
    }
}
void host_empty_box(float* out6) {
    const rtbvh::Box e = rtbvh::empty_box();
    for (int i = 0; i < 3; i++) { out6[i] = e.lo[i]; out6[3 + i] = e.hi[i]; }
}
// growing the empty box by one point, as the builder's centroid bounds start (aabb.rs:340-350)
void host_empty_grow(const float* pt3, float* out6) {
    rtbvh::Box pt;
    for (int i = 0; i < 3; i++) pt.lo[i] = pt.hi[i] = pt3[i];
    const rtbvh::Box g = rtbvh::join(rtbvh::empty_box(), pt);
    for (int i = 0; i < 3; i++) { out6[i] = g.lo[i]; out6[3 + i] = g.hi[i]; }
}
// the walk's slab test (the restatement above of ray.rs:174-194, as the kernels evaluate it) on one box
int host_ray_hits_box(const float* origin, const float* dir, const float* box6) {
    const float len = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    HRay r;
    for (int a = 0; a < 3; a++) {
        const float d = dir[a] / len;
        r.o[a] = origin[a];
        r.inv[a] = 1.0f / d;
        r.s[a] = d < 0.0f;
    }
    return hits(r, box6, box6 + 3) ? 1 : 0;
}

// the same ray turned round as the crate's property tests do (ray.rs:420-423): direction and inverse direction negated,
// the cached signs kept
int host_ray_hits_box_flipped(const float* origin, const float* dir, const float* box6) {
    const float len = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
    HRay r;
    for (int a = 0; a < 3; a++) {
        const float d = dir[a] / len;
        r.o[a] = origin[a];
        r.inv[a] = -(1.0f / d);
        r.s[a] = d < 0.0f;
    }
    return hits(r, box6, box6 + 3) ? 1 : 0;
}

}  // extern "C"
