// rt_bvh.h — host-side construction of the reference's candidate-filter BVH (product code).
//
// The reference slave never tests a primitive the `bvh` crate's traversal did not return
// (ray-tracer-slave/src/main.rs:112-114): a primitive is a candidate only if the ray passes the
// AABB test of every node on its root-to-leaf path (bvh_impl.rs:373-398, ray.rs:174-194).
// At large distances the reference's sphere roots lose precision (sphere.rs:45 squares a length
// of ~1e2 to find a radius of ~1e-1) and report hits for rays that miss the sphere's AABB;
// the BVH silently drops those.  To be pixel-for-pixel with the reference the GPU kernel therefore
// validates an accepted hit against that same chain of AABB tests, and breaks exact distance
// ties by DFS leaf order like `min_by` over the traversal output (shapes/mod.rs:177-182).
//
// This file builds the tree with the crate's algorithm (bvh_impl.rs:229-364: centroid-bounds
// largest axis, 6 buckets, SAH cost, f32 arithmetic throughout) and flattens it to what the
// kernel needs: for every node its parent and the AABB its parent stores for it, and for every
// primitive its leaf node.  Node indices are assigned in the crate's order (pre-order, left
// first), so a leaf's node index is also its DFS rank.
#pragma once
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <limits>
#include <thread>
#include <vector>

namespace rtbvh {

struct Box {
    float lo[3], hi[3];
};

struct FlatNode {          // 32 bytes: two float4 on the device
    float lo[3];
    uint32_t parent;       // 0xffffffff for the root
    float hi[3];
    uint32_t pad;
};

// Traversal node (internal nodes only, 64 bytes = four float4 on the device): the two child AABBs the
// crate stores in a `BVHNode::Node` plus child references.  A reference is either the index of another
// TravNode or LEAF_BIT | primitive index.
constexpr uint32_t LEAF_BIT = 0x80000000u;
struct TravNode {
    float l_lo[3];
    uint32_t left;
    float l_hi[3];
    uint32_t right;
    float r_lo[3];
    uint32_t pad0;
    float r_hi[3];
    uint32_t pad1;
};

// Quantised traversal node (32 bytes = two uint4): the same topology as TravNode, child boxes rounded OUTWARDS onto
// a 16-bit grid over the scene box (coordinate = base + q*step) with at least one whole grid unit of slack.  Only a conservative pre-filter: the
// kernel validates every leaf it reaches with the exact box.  A box is held as CENTRE and HALF-EXTENT per axis (integers, lo = c - h,
// hi = c + h, the upper plane at most one more unit out): the kernel then gets the near / far slab values as tc -+ h |ig| with
// three fused multiply-adds per axis and child, and no min / max per plane (rt_kernel.hip.h).
struct QNode {
    uint16_t l_c[3], l_h[3], r_c[3], r_h[3];
    uint32_t left, right;
};

struct QGrid {
    float base[3] = {0.f, 0.f, 0.f}, step[3] = {1.f, 1.f, 1.f};
    bool ok = false;                 // false: degenerate / non-finite scene box, use the exact nodes
};

struct FlatBVH {
    std::vector<FlatNode> nodes;
    std::vector<uint32_t> leaf_of;   // primitive index -> node index (= DFS rank)
    std::vector<TravNode> trav;      // internal nodes, root first (empty when the root is a leaf)
    uint32_t root_ref = 0;           // reference of the root (LEAF_BIT | 0 for a single primitive)
    uint32_t depth = 0;              // edges on the longest root-to-leaf path
    std::vector<QNode> travq;        // quantised twin of trav
    QGrid grid;
};

inline Box empty_box() {
    const float inf = std::numeric_limits<float>::infinity();
    return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
}
// aabb.rs:268-282 join / :357-372 grow use f32::min / f32::max (IEEE minNum/maxNum)
inline Box join(const Box& a, const Box& b) {
    Box r;
    for (int i = 0; i < 3; i++) {
        r.lo[i] = fminf(a.lo[i], b.lo[i]);
        r.hi[i] = fmaxf(a.hi[i], b.hi[i]);
    }
    return r;
}
inline void center(const Box& b, float c[3]) {      // aabb.rs:458-484: min + (max - min) / 2
    for (int i = 0; i < 3; i++) c[i] = b.lo[i] + ((b.hi[i] - b.lo[i]) / 2.0f);
}
inline float surface_area(const Box& b) {           // aabb.rs:525-528
    const float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    return 2.0f * (sx * sy + sx * sz + sy * sz);
}
inline int largest_axis(const Box& b) {             // aabb.rs:570-580
    const float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    if (sx > sy && sx > sz) return 0;
    if (sy > sz) return 1;
    return 2;
}

// Build over the primitives' AABBs (`Bounded::aabb`, sphere.rs:65-72 / mesh.rs:46-96).
// `order` (optional): order[w] = the primitive at position w of the reference's `world: Vec<Object>` (lib.rs:11).  BVH::build
// starts from `indices = 0..shapes.len()` over THAT list (bvh_impl.rs:421-427), and everything downstream keeps the list
// order (bucket member lists, the halves of the split_at(len / 2) fallback :277-291, hence the leaves' depth-first order),
// so the build starts from the primitives in world order; nullptr: primitive order.  Primitive numbers (boxes, leaf
// references, leaf_of) stay those of `prim`.
inline FlatBVH build(const std::vector<Box>& prim, const uint32_t* order = nullptr) {
    FlatBVH out;
    out.leaf_of.assign(prim.size(), 0);
    if (prim.empty()) return out;                   // the reference recurses without bound here
    // Work list of index RANGES over one array: a node's primitives are idx[begin, end), in the order the
    // reference's recursion would hold them (children = the buckets' member lists concatenated in bucket order, each
    // in its original relative order = a stable partition), so the halving fallback and the leaf order are the same.
    struct Item {
        uint32_t begin, end;
        uint32_t parent;
        Box as_seen_by_parent;
        bool is_right;
        uint32_t depth;
        uint32_t me;                 // node number
    };
    const uint32_t np = (uint32_t)prim.size();
    std::vector<uint32_t> idx(np), tmp(np);
    std::vector<uint8_t> bucket_of(np);
    std::vector<float> cx(np), cy(np), cz(np);          // centroids once (aabb.rs:458-484)
    for (uint32_t i = 0; i < np; i++) {
        idx[i] = order ? order[i] : i;
        float c[3];
        center(prim[i], c);
        cx[i] = c[0];
        cy[i] = c[1];
        cz[i] = c[2];
    }
    const float* cen_of[3] = {cx.data(), cy.data(), cz.data()};
    // Node numbers are the reference's creation order = depth first, left subtree first.  A subtree over m
    // primitives has 2m - 1 nodes, so a node's number follows from its position (left child = me + 1, right child =
    // me + 2 * m_left) and disjoint subtrees can be built by different threads into preallocated arrays.
    const size_t n_nodes = 2 * (size_t)np - 1;
    std::vector<uint32_t> left_of(n_nodes, 0xffffffffu), right_of(n_nodes, 0xffffffffu), prim_of(n_nodes, 0xffffffffu);
    out.nodes.assign(n_nodes, FlatNode{{0, 0, 0}, 0, {0, 0, 0}, 0});
    const float EPS = 0.00001f;                     // bvh lib.rs:80
    // splits one node; returns false for a leaf, else fills L and R
    auto split = [&](const Item& it, Item& L, Item& R) -> bool {
        const uint32_t me = it.me;
        FlatNode fn;
        for (int i = 0; i < 3; i++) {
            fn.lo[i] = it.as_seen_by_parent.lo[i];
            fn.hi[i] = it.as_seen_by_parent.hi[i];
        }
        fn.parent = it.parent;
        fn.pad = 0;
        out.nodes[me] = fn;
        if (it.parent != 0xffffffffu) (it.is_right ? right_of : left_of)[it.parent] = me;
        const uint32_t cnt_all = it.end - it.begin;
        if (cnt_all == 1) {                         // bvh_impl.rs:254-265
            out.leaf_of[idx[it.begin]] = me;
            prim_of[me] = idx[it.begin];
            return false;
        }
        // convex hull of the shapes and of their centroids (:247-251)
        Box all = empty_box(), cen = empty_box();
        for (uint32_t k = it.begin; k < it.end; k++) {
            const uint32_t s = idx[k];
            all = join(all, prim[s]);
            cen.lo[0] = fminf(cen.lo[0], cx[s]);
            cen.hi[0] = fmaxf(cen.hi[0], cx[s]);
            cen.lo[1] = fminf(cen.lo[1], cy[s]);
            cen.hi[1] = fmaxf(cen.hi[1], cy[s]);
            cen.lo[2] = fminf(cen.lo[2], cz[s]);
            cen.hi[2] = fmaxf(cen.hi[2], cz[s]);
        }
        const int ax = largest_axis(cen);
        const float extent = cen.hi[ax] - cen.lo[ax];
        L = Item{0, 0, me, empty_box(), false, it.depth + 1, 0};
        R = Item{0, 0, me, empty_box(), true, it.depth + 1, 0};
        auto halve = [&]() {                        // :277-291
            const uint32_t h = cnt_all / 2;
            L.begin = it.begin;
            L.end = R.begin = it.begin + h;
            R.end = it.end;
            L.as_seen_by_parent = empty_box();
            for (uint32_t k = L.begin; k < L.end; k++) L.as_seen_by_parent = join(L.as_seen_by_parent, prim[idx[k]]);
            R.as_seen_by_parent = empty_box();
            for (uint32_t k = R.begin; k < R.end; k++) R.as_seen_by_parent = join(R.as_seen_by_parent, prim[idx[k]]);
        };
        if (extent < EPS) {
            halve();
        } else {                                    // :293-349, six SAH buckets
            constexpr int NB = 6;
            size_t cnt[NB] = {0, 0, 0, 0, 0, 0};
            Box bb[NB];
            for (int b = 0; b < NB; b++) bb[b] = empty_box();
            const float* ca = cen_of[ax];
            for (uint32_t k = it.begin; k < it.end; k++) {
                const uint32_t s = idx[k];
                const float rel = (ca[s] - cen.lo[ax]) / extent;
                const float fb = rel * ((float)NB - 0.01f);
                size_t b = 0;                       // Rust `as usize`: truncate, saturate, NaN -> 0
                if (fb == fb && fb > 0.0f) b = fb >= 1.8e19f ? (size_t)-1 : (size_t)fb;
                if (b >= (size_t)NB) b = NB - 1;    // (the crate would index out of bounds)
                cnt[b]++;
                bb[b] = join(bb[b], prim[s]);
                bucket_of[k] = (uint8_t)b;
            }
            int best = 0;
            float best_cost = std::numeric_limits<float>::infinity();
            Box best_l = empty_box(), best_r = empty_box();
            for (int i = 0; i < NB - 1; i++) {
                size_t nl = 0, nr = 0;
                Box l = empty_box(), r = empty_box();
                for (int b = 0; b <= i; b++) {
                    nl += cnt[b];
                    l = join(l, bb[b]);
                }
                for (int b = i + 1; b < NB; b++) {
                    nr += cnt[b];
                    r = join(r, bb[b]);
                }
                const float cost = ((float)nl * surface_area(l) + (float)nr * surface_area(r)) / surface_area(all);
                if (cost < best_cost) {
                    best = i;
                    best_cost = cost;
                    best_l = l;
                    best_r = r;
                }
            }
            // stable counting sort of the range by bucket: the members of bucket 0, then 1, ... each in list order
            uint32_t start[NB + 1];
            start[0] = it.begin;
            for (int b = 0; b < NB; b++) start[b + 1] = start[b] + (uint32_t)cnt[b];
            uint32_t fill[NB];
            for (int b = 0; b < NB; b++) fill[b] = start[b];
            for (uint32_t k = it.begin; k < it.end; k++) tmp[fill[bucket_of[k]]++] = idx[k];
            for (uint32_t k = it.begin; k < it.end; k++) idx[k] = tmp[k];
            L.begin = it.begin;
            L.end = R.begin = start[best + 1];
            R.end = it.end;
            L.as_seen_by_parent = best_l;
            R.as_seen_by_parent = best_r;
            if (L.begin == L.end || R.begin == R.end) halve();   // unreachable for finite centroids
        }
        L.me = me + 1;
        R.me = me + 2 * (L.end - L.begin);
        return true;
    };
    // whole subtree below one item, depth first; returns the deepest level reached
    auto subtree = [&](const Item& top) -> uint32_t {
        uint32_t deepest = top.depth;
        std::vector<Item> stack{top};
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (it.depth > deepest) deepest = it.depth;
            Item L, R;
            if (split(it, L, R)) {
                stack.push_back(R);
                stack.push_back(L);
            }
        }
        return deepest;
    };
    const Item root{0u, np, 0xffffffffu, empty_box(), false, 0u, 0u};
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = np < 16384 ? 1u : std::min(8u, hw ? hw : 1u);
    if (n_thr <= 1) {
        out.depth = subtree(root);
    } else {
        // the top of the tree on this thread until there are enough subtrees to share out, then one worker per chunk
        std::vector<Item> open{root}, work;
        while (!open.empty() && open.size() + work.size() < 8 * n_thr) {
            size_t big = 0;                          // split the largest open item next
            for (size_t i = 1; i < open.size(); i++)
                if (open[i].end - open[i].begin > open[big].end - open[big].begin) big = i;
            const Item it = open[big];
            open.erase(open.begin() + (long)big);
            if (it.depth > out.depth) out.depth = it.depth;
            if (it.end - it.begin < 1024) {
                work.push_back(it);
                continue;
            }
            Item L, R;
            if (split(it, L, R)) {
                open.push_back(L);
                open.push_back(R);
            }
        }
        work.insert(work.end(), open.begin(), open.end());
        std::sort(work.begin(), work.end(), [](const Item& a, const Item& b) { return a.end - a.begin > b.end - b.begin; });
        std::atomic<size_t> next{0};
        std::vector<uint32_t> deepest(n_thr, 0);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_thr; t++)
            pool.emplace_back([&, t] {
                for (size_t i = next.fetch_add(1); i < work.size(); i = next.fetch_add(1))
                    deepest[t] = std::max(deepest[t], subtree(work[i]));
            });
        for (std::thread& th : pool) th.join();
        for (uint32_t dpt : deepest) out.depth = std::max(out.depth, dpt);
    }
    // traversal form: internal nodes only, numbered in creation order
    std::vector<uint32_t> trav_id(out.nodes.size(), 0xffffffffu);
    uint32_t n_int = 0;
    for (size_t n = 0; n < out.nodes.size(); n++)
        if (prim_of[n] == 0xffffffffu) trav_id[n] = n_int++;
    out.trav.resize(n_int);
    auto ref_of = [&](uint32_t n) { return prim_of[n] != 0xffffffffu ? (LEAF_BIT | prim_of[n]) : trav_id[n]; };
    for (size_t n = 0; n < out.nodes.size(); n++) {
        if (prim_of[n] != 0xffffffffu) continue;
        TravNode& t = out.trav[trav_id[n]];
        const FlatNode& l = out.nodes[left_of[n]];
        const FlatNode& r = out.nodes[right_of[n]];
        for (int i = 0; i < 3; i++) {
            t.l_lo[i] = l.lo[i];
            t.l_hi[i] = l.hi[i];
            t.r_lo[i] = r.lo[i];
            t.r_hi[i] = r.hi[i];
        }
        t.left = ref_of(left_of[n]);
        t.right = ref_of(right_of[n]);
        t.pad0 = t.pad1 = 0;
    }
    out.root_ref = ref_of(0);

    // ---- quantised twin.  Grid over the union of the root's child boxes, 4 units of margin below and 11 above.
    // The device never decodes a coordinate: it evaluates the slab test in grid units (rt_kernel.hip.h), with
    // < 0.15 unit of rounding against the >= 1 unit of outward slack given here.
    if (!out.trav.empty()) {
        QGrid g;
        g.ok = true;
        const TravNode& r0 = out.trav[0];
        for (int a = 0; a < 3; a++) {
            const float lo = fminf(r0.l_lo[a], r0.r_lo[a]), hi = fmaxf(r0.l_hi[a], r0.r_hi[a]);
            if (!(std::isfinite(lo) && std::isfinite(hi)) || !(hi >= lo)) g.ok = false;
            const double ext = (double)hi - (double)lo;
            float st = (float)(ext / 65520.0);                // 16 units of head-room: no box is ever clamped
            if (!(st > 0.f)) st = 1e-30f;                     // flat scene along this axis
            g.step[a] = st;
            g.base[a] = (float)((double)lo - 4.0 * (double)st);
            if (!std::isfinite(st) || !std::isfinite(g.base[a])) g.ok = false;
        }
        if (g.ok) {
            out.travq.resize(out.trav.size());
            // grid coordinate with at least one whole unit of outward slack, in exact (double) arithmetic:
            // base + q_lo*step <= v - step  and  base + q_hi*step >= v + step
            auto q_lo = [&](float v, int a, bool& ok) {
                const double q = std::floor(((double)v - (double)g.base[a]) / (double)g.step[a]) - 1.0;
                if (!(q >= 0.0 && q <= 65535.0)) ok = false;
                return (uint16_t)(q >= 0.0 && q <= 65535.0 ? q : 0.0);
            };
            auto q_hi = [&](float v, int a, bool& ok) {
                const double q = std::ceil(((double)v - (double)g.base[a]) / (double)g.step[a]) + 1.0;
                if (!(q >= 0.0 && q <= 65535.0)) ok = false;
                return (uint16_t)(q >= 0.0 && q <= 65535.0 ? q : 65535.0);
            };
            for (size_t n = 0; n < out.trav.size() && g.ok; n++) {
                const TravNode& tn = out.trav[n];
                QNode& qn = out.travq[n];
                for (int a = 0; a < 3; a++) {
                    // centre / half-extent: h = ceil((hi - lo) / 2), c = lo + h, so that c - h = lo and c + h = hi or hi + 1
                    auto put = [&](float flo, float fhi, uint16_t& c, uint16_t& h) {
                        const uint32_t lo = q_lo(flo, a, g.ok), hi = q_hi(fhi, a, g.ok);
                        const uint32_t hh = hi >= lo ? (hi - lo + 1u) >> 1 : 0u;
                        if (lo + hh > 65535u) g.ok = false;
                        h = (uint16_t)hh;
                        c = (uint16_t)(lo + hh);
                    };
                    put(tn.l_lo[a], tn.l_hi[a], qn.l_c[a], qn.l_h[a]);
                    put(tn.r_lo[a], tn.r_hi[a], qn.r_c[a], qn.r_h[a]);
                }
                qn.left = tn.left;
                qn.right = tn.right;
            }
        }
        out.grid = g;
    }
    return out;
}

}  // namespace rtbvh
