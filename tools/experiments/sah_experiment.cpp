// Experiment (CPU): how many node steps would a better-built tree save?  The kernels only need a conservative walk
// (leaf validation restores the reference's candidate set), so the traversal tree need not be the reference's 6-bucket
// one.  Compares the reference tree (rt_bvh.h) with a 32-bin, three-axis SAH build on the same boxes: SAH cost and
// measured internal-node visits for camera-like and bounce-like rays.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "rt_bvh.h"

using rtbvh::Box;
struct Node2 {
    Box lb, rb;
    uint32_t l, r;   // LEAF_BIT | prim or node index
};
static float area(const Box& b) {
    const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return 2.f * (x * y + y * z + z * x);
}
static Box hull(const std::vector<Box>& p, const uint32_t* idx, size_t n) {
    Box b = rtbvh::empty_box();
    for (size_t i = 0; i < n; i++) b = rtbvh::join(b, p[idx[i]]);
    return b;
}
// top-down SAH, NB bins on each of the three axes, single-primitive leaves
static uint32_t build_alt(const std::vector<Box>& p, uint32_t* idx, size_t n, std::vector<Node2>& out, int NB) {
    if (n == 1) return rtbvh::LEAF_BIT | idx[0];
    Box cen = rtbvh::empty_box();
    for (size_t i = 0; i < n; i++) {
        float c[3];
        rtbvh::center(p[idx[i]], c);
        for (int a = 0; a < 3; a++) {
            cen.lo[a] = fminf(cen.lo[a], c[a]);
            cen.hi[a] = fmaxf(cen.hi[a], c[a]);
        }
    }
    int best_ax = -1, best_split = 0;
    float best_cost = INFINITY;
    for (int a = 0; a < 3; a++) {
        const float ext = cen.hi[a] - cen.lo[a];
        if (!(ext > 1e-9f)) continue;
        std::vector<Box> bb(NB, rtbvh::empty_box());
        std::vector<size_t> cnt(NB, 0);
        for (size_t i = 0; i < n; i++) {
            float c[3];
            rtbvh::center(p[idx[i]], c);
            int b = (int)((c[a] - cen.lo[a]) / ext * NB);
            b = std::min(std::max(b, 0), NB - 1);
            bb[b] = rtbvh::join(bb[b], p[idx[i]]);
            cnt[b]++;
        }
        std::vector<Box> rb(NB, rtbvh::empty_box());
        std::vector<size_t> rc(NB, 0);
        Box acc = rtbvh::empty_box();
        size_t c = 0;
        for (int b = NB - 1; b > 0; b--) {
            acc = rtbvh::join(acc, bb[b]);
            c += cnt[b];
            rb[b] = acc;
            rc[b] = c;
        }
        acc = rtbvh::empty_box();
        c = 0;
        for (int b = 0; b < NB - 1; b++) {
            acc = rtbvh::join(acc, bb[b]);
            c += cnt[b];
            if (c == 0 || rc[b + 1] == 0) continue;
            const float cost = c * area(acc) + rc[b + 1] * area(rb[b + 1]);
            if (cost < best_cost) {
                best_cost = cost;
                best_ax = a;
                best_split = b;
            }
        }
    }
    size_t mid;
    if (best_ax < 0) {
        mid = n / 2;
    } else {
        const float ext = cen.hi[best_ax] - cen.lo[best_ax];
        mid = std::partition(idx, idx + n, [&](uint32_t s) {
                  float c[3];
                  rtbvh::center(p[s], c);
                  int b = (int)((c[best_ax] - cen.lo[best_ax]) / ext * NB);
                  b = std::min(std::max(b, 0), NB - 1);
                  return b <= best_split;
              }) - idx;
        if (mid == 0 || mid == n) mid = n / 2;
    }
    const uint32_t me = (uint32_t)out.size();
    out.push_back(Node2{});
    Node2 nd;
    nd.lb = hull(p, idx, mid);
    nd.rb = hull(p, idx + mid, n - mid);
    nd.l = build_alt(p, idx, mid, out, NB);
    nd.r = build_alt(p, idx + mid, n - mid, out, NB);
    out[me] = nd;
    return me;
}
static bool hit(const float* o, const float* inv, const Box& b) {
    float tmin = 0.f, tmax = INFINITY;
    for (int a = 0; a < 3; a++) {
        const float t0 = (b.lo[a] - o[a]) * inv[a], t1 = (b.hi[a] - o[a]) * inv[a];
        tmin = fmaxf(tmin, fminf(t0, t1));
        tmax = fminf(tmax, fmaxf(t0, t1));
    }
    return tmin <= tmax;
}
static uint64_t visits(const std::vector<Node2>& t, uint32_t root, const float* o, const float* d, uint64_t* leaves) {
    if (root & rtbvh::LEAF_BIT) return 0;
    float inv[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    std::vector<uint32_t> st{root};
    uint64_t v = 0;
    while (!st.empty()) {
        const uint32_t r = st.back();
        st.pop_back();
        if (r & rtbvh::LEAF_BIT) {
            (*leaves)++;
            continue;
        }
        v++;
        if (hit(o, inv, t[r].lb)) st.push_back(t[r].l);
        if (hit(o, inv, t[r].rb)) st.push_back(t[r].r);
    }
    return v;
}
extern "C" void sah_experiment(const float* boxes, uint32_t n, int nb, double* out) {
    std::vector<Box> p(n);
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            p[i].lo[a] = boxes[6 * i + a];
            p[i].hi[a] = boxes[6 * i + 3 + a];
        }
    const rtbvh::FlatBVH ref = rtbvh::build(p);
    std::vector<Node2> a(ref.trav.size());
    for (size_t k = 0; k < ref.trav.size(); k++) {
        const rtbvh::TravNode& e = ref.trav[k];
        for (int x = 0; x < 3; x++) {
            a[k].lb.lo[x] = e.l_lo[x];
            a[k].lb.hi[x] = e.l_hi[x];
            a[k].rb.lo[x] = e.r_lo[x];
            a[k].rb.hi[x] = e.r_hi[x];
        }
        a[k].l = e.left;
        a[k].r = e.right;
    }
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    std::vector<Node2> b;
    b.reserve(n);
    const uint32_t broot = build_alt(p, idx.data(), n, b, nb);
    auto sah = [&](const std::vector<Node2>& t) {
        double s = 0;
        for (const Node2& nd : t) s += area(nd.lb) + area(nd.rb);
        return s;
    };
    const Box all = hull(p, idx.data(), n);
    out[0] = sah(a) / area(all);
    out[1] = sah(b) / area(all);
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    uint64_t va = 0, vb = 0, la = 0, lb = 0, va2 = 0, vb2 = 0;
    const int NR = 200000;
    for (int i = 0; i < NR; i++) {            // camera-like: origin 0, direction through the fov-90 image plane
        float o[3] = {0, 0, 0}, d[3] = {U(g) * 1.777f, U(g), -1.f};
        const float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (float& x : d) x /= l;
        va += visits(a, ref.root_ref, o, d, &la);
        vb += visits(b, broot, o, d, &lb);
    }
    for (int i = 0; i < NR; i++) {            // bounce-like: from the top of a random primitive box, upper hemisphere-ish
        const Box& s = p[g() % n];
        float o[3] = {0.5f * (s.lo[0] + s.hi[0]), s.hi[1] + 1e-3f, 0.5f * (s.lo[2] + s.hi[2])};
        float d[3] = {U(g), fabsf(U(g)) * 0.7f + 0.01f, U(g)};
        const float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (float& x : d) x /= l;
        va2 += visits(a, ref.root_ref, o, d, &la);
        vb2 += visits(b, broot, o, d, &lb);
    }
    out[2] = (double)va / NR;
    out[3] = (double)vb / NR;
    out[4] = (double)va2 / NR;
    out[5] = (double)vb2 / NR;
    out[6] = (double)la / (2.0 * NR);
    out[7] = (double)lb / (2.0 * NR);
}
