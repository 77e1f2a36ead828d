// CPU study for the treelet walk: how many node visits does a c5-class ray make when the tree is cut
// into treelets whose nodes hold 8-bit planes on a lattice derived from the treelet root's box (16-byte nodes = one
// gather instruction), against the 16-bit quantised walk (32-byte nodes = two) and the exact tree.
//   g++ -O2 -std=c++17 -I ray_tracer_s8_amd/csrc -o /tmp/treelet_sim tools/experiments/treelet_sim.cpp -lpthread
//   /tmp/treelet_sim [T_bottom] [T_mid or 0]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "rt_bvh.h"

struct Eff { int lo[3], hi[3]; };                 // a child box in global 16-bit grid units
struct Frame { int o[3], s[3]; };

int main(int argc, char** argv) {
    const uint32_t TB = argc > 1 ? atoi(argv[1]) : 64;      // leaves per bottom treelet
    const uint32_t TM = argc > 2 ? atoi(argv[2]) : 0;       // leaves per mid treelet (0: two levels only)
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<rtbvh::Box> boxes;
    auto add = [&](float cx, float cy, float cz, float r) { rtbvh::Box b; b.lo[0]=cx-r; b.lo[1]=cy-r; b.lo[2]=cz-r; b.hi[0]=cx+r; b.hi[1]=cy+r; b.hi[2]=cz+r; boxes.push_back(b); };
    add(0, -101, -20, 100);
    while (boxes.size() < 65536) { float x = -96 + 192*U(g), y = -1 + 41*U(g), z = -192 + 189*U(g), r = 0.1f + 0.3f*U(g); if (std::sqrt(x*x+y*y+z*z) - r < 0.5f) continue; add(x,y,z,r); }
    rtbvh::FlatBVH t = rtbvh::build(boxes);
    const size_t NI = t.trav.size();
    printf("internal nodes %zu depth %u   T_bottom %u  T_mid %u\n", NI, t.depth, TB, TM);
    // leaves per subtree
    std::vector<uint32_t> leaves(NI, 0);
    for (size_t n = NI; n-- > 0;) {              // children have larger numbers (creation order = DFS)
        auto cnt = [&](uint32_t r) { return (r & rtbvh::LEAF_BIT) ? 1u : leaves[r]; };
        leaves[n] = cnt(t.trav[n].left) + cnt(t.trav[n].right);
    }
    // level of a subtree with L leaves: 0 = root treelet, 1 = mid, 2 = bottom
    auto level_of = [&](uint32_t L) { return L > (TM ? TM : TB) ? 0 : (TM && L > TB) ? 1 : 2; };
    // frames and effective child boxes
    std::vector<Frame> frame(NI);
    std::vector<Eff> effl(NI), effr(NI);
    std::vector<int> lvl(NI);
    size_t n_lvl[3] = {0, 0, 0}, n_treelets[3] = {0, 0, 0};
    struct W { uint32_t n; Frame f; int lv; };
    std::vector<W> st;
    st.push_back({0, Frame{{0, 0, 0}, {259, 259, 259}}, 0});
    n_treelets[0] = 1;
    auto ceil_div = [](int a, int b) { return (a + b - 1) / b; };
    while (!st.empty()) {
        W w = st.back(); st.pop_back();
        const rtbvh::QNode& q = t.travq[w.n];
        frame[w.n] = w.f; lvl[w.n] = w.lv; n_lvl[w.lv]++;
        auto quant = [&](const uint16_t* lo, const uint16_t* hi, Eff& e) {
            for (int a = 0; a < 3; a++) {
                int l8 = ((int)lo[a] - w.f.o[a]) / w.f.s[a];          // floor (operands >= 0)
                int h8 = ceil_div((int)hi[a] - w.f.o[a], w.f.s[a]);
                if (l8 < 0 || h8 > 255 || (int)lo[a] < w.f.o[a]) { printf("range! node %u axis %d l8 %d h8 %d\n", w.n, a, l8, h8); exit(1); }
                e.lo[a] = w.f.o[a] + l8 * w.f.s[a];
                e.hi[a] = w.f.o[a] + h8 * w.f.s[a];
            }
        };
        quant(q.l_lo, q.l_hi, effl[w.n]);
        quant(q.r_lo, q.r_hi, effr[w.n]);
        auto child = [&](uint32_t r, const Eff& e) {
            if (r & rtbvh::LEAF_BIT) return;
            const int cl = level_of(leaves[r]);
            if (cl == w.lv) { st.push_back({r, w.f, cl}); return; }
            Frame f;
            for (int a = 0; a < 3; a++) { f.o[a] = e.lo[a]; f.s[a] = std::max(1, ceil_div(e.hi[a] - e.lo[a], 255)); }
            n_treelets[cl]++;
            st.push_back({r, f, cl});
        };
        child(q.left, effl[w.n]);
        child(q.right, effr[w.n]);
    }
    printf("nodes per level: root %zu  mid %zu  bottom %zu ; treelets: mid %zu bottom %zu\n", n_lvl[0], n_lvl[1], n_lvl[2], n_treelets[1], n_treelets[2]);
    // rays
    const int nr = 40000;
    double v_exact = 0, v_q16 = 0, v8[3] = {0, 0, 0}, enter[3] = {0, 0, 0}, leaf_exact = 0, leaf_q16 = 0, leaf_8 = 0;
    for (int i = 0; i < nr; i++) {
        float o[3], d[3];
        if (i & 1) { float u = 2*U(g)-1, v = (2*U(g)-1)*0.5625f; d[0]=u; d[1]=v; d[2]=-1; o[0]=o[1]=o[2]=0; }
        else { o[0] = -96 + 192*U(g); o[1] = -1 + 20*U(g); o[2] = -192 + 189*U(g); float z = 2*U(g)-1, ph = 6.2831853f*U(g), s = std::sqrt(1-z*z); d[0]=s*std::cos(ph); d[1]=std::fabs(z); d[2]=s*std::sin(ph); }
        double len = std::sqrt((double)d[0]*d[0]+(double)d[1]*d[1]+(double)d[2]*d[2]);
        double inv[3], og[3], ig[3];
        for (int a = 0; a < 3; a++) { inv[a] = len / d[a]; og[a] = ((double)o[a] - t.grid.base[a]) / t.grid.step[a]; ig[a] = t.grid.step[a] * inv[a]; }
        auto hit_f = [&](const float* lo, const float* hi) {
            double tn = 0, tf = 1e300;
            for (int a = 0; a < 3; a++) { double t0 = (lo[a] - o[a]) * inv[a], t1 = (hi[a] - o[a]) * inv[a]; tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1)); }
            return tn <= tf;
        };
        auto hit_g = [&](const int* lo, const int* hi) {       // grid units
            double tn = 0, tf = 1e300;
            for (int a = 0; a < 3; a++) { double t0 = (lo[a] - og[a]) * ig[a], t1 = (hi[a] - og[a]) * ig[a]; tn = std::max(tn, std::min(t0, t1)); tf = std::min(tf, std::max(t0, t1)); }
            return tn <= tf;
        };
        for (int mode = 0; mode < 3; mode++) {
            std::vector<uint32_t> s; uint32_t ref = t.root_ref;
            for (;;) {
                if (ref & rtbvh::LEAF_BIT) { (mode == 0 ? leaf_exact : mode == 1 ? leaf_q16 : leaf_8) += 1; if (s.empty()) break; ref = s.back(); s.pop_back(); continue; }
                bool hl, hr;
                if (mode == 0) { const rtbvh::TravNode& nd = t.trav[ref]; v_exact += 1; hl = hit_f(nd.l_lo, nd.l_hi); hr = hit_f(nd.r_lo, nd.r_hi); }
                else if (mode == 1) {
                    const rtbvh::QNode& q = t.travq[ref]; v_q16 += 1;
                    int a0[3] = {q.l_lo[0], q.l_lo[1], q.l_lo[2]}, a1[3] = {q.l_hi[0], q.l_hi[1], q.l_hi[2]}, b0[3] = {q.r_lo[0], q.r_lo[1], q.r_lo[2]}, b1[3] = {q.r_hi[0], q.r_hi[1], q.r_hi[2]};
                    hl = hit_g(a0, a1); hr = hit_g(b0, b1);
                } else { v8[lvl[ref]] += 1; hl = hit_g(effl[ref].lo, effl[ref].hi); hr = hit_g(effr[ref].lo, effr[ref].hi); }
                const rtbvh::TravNode& nd = t.trav[ref];
                if (mode == 2) {
                    auto ent = [&](uint32_t c) { if (!(c & rtbvh::LEAF_BIT) && lvl[c] != lvl[ref]) enter[lvl[c]] += 1; };
                    if (hl) ent(nd.left);
                    if (hr) ent(nd.right);
                }
                if (hl) { if (hr) s.push_back(nd.right); ref = nd.left; } else if (hr) ref = nd.right; else if (s.empty()) break; else { ref = s.back(); s.pop_back(); }
            }
        }
    }
    printf("per ray: exact %.1f visits %.2f leaves | 16-bit %.1f visits %.2f leaves | 8-bit treelets: root %.1f mid %.1f bottom %.1f = %.1f visits, %.2f leaves; treelet entries mid %.2f bottom %.2f\n",
           v_exact / nr, leaf_exact / nr, v_q16 / nr, leaf_q16 / nr, v8[0] / nr, v8[1] / nr, v8[2] / nr, (v8[0] + v8[1] + v8[2]) / nr, leaf_8 / nr,
           enter[1] / nr, enter[2] / nr);
    printf("gather instructions per ray (nodes only): 16-bit walk %.0f ; treelets with root+mid in LDS %.0f ; with only root in LDS %.0f\n",
           2 * v_q16 / nr, v8[2] / nr, (v8[1] + v8[2]) / nr);
    printf("LDS bytes: root %zu, root+mid %zu\n", n_lvl[0] * 16, (n_lvl[0] + n_lvl[1]) * 16);
    return 0;
}
