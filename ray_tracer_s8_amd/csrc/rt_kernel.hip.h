// rt_kernel.hip.h — gfx950 path-trace tile kernel (device code).
//
// One work-item per pixel SAMPLE: a lane owns one sample of a pixel at a time and runs the reference slave's
// per-sample loop body for it — thin-lens ray generation (camera.rs:109-129), closest hit
// over the primitive list (shapes/mod.rs:158-191), shade + bounce (main.rs:108-146) — and the wave then
// sums a pixel's sample colours in the reference's order and takes the mean, gamma and RGB8 quantise
// (main.rs:73-81, color.rs:13-19).
//
// Arithmetic contract: every value that reaches the image is computed with the SAME
// IEEE-754 binary32 operations in the SAME order as the reference (oracle/rt_oracle.cpp):
// this file must be compiled with -ffp-contract=off and correctly rounded sqrt/div
// (hipcc default).  FMA is used only where written explicitly and only inside the
// conservative broad phase, whose value never reaches the image.
//
// Structure (DESIGN.md "Kernel"):
//   * persistent waves: the grid is sized to the chip, every wave pulls tiles of 64x1 pixels of
//     the requested strips from a global atomic queue; a tile is 64 * spp SAMPLE UNITS in pixel-major
//     order, a lane whose path ends deposits the sample's colour in the wave's ring and takes the wave's
//     next unit at once — the scan / walk runs with a nearly full EXEC mask although path lengths vary
//     from 1 to depth+1 segments, and no lane ever works through a pixel's samples one after the other;
//     the wave commits finished units IN ORDER: a pixel's colours are summed s = 0 .. spp-1 as main.rs:73-77 does;
//   * broad phase: wave-uniform LDS broadcast reads (ds_read_b128) of sphere pairs,
//     packed-FP32 (v_pk_fma_f32) conservative "line misses inflated sphere" test;
//     survivors go to a per-lane candidate list in LDS, in index order;
//   * narrow phase: the reference's exact root computation (sphere.rs:42-47 +
//     roots::find_roots_quadratic + shapes/mod.rs:106-129) on the candidates only;
//   * scenes larger than one LDS chunk are streamed chunk by chunk through LDS
//     (STREAMED=true), workgroup-synchronously;
//   * one launch serves a batch of strips (same frame, any division_no / seed).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "rt_cull.h"

namespace rtk {

constexpr int BLOCK = 256;       // 4 waves
#ifndef RT_MAXC
#define RT_MAXC 16
#endif
#ifndef RT_MINWAVES
#define RT_MINWAVES 4
#endif
#ifndef RT_MINWAVES_TRAV
#define RT_MINWAVES_TRAV 4       // exact-node L2 kernel: no spills.  History: 6 waves / 80 VGPRs / 10 spilled registers was +1 % on c3 when c3
                                 // still ran here; on what it serves now (meshes — whose deep trees leave LDS for four workgroups per CU
                                 // anyway — and mid-size fields) 5 was +3...8 % / +0...1 % over 6, and with the compacted root tests and the
                                 // straight-line step 4 another +1...6 % / +0...3 % over 5 (tools/ab_trav.sh)
#endif
#ifndef RT_MINWAVES_LTREE       // LDS-resident tree: one workgroup of 16 waves per CU = 4 per SIMD, 128 VGPRs
#define RT_MINWAVES_LTREE 4
#endif
#ifndef RT_MINWAVES_CULL
#define RT_MINWAVES_CULL 5
#endif
#ifndef RT_STEPS_PER_CHECK_X    // exact-node L2 kernel, straight-line step
#define RT_STEPS_PER_CHECK_X 16
#endif
#ifndef RT_QUEUE_TAKE_MIN       // tiles a workgroup takes from the launch's queue per (memory-side) atomic, at least
#define RT_QUEUE_TAKE_MIN 4
#endif
#ifndef RT_BF2                  // straight-line step in the uncapped quantised walks (see there)
#define RT_BF2 1
#endif
#ifndef RT_CULL_FLUSH_MIN       // culled walk: a lane without a hit root-tests its leaves at a check once it holds this many
#define RT_CULL_FLUSH_MIN 1
#endif
#ifndef RT_MINWAVES_QTRAV       // quantised-node kernels: 96 VGPRs, no spill slots (unbounded they take 97-99 = 4 waves/SIMD)
#define RT_MINWAVES_QTRAV 5
#endif
constexpr int MAXC = RT_MAXC;    // candidate list slots per lane (per chunk)
constexpr int CHUNK = 2048;      // max spheres per LDS chunk (32 KiB): list entries carry an 8-bit group index
constexpr int UNROLL = 8;        // broad-phase unroll; chunk sizes are padded to this
constexpr int MAX_BATCH = 64;    // strips per launch
constexpr int TRAV_STACK = 64;   // traversal stack entries per lane (host falls back to the linear scan beyond)
#ifndef RT_MAXL
#define RT_MAXL 8
#endif
#ifndef RT_MINL                 // the host may shrink the leaf lists down to this many slots to fit one more workgroup per CU
#define RT_MINL 4
#endif
#ifndef RT_STEPS_PER_CHECK
#define RT_STEPS_PER_CHECK 8
#endif
#ifndef RT_STEPS_PER_CHECK_LTREE   // LDS-tree kernel: steps per block of its branch-free walk (tools/sweep_steps.sh: 8 at refill 2/8)
#define RT_STEPS_PER_CHECK_LTREE 8
#endif
#ifndef RT_STEPS_PER_CHECK_Q    // quantised-node kernel (large scenes, long walks): c5 +1 % over 8
#define RT_STEPS_PER_CHECK_Q 16
#endif
constexpr int MAXL = RT_MAXL;    // leaf-candidate slots per lane in traversal mode (flushed when full)
constexpr int MINL = RT_MINL;
constexpr int MAXL_EXACT = 7;     // exact-node kernel: fixed (see the kernel)
// bias of the LDS-tree kernel's node references: reference 0x8000 = the dword behind node DONE (see the staging code)
__host__ __device__ inline uint32_t lt_r0(uint32_t n_internal) { return 0x8000u - (n_internal + 1u) * 19u; }
constexpr int LNODE_DW = 19;      // LDS-tree kernel: dwords per staged node (see the staging code); odd, so that the
                                  // nodes start on all 32 banks
#ifndef RT_LT_PARTIAL           // LDS-tree kernel: lanes with a pending candidate that make a partial root-test round (64: never)
#define RT_LT_PARTIAL 40
#endif
#ifndef RT_LT_EAGER             // culled LDS-tree kernel: lanes with candidates but no hit yet that make a partial root-test round
#define RT_LT_EAGER 12
#endif
#ifndef RT_MAXL_LTREE
#define RT_MAXL_LTREE 12
#endif
constexpr int MAXL_LTREE = RT_MAXL_LTREE;     // LDS-tree kernel (16-bit entries): a block of RT_STEPS_PER_CHECK_LTREE appends always fits;
constexpr int MAXL_LTREE_MAX = 16;            // the host gives a tree that leaves room up to this many slots (KParams::maxl; c3 14: +0.5 %)
constexpr uint32_t LEAF_BIT = 0x80000000u;
// Output staging (north_star: "coalesced HBM stores of the tile"): a wave collects the RGB8 bytes of up to STAGE_SLOTS of
// its 64x1 tiles in LDS and writes a finished tile as 48 whole dwords = three whole 64-byte lines.  Byte stores of
// single pixels reached HBM as partial lines: 1.3x (c3) to 13x (c5) write amplification (profiles/r01_*, r02_*).
constexpr uint32_t STAGE_TILE_BYTES = 192;

// ---- Sample units (round 4).  The unit of work a lane takes is ONE SAMPLE of a pixel.  A wave keeps up to
// n_slots PIXEL SLOTS; a slot holds a group of `grp` neighbouring pixels of a tile (one pixel from 8 samples per pixel up) = grp x spp
// units in pixel-major order, and owns a header plus one 12-byte record per unit in the wave's scratch (global memory, L2-resident:
// a slot is reused as soon as it is free).  Lanes take the units of the open slot, then of the next free one; a finished unit's
// colour goes to its record and counts the slot's LDS counter down.  A slot whose counter has reached zero is COMMITTED — out of
// order with respect to other slots, which is what keeps a long path from holding back anything but its own pixel: each of its
// pixels is summed s = 0 .. spp-1 (the f32 sum order of main.rs:73-77), then mean, gamma, quantise — and freed.
constexpr uint32_t SLOTS_MAX = 32;                // pixel slots per wave
constexpr uint32_t SLOT_FREE = 0x80000000u;       // counter value of a free slot
constexpr uint32_t STAGE_TILES = 3;               // output staging: tiles a wave may have open
struct WaveQ {
    uint32_t cnt[SLOTS_MAX];                     // per slot: bits 0-12 units not yet finished (0: complete, waiting for its commit), bits 13-30 the ray
                                                 //   segments its finished units traced (per-strip cost), bit 31 = SLOT_FREE
    // per-strip cost (KParams::strip_cost): ray segments of committed slots not yet added to the launch's array, for TWO strips — [0] the
    // one the wave issues from, [1] the one before it (slots of both are in flight when the wave crosses a strip boundary)
    uint32_t cost_acc[2];
    uint32_t cost_strip[2];
};
static_assert(sizeof(WaveQ) == 144, "WaveQ layout (rt_api.hip: LDS_LIMIT leaves 3 KiB of static LDS)");
constexpr uint32_t COST_COPIES = 16;
constexpr uint32_t SLOT_UNIT_BITS = 13;           // units of a slot < 2^13 (spp <= RT_MAX_SPP = 4096), segments of a slot < 2^18 (x 63 bounces)
struct WaveStage {                               // kernels with output staging only
    int left[STAGE_TILES + 1];                   // pixels of the staged tile not yet committed (< 0: stage slot free)
    uint32_t dst_lo[STAGE_TILES + 1], dst_hi[STAGE_TILES + 1];   // the tile's first byte in the strip
};

struct StripDesc {
    uint64_t seed;
    uint8_t* rgb;                // [Hs*W*3]
    float* f32;                  // optional [Hs*W*3]
    uint32_t y0;                 // first global row of the strip = Hs * division_no
    uint32_t pad;
};

struct KParams {
    uint32_t W, H, Hs;           // image size, rows per strip
    uint32_t spp, depth;         // samples per pixel; ray_color entry depth = max_bounces+1
    uint32_t n_sph, n_sph_pad;   // spheres, padded to UNROLL with never-hit dummies
    uint32_t n_tri;
    uint32_t chunk;              // spheres per LDS chunk actually used (multiple of UNROLL)
    uint32_t n_chunks;
    uint32_t flags;
    uint32_t path32;             // 1: path stack entries are u32, 0: u16
    uint32_t lds_cand_off;       // byte offsets into dynamic LDS
    uint32_t lds_path_off;
    uint32_t lds_rr_off;
    uint32_t lds_stack_off;      // traversal engine: per-lane stack, (bvh depth + 1) x 256 x u32
    uint32_t lds_cmp_off;        // compacted root tests (ISECT 2): 1 KiB per wave (offsets, distances, roots, hit flags); 0xffffffff: per-lane flush
    uint32_t lds_stage_off;      // output staging (STAGE_TILE_BYTES per wave); 0xffffffff: every pixel is stored directly
    uint32_t n_strips;           // strips in this launch
    uint32_t tiles_x, tiles_per_strip;            // tiles of 64x1 pixels (three whole 64-B lines of RGB8 per tile)
    uint32_t tiles_total;        // tiles of the launch (tiles_per_strip * n_strips)
    uint32_t tiles_big;          // the first tiles_big entries of the launch's queue are whole tiles, every later one a PART of one: a quarter
                                 // (16 pixels; sub_shift 2) or a sixteenth (4 pixels; sub_shift 4: many samples per pixel)
    uint32_t sub_shift;
                                 //   of one of the remaining tiles: the launch ends on small pieces (its tail is one piece long)
    uint32_t n_tiles;            // queue entries: tiles_big + ((tiles_total - tiles_big) << sub_shift)
    uint32_t n_slots;            // sample units: pixel slots per wave (<= SLOTS_MAX)
    uint32_t grp;                //   pixels per slot (1 from 8 spp up; 8 / spp below, so that a slot is at least 8 units)
    uint32_t grp_magic;          //   floor(2^32 / grp) + 1 (grp > 1)
    uint32_t slot_stride;        //   12-byte records per slot in the scratch: header + grp x spp
    uint32_t commit_slots;       //   a commit is worth its instructions once this many slots are complete
    uint32_t spp_magic;          //   floor(2^32 / spp) + 1: q / spp == mulhi(q, magic) for q < 65 * spp (spp <= RT_MAX_SPP = 4096); 0 for spp 1
    uint32_t slotu_magic;        //   floor(2^32 / U) + 1, U = grp * spp the units of a full slot: q / U == mulhi(q, magic) for q < 65 * U
    float* ring;                 //   [waves of the grid][n_slots][slot_stride][3]: slot header (x, row, meta), then (r, g, b) per unit
    float org[3], llc[3], hor[3], ver[3];   // Camera::new (camera.rs:19-47), host-computed
    float lens_radius, focus_distance;
    float u_den, v_den;          // aspect*H_f - 1, H_f - 1 (camera.rs:115-117)
    float t_min, t_max;
    float spp_f;
    float spp_rcp;               // 1 / spp when spp is a power of two (then x / spp == x * spp_rcp bit for bit), else 0
    const float4* geom_pk;       // [n_sph_pad/2][2]: (c0x,c1x,c0y,c1y) (c0z,c1z,rr0,rr1)
    const float4* geom_px;       // expanded form: (c0x,c1x,c0y,c1y) (c0z,c1z,w0,w1),
                                 //   w = |c|^2 - rr - 2^-16 (|c|^2 + rr)
    const float4* geom;          // [n_sph_pad] (cx,cy,cz, RN(r*r))
    const float4* mat;           // [n_sph+n_tri] (albedo r,g,b, roughness)
    const float* emis;           // [n_sph+n_tri]
    const float* tri;            // [n_tri*9] a,b,c
    const float4* tri_box;       // [2*n_tri] Triangle::aabb (lo, hi) as the BVH sees it
    const float4* trav;          // [4*n_internal] rtbvh::TravNode: (l_lo, left)(l_hi, right)(r_lo,-)(r_hi,-)
    const uint4* travq;          // [2*n_internal] rtbvh::QNode: 12 x u16 grid coordinates, left, right
    const float4* geom_r;        // [n_sph] (cx,cy,cz, radius): exact Sphere::aabb on the fly for leaf validation
    float q_base[3], q_step[3], q_rstep[3];   // grid: coordinate = q_base + q * q_step; q_rstep = 1 / q_step
    uint32_t root_ref;           // root reference (LEAF_BIT | prim when the tree is a single leaf)
    uint32_t maxl;               // traversal: leaf-list slots per lane (MINL..MAXL)
    uint32_t list16;             // L2-gather traversal kernels: leaf-list entries are 16-bit (scenes of <= 65536 primitives)
    uint32_t stack_lds;          // capped quantised-node kernel: stack entries per lane kept in LDS, deeper ones go to stack_ovf
    uint32_t ovf_stride;         //   threads in the grid (stride of the overflow area)
    uint32_t* stack_ovf;         //   [entries beyond stack_lds][ovf_stride]
    const uint32_t* big;         // culled walk (ISECT 7): [n_big] spheres too large for the culling slack, root-tested at query start
    uint32_t n_big;
    float r_slack;               //   largest radius among the spheres NOT in big[]
    float tri_k, tri_diag, tri_es, tri_e;   // culled walk over the exact nodes (ISECT 9), triangles NOT in big[]: largest |e1||e2|, largest box
                                 //   diagonal, largest |e1| + |e2|, largest edge (cull_bound_tri); all 0 without triangles
    uint32_t refill_eighths;     // traversal: finished lanes are refilled once <= this many eighths of the live lanes still walk
    uint32_t n_internal;         // internal nodes of the tree (= TravNode count)
    uint32_t lds_node_off;       // LDS-resident tree (ISECT 5): byte offsets of the staged nodes ...
    const float4* bvh_nodes;     // [2*n_nodes]: (lo.xyz, parent as bits) (hi.xyz, -) — rt_bvh.h FlatNode
    const uint32_t* leaf_of;     // [n_sph+n_tri] primitive -> leaf node index (= DFS rank)
    const uint32_t* world_rank;  // [n_sph+n_tri] primitive -> position in RenderInfo.world, or nullptr (= primitive order): the
                                 //   tie order of plain linear-scan semantics (RT_FLAG_NO_BVH_CULL)
    unsigned long long* strip_cost;  // optional [COST_COPIES][MAX_BATCH]: ray segments per strip of the batch (what the frame context balances its
                                 //   devices by), in COST_COPIES partial sums (workgroup % COST_COPIES) so that the waves' atomics do not all
                                 //   meet on one address per strip
    unsigned long long* counters;// [0] segments [1] candidates [2] fallbacks
    unsigned long long* queue;   // tile queue head of this launch (zeroed on the stream before it)
    StripDesc strips[MAX_BATCH];
};

// ------------------------------------------------------------------ vector helpers
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// sqrtf, correctly rounded as IEEE 754 demands, for less than the compiler's sequence.  That sequence is v_sqrt_f32, two one-ulp
// residual tests and, around them, a scaling of small operands and a class fix-up: 16 instructions, nine of them in the 4-cycle class
// (tools/ubench/valu_classes).  Here: v_rsq_f32 and one coupled Newton step — seven fast-class instructions — which gives the
// correctly rounded root for EVERY operand of magnitude 2^-96 ... below infinity (tools/ubench/ieee_cores.hip compares all 1.88e9 of
// them with __builtin_sqrtf on the device; a negative operand gives NaN either way).  Lanes with any other operand (zero, tiny,
// infinite, NaN) take the compiler's sequence behind a branch that is skipped when no lane of the wave needs it.
// -DRT_IEEE_SQRT_PLAIN: the compiler's sequence everywhere (A/B runs).
// NEG_OK: a negative operand of in-range magnitude stays on the fast path (NaN from either sequence; for results that only feed
// comparisons — the discriminant of a miss); otherwise negative operands take the compiler's sequence too, whose NaN they keep.
template <bool NEG_OK = false>
__device__ __forceinline__ float sqrt_rn(float x) {
#ifdef RT_IEEE_SQRT_PLAIN
    return __builtin_sqrtf(x);
#else
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    const float dd = __builtin_fmaf(-g, g, x);
    float y = __builtin_fmaf(dd, h, g);
    const bool odd = ((__float_as_uint(x) & (NEG_OK ? 0x7fffffffu : 0xffffffffu)) - 0x0f800000u) >= (0x7f800000u - 0x0f800000u);
    if (__ballot(odd)) {
        if (odd) {
            asm volatile("" : "+v"(x));        // (not to be speculated: without it the compiler computes both sequences and selects)
            y = __builtin_sqrtf(x);
        }
    }
    return y;
#endif
}
#define RT_SQRT(x_) sqrt_rn(x_)
#define RT_SQRT_NEG_OK(x_) sqrt_rn<true>(x_)
#define RT_DIV(a_, b_) ((a_) / (b_))
// Hooks of the timing probes (tools/probes/rt_probes.h: extra loads / ALU work per node step, bare hardware sqrt / rcp — some of them
// render WRONG images by design).  They expand to nothing here; a measurement build adds -DRT_PROBES -Itools/probes, which no product
// build does (ray_tracer_s8_amd/build.py records every library's flags, bench.py and the tests refuse a non-default record).
#define RT_HOOK_LT_STEP_LOADS(nd_, ni_, lnodes_)
#define RT_HOOK_LT_STEP_ALU(p0_, p1_, p2_, p3_, oy_)
#define RT_HOOK_LT_STEP_END
#define RT_HOOK_Q_GATHER(t_ref_, cl_, travq_, n_internal_)
#ifdef RT_PROBES
#include "rt_probes.h"
#endif
__device__ __forceinline__ V3 operator/(V3 a, float s) { return {RT_DIV(a.x, s), RT_DIV(a.y, s), RT_DIV(a.z, s)}; }
// glam sse2 dot3 order: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float vlength(V3 a) { return RT_SQRT(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { return a / vlength(a); }   // glam normalize: divide
__device__ __forceinline__ bool try_normalize(V3 a, V3& out) {            // glam try_normalize
    float rcp = RT_DIV(1.0f, RT_SQRT(dot(a, a)));
    if (rcp > 0.0f && rcp < __builtin_inff()) {   // is_finite() && > 0
        out = a * rcp;
        return true;
    }
    return false;
}
__device__ __forceinline__ V3 normalize_or_zero(V3 a) {
    V3 r;
    return try_normalize(a, r) ? r : mk(0.f, 0.f, 0.f);
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ------------------------------------------------------------------ RNG: rand 0.8.5 SmallRng
struct Rng {
    uint64_t s0, s1, s2, s3;
};
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ uint32_t next_u32(Rng& r) {   // xoshiro256++ next_u64 >> 32
    uint64_t result = rotl64(r.s0 + r.s3, 23) + r.s0;
    uint64_t t = r.s1 << 17;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return (uint32_t)(result >> 32);
}
__device__ __forceinline__ uint64_t splitmix_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
constexpr uint64_t PHI = 0x9e3779b97f4a7c15ull;
// DESIGN.md "RNG": sample s of pixel p (global index) of a job with S samples per pixel draws from
// SmallRng::seed_from_u64(job_seed + 4 * PHI * (p * S + s)) — block p * S + s of four consecutive SplitMix64 outputs of the job seed
__device__ __forceinline__ Rng seed_state(uint64_t st) {               // st = job_seed + 4 * PHI * (p * S + s)
    Rng r;                                                              // seed_from_u64
    st += PHI; r.s0 = splitmix_mix(st);
    st += PHI; r.s1 = splitmix_mix(st);
    st += PHI; r.s2 = splitmix_mix(st);
    st += PHI; r.s3 = splitmix_mix(st);
    return r;
}
__device__ __forceinline__ float u01(Rng& r) {           // [1,2) mantissa trick, minus 1
    return __uint_as_float((next_u32(r) >> 9) | 0x3f800000u) - 1.0f;
}
__device__ __forceinline__ float gen_range_01(Rng& r) { return u01(r) * 1.0f + 0.0f; }
__device__ __forceinline__ float uniform_m1_1(Rng& r) { return u01(r) * 2.0f + -1.0f; }

// ------------------------------------------------------------------ exact sphere test
// sphere.rs:42-47 -> roots::find_roots_quadratic(1, b, c) -> shapes/mod.rs:106-129.
// c = center, rr = RN(r*r).  td = 2*d.  Returns true and t when a root lies in [t_min, t_max).
__device__ __forceinline__ bool exact_sphere(V3 o, V3 td, V3 cen, float rr, float t_min, float t_max,
                                             float& t_out) {
    // Straight-line form: every lane computes both quotient roots and selects.  As nested branches (disc < 0, disc == 0,
    // sign of b, the two "do not use the smallest divisor" tests, the ordering, the window cases) this was 8 branches and 13
    // exec-mask regions per candidate — more scalar bookkeeping than arithmetic.  The values selected are
    // exactly those of the nested form; a NaN discriminant (non-finite operands) fails every comparison below, as there.
    const V3 oc = o - cen;
    const float b = dot(td, oc);
    const float len = RT_SQRT(dot(oc, oc));
    const float c = len * len - rr;
    const float disc = b * b - 4.0f * c;            // a1*a1 - _4*a2*a0, a2 = 1
    const float sq = RT_SQRT_NEG_OK(disc);         // (NaN for a negative discriminant: that lane reports a miss below)
    const bool bneg = b < 0.0f;
    const float same_sign = bneg ? -b + sq : -b - sq;
    const float diff_sign = bneg ? -b - sq : -b + sq;
    const float a0x2 = 2.0f * c;
    const bool big_s = __builtin_fabsf(same_sign) > 2.0f, big_d = __builtin_fabsf(diff_sign) > 2.0f;
    const float q1 = RT_DIV(a0x2, same_sign), q2 = RT_DIV(a0x2, diff_sign), hs = same_sign / 2.0f, hd = diff_sign / 2.0f;
    const float x1 = big_s ? q1 : hd;
    const float x2 = big_s ? (big_d ? q2 : hs) : hs;
    const bool one = disc == 0.0f;                  // Roots::One([-a1 / (2 a2)])
    const float r1 = -b / 2.0f;
    const float x = one ? r1 : (x1 < x2 ? x1 : x2);
    const float y = one ? r1 : (x1 < x2 ? x2 : x1);
    const bool xin = (x >= t_min) && (x < t_max);
    const bool yin = (y >= t_min) && (y < t_max);
    // Two([x, y]): both in range -> the smaller, else the one in range; One([x]): x if in range (shapes/mod.rs:106-129)
    t_out = (xin && yin) ? (x < y ? x : y) : (xin ? x : y);
    return !(disc < 0.0f) && (xin || yin);
}

// mesh.rs:109-161 (two-sided Moller-Trumbore) -> Roots::One([dist]) -> shapes/mod.rs:109-115
__device__ __forceinline__ bool exact_triangle(V3 o, V3 d, const float* __restrict__ tv, float t_min, float t_max,
                                               float& t_out) {
    const float EPSILON = 0.00001f;
    V3 A = mk(tv[0], tv[1], tv[2]), B = mk(tv[3], tv[4], tv[5]), C = mk(tv[6], tv[7], tv[8]);
    V3 a_to_b = B - A;
    V3 a_to_c = C - A;
    V3 u_vec = cross(d, a_to_c);
    float det = dot(a_to_b, u_vec);
    if (det < EPSILON && det > -EPSILON) return false;
    float inv_det = 1.0f / det;
    V3 a_to_origin = o - A;
    float u = dot(a_to_origin, u_vec) * inv_det;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    V3 v_vec = cross(a_to_origin, a_to_b);
    float v = dot(d, v_vec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float dist = dot(a_to_c, v_vec) * inv_det;
    if (!(dist > EPSILON)) return false;
    t_out = dist;
    return (dist >= t_min) && (dist < t_max);
}

// bvh::ray::Ray cached values (ray.rs:133-143) needed by intersects_aabb
struct RayAux {
    V3 inv;          // 1 / direction
    bool sx, sy, sz; // direction < 0
    bool full_chain; // RT_FLAG_FULL_CHAIN: never use the leaf-box shortcut (A/B testing)
    bool finite;     // all three inverse components finite (and the flag above clear)
};
__device__ __forceinline__ RayAux ray_aux(V3 d, bool full_chain) {
    RayAux a;
    a.full_chain = full_chain;
    a.inv = mk(RT_DIV(1.0f, d.x), RT_DIV(1.0f, d.y), RT_DIV(1.0f, d.z));
    a.sx = d.x < 0.0f;
    a.sy = d.y < 0.0f;
    a.sz = d.z < 0.0f;
    const float big = __builtin_inff();
    a.finite = !full_chain && __builtin_fabsf(a.inv.x) < big && __builtin_fabsf(a.inv.y) < big &&
               __builtin_fabsf(a.inv.z) < big;
    return a;
}
// ray.rs:81-112: min/max with `if x < y {x} else {y}` semantics (not IEEE minNum)
__device__ __forceinline__ float rmin(float x, float y) { return x < y ? x : y; }
__device__ __forceinline__ float rmax(float x, float y) { return x > y ? x : y; }
// Ray::intersects_aabb (ray.rs:174-194)
__device__ __forceinline__ bool intersects_aabb(V3 o, const RayAux& a, float4 lo, float4 hi) {
    float ray_min = ((a.sx ? hi.x : lo.x) - o.x) * a.inv.x;
    float ray_max = ((a.sx ? lo.x : hi.x) - o.x) * a.inv.x;
    float y_min = ((a.sy ? hi.y : lo.y) - o.y) * a.inv.y;
    float y_max = ((a.sy ? lo.y : hi.y) - o.y) * a.inv.y;
    ray_min = rmax(ray_min, y_min);
    ray_max = rmin(ray_max, y_max);
    float z_min = ((a.sz ? hi.z : lo.z) - o.z) * a.inv.z;
    float z_max = ((a.sz ? lo.z : hi.z) - o.z) * a.inv.z;
    ray_min = rmax(ray_min, z_min);
    ray_max = rmin(ray_max, z_max);
    return rmax(ray_min, 0.0f) <= ray_max;
}
// The same test for a ray whose inverse direction is finite (RayAux::finite): then inv has the sign of the
// direction, lo <= hi gives (lo-o)*inv <= (hi-o)*inv for inv > 0 and >= for inv < 0 under monotone rounding, so
// the sign-selected ray_min / ray_max are the min / max of the two products, no NaN can arise (no 0*inf), and
// min/max differ from the crate's `if x < y` forms only in the sign of a zero, which no comparison sees.
__device__ __forceinline__ bool intersects_aabb_finite(V3 o, const RayAux& a, float4 lo, float4 hi) {
    const float x0 = (lo.x - o.x) * a.inv.x, x1 = (hi.x - o.x) * a.inv.x;
    const float y0 = (lo.y - o.y) * a.inv.y, y1 = (hi.y - o.y) * a.inv.y;
    const float z0 = (lo.z - o.z) * a.inv.z, z1 = (hi.z - o.z) * a.inv.z;
    const float ray_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)),
                                          __builtin_fminf(z0, z1));
    const float ray_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)),
                                          __builtin_fmaxf(z0, z1));
    return __builtin_fmaxf(ray_min, 0.0f) <= ray_max;
}
// intersects_aabb_finite with the slab entry and exit returned (the culled walk over the exact nodes orders and skips by them)
__device__ __forceinline__ void slabs_finite(V3 o, const RayAux& a, float4 lo, float4 hi, float& ray_min, float& ray_max) {
    const float x0 = (lo.x - o.x) * a.inv.x, x1 = (hi.x - o.x) * a.inv.x;
    const float y0 = (lo.y - o.y) * a.inv.y, y1 = (hi.y - o.y) * a.inv.y;
    const float z0 = (lo.z - o.z) * a.inv.z, z1 = (hi.z - o.z) * a.inv.z;
    ray_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    ray_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
}
// Would BVH::traverse (bvh_impl.rs:373-398) have returned this primitive?  Every node on the
// leaf's path to the root must pass the AABB test its parent stores for it.
//
// Shortcut: every ancestor box contains the leaf's box exactly (f32 min/max joins do not round), and
// with a FINITE inverse direction each step of intersects_aabb is monotone under IEEE rounding:
// lo_a <= lo  =>  RN(lo_a - o) <= RN(lo - o)  =>  RN(.. * inv) ordered by the sign of inv, which is the
// sign the slab selection uses; the custom min/max are monotone for non-NaN operands and no 0*inf can
// occur.  Hence ray_min_a <= ray_min_leaf and ray_max_a >= ray_max_leaf: if the leaf's own box passes,
// every ancestor passes.  Only when a direction component is +-0 (inv = +-inf) is the chain walked.
__device__ __forceinline__ bool bvh_reaches(const float4* __restrict__ nodes, uint32_t node, V3 o, const RayAux& a) {
    const bool finite_inv = a.finite;
    for (;;) {
        const float4 lo = nodes[2 * (size_t)node];
        const uint32_t parent = __float_as_uint(lo.w);
        if (parent == 0xffffffffu) return true;              // root: no test (N = 1: always a candidate)
        const float4 hi = nodes[2 * (size_t)node + 1];
        if (!intersects_aabb(o, a, lo, hi)) return false;
        if (finite_inv) return true;                         // leaf box passed => all ancestors pass
        node = parent;
    }
}

// closest-hit bookkeeping: min_by on |P - origin| over the traversal output; the FIRST minimum in
// DFS leaf order wins, NaN keeps the running one (shapes/mod.rs:177-182)
struct Hit {
    int idx;
    float dist;
    float t;       // the root; the hit point is formed again where it is used, P = o + t * d (Ray::at): the same operation on
                   // the same operands gives the same bits, and the walk carries one register instead of three
};
// MODE 0: plain index order (RT_FLAG_NO_BVH_CULL).  MODE 1: BVH semantics, chain validation deferred
// to the winner (ties still go to the earlier DFS leaf).  MODE 2: BVH semantics, every improving hit is
// validated at once.
template <int MODE>
__device__ __forceinline__ void consider(Hit& h, int idx, V3 o, V3 d, float t, const RayAux& a,
                                         const float4* __restrict__ nodes, const uint32_t* __restrict__ leaf_of) {
    V3 p = o + t * d;                    // Ray::at (ray.rs:147-149)
    float dist = vlength(p - o);
    const bool better = h.idx < 0 || h.dist > dist;
    const bool tie = h.idx >= 0 && h.dist == dist;
    if (!(better || tie)) return;
    if (MODE == 0) {
        if (!better) return;                                 // index order: first minimum wins
    } else if (MODE == 1) {
        if (!better && leaf_of[idx] > leaf_of[h.idx]) return;   // equal distance: earlier DFS leaf wins
    } else {
        const uint32_t rank = leaf_of[idx];
        if (!better && rank > leaf_of[h.idx]) return;
        if (!bvh_reaches(nodes, rank, o, a)) return;         // the reference never saw this primitive
    }
    h.idx = idx;
    h.dist = dist;
    h.t = t;
}

// index-order first minimum over the primitives that satisfy `admitted` (evaluated only for an improving hit)
// RANKED: the candidates arrive in any order (culled walk), so an equal distance goes to the earlier depth-first leaf explicitly.
template <bool RANKED = false, class Pred>
__device__ __forceinline__ void consider_if(Hit& h, int idx, V3 o, V3 d, float t, Pred admitted, const uint32_t* __restrict__ leaf_of = nullptr) {
    V3 p = o + t * d;
    float dist = vlength(p - o);
    if (!(h.idx < 0 || h.dist > dist)) {
        if (!(RANKED && h.dist == dist && leaf_of[idx] < leaf_of[h.idx])) return;
    }
    if (!admitted()) return;
    h.idx = idx;
    h.dist = dist;
    h.t = t;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Rust `as u8` from f32: truncate, saturate, NaN -> 0
__device__ __forceinline__ uint8_t f32_as_u8(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}

// Phase census for tuning (compile with -DRT_PROFILE_PHASES; tools/phase_census.py): one count per wave
// each time the code is reached by at least one lane.  Counters live in the spare queue slots [8192..].
#ifdef RT_PROFILE_PHASES
#define WCOUNT(slot)                                                                         \
    do {                                                                                     \
        unsigned long long _m = __ballot(1);                                                 \
        if ((int)(threadIdx.x & 63) == (int)__builtin_ctzll(_m)) atomicAdd(&p.counters[4 + 8192 + (slot)], 1ull); \
    } while (0)
#define LCOUNT(slot)                                                                         \
    do {                                                                                     \
        unsigned long long _m = __ballot(1);                                                 \
        if ((int)(threadIdx.x & 63) == (int)__builtin_ctzll(_m)) {                           \
            atomicAdd(&p.counters[4 + 8192 + 32 + (slot)], (unsigned long long)__builtin_popcountll(_m)); \
            atomicAdd(&p.counters[4 + 8192 + 64 + (slot)], 1ull);                            \
        }                                                                                    \
    } while (0)
#else
#define WCOUNT(slot) do { } while (0)
#define LCOUNT(slot) do { } while (0)
#endif
// Phase clock for tuning (compile with -DRT_PROFILE_TIME; tools/phase_time.py): wave cycles (s_memtime) between
// consecutive stamps are charged to the phase named by the stamp that ends the interval; one total per phase in
// a block of the wave's own (TFLUSH).
#ifdef RT_PROFILE_TIME
#define TDECL unsigned long long _tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long _tlast = __builtin_amdgcn_s_memtime(); const unsigned long long _t0 = __builtin_amdgcn_s_memrealtime(); unsigned long long _tq = 0, _nq = 0
#define TDRAINED do { _tq = __builtin_amdgcn_s_memrealtime(); } while (0)
// (+ for every 20th wave, the start of each of its first 16 rounds after the queue ran dry and the lanes that still hold a unit: [4400 + (w / 20) * 16 + round])
#define TROUND do { if (_tq) { const uint32_t _w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); const unsigned long long _lv = __builtin_popcountll(__ballot(have_unit)); \
    if ((threadIdx.x & 63) == 0 && _w % 20u == 0u && _w / 20u < 215u && _nq < 16ull) p.counters[4 + 8192 + 4400 + (_w / 20u) * 16u + _nq] = (__builtin_amdgcn_s_memrealtime() & 0xfffffffull) | (_lv << 28); \
    _nq++; } } while (0)
#define TSTAMP(ph) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); _tacc[ph] += _n - _tlast; _tlast = _n; } while (0)
// At its end every wave writes ONE block of 16 words of its own behind the queue slots — [4 + 16384 + 16 w ..]: the eight phase totals, its
// end, start and queue-empty times (s_memrealtime: the 100 MHz counter all XCDs share; s_memtime runs per XCD) and its rounds after the
// queue ran dry.  Plain stores: the first version added them to shared words, and 50 000 same-address atomics at the end of a launch held
// up the very rounds they were timing (drain rounds of 100-200 us among 10 us ones).
#define TFLUSH do { if ((threadIdx.x & 63) == 0) { const uint32_t _w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if (_w < 8192u) { unsigned long long* _b = p.counters + 4 + 16384 + 16u * _w; for (int _i = 0; _i < 8; _i++) _b[_i] = _tacc[_i]; \
    _b[8] = __builtin_amdgcn_s_memrealtime(); _b[9] = _t0; _b[10] = _tq; _b[11] = _nq; } } } while (0)
#else
#define TDECL do { } while (0)
#define TDRAINED do { } while (0)
#define TROUND do { } while (0)
#define TSTAMP(ph) do { } while (0)
#define TFLUSH do { } while (0)
#endif

// The sphere record of the traversal kernels' root tests: (centre, RN(r * r)), formed from the (centre, radius) array that the leaf
// validation and the culled walk read anyway — one rounded multiply, the host's own `radius * radius` (sphere.rs:45 `radius.powi(2)`)
// — so that a large scene keeps ONE 16-byte record per sphere in L2 instead of two (round 3: c5's hot set is 2 MB of nodes + 1 MB of
// materials + these records against 4 MB of L2 per XCD; c5 4 460 -> 4 710 Mrays/s).  -DRT_GEOM_RR reads the (centre, r^2) array as before.
#ifndef RT_GEOM_RR
#define RT_SPHERE_REC(p_, prim_) ([&]() { const float4 r_ = at32((p_).geom_r, (prim_)); return make_float4(r_.x, r_.y, r_.z, r_.w * r_.w); }())
#else
#define RT_SPHERE_REC(p_, prim_) at32((p_).geom, (prim_))
#endif

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// Element `i` of a device array reached through a 32-bit byte offset from its (wave-uniform) base pointer: one shift /
// multiply and the scalar-base addressing mode instead of 64-bit address arithmetic per lane.  Arrays stay < 4 GiB.
template <class T>
__device__ __forceinline__ const T& at32(const T* __restrict__ base, uint32_t i) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + i * (uint32_t)sizeof(T));
}

// A loop-invariant index used in a COLD place of the loop (the path stack: once per round): opaque to the compiler, so that the address
// built from it is formed where it is used — two instructions a round — instead of living in a vector register across the walk
// (the register-bound quantised kernels spilled exactly such addresses)
// ON only in the kernels that are bound by their registers (the quantised walks, 96 VGPRs = five waves per SIMD: c5 +1.4 %; in the
// others the two instructions cost 1 %).
template <bool ON>
__device__ __forceinline__ int cold(int v) {
    if (ON) asm volatile("" : "+v"(v));
    return v;
}
// A word of the wave's sample scratch, written by another lane of THIS wave in an earlier round (its store has completed: vmcnt(0)).
// Wavefronts of one workgroup share the CU's vector L1, which is write-through: no cache action is needed between them (the compiler's
// memory model for this chip says so in as many words), so these are plain loads the compiler may merge into dwordx3 / dwordx4 —
// RT_COMMIT_AGENT_LOADS=1 restores the agent-scope loads of the first version (every word its own trip past the L1).
#ifndef RT_COMMIT_BATCH          // samples whose records a committing lane asks for in one trip
#define RT_COMMIT_BATCH 8
#endif
#ifndef RT_COMMIT_AGENT_LOADS
#define RT_COMMIT_AGENT_LOADS 0
#endif
template <class T>
__device__ __forceinline__ T ring_load(const T* q) {
#if RT_COMMIT_AGENT_LOADS
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *q;
#endif
}
// a wave-uniform value read through a vector register (LDS): back into a scalar register
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Slab values of the two child boxes of a QNode (rt_bvh.h: centre c and half-extent h per axis, 16-bit grid units) for a ray
// carried in grid units, t(q) = q * ig + cq: per axis and child tc = fma(c, ig, cq) is the slab centre and
// near / far = tc -+ h |ig| — whichever the sign of the direction — so the slab test needs no min / max per plane: three
// fused multiply-adds (|ig| and -h are operand modifiers) instead of two and two min / max, which cost twice a fused
// multiply-add each on this chip.  Real values up to < 0.2 grid unit of rounding (tc as before < 0.15, one more rounding
// of a value below 2^18 + 2^16 units), against >= 1 unit of outward slack in the node: conservative.
__device__ __forceinline__ void qslabs(const uint4 qa, const uint4 qb, V3 ig, V3 cq, float& lmin, float& lmax, float& rmin_, float& rmax_) {
#define RT_Q(word, hi16) ((float)((hi16) ? ((word) >> 16) : ((word) & 0xffffu)))
    const float ax = __builtin_fabsf(ig.x), ay = __builtin_fabsf(ig.y), az = __builtin_fabsf(ig.z);
    const float lcx = __builtin_fmaf(RT_Q(qa.x, 0), ig.x, cq.x), lcy = __builtin_fmaf(RT_Q(qa.x, 1), ig.y, cq.y), lcz = __builtin_fmaf(RT_Q(qa.y, 0), ig.z, cq.z);
    const float lhx = RT_Q(qa.y, 1), lhy = RT_Q(qa.z, 0), lhz = RT_Q(qa.z, 1);
    const float rcx = __builtin_fmaf(RT_Q(qa.w, 0), ig.x, cq.x), rcy = __builtin_fmaf(RT_Q(qa.w, 1), ig.y, cq.y), rcz = __builtin_fmaf(RT_Q(qb.x, 0), ig.z, cq.z);
    const float rhx = RT_Q(qb.x, 1), rhy = RT_Q(qb.y, 0), rhz = RT_Q(qb.y, 1);
#undef RT_Q
    lmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(-lhx, ax, lcx), __builtin_fmaf(-lhy, ay, lcy)), __builtin_fmaf(-lhz, az, lcz));
    lmax = __builtin_fminf(__builtin_fminf(__builtin_fmaf(lhx, ax, lcx), __builtin_fmaf(lhy, ay, lcy)), __builtin_fmaf(lhz, az, lcz));
    rmin_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(-rhx, ax, rcx), __builtin_fmaf(-rhy, ay, rcy)), __builtin_fmaf(-rhz, az, rcz));
    rmax_ = __builtin_fminf(__builtin_fminf(__builtin_fmaf(rhx, ax, rcx), __builtin_fmaf(rhy, ay, rcy)), __builtin_fmaf(rhz, az, rcz));
}

// The distance-culling bounds cull_bound / cull_bound_tri of the culled walks live in rt_cull.h (shared with the CPU harness
// that checks what they claim, tests/test_cull_lemma.py).
__device__ __forceinline__ float cull_bound(float best, V3 o, float r_slack) { return cull_bound(best, o.x, o.y, o.z, r_slack); }
__device__ __forceinline__ float cull_bound_tri(float best, V3 o, float k, float diag, float es, float e) {
    return cull_bound_tri(best, o.x, o.y, o.z, k, diag, es, e);
}

// ------------------------------------------------------------------ the kernel
// ISECT selects the closest-hit engine: 0 = linear scan, scene resident in LDS; 1 = linear scan, scene streamed
// through LDS in chunks; 2 = per-lane traversal of the reference BVH (exact 64-byte nodes); 3 = the same walk over
// 32-byte nodes whose boxes are rounded outwards onto a 16-bit grid, every reached leaf being validated with the
// reference's exact own-leaf AABB test.
template <int ISECT, bool EXPANDED, int BS = BLOCK, bool STATS = false>
__global__ __launch_bounds__(BS, (ISECT == 5 || ISECT == 6) ? RT_MINWAVES_LTREE : ISECT == 9 ? RT_MINWAVES_TRAV : ISECT >= 7 ? RT_MINWAVES_CULL : ISECT >= 3 ? RT_MINWAVES_QTRAV : ISECT == 2 ? RT_MINWAVES_TRAV : RT_MINWAVES) void rt_tile_kernel(const KParams p) {
    constexpr int BLOCK = BS;                        // threads per workgroup = stride of the per-lane LDS arrays
    constexpr bool STREAMED = (ISECT == 1);
    constexpr bool TRAVERSE = (ISECT >= 2);
    constexpr bool QNODES = (ISECT == 3 || ISECT == 4 || ISECT == 7 || ISECT == 8);   // traversal over 32-byte conservatively quantised nodes
    // ISECT 7: ... nearer child first, and a subtree whose box the ray enters beyond the running closest hit (plus a proven
    // slack, cull_bound) is not entered.  Spheres only.  Candidates then arrive out of depth-first order: ties by rank.
    constexpr bool CULL = (ISECT == 6 || ISECT == 7 || ISECT == 8 || ISECT == 9);   // (8: with the capped LDS stack; 9: over the EXACT nodes,
                                                                         //  where triangles may take part — cull_bound_tri; 6: the LDS-resident tree)
    constexpr bool XNODES = (ISECT == 2 || ISECT == 9);    // exact 64-byte nodes gathered from L2
    constexpr bool CAPPED = (ISECT == 4 || ISECT == 8);            // ... whose stack keeps p.stack_lds entries in LDS, deeper ones in HBM
    // ISECT 5: the exact-node walk with the WHOLE tree (and the materials) resident in LDS, one 1024-thread workgroup
    // per CU; references, stack and leaf lists are 16-bit
    constexpr bool LTREE = (ISECT == 5 || ISECT == 6);     // (6: nearer child first + distance culling, round 4)
    constexpr uint32_t LB = LTREE ? 0x8000u : LEAF_BIT;   // leaf flag of a node reference
    // the branch-free node step (see there) pays where a step is bound by the wave's instruction stream = nodes in LDS;
    // the L2-gather engines are bound by the gathers and keep the step that skips them at leaves (c3 -14 %, c5 -11 %)
    constexpr bool BFSTEP = LTREE;
    // leaf-list slots per lane: the exact-node kernel's are fixed (7 KiB lets six of its workgroups share a CU's LDS on
    // c3-class trees; a run-time count cost it 1 %), the quantised kernels' are chosen by the host's LDS plan
    const uint32_t ML = QNODES ? p.maxl : (uint32_t)MAXL_EXACT;   // (LTREE: MAXL_LTREE, see its step)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // the workgroup's share of the tile queue: [next | end << 32], a refill lock, and "the launch's queue is empty"
    __shared__ unsigned long long wg_tiles;
    __shared__ unsigned int wg_lock, wg_drained;
    __shared__ WaveQ wave_q[BS / 64];                    // sample units: per-wave bookkeeping (see WaveQ)
    constexpr bool CAN_STAGE = (ISECT >= 1 && ISECT <= 4) || ISECT >= 7;     // output staging compiled in (see there)
    __shared__ WaveStage wave_st[CAN_STAGE ? BS / 64 : 1];
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(&wave_q[threadIdx.x >> 6]);
        // every slot free
        if ((threadIdx.x & 63u) < SLOTS_MAX) wave_q[threadIdx.x >> 6].cnt[threadIdx.x & 63u] = SLOT_FREE;
        if ((threadIdx.x & 63u) == 0u) {
            wave_q[threadIdx.x >> 6].cost_acc[0] = wave_q[threadIdx.x >> 6].cost_acc[1] = 0u;
            wave_q[threadIdx.x >> 6].cost_strip[0] = wave_q[threadIdx.x >> 6].cost_strip[1] = 0xffffffffu;
        }
        (void)z;
        if (CAN_STAGE && (threadIdx.x & 63u) <= STAGE_TILES) wave_st[threadIdx.x >> 6].left[threadIdx.x & 63u] = -1;
    }
    if (threadIdx.x == 0) {
        wg_tiles = 0ull;
        wg_lock = 0u;
        wg_drained = 0u;
    }
    __syncthreads();
    float4* lgeom = reinterpret_cast<float4*>(lds_raw);          // pair layout, see KParams::geom_pk / geom_px
    const float* lgeomf = reinterpret_cast<const float*>(lds_raw);
    float* lrr = reinterpret_cast<float*>(lds_raw + p.lds_rr_off);   // EXPANDED: exact r^2 per sphere of the chunk
    const float4* __restrict__ gsrc = EXPANDED ? p.geom_px : p.geom_pk;
    uint16_t* lcand = reinterpret_cast<uint16_t*>(lds_raw + p.lds_cand_off);
    unsigned char* lpath = lds_raw + p.lds_path_off;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // Column of this lane in the per-lane arrays of 16-bit entries ([slot][BLOCK]): lanes 0-31 of a wave take the low
    // halves of 32 consecutive dwords, lanes 32-63 the high halves, so the 32 lanes the LDS serves together touch 32
    // different banks whatever their slots (with column = tid, lanes 2k and 2k+1 shared a bank: 2-way conflicts
    // whenever neighbours differed in stack depth)
    const int tid16 = (tid & ~63) | ((tid & 31) << 1) | ((tid >> 5) & 1);

    if (ISECT == 0) {
        // resident scene: stage the whole primitive list into LDS once
        for (uint32_t i = tid; i < p.n_sph_pad; i += BLOCK) lgeom[i] = gsrc[i];
        if (EXPANDED)
            for (uint32_t i = tid; i < p.n_sph_pad; i += BLOCK) lrr[i] = p.geom[i].w;
        __syncthreads();
    }

    if (LTREE) {
        // Stage the whole tree once per workgroup, 76 bytes per node, laid out for SIGN-SELECTED plane fetches: for
        // each axis and child the three dwords (lo, hi, lo), so that a two-dword read at dword offset s = (d.axis < 0)
        // returns (near, far) = (aabb[sign], aabb[1 - sign]) — literally ray.rs:175-176 — with no min / max / select in
        // the step.  Dwords: l.x 0-2, r.x 3-5, l.y 6-8, r.y 9-11, l.z 12-14, r.z 15-17, 18 = left | right << 16.
        // References are 16 bits: a leaf is 0x8000 | primitive, a node is LT_R0 + its offset in dwords, with the bias LT_R0
        // chosen so that the dword after the last node has reference 0x8000.  After the tree comes node DONE (see the step: its
        // left box is all of space and its left child DONE itself, its right box NaN) and then n_prims + 19 dwords of NaN:
        // the address formed from a LEAF reference, base + 4 * (0x8000 + primitive), falls into that field, every 19-dword window of
        // which is a node no ray enters — so a lane at a leaf gathers through the same address arithmetic as any other and
        // fails both slab tests, without a clamp of the reference (round 3; before: one MISS node and min(reference, MISS)).
        float* ln = reinterpret_cast<float*>(lds_raw + p.lds_node_off);
        const uint32_t r0 = lt_r0(p.n_internal);
        for (uint32_t n = tid; n <= p.n_internal; n += BLOCK) {
            float* q = ln + LNODE_DW * n;
            if (n < p.n_internal) {
                const float4 a0 = p.trav[4u * n], a1 = p.trav[4u * n + 1], a2 = p.trav[4u * n + 2], a3 = p.trav[4u * n + 3];
                auto ref16 = [r0](uint32_t r) { return (r & LEAF_BIT) ? (0x8000u | (r & 0x7fffu)) : r0 + r * (uint32_t)LNODE_DW; };
                q[0] = a0.x; q[1] = a1.x; q[2] = a0.x;   q[3] = a2.x; q[4] = a3.x; q[5] = a2.x;
                q[6] = a0.y; q[7] = a1.y; q[8] = a0.y;   q[9] = a2.y; q[10] = a3.y; q[11] = a2.y;
                q[12] = a0.z; q[13] = a1.z; q[14] = a0.z; q[15] = a2.z; q[16] = a3.z; q[17] = a2.z;
                q[18] = __uint_as_float(ref16(__float_as_uint(a0.w)) | (ref16(__float_as_uint(a1.w)) << 16));
            } else {
                const float qn = __builtin_nanf(""), inf = __builtin_inff();
                for (int i = 0; i < 18; i++) q[i] = qn;
                for (int a = 0; a < 3; a++) { q[6 * a] = -inf; q[6 * a + 1] = inf; q[6 * a + 2] = -inf; }   // DONE: the left box (lo, hi, lo) per axis
                q[18] = __uint_as_float((r0 + p.n_internal * (uint32_t)LNODE_DW) * 0x10001u);
            }
        }
        float* nanf_ = ln + LNODE_DW * (p.n_internal + 1u);
        for (uint32_t i = tid; i < p.n_sph + p.n_tri + (uint32_t)LNODE_DW; i += BLOCK) nanf_[i] = __builtin_nanf("");
        __syncthreads();
    }
    // (LTREE: biased by LT_R0 dwords, so that base + 4 * reference is the node's address; a 32-bit LDS address may wrap below zero and back)
    const float* const lnodes = reinterpret_cast<const float*>(lds_raw + p.lds_node_off) - (LTREE ? (int)lt_r0(p.n_internal) : 0);
    const V3 corg = mk(p.org[0], p.org[1], p.org[2]);
    const V3 llc = mk(p.llc[0], p.llc[1], p.llc[2]);
    const V3 hor = mk(p.hor[0], p.hor[1], p.hor[2]);
    const V3 ver = mk(p.ver[0], p.ver[1], p.ver[2]);
    const bool exact_scan = (p.flags & 1u) != 0;
    const bool use_bvh = (p.flags & 2u) == 0;       // RT_FLAG_NO_BVH_CULL clears the reference's AABB-chain validation
    const float KMf = 1.0f - 0x1p-17f;              // broad-phase margin
    const v2f NKM = {-KMf, -KMf};

    // ---- sample units: wave-uniform cursors (identical in every lane; all 64 lanes stay in the loop until the wave is done)
    const uint32_t wave = (uint32_t)tid >> 6;
    // (wave_q is always addressed as the __shared__ array it is: through a volatile reference the accesses became flat loads
    // with a 64-bit address register pair each; the lanes' writes and reads are ordered by wavefront-scope fences instead)
#define wq wave_q[wave]
#define wst wave_st[CAN_STAGE ? wave : 0u]
    float* const ring = p.ring + (size_t)(blockIdx.x * (uint32_t)(BLOCK / 64) + wave) * ((size_t)p.n_slots * p.slot_stride * 3u);
    const uint32_t all_free = p.n_slots >= 32u ? 0xffffffffu : (1u << p.n_slots) - 1u;
    uint32_t freem = all_free;                // the free pixel slots.  The LOWEST free slot is taken first: the slots in use — and with them the
                                              // part of the scratch that L2 has to hold — are the low ones unless a burst of long paths needs more
    uint32_t cur_slot = 0;                    // the open slot: the one the tile's next unit belongs to (unless that unit starts a slot)
    uint32_t tile_u = 0, tile_units = 0;      // issue tile: its next unit, its units (npix * spp); pixel-major: unit = pixel-in-tile * spp + sample
    uint32_t tile_x0 = 0, tile_row = 0, tile_yg = 0;   // ... decoded once: first column, row within the strip, GLOBAL row (main.rs:66-68)
    uint32_t tile_meta = STAGE_TILES << 16;   // ... strip in the batch | stage slot << 16 (STAGE_TILES: its pixels are stored directly)
    uint64_t tile_seed = 0;                   // ... SplitMix64 state of its first unit's stream
    bool q_drained = false;                   // the launch's queue has no tile left for this wave
    // ---- per-lane state
    bool have_unit = false, need_ray = false;
    uint32_t useq = 0;                        // the lane's unit: slot << 24 | unit within the slot
    Rng rng = {0, 0, 0, 0};
    uint32_t depth_left = 0, k = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    bool bounce = false;         // need_ray kind: false = camera ray of a new sample, true = scattered ray
    V3 bn = mk(0, 0, 0);         // bounce: surface normal at the hit
    float brough = 0.f;          // bounce: roughness of the hit material
    // per-lane statistics.  Traversal kernels (register-bound): 32-bit, drained into the 64-bit totals before they can wrap
    using Cnt = typename std::conditional<(ISECT >= 2), uint32_t, unsigned long long>::type;
    Cnt n_seg = 0, n_cand = 0, n_fall = 0;
    Cnt n_int = 0;               // STATS variants (RT_FLAG_COUNT_STEPS): internal nodes visited = pairs of slab tests
    // ---- closest-hit query state.  The linear engines finish a query inside one loop iteration; the traversal
    // engine keeps it across iterations (in_trav) so that lanes whose traversal ended can be refilled while
    // stragglers keep walking.
    Hit h;
    h.idx = -1;
    h.dist = 0.f;
    h.t = 0.f;
    RayAux aux = ray_aux(mk(1.f, 1.f, 1.f), false);
    // The quantised walks carry the ray in grid units and need the inverse direction only where a leaf is validated or a
    // fallback lane walks the exact nodes: it is formed again there (three divisions) instead of living in three
    // registers across the walk (the culled kernel spilled seven registers without this, which showed as 2.5 x the
    // fetch traffic past L2).
    auto AUX = [&]() -> RayAux { return QNODES ? ray_aux(d, (p.flags & 8u) != 0) : aux; };
    // culled walks: nothing entered beyond this distance can beat or tie the hit at distance `best` (spheres: cull_bound;
    // the exact-node variant also holds triangles: cull_bound_tri; a scene of both takes the larger)
    auto far_bound = [&](float best) -> float {
        float b = cull_bound(best, o, p.r_slack);
        if ((ISECT == 9 || ISECT == 6) && p.n_tri) b = __builtin_fmaxf(b, cull_bound_tri(best, o, p.tri_k, p.tri_diag, p.tri_es, p.tri_e));
        return b;
    };
    V3 td = mk(0, 0, 0);                     // linear engines: 2 * d of the current segment
    bool in_trav = false;
    uint32_t t_ref = 0, t_sp = 0, t_cnt = 0;
    uint32_t t_head = 0;                      // LTREE: first list entry not root-tested yet (partial rounds, see the walk loop)
    V3 ig = mk(0, 0, 0), cq = mk(0, 0, 0);   // QNODES: the ray in grid units, t(q) = q * ig + cq
    bool qfin = false;                       // QNODES: this lane may use the quantised boxes
    float t_far = __builtin_inff();          // CULL: no primitive entered beyond this distance can beat the running hit
    uint32_t* lc32 = reinterpret_cast<uint32_t*>(lds_raw + p.lds_cand_off);     // TRAVERSE: leaf candidates (u32)
    uint32_t* lstack = reinterpret_cast<uint32_t*>(lds_raw + p.lds_stack_off);   // TRAVERSE: per-lane stack
    uint32_t sgx = 0, sgy = 0, sgz = 0;      // LTREE: (direction.axis < 0) of the current query, ray.rs:139-141
    uint16_t* lc16 = reinterpret_cast<uint16_t*>(lds_raw + p.lds_cand_off);     // LTREE: both 16-bit
    uint16_t* lstack16 = reinterpret_cast<uint16_t*>(lds_raw + p.lds_stack_off);

    if (BFSTEP) lstack16[tid16] = (uint16_t)(lt_r0(p.n_internal) + p.n_internal * (uint32_t)LNODE_DW);  // stack slot 0: popping an empty stack yields DONE

    auto drain_counters = [&]() {
        const unsigned long long ws = wave_sum(n_seg), wc = wave_sum(n_cand), wf = wave_sum(n_fall);
        const unsigned long long wi = STATS ? wave_sum(n_int) : 0ull;
        if (lane == (int)__builtin_ctzll(__ballot(true))) {
            atomicAdd(&p.counters[0], ws);
            atomicAdd(&p.counters[1], wc);
            atomicAdd(&p.counters[2], wf);
            if (STATS) atomicAdd(&p.counters[3], wi);
        }
        n_seg = n_cand = n_fall = 0;
        n_int = 0;
    };
    // ---- output staging (north_star: "coalesced HBM stores of the tile"): a wave collects the RGB8 bytes of up to STAGE_TILES of
    // its 64x1 tiles in LDS (192 bytes each) and writes a finished tile as 48 dwords = three whole 64-byte lines; a tile that finds
    // no free stage slot is stored pixel by pixel.  Compiled into the kernels whose gathers churn L2 (the L2-gather walks, the
    // streamed scan): there a pixel's line is evicted half written (1.3 x ... 13 x write amplification, profiles/r01_*).  The
    // LDS-tree and resident-scan kernels leave L2 to the frame: their byte stores merge there into whole lines.
    const bool staging = CAN_STAGE && p.lds_stage_off != 0xffffffffu;
    unsigned char* const stage_base = lds_raw + (staging ? p.lds_stage_off + wave * (STAGE_TILES * STAGE_TILE_BYTES) : 0u);
    // tile number -> strip in the batch, first column, row within the strip
    auto decode_tile = [&](uint32_t t, uint32_t& st, uint32_t& x0, uint32_t& row) {
        st = t / p.tiles_per_strip;
        const uint32_t rem = t - st * p.tiles_per_strip;
        row = rem / p.tiles_x;
        x0 = (rem - row * p.tiles_x) << 6;
    };
    // q / spp for q < 65 * spp (a unit's place in its tile -> its pixel): a multiply-high (spp 1: the unit itself)
    auto div_spp = [&](uint32_t q) -> uint32_t { return p.spp_magic ? __umulhi(q, p.spp_magic) : q; };

    TDECL;
    for (;;) {
        WCOUNT(0);
        TSTAMP(5);
        TROUND;
        uint32_t px = 0, pyg = 0;                 // a new unit's pixel: column, GLOBAL row — from the acquisition to the camera ray of this round
        // (the cursors are wave-uniform by construction; every assignment says so — uni() — so that they live in scalar registers
        // across the loop: left to the compiler they travelled through vector registers, a dozen moves per round)
        // ================= commit: complete slots -> pixels (main.rs:73-81)
        if (freem != all_free) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the lanes' deposits of the last round
            const bool slot_done = (uint32_t)lane < SLOTS_MAX && (wq.cnt[cold<QNODES>(lane) & (int)(SLOTS_MAX - 1u)] & (SLOT_FREE | ((1u << SLOT_UNIT_BITS) - 1u))) == 0u;
            const uint32_t complete = (uint32_t)__ballot(slot_done);         // (an open slot's counter still holds its unissued units)
            const uint32_t n_complete = (uint32_t)__builtin_popcount(complete);
            // Worth the instructions?  A commit runs at one lane per pixel, so it waits until commit_slots are complete — unless
            // the wave is about to run out of slots (lanes would go without units) or nothing is left to issue.
            const bool issue_over = q_drained && tile_u == tile_units;
            if (n_complete >= p.commit_slots || (n_complete != 0u && ((uint32_t)__builtin_popcount(freem) < 2u || issue_over))) {
                WCOUNT(14);
                // the colours were stored by other lanes of this wave in earlier rounds: every store has completed before the loads
                // are issued (ring_load: same CU, same write-through L1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t n_lanes = p.n_slots * p.grp;               // lane L <-> pixel L % grp of slot L / grp
                for (uint32_t l0 = 0; l0 < n_lanes; l0 += 64u) {
                    const uint32_t L = l0 + (uint32_t)lane;
                    const uint32_t slot = p.grp == 1u ? L : __umulhi(L, p.grp_magic), g = L - slot * p.grp;      // (grp > 1: below 8 spp only)
                    const bool mine = slot < p.n_slots && ((complete >> (slot & 31u)) & 1u) != 0u;
                    if (__ballot(mine) == 0ull) continue;
                    uint32_t stg = STAGE_TILES, cstrip = 0;
                    bool fin = false;
                    uint32_t pin = 0;
                    if (mine) {
                        const float* sb = ring + __umul24(slot, p.slot_stride) * 3u;
                        // (header and the first samples are asked for together: one trip to L2, not two; a lane beyond the slot's
                        // pixels — the short last slot of a tile — sums records nobody wrote and stores nothing)
                        const uint32_t hx = ring_load(reinterpret_cast<const uint32_t*>(sb) + 0);
                        const uint32_t hrow = ring_load(reinterpret_cast<const uint32_t*>(sb) + 1);
                        const uint32_t hmeta = ring_load(reinterpret_cast<const uint32_t*>(sb) + 2);
                        const float* r = sb + 3u + g * p.spp * 3u;
                        float sum_r = 0.f, sum_g = 0.f, sum_b = 0.f;
                        cstrip = hmeta & 0xffu;
                        {
                            // pix_color += (main.rs:75), s = 0 .. spp - 1; the loads of four samples in flight together
                            uint32_t i = 0;
                            // (batches of RT_COMMIT_BATCH, then of four, then single samples: at the reference's 100 samples per pixel a
                            // pixel's sum is 13 + 1 trips instead of 25)
                            auto batch = [&](auto cb_tag) {
                                constexpr int CB = decltype(cb_tag)::value;
#pragma clang loop unroll(disable)
                                for (; i + (uint32_t)CB <= p.spp; i += (uint32_t)CB) {
                                    LCOUNT(12);
                                    float c[3 * CB];
#pragma unroll
                                    for (int e = 0; e < 3 * CB; e++) c[e] = ring_load(r + e);
#pragma unroll
                                    for (int e = 0; e < CB; e++) {
                                        sum_r = sum_r + c[3 * e + 0];
                                        sum_g = sum_g + c[3 * e + 1];
                                        sum_b = sum_b + c[3 * e + 2];
                                    }
                                    r += 3 * CB;
                                }
                            };
                            if (RT_COMMIT_BATCH > 4 && !QNODES) batch(std::integral_constant<int, RT_COMMIT_BATCH>{});      // (the 96-register quantised kernels would spill)
                            batch(std::integral_constant<int, 4>{});
#pragma clang loop unroll(disable)
                            for (; i < p.spp; i++) {
                                LCOUNT(12);
                                const float cr = ring_load(r + 0);
                                const float cg = ring_load(r + 1);
                                const float cb = ring_load(r + 2);
                                sum_r = sum_r + cr;
                                sum_g = sum_g + cg;
                                sum_b = sum_b + cb;
                                r += 3;
                            }
                        }
                        if (g < ((hmeta >> 8) & 0xffu)) {                    // (the last slot of a ragged tile holds fewer pixels)
                            // ---- mean, gamma, quantise, store (main.rs:78-81)
                            // pix_color / sample_count (main.rs:78-80).  When the sample count is a power of two the quotient IS the
                            // product with its exact reciprocal (one rounding of the same real value either way, subnormal results
                            // included), and a multiply is a tenth of an IEEE division's instructions: wave-uniform choice.
                            float cr_, cg_, cb_;
                            if (p.spp_rcp != 0.0f) {
                                cr_ = __builtin_sqrtf(sum_r * p.spp_rcp);
                                cg_ = __builtin_sqrtf(sum_g * p.spp_rcp);
                                cb_ = __builtin_sqrtf(sum_b * p.spp_rcp);
                            } else {
                                cr_ = __builtin_sqrtf(sum_r / p.spp_f);
                                cg_ = __builtin_sqrtf(sum_g / p.spp_f);
                                cb_ = __builtin_sqrtf(sum_b / p.spp_f);
                            }
                            const uint32_t sidx = hmeta & 0xffu;
                            stg = (hmeta >> 16) & 0xffu;
                            const size_t oidx = ((size_t)hrow * p.W + hx + g) * 3;          // row within the strip
                            if (!CAN_STAGE || stg == STAGE_TILES) {
                                uint8_t* orgb = p.strips[sidx].rgb;
                                orgb[oidx + 0] = f32_as_u8(cr_ * 255.999f);
                                orgb[oidx + 1] = f32_as_u8(cg_ * 255.999f);
                                orgb[oidx + 2] = f32_as_u8(cb_ * 255.999f);
                            } else {
                                uint8_t* sd = stage_base + stg * STAGE_TILE_BYTES + ((hx + g) & 63u) * 3u;
                                sd[0] = f32_as_u8(cr_ * 255.999f);
                                sd[1] = f32_as_u8(cg_ * 255.999f);
                                sd[2] = f32_as_u8(cb_ * 255.999f);
                                fin = true;
                            }
                            float* of = p.strips[sidx].f32;
                            if (of) {
                                of[oidx + 0] = cr_;
                                of[oidx + 1] = cg_;
                                of[oidx + 2] = cb_;
                            }
                            (void)pin;
                        }
                    }
                    if (p.strip_cost && mine && g == 0u) {
                        // the slot's ray segments -> per-strip cost: into the wave's running sum for its strip (LDS: the strip the wave
                        // issues from, or the one before), else — a straggler from further back — straight to the launch's array
                        const uint32_t sg = wq.cnt[slot & (SLOTS_MAX - 1u)] >> SLOT_UNIT_BITS;
                        if (cstrip == wq.cost_strip[0]) (void)__hip_atomic_fetch_add(&wq.cost_acc[0], sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        else if (cstrip == wq.cost_strip[1]) (void)__hip_atomic_fetch_add(&wq.cost_acc[1], sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        else if (sg) atomicAdd(&p.strip_cost[(blockIdx.x % COST_COPIES) * MAX_BATCH + cstrip], (unsigned long long)sg);
                    }
                    if (CAN_STAGE && __ballot(fin)) {
                        // staged tiles that are complete now: LDS -> three whole lines of the strip
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the byte stores of this wave's other lanes
#pragma unroll
                        for (uint32_t k = 0; k < STAGE_TILES; k++) {
                            const unsigned long long mk = __ballot(fin && stg == k);
                            if (mk == 0ull) continue;
                            const int left = (int)uni((uint32_t)wst.left[k]) - (int)__builtin_popcountll(mk);
                            if (left == 0) {
                                if (lane < (int)(STAGE_TILE_BYTES / 4)) {
                                    uint32_t* d32 = reinterpret_cast<uint32_t*>(((uintptr_t)uni(wst.dst_hi[k]) << 32) | uni(wst.dst_lo[k]));
                                    d32[lane] = reinterpret_cast<const uint32_t*>(stage_base + k * STAGE_TILE_BYTES)[lane];
                                }
                            }
                            if (lane == 0) wst.left[k] = left == 0 ? -1 : left;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // reads done before a slot is written again
                    }
                }
                // the committed slots are free again
                if (slot_done) wq.cnt[cold<QNODES>(lane)] = SLOT_FREE;
                freem = uni(freem | complete);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        TSTAMP(6);
        if (TRAVERSE && __ballot(((n_seg | n_cand | n_fall | n_int) & 0x80000000u) != 0)) drain_counters();
        // ================= unit acquisition: lanes without a unit take the next units of the open slot, then of the next free slot
        {
            bool need = !have_unit;
            for (;;) {                                        // (one pass per tile: nearly always one)
                if (__ballot(need) == 0ull) break;
                if (tile_u == tile_units) {                       // wave-uniform: the issue tile is exhausted, fetch the next
                    if (q_drained) break;
                    // Two-level tile queue.  The launch's queue head is shared by all eight XCDs, so every atomic on it
                    // is a memory-side operation (one per tile: 6 MB of "write" traffic per 4K frame on top of 25 MB of
                    // pixels, profiles/).  A workgroup therefore takes WGC = one tile per wave at a time from it and hands
                    // them to its waves through an LDS counter pair: tile-granular balance inside the workgroup, and the
                    // launch tail still one tile per wave long (waves taking private runs of tiles left 10-25 % between
                    // the average and the last wave, tools/ab.sh).
                    uint32_t t = 0xffffffffu;
                    if (lane == 0) {
                        constexpr uint32_t WGC = (uint32_t)(BLOCK / 64) < (uint32_t)RT_QUEUE_TAKE_MIN ? (uint32_t)RT_QUEUE_TAKE_MIN : (uint32_t)(BLOCK / 64);
                        for (;;) {
                            const unsigned long long old = __hip_atomic_fetch_add(&wg_tiles, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if ((uint32_t)old < (uint32_t)(old >> 32)) {
                                t = (uint32_t)old;
                                break;
                            }
                            if (__hip_atomic_load(&wg_drained, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                            if (__hip_atomic_exchange(&wg_lock, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                                // this wave refills (unless another one just did)
                                const unsigned long long cur = __hip_atomic_load(&wg_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                bool done = false;
                                if ((uint32_t)cur >= (uint32_t)(cur >> 32)) {
                                    const uint32_t g = (uint32_t)atomicAdd(p.queue, (unsigned long long)WGC);
                                    if (g >= p.n_tiles) {
                                        __hip_atomic_store(&wg_drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    } else {
                                        t = g;                                    // the first one is this wave's
                                        const uint32_t ge = min(g + WGC, p.n_tiles);
                                        __hip_atomic_store(&wg_tiles, (unsigned long long)(g + 1u) | ((unsigned long long)ge << 32),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                    done = true;
                                }
                                __hip_atomic_store(&wg_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (done) break;
                            } else {
                                __builtin_amdgcn_s_sleep(2);                      // another wave is refilling
                            }
                        }
                    }
                    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                    if (t == 0xffffffffu) {                   // queue drained: the wave ends when its units are committed
                        q_drained = true;
                        TDRAINED;
                        break;
                    }
                    // queue entry -> tile (and quarter of it): whole tiles first, the last ones in quarters
                    uint32_t sub = 0u, tw = 64u;
                    if (t >= p.tiles_big) {
                        const uint32_t e = t - p.tiles_big;
                        t = p.tiles_big + (e >> p.sub_shift);
                        tw = 64u >> p.sub_shift;
                        sub = (e & ((1u << p.sub_shift) - 1u)) * tw;
                    }
#ifndef RT_TILES_TOP_DOWN
                    // Tiles are handed out from the END of the batch backwards: strips are listed top to bottom and the
                    // rows near the top of a frame are mostly sky (one segment per sample), so the launch ends on its
                    // cheapest tiles and the tail in which waves run half empty is shorter.
                    t = p.tiles_total - 1u - t;
#endif
                    // decode the tile once, wave-uniformly: strip, tile origin, global row and seed
                    uint32_t tstrip;
                    decode_tile(t, tstrip, tile_x0, tile_row);
                    if (p.strip_cost && lane == 0 && (wq.cost_strip[0] != tstrip || wq.cost_acc[0] >= 0x80000000u)) {
                        // per-strip cost: the wave issues from another strip now; the sum of the strip before the last goes to the array
                        if (wq.cost_acc[1]) atomicAdd(&p.strip_cost[(blockIdx.x % COST_COPIES) * MAX_BATCH + wq.cost_strip[1]], (unsigned long long)wq.cost_acc[1]);
                        wq.cost_acc[1] = wq.cost_acc[0];
                        wq.cost_strip[1] = wq.cost_strip[0];
                        wq.cost_acc[0] = 0u;
                        wq.cost_strip[0] = tstrip;
                    }
                    tile_x0 = uni(tile_x0 + sub);
                    tile_row = uni(tile_row);
                    if (tile_x0 >= p.W) continue;                 // (a quarter beyond the right edge of a ragged tile: nothing in it)
                    tile_yg = uni(p.strips[tstrip].y0 + tile_row);                // main.rs:66-68
                    // SplitMix64 state of the tile's first stream: seed + 4 PHI * (p * S), p = row * W + x0 the tile's first pixel
                    tile_seed = p.strips[tstrip].seed + (((uint64_t)tile_yg * p.W + tile_x0) * p.spp) * (4ull * PHI);
                    tile_seed = (uint64_t)uni((uint32_t)tile_seed) | ((uint64_t)uni((uint32_t)(tile_seed >> 32)) << 32);
                    const uint32_t npix = min(tw, p.W - tile_x0);
                    tile_units = uni(npix * p.spp);
                    tile_u = 0u;
                    uint32_t tstage = STAGE_TILES;
                    if (staging && npix == 64u) {
                        // a whole 64x1 tile whose 192 bytes start dword-aligned: stage it if a stage slot is free
                        uint8_t* dst = p.strips[tstrip].rgb + ((size_t)tile_row * p.W + tile_x0) * 3;
                        if ((reinterpret_cast<uintptr_t>(dst) & 3u) == 0) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            for (uint32_t k = 0; k < STAGE_TILES; k++)
                                if ((int)uni((uint32_t)wst.left[k]) < 0) {
                                    tstage = k;
                                    break;
                                }
                            if (tstage < STAGE_TILES && lane == 0) {
                                wst.left[tstage] = 64;
                                wst.dst_lo[tstage] = (uint32_t)reinterpret_cast<uintptr_t>(dst);
                                wst.dst_hi[tstage] = (uint32_t)(reinterpret_cast<uintptr_t>(dst) >> 32);
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        }
                    }
                    tile_meta = uni(tstrip | (tstage << 16));
                }
                // ---- The tile's next units, all needy lanes at once: the lane of rank r takes unit tv = tile_u + r.  The tile's units
                // are cut into slots of U = grp * spp (the last one may be short): unit tv belongs to the tile's slot number tv / U.
                // Slots up to the one tile_u - 1 lies in are open already (that one is cur_slot); the others are the lowest free slots in turn,
                // and the lane that takes a slot's first unit sets the slot up (counter, header).
                bool got = false;
                uint32_t tv = 0;
                bool stall = false;
                {
                    const unsigned long long mask = __ballot(need);
                    const uint32_t want = (uint32_t)__builtin_popcountll(mask);
                    const uint32_t U = p.slot_stride - 1u;
                    const uint32_t k_open = __umulhi(tile_u + U - 1u, p.slotu_magic);    // slots of this tile opened so far = ceil(tile_u / U)
                    const uint32_t n_free = (uint32_t)__builtin_popcount(freem);
                    const uint32_t slot_room = (k_open + n_free) * U - tile_u;            // units until the free slots run out
                    const uint32_t take = min(want, min(tile_units - tile_u, slot_room));
                    stall = take < want && take == slot_room;                             // out of slots: wait for commits
                    const uint32_t n_new = __umulhi(tile_u + take + U - 1u, p.slotu_magic) - k_open;   // slots this step opens (wave-uniform)
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    got = need && rank < take;
                    tv = tile_u + rank;
                    const uint32_t k = __umulhi(tv, p.slotu_magic);                       // tv / U
                    const uint32_t unit = tv - __umul24(k, U);
                    uint32_t m = freem;                                                   // the (k - k_open)-th lowest free slot (where k >= k_open)
                    for (uint32_t i = 0; i + 1u < n_new; i++) m = k > k_open + i ? (m & (m - 1u)) : m;
                    const uint32_t slot = k >= k_open ? (uint32_t)__builtin_ctz(m | 0x80000000u) : cur_slot;
                    if (got) {
                        useq = (slot << 24) | unit;
                        if (unit == 0u) {
                            const uint32_t un = min(U, tile_units - tv);                  // its units; its pixels: un / spp
                            wq.cnt[slot] = un;
                            uint32_t* hdr = reinterpret_cast<uint32_t*>(ring + __umul24(slot, p.slot_stride) * 3u);
                            hdr[0] = tile_x0 + __umul24(k, p.grp);
                            hdr[1] = tile_row;
                            hdr[2] = (tile_meta & 0xffu) | (div_spp(un) << 8) | (tile_meta & 0xff0000u);
                        }
                    }
                    // the cursors, wave-uniformly
                    tile_u = uni(tile_u + take);
                    if (n_new) {
                        uint32_t fm = freem;
                        for (uint32_t i = 0; i + 1u < n_new; i++) fm &= fm - 1u;
                        cur_slot = uni((uint32_t)__builtin_ctz(fm));
                        freem = uni(fm & (fm - 1u));
                    }
                }
                TSTAMP(7);
                if (got) {
                    WCOUNT(1);
                    LCOUNT(0);
                    px = tile_x0 + div_spp(tv);                                   // pixel-major: pixel tv / spp, sample tv % spp
                    pyg = tile_yg;
                    // stream p * S + s: the tile's first stream + tv (tile_seed holds that one's SplitMix64 state)
                    rng = seed_state(tile_seed + (uint64_t)tv * (4ull * PHI));
                    have_unit = true;
                    need_ray = true;
                    bounce = false;
                    need = false;
                }
                if (stall) { LCOUNT(14); break; }          // (census: rounds in which lanes go without a unit for want of a free slot)
            }
        }
        const bool active = have_unit;
        if (!active) { LCOUNT(13); }                       // (census: lanes that idle through this round)
        TSTAMP(0);
        if (active && need_ray) {
            // ---- next ray of the lane: the camera ray of a new sample (Camera::get_ray, camera.rs:109-129) or the
            // scattered ray of a bounce (main.rs:119-127).  Both start with a rejection-sampled pair of
            // Uniform(-1,1) draws — UnitDisc accepts x1^2+x2^2 <= 1, UnitSphere (Marsaglia) rejects >= 1 — and both
            // end in Ray::new's normalisation, so the two kinds share one sampler loop and one normalize.
            // RNG draw order per lane is the reference's.
            float x1, x2, sm;
            WCOUNT(2);
            for (;;) {
                WCOUNT(3);
                LCOUNT(1);
                x1 = uniform_m1_1(rng);
                x2 = uniform_m1_1(rng);
                sm = x1 * x1 + x2 * x2;
                if (bounce ? !(sm >= 1.0f) : (sm <= 1.0f)) break;
            }
            // (the two kinds also END alike — bounce: try_normalize(scatter).unwrap_or(n), main.rs:126; camera:
            // normalize_or_zero(focal_point - o), camera.rs:127 — so that one copy of that normalisation serves both arms: the
            // arms run one after the other at about half the lanes each, what follows them at all of them)
            V3 xdir, pre, fallback;
            LCOUNT(2);
            if (bounce) {
                LCOUNT(3);
                const float factor = 2.0f * RT_SQRT(1.0f - sm);                // UnitSphere, main.rs:119
                const V3 us = mk(x1 * factor, x2 * factor, 1.0f - 2.0f * sm);
                const V3 diffuse_dir = us + bn;
                const V3 glossy_dir = d - (2.0f * dot(d, bn)) * bn;                    // main.rs:120-121
                pre = diffuse_dir + brough * (glossy_dir - diffuse_dir);               // main.rs:122
                fallback = bn;                                                         // main.rs:126
                // o is already the hit point P (origin exactly P)
            } else {
                LCOUNT(4);
                const V3 offset = mk(x1 * p.lens_radius, x2 * p.lens_radius, 0.0f);
                const float u = ((float)px + gen_range_01(rng)) / p.u_den;
                const float v = ((float)(p.H - pyg - 1) + gen_range_01(rng)) / p.v_den;   // camera row, main.rs:71
                const V3 dir0 = normalize_or_zero(llc + u * hor + v * ver - corg);
                const V3 d1 = normalize(dir0);                     // Ray::new re-normalises (ray.rs:134)
                const V3 focal_point = corg + p.focus_distance * d1;
                o = corg + offset;
                pre = focal_point - o;
                fallback = mk(0.f, 0.f, 0.f);                      // normalize_or_zero
                depth_left = p.depth;
                k = 0;
            }
            if (!try_normalize(pre, xdir)) xdir = fallback;
            d = normalize(xdir);                                   // Ray::new (ray.rs:134)
            need_ray = false;
            bounce = false;
        }
        TSTAMP(1);
        // the wave is done when nothing is left to issue and everything issued is committed (wave-uniform: the lanes leave together)
        const bool wave_busy = __ballot(active) != 0ull || freem != all_free || !q_drained || tile_u != tile_units;
        if (STREAMED) {
            if (!__syncthreads_or(wave_busy ? 1 : 0)) break;
        } else {
            if (!wave_busy) break;
        }

        // ================= closest hit (shapes/mod.rs:158-191) =================
        // Round 1 scans without the BVH chain and validates only the winner (an argmin of the
        // root-test-pass set that is itself a BVH candidate is the argmin of the BVH candidates).
        // If the winner is not a BVH candidate (the reference's false far hits, ~1e-5 of segments),
        // round 2 rescans for that lane validating every improving hit.
        if (active && !in_trav) {                            // a new closest-hit query starts
            h.idx = -1;
            aux = ray_aux(d, (p.flags & 8u) != 0);
            if (!TRAVERSE) td = 2.0f * d;                    // (2f32 * ray.direction), sphere.rs:44
            n_seg++;
            if (QNODES) {
                // the ray in grid units: t(q) = q * ig + cq with ig = step * inv, cq = -((o - base) / step) * ig
                const V3 og = mk((o.x - p.q_base[0]) * p.q_rstep[0], (o.y - p.q_base[1]) * p.q_rstep[1], (o.z - p.q_base[2]) * p.q_rstep[2]);
                ig = mk(p.q_step[0] * aux.inv.x, p.q_step[1] * aux.inv.y, p.q_step[2] * aux.inv.z);
                cq = mk(-(og.x * ig.x), -(og.y * ig.y), -(og.z * ig.z));
                const float big = 0x1p100f, tiny = 0x1p-60f;
                const float ax = __builtin_fabsf(ig.x), ay = __builtin_fabsf(ig.y), az = __builtin_fabsf(ig.z);
                qfin = aux.finite && ax < big && ay < big && az < big && ax > tiny && ay > tiny && az > tiny &&
                       __builtin_fabsf(og.x) < 0x1p18f && __builtin_fabsf(og.y) < 0x1p18f && __builtin_fabsf(og.z) < 0x1p18f;
            }
            if (LTREE) {
                // (the fast step's finished lanes idle on node DONE, whose all-of-space box is entered by every ray with a FINITE origin
                // and a finite non-zero inverse direction; any other lane walks the SLOW step, which clamps its stack pointer — round-3
                // advisor: an origin at infinity, (-inf - o) * inv = NaN, would otherwise pop below slot 0)
                aux.finite = aux.finite && __builtin_fabsf(o.x) < __builtin_inff() && __builtin_fabsf(o.y) < __builtin_inff() &&
                             __builtin_fabsf(o.z) < __builtin_inff();
                sgx = aux.sx ? 1u : 0u;
                sgy = aux.sy ? 1u : 0u;
                sgz = aux.sz ? 1u : 0u;
            }
            if (CULL) {
                // spheres too large for the culling slack (a ground sphere) are root-tested here, under the same candidate
                // rule as any leaf; meeting them again in the walk changes nothing
                t_far = __builtin_inff();
                for (uint32_t j = 0; j < p.n_big; j++) {
                    const uint32_t prim = p.big[j];
                    float t;
                    bool hit;
                    if (prim < p.n_sph) {
                        const float4 g = RT_SPHERE_REC(p, prim);
                        hit = exact_sphere(o, 2.0f * d, mk(g.x, g.y, g.z), g.w, p.t_min, p.t_max, t);
                    } else {
                        hit = exact_triangle(o, d, p.tri + 9 * (size_t)(prim - p.n_sph), p.t_min, p.t_max, t);
                    }
                    if (hit)
                        consider_if<true>(h, (int)prim, o, d, t, [&]() { return bvh_reaches(p.bvh_nodes, p.leaf_of[prim], o, AUX()); }, p.leaf_of);
                }
                if (h.idx >= 0) t_far = far_bound(h.dist);
            }
            if (TRAVERSE) {
                t_ref = p.root_ref;
                t_sp = BFSTEP ? 1u : 0u;                     // slot 0 of the branch-free step's stack holds the DONE sentinel
                t_cnt = 0;
                in_trav = (p.n_sph + p.n_tri) > 0;
            }
        }
        if constexpr (TRAVERSE) {
            // ---- BVH::traverse (bvh_impl.rs:373-398) per lane, iteratively: depth-first, left child first, a
            // child is entered iff the ray passes the AABB its parent stores for it.  The leaves reached ARE the
            // reference's candidate list, in its order, so no conservative filter and no validation are needed:
            // exact root tests on them, first minimum wins (shapes/mod.rs:158-191).
            // Steps run until at most half of the wave's live lanes are still walking; the finished lanes are then
            // shaded / refilled while the stragglers keep their stack (LDS) and resume in the next round.
            auto root_test = [&](uint32_t i) {
                    LCOUNT(6);
                    const uint32_t prim = LTREE ? (uint32_t)lc16[i * BLOCK + tid16] & 0x7fffu
                                          : p.list16 ? (uint32_t)lc16[i * BLOCK + tid16] : lc32[i * BLOCK + tid];
                    float t;
                    // The quantised walk only over-approximates BVH::traverse, so a leaf it delivers counts iff
                    // the reference would have reached it = its own exact box passes (leaf-box lemma, bvh_reaches;
                    // Sphere::aabb = c -+ r).  The filter is a pure predicate of (ray, primitive), so it is applied
                    // lazily: only to a hit that would replace the running closest one.  (A lane without a finite
                    // inverse direction walked the exact nodes: nothing to validate.)
                    if (prim < p.n_sph) {
                        const float4 g = RT_SPHERE_REC(p, prim);
                        if (exact_sphere(o, 2.0f * d, mk(g.x, g.y, g.z), g.w, p.t_min, p.t_max, t)) {   // (2f32 * ray.direction), sphere.rs:44
                            if (QNODES)
                                consider_if<CULL>(h, (int)prim, o, d, t, [&]() {
                                    if (!qfin || (p.n_sph + p.n_tri) == 1) return true;
                                    const float4 s = at32(p.geom_r, prim);
                                    return intersects_aabb_finite(o, AUX(), make_float4(s.x - s.w, s.y - s.w, s.z - s.w, 0.f),
                                                                  make_float4(s.x + s.w, s.y + s.w, s.z + s.w, 0.f));
                                }, p.leaf_of);
                            else if (CULL)
                                consider<1>(h, (int)prim, o, d, t, aux, p.bvh_nodes, p.leaf_of);       // (out of depth-first order: ties by rank)
                            else
                                consider<0>(h, (int)prim, o, d, t, aux, p.bvh_nodes, p.leaf_of);
                        }
                    } else {
                        if (exact_triangle(o, d, p.tri + 9 * (size_t)(prim - p.n_sph), p.t_min, p.t_max, t)) {
                            if (QNODES)
                                consider_if(h, (int)prim, o, d, t,
                                            [&]() { return !qfin || bvh_reaches(p.bvh_nodes, p.leaf_of[prim], o, AUX()); });
                            else if (CULL)
                                consider<1>(h, (int)prim, o, d, t, aux, p.bvh_nodes, p.leaf_of);
                            else
                                consider<0>(h, (int)prim, o, d, t, aux, p.bvh_nodes, p.leaf_of);
                        }
                    }
            };
            auto flush = [&]() {
                if (BFSTEP) n_cand += t_cnt - t_head;            // (the other steps count at the append)
                // (Tried in round 3 and dropped on the LDS-tree kernel, tools/experiments/: striking the sure misses from the leaf
                // lists first with the linear engines' conservative broad-phase test, r03_leaf_prefilter.patch, c3 14 560 -> 14 290;
                // and the compacted root tests of the exact-node L2 kernel below in the LDS two list slots give back,
                // r03_ltree_compacted_root_tests.patch: 2.7 % slower than its own per-lane flush, and the kernel that carries both
                // flushes 12 % slower than the one that carries one — 109 instead of 105 VGPRs, 17 instead of 7 spilled SGPRs.
                // Also: the next candidate's sphere record fetched while this one is tested, r03_ltree_flush_prefetch.patch, -1.8 %:
                // the rounds do not wait for their records.)
#pragma clang loop unroll(disable)
                for (uint32_t i = BFSTEP ? t_head : 0u; i < t_cnt; i++) root_test(i);
                t_cnt = 0;
                if (BFSTEP) t_head = 0;
                if (CULL && h.idx >= 0) t_far = far_bound(h.dist);
            };
            // LTREE: one root test for every lane with a pending candidate, its oldest (list order kept: first minimum wins)
            auto flush_one = [&]() {
                n_cand++;
                root_test(t_head);
                t_head++;
                if (t_head == t_cnt) t_head = t_cnt = 0;
                if (CULL && h.idx >= 0) t_far = far_bound(h.dist);
            };

            // ---- Compacted root tests (exact-node L2 kernel).  The per-lane flush above runs as many rounds as the longest
            // list of the wave, each at a handful of lanes (mesh workload: 17 rounds of 5.7 lanes per loop round, half of the
            // kernel's instructions).  Here the (lane, candidate) PAIRS of all flushing lanes are numbered consecutively and
            // pair j is tested by lane j mod 64 — full rounds of 64 — with the owner's ray fetched across lanes; the owner then
            // takes the first minimum of its own pairs in list order, exactly as `consider<0>` does.  Wave-uniform call: every
            // lane still in the loop takes part as a worker, `want` marks the lanes whose lists are flushed.
            constexpr bool COMPACT = XNODES;
            auto flush_c = [&](bool want) {
                const uint32_t cnt = want ? t_cnt : 0u;
                // exclusive prefix sum and total of the (at most 4-bit) counts from ballots: no cross-lane traffic
                uint32_t excl = 0, total = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const unsigned long long m = __ballot(((cnt >> b) & 1u) != 0);
                    excl += (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
                    total += (uint32_t)__builtin_popcountll(m) << b;
                }
                // lanes that left the loop take no part: workers and table entries are numbered by RANK among the live lanes
                const unsigned long long act = __ballot(true);
                const uint32_t n_act = (uint32_t)__builtin_popcountll(act);
                const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                uint32_t* const w_pref = reinterpret_cast<uint32_t*>(lds_raw + p.lds_cmp_off) + (tid >> 6) * 256;   // offset | lane << 16
                float* const w_dist = reinterpret_cast<float*>(w_pref + 64);
                float* const w_root = reinterpret_cast<float*>(w_pref + 128);
                uint32_t* const w_hit = w_pref + 192;
                w_pref[rank] = excl | ((uint32_t)lane << 16);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                int best_k = -1;                                     // owner: slot of the winner among this flush's pairs
                float best_t = 0.f;
                auto own_prim = [&](uint32_t k) -> uint32_t {
                    return p.list16 ? (uint32_t)lc16[k * BLOCK + tid16] : lc32[k * BLOCK + tid];
                };
                for (uint32_t base = 0; base < total; base += n_act) {
                    const uint32_t j = base + rank;
                    const bool valid = j < total;
                    // owner of pair j = the last live lane whose offset is <= j (lanes without candidates share the next one's)
                    uint32_t R = 0, ent = w_pref[0];
                    if (valid) {
#pragma unroll
                        for (uint32_t stp = 32; stp > 0; stp >>= 1) {
                            const uint32_t c = R + stp;
                            if (c < n_act) {
                                const uint32_t e = w_pref[c];
                                if ((e & 0xffffu) <= j) { R = c; ent = e; }
                            }
                        }
                    }
                    const uint32_t L = ent >> 16, slot = j - (ent & 0xffffu);
                    // the owner's ray (cross-lane reads are executed by every lane of the round)
                    const V3 ro = mk(__shfl(o.x, (int)L, 64), __shfl(o.y, (int)L, 64), __shfl(o.z, (int)L, 64));
                    const V3 rd = mk(__shfl(d.x, (int)L, 64), __shfl(d.y, (int)L, 64), __shfl(d.z, (int)L, 64));
                    bool hit = false;
                    float t = 0.f, dist = 0.f;
                    if (valid) {
                        LCOUNT(6);
                        const uint32_t ot = ((uint32_t)tid & ~63u) | L;                              // the owner's thread and 16-bit column
                        const uint32_t ot16 = (ot & ~63u) | ((ot & 31u) << 1) | ((ot >> 5) & 1u);
                        const uint32_t prim = p.list16 ? (uint32_t)lc16[slot * BLOCK + ot16] : lc32[slot * BLOCK + ot];
                        if (prim < p.n_sph) {
                            const float4 g = RT_SPHERE_REC(p, prim);
                            hit = exact_sphere(ro, 2.0f * rd, mk(g.x, g.y, g.z), g.w, p.t_min, p.t_max, t);   // (2f32 * ray.direction), sphere.rs:44
                        } else {
                            hit = exact_triangle(ro, rd, p.tri + 9 * (size_t)(prim - p.n_sph), p.t_min, p.t_max, t);
                        }
                        if (hit) {
                            const V3 pp = ro + t * rd;               // Ray::at, then |P - o| (consider)
                            dist = vlength(pp - ro);
                        }
                    }
                    w_dist[rank] = dist;
                    w_root[rank] = t;
                    w_hit[rank] = hit ? 1u : 0u;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    // owners: first minimum over their pairs of this round, in list order (consider<0>)
                    if (cnt) {
                        const uint32_t k0 = excl > base ? excl : base, k1 = min(excl + cnt, base + n_act);
#pragma clang loop unroll(disable)
                        for (uint32_t k = k0; k < k1; k++) {
                            if (!w_hit[k - base]) continue;
                            const float dk = w_dist[k - base];
                            bool take = best_k < 0 ? (h.idx < 0 || h.dist > dk) : (h.dist > dk);
                            if (CULL && !take && h.dist == dk && (best_k >= 0 || h.idx >= 0)) {
                                // culled walk: candidates arrive out of depth-first order, an equal distance goes to the earlier leaf
                                const uint32_t cur = best_k >= 0 ? own_prim((uint32_t)best_k) : (uint32_t)h.idx;
                                take = p.leaf_of[own_prim(k - excl)] < p.leaf_of[cur];
                            }
                            if (take) {
                                h.dist = dk;
                                best_k = (int)(k - excl);
                                best_t = w_root[k - base];
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // reads done before the next round's writes
                }
                if (best_k >= 0) {
                    h.idx = (int)own_prim((uint32_t)best_k);
                    h.t = best_t;
                }
                if (want) t_cnt = 0;
                if (CULL && want && h.idx >= 0) t_far = far_bound(h.dist);
            };
            // Per-lane stack.  The exact-node kernel keeps all of it in LDS ((depth + 1) KiB per workgroup); the quantised
            // kernel (large scenes, deep trees) keeps p.stack_lds entries there and the rare deeper ones in HBM, so
            // that the tree's depth does not take the CU's LDS away from its occupancy.
            auto push = [&](uint32_t v) {
                if (LTREE) lstack16[t_sp * BLOCK + tid16] = (uint16_t)v;
                else if (!CAPPED || t_sp < p.stack_lds) lstack[t_sp * BLOCK + tid] = v;
                else p.stack_ovf[(size_t)(t_sp - p.stack_lds) * p.ovf_stride + (blockIdx.x * BLOCK + tid)] = v;
                t_sp++;
            };
            auto pop = [&]() -> uint32_t {
                --t_sp;
                if (LTREE) return (uint32_t)lstack16[t_sp * BLOCK + tid16];
                if (!CAPPED || t_sp < p.stack_lds) return lstack[t_sp * BLOCK + tid];
                return p.stack_ovf[(size_t)(t_sp - p.stack_lds) * p.ovf_stride + (blockIdx.x * BLOCK + tid)];
            };
            if constexpr (BFSTEP) {
            // ---- Branch-free node step of the LDS-resident tree.  With the nodes a few dozen cycles
            // away a wave's time per step is the length of its serial instruction stream, scalar exec-mask bookkeeping
            // and branches included, and the earlier step (a dozen exec regions: leaf / internal, push, pop, list
            // append, emptiness and fullness tests) spent more instructions on control than on the two slab tests.
            // Here every lane of the block runs the same straight line and the state advances through selects:
            //   * stack slot 0 holds the reference DONE for good and t_sp >= 1, so popping the empty stack yields DONE.  Node
            //     DONE (= n_internal) is a dummy that leads back to itself: its left box is all of space ((-inf - o) * inv and
            //     (+inf - o) * inv are -+inf for every finite non-zero inv), its left child is DONE, its right box is NaN.  A
            //     finished lane idles there until the block ends — no emptiness test, no per-lane exit, and (round 3) no clamp
            //     of the stack pointer: the lane neither pushes nor pops.  A lane whose inverse direction is not finite may
            //     miss even that box (-inf * -inf); it walks the SLOW variant, which keeps the clamp, and there DONE pops DONE;
            //   * a lane at a leaf forms its node address like any other — base + 4 * reference — and lands in the NaN field behind the
            //     tree (staging code above): NaN planes, both slab results false, it pops; no clamp of the reference (round 3);
            //   * the right child is stored to stack[t_sp], the next free slot, pushed or not (t_sp += both);
            //   * every reference is stored to list[t_cnt], the next free list slot, and only a leaf advances t_cnt;
            //   * the leaf list has room for a whole block of appends (checked between blocks): no fullness test.
            // The crate's literal slab test (a +-0 direction component, or RT_FLAG_FULL_CHAIN) is chosen per BLOCK of
            // steps for the whole wave: it is the reference's own test, valid for every lane.
            const uint32_t DONE = lt_r0(p.n_internal) + p.n_internal * (uint32_t)LNODE_DW;   // node references: LT_R0 + offset in dwords
            constexpr int STEPS = RT_STEPS_PER_CHECK_LTREE;
            static_assert(MAXL_LTREE > STEPS, "the leaf list must take a block of appends");
            // Inside a block the stack pointer and the list length are carried as LDS byte addresses (top_a: the lane's top
            // stack slot; cnt_m: one slot below the lane's next list slot), so that the three accesses of a step are a
            // register plus an immediate offset: no index arithmetic in the step.
            constexpr uint32_t SLOT = (uint32_t)BLOCK * 2u;      // bytes between two slots of a lane's column
            typedef __attribute__((address_space(3))) unsigned char lds_byte;
            typedef __attribute__((address_space(3))) uint16_t lds_u16;
            const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_byte*)lds_raw;               // LDS address of the dynamic area
            const uint32_t stack0_a = lds0 + p.lds_stack_off + (uint32_t)tid16 * 2u;     // slot 0 (DONE) of this lane
            const uint32_t list0_m = lds0 + p.lds_cand_off + (uint32_t)tid16 * 2u - SLOT; // one slot below list slot 0
            uint32_t top_a = 0, cnt_m = 0;
            auto lds16 = [&](uint32_t a) -> lds_u16& { return *(lds_u16*)(uintptr_t)a; };
            auto step = [&](auto slow_tag) {
                constexpr bool SLOW = decltype(slow_tag)::value;
                // the leaf flag (bit 15) moved to the weight of one list slot: two fast-class instructions (shift right, and) where
                // a compare and two selects were (tools/ubench/valu_classes: 2.7 against 4.4 cycles per wave-instruction)
                static_assert(SLOT == 0x800u, "leaf flag 0x8000 >> 4 must be one slot");
                const uint32_t leaf_slot = (t_ref >> 4) & SLOT;
                const uint32_t ni = t_ref;                          // (a leaf's address lies in the NaN field behind the tree)
                if (STATS) n_int += (ni < DONE) ? 1u : 0u;
                const uint32_t top = (uint32_t)lds16(top_a);
                WCOUNT(5);
                LCOUNT(5);
                // every reference goes to the next list slot; only a leaf moves the list's end past it
                lds16(cnt_m + SLOT) = (uint16_t)t_ref;                           // (the flush masks the leaf flag off)
                // Ray::intersects_aabb (ray.rs:174-194) on both child boxes; (near, far) planes fetched by sign
                const float* __restrict__ nd = lnodes + ni;
                const float* __restrict__ fx = nd + sgx;
                const float* __restrict__ fy = nd + 6 + sgy;
                const float* __restrict__ fz = nd + 12 + sgz;
                const float p0 = fx[0], p1 = fx[1], p2 = fx[3], p3 = fx[4], p4 = fy[0], p5 = fy[1], p6 = fy[3], p7 = fy[4];
                const float p8 = fz[0], p9 = fz[1], p10 = fz[3], p11 = fz[4];
                const uint32_t refs = __float_as_uint(nd[18]);
                RT_HOOK_LT_STEP_LOADS(nd, ni, lnodes);
                // all seven reads in flight before the first use (the scheduler, short of registers, serialised them)
                __builtin_amdgcn_sched_barrier(0);
                RT_HOOK_LT_STEP_ALU(p0, p1, p2, p3, o.y);
                const float lxn = (p0 - o.x) * aux.inv.x, lxf = (p1 - o.x) * aux.inv.x;
                const float rxn = (p2 - o.x) * aux.inv.x, rxf = (p3 - o.x) * aux.inv.x;
                const float lyn = (p4 - o.y) * aux.inv.y, lyf = (p5 - o.y) * aux.inv.y;
                const float ryn = (p6 - o.y) * aux.inv.y, ryf = (p7 - o.y) * aux.inv.y;
                const float lzn = (p8 - o.z) * aux.inv.z, lzf = (p9 - o.z) * aux.inv.z;
                const float rzn = (p10 - o.z) * aux.inv.z, rzf = (p11 - o.z) * aux.inv.z;
                bool hl, hr;
                float le, re;                                    // entry distances of the two boxes (max(ray_min, 0), ray.rs:193)
                if (SLOW) {                                      // the crate's min / max (ray.rs:81-112): NaN-aware order
                    le = rmax(rmax(rmax(lxn, lyn), lzn), 0.0f);
                    re = rmax(rmax(rmax(rxn, ryn), rzn), 0.0f);
                    hl = le <= rmin(rmin(lxf, lyf), lzf);
                    hr = re <= rmin(rmin(rxf, ryf), rzf);
                } else {                                         // finite inverse direction: no NaN can arise, and min / max
                    // differ from the crate's forms only in the sign of a zero, which no comparison sees
                    le = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(lxn, lyn), lzn), 0.0f);
                    re = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(rxn, ryn), rzn), 0.0f);
                    hl = le <= __builtin_fminf(__builtin_fminf(lxf, lyf), lzf);
                    hr = re <= __builtin_fminf(__builtin_fminf(rxf, ryf), rzf);
                }
                uint32_t cl = refs & 0xffffu, cr = refs >> 16;
                if (CULL) {
                    // Culled walk (ISECT 6): a box the ray enters beyond t_far cannot hold a closer or tying root (cull_bound, rt_cull.h)
                    // and is not entered; of two boxes the NEARER goes first, so that t_far shrinks early.  (A second compare, not
                    // min(exit, t_far): the NaN planes of leaves and of node DONE's right box must keep failing.)
                    hl = hl && le <= t_far;
                    hr = hr && re <= t_far;
                    const bool swp = hl && hr && re < le;
                    const uint32_t c0 = cl;
                    cl = swp ? cr : cl;
                    cr = swp ? c0 : cr;
                }
                lds16(top_a + SLOT) = (uint16_t)cr;                // right subtree after the whole left subtree (CULL: the farther one)
                const bool any = hl || hr;
                t_ref = any ? (hl ? cl : cr) : top;
                uint32_t delta = any ? 0u : 0u - SLOT;             // (two selects and a fast-class add; the sum of two selects was a v_add3)
                delta = (hl && hr) ? SLOT : delta;
                asm volatile("" : "+v"(delta));
                top_a += delta;
                if (SLOW) top_a = max(top_a, stack0_a);          // (DONE may pop here: see above)
                cnt_m += leaf_slot;
                RT_HOOK_LT_STEP_END;
            };
            for (;;) {
                const uint32_t walking = (uint32_t)__builtin_popcountll(__ballot(in_trav));
                const uint32_t live = (uint32_t)__builtin_popcountll(__ballot(active));     // (lanes without a unit idle in the loop)
                if (walking == 0 || (walking * 8 <= live * p.refill_eighths && walking < live)) break;
                if (in_trav && t_cnt > p.maxl - (uint32_t)STEPS) flush();          // room for a block of appends
                // Partial rounds of root tests.  The flush after the walk runs as many rounds as the LONGEST list among the
                // finished lanes, most of them at a handful of lanes (round 3 census: 4.0 rounds at 17.4 lanes per loop round,
                // 18 % of the kernel's time for 1.34 candidates per query).  Here, between two blocks of steps, one candidate per
                // lane — its oldest: list order is kept, the first minimum still wins — is tested as soon as RT_LT_PARTIAL lanes have
                // one pending: 3.4 rounds at 21 lanes.
                if (RT_LT_PARTIAL < 64) {
                    const bool pend = t_cnt > t_head;
                    // (culled walk: a lane that has candidates but no hit yet cannot skip anything: such lanes are tested as soon as
                    // there are RT_LT_EAGER of them)
                    if ((int)__builtin_popcountll(__ballot(pend)) >= RT_LT_PARTIAL ||
                        (CULL && (int)__builtin_popcountll(__ballot(pend && h.idx < 0)) >= RT_LT_EAGER)) {
                        if (pend) flush_one();
                    }
                }
                const bool slow = __ballot(in_trav && !aux.finite) != 0;         // wave-uniform
                if (in_trav) {
                    top_a = stack0_a + (t_sp - 1u) * SLOT;
                    cnt_m = list0_m + t_cnt * SLOT;
                    if (slow) {
#pragma unroll
                        for (int rep = 0; rep < STEPS; rep++) step(std::true_type{});
                    } else {
#pragma unroll
                        for (int rep = 0; rep < STEPS; rep++) step(std::false_type{});
                    }
                    t_sp = (top_a + SLOT - stack0_a) / SLOT;       // (0 for a lane that finished: its pop of DONE went below slot 0)
                    t_cnt = (cnt_m - list0_m) / SLOT;
                    in_trav = t_ref != DONE;
                }
            }
            } else {
            for (;;) {
                const uint32_t walking = (uint32_t)__builtin_popcountll(__ballot(in_trav));
                const uint32_t live = (uint32_t)__builtin_popcountll(__ballot(active));
                if (walking == 0 || (walking * 8 <= live * p.refill_eighths && walking < live)) break;
#ifndef RT_FLUSH_INLINE
                // a lane whose leaf list is full waits at its leaf until this point (keeps the root tests out of the
                // unrolled step code: one copy instead of RT_STEPS_PER_CHECK; c3 +1 %, 45 % less code)
                // (CULL: a lane without a hit yet tests its leaves now, so that the walk can start skipping)
                if (COMPACT && p.lds_cmp_off != 0xffffffffu) {
                    const bool want = in_trav && (t_cnt == ML || (CULL && t_cnt >= (uint32_t)RT_CULL_FLUSH_MIN && h.idx < 0));
                    if (__ballot(want)) flush_c(want);
                } else
                if (in_trav && (t_cnt == ML || (CULL && t_cnt >= (uint32_t)RT_CULL_FLUSH_MIN && h.idx < 0))) flush();
#endif
                // ---- Straight-line step of the uncapped quantised walks (RT_BF2).  The step below this one spends more
                // instructions on exec-mask bookkeeping (14 regions) than on the two slab tests; with the gathers no longer the
                // only bound (culled walk) the wave's serial instruction stream matters here as it did for the LDS tree.  One
                // region per step (the lane still walks), everything else through selects: a lane at a leaf fetches node 0 (a
                // hot line) and fails both tests, so it appends and pops; a full list stalls the lane until the flush between
                // blocks; the right child is stored to the free stack slot pushed or not.  Taken when every walking lane of the
                // wave carries its ray in grid units; a wave with a fallback lane runs the step below.
                constexpr bool BF2 = (RT_BF2 != 0) && (XNODES || QNODES);
                if (BF2 && !__ballot(in_trav && !(QNODES ? qfin : aux.finite))) {
#pragma unroll
                    for (int rep = 0; rep < (QNODES ? RT_STEPS_PER_CHECK_Q : RT_STEPS_PER_CHECK_X); rep++)
                    if (in_trav) {
                        WCOUNT(5);
                        LCOUNT(5);
                        const bool is_leaf = (t_ref & LEAF_BIT) != 0;
                        if (STATS) n_int += is_leaf ? 0u : 1u;
                        // what a pop would yield (capped stack: entries from p.stack_lds up live in HBM, fetched below when needed)
                        const uint32_t tsl = t_sp - (t_sp ? 1u : 0u);
                        uint32_t top = lstack[(CAPPED ? min(tsl, p.stack_lds - 1u) : tsl) * BLOCK + tid];
                        bool hl, hr;
                        float le = 0.f, re = 0.f;
                        uint32_t c0, c1;                                            // left, right child reference
                        if constexpr (QNODES) {
                        const uint4* __restrict__ nq = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.travq) + ((is_leaf ? 0u : t_ref) << 5));
                        uint4 qa = nq[0], qb = nq[1];
                        asm volatile("" : "+v"(qb.x), "+v"(qb.y), "+v"(qb.z), "+v"(qb.w));   // keep the two 16-byte loads whole
                        float lmin, lmax, rmin_, rmax_;
                        qslabs(qa, qb, ig, cq, lmin, lmax, rmin_, rmax_);
                        le = __builtin_fmaxf(lmin, 0.0f);
                        re = __builtin_fmaxf(rmin_, 0.0f);
                        hl = le <= (CULL ? __builtin_fminf(lmax, t_far) : lmax);
                        hr = re <= (CULL ? __builtin_fminf(rmax_, t_far) : rmax_);
                        c0 = qb.z;
                        c1 = qb.w;
                        } else {
                        const float4* __restrict__ nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.trav) + ((is_leaf ? 0u : t_ref) << 6));
                        const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
                        if (CULL) {
                            // the crate's test for a finite inverse direction (intersects_aabb_finite), with the entry distances kept:
                            // a box entered beyond t_far is skipped, the nearer child goes first
                            float lmin, lmax, rmin_, rmax_;
                            slabs_finite(o, aux, n0, n1, lmin, lmax);
                            slabs_finite(o, aux, n2, n3, rmin_, rmax_);
                            le = __builtin_fmaxf(lmin, 0.0f);
                            re = __builtin_fmaxf(rmin_, 0.0f);
                            hl = le <= lmax && le <= t_far;
                            hr = re <= rmax_ && re <= t_far;
                        } else {
                            hl = intersects_aabb_finite(o, aux, n0, n1);
                            hr = intersects_aabb_finite(o, aux, n2, n3);
                        }
                        c0 = __float_as_uint(n0.w);
                        c1 = __float_as_uint(n1.w);
                        }
                        hl = hl && !is_leaf;
                        hr = hr && !is_leaf;
                        const bool both = hl && hr, any = hl || hr;
                        const bool swp = CULL && both && re < le;                 // nearer child first
                        const uint32_t cl = swp ? c1 : c0, cr = swp ? c0 : c1;
                        if (!CAPPED || t_sp < p.stack_lds) lstack[t_sp * BLOCK + tid] = cr;       // the next free slot, pushed or not
                        else if (both) p.stack_ovf[(size_t)(t_sp - p.stack_lds) * p.ovf_stride + (blockIdx.x * BLOCK + tid)] = cr;
                        const bool can = t_cnt < ML, app = is_leaf && can, stall = is_leaf && !can;
                        if (app) {
                            if (p.list16) lc16[t_cnt * BLOCK + tid16] = (uint16_t)t_ref;
                            else lc32[t_cnt * BLOCK + tid] = t_ref & ~LEAF_BIT;
                        }
                        t_cnt += app ? 1u : 0u;
                        n_cand += app ? 1u : 0u;
                        const bool pop_ = !any && !stall, empty = t_sp == 0;
                        if (CAPPED && pop_ && t_sp > p.stack_lds)
                            top = p.stack_ovf[(size_t)(t_sp - 1u - p.stack_lds) * p.ovf_stride + (blockIdx.x * BLOCK + tid)];
                        t_ref = stall ? t_ref : any ? (hl ? cl : cr) : top;
                        in_trav = !(pop_ && empty);
                        t_sp = t_sp + (both ? 1u : 0u) - ((pop_ && !empty) ? 1u : 0u);
                    }
                } else
                // (the step of a wave with a fallback lane: rare, kept rolled when the straight-line step is compiled in)
#if defined(RT_ROLL_STEPS) || RT_BF2
#pragma clang loop unroll(disable)
#else
#pragma unroll
#endif
                for (int rep = 0; rep < (QNODES ? RT_STEPS_PER_CHECK_Q : RT_STEPS_PER_CHECK); rep++)
                if (in_trav) {
                    WCOUNT(5);
                    LCOUNT(5);
                    if (t_ref & LB) {
#ifdef RT_FLUSH_INLINE
                        if (t_cnt == ML) flush();
#else
                        if (t_cnt == ML) continue;
#endif
                        if (p.list16) lc16[t_cnt * BLOCK + tid16] = (uint16_t)t_ref;        // (prims <= 65536: the id's low 16 bits)
                        else lc32[t_cnt * BLOCK + tid] = t_ref & ~LEAF_BIT;
                        t_cnt++;
                        n_cand++;
                        if (t_sp == 0) {
                            in_trav = false;                 // candidates are tested together after the walk
                        } else {
                            t_ref = pop();
                        }
                    } else {
                        bool hl, hr;
                        uint32_t cl, cr;
                        if (STATS) n_int++;
                        if (QNODES) {
                            // (32-bit byte offset from the uniform base: one shift and the scalar-base addressing mode)
                            const uint4* __restrict__ nq = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.travq) + (t_ref << 5));
                            uint4 qa = nq[0], qb = nq[1];
                            asm volatile("" : "+v"(qb.x), "+v"(qb.y), "+v"(qb.z), "+v"(qb.w));   // keep the two 16-byte loads whole
                            // (fetching the second half only after the first has arrived — an L1 hit instead of a second miss on a
                            // line in flight — costs 7 % on c5: the step's latency matters more than the texture path's time)
                            cl = qb.z;
                            cr = qb.w;
                            RT_HOOK_Q_GATHER(t_ref, cl, p.travq, p.n_internal);
                            if (qfin) {
                                // slab test in grid units: t = fma(q, ig, cq), the same real value as
                                // ((base + q*step) - o) * inv up to < 0.15 grid unit of rounding; the boxes carry >= 1 grid
                                // unit of outward slack: conservative, exactness restored at the leaves
                                float lmin, lmax, rmin_, rmax_;
                                qslabs(qa, qb, ig, cq, lmin, lmax, rmin_, rmax_);
                                if (CULL) {
                                    // the exit clipped to t_far: a box entered beyond it is skipped; the nearer child first
                                    const float le = __builtin_fmaxf(lmin, 0.0f), re = __builtin_fmaxf(rmin_, 0.0f);
                                    hl = le <= __builtin_fminf(lmax, t_far);
                                    hr = re <= __builtin_fminf(rmax_, t_far);
                                    if (hl && hr && re < le) {
                                        const uint32_t t = cl;
                                        cl = cr;
                                        cr = t;
                                    }
                                } else {
                                    hl = __builtin_fmaxf(lmin, 0.0f) <= lmax;
                                    hr = __builtin_fmaxf(rmin_, 0.0f) <= rmax_;
                                }
                            } else {
                                // +-0 direction component (inverse = +-inf): the monotonicity argument does not hold, so
                                // this lane walks the exact nodes with the crate's literal test; its leaves need no validation
                                const float4* __restrict__ nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.trav) + (t_ref << 6));
                                const RayAux ax = AUX();
                                hl = intersects_aabb(o, ax, nd[0], nd[1]);
                                hr = intersects_aabb(o, ax, nd[2], nd[3]);
                            }
                        } else {
                            const float4* __restrict__ nd = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.trav) + (t_ref << 6));
                            const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
                            if (aux.finite) {                    // (RT_FLAG_FULL_CHAIN also forces the crate's literal form)
                                hl = intersects_aabb_finite(o, aux, n0, n1);
                                hr = intersects_aabb_finite(o, aux, n2, n3);
                            } else {
                                hl = intersects_aabb(o, aux, n0, n1);
                                hr = intersects_aabb(o, aux, n2, n3);
                            }
                            cl = __float_as_uint(n0.w);
                            cr = __float_as_uint(n1.w);
                        }
                        if (hl) {
                            if (hr) push(cr);                            // right subtree after the whole left subtree
                            t_ref = cl;
                        } else if (hr) {
                            t_ref = cr;
                        } else if (t_sp == 0) {
                            in_trav = false;
                        } else {
                            t_ref = pop();
                        }
                    }
                }
            }
            }
            TSTAMP(2);
            if (COMPACT && p.lds_cmp_off != 0xffffffffu) {
                const bool want = active && !in_trav && t_cnt > 0;
                if (__ballot(want)) flush_c(want);
            } else
            if (active && !in_trav) flush();                 // exact root tests of the finished lanes, together
            TSTAMP(3);
        } else {
            bool seg_active = active;
            bool inline_chain = false;
            for (;;) {
                if (seg_active) h.idx = -1;
                for (uint32_t ch = 0; ch < p.n_chunks; ch++) {
                    const uint32_t base = ch * p.chunk;
                    const uint32_t cn = min(p.chunk, p.n_sph_pad - base);   // multiple of UNROLL
                    if (STREAMED) {
                        __syncthreads();
                        for (uint32_t i = tid; i < cn; i += BLOCK) lgeom[i] = gsrc[base + i];
                        if (EXPANDED)
                            for (uint32_t i = tid; i < cn; i += BLOCK) lrr[i] = p.geom[base + i].w;
                        __syncthreads();
                    }
                    if (seg_active) {
                        uint32_t cnt = 0;
                        if (!exact_scan) {
                            uint32_t ncand_it = 0;
                            // ---- broad phase: conservative "line misses sphere" rejection on sphere pairs.
                            // The value only selects candidates (FMA allowed); the narrow phase decides.
                            // pass = !(t < 0).  Two forms, chosen per scene by the host:
                            //  oc form (11 packed ops / pair):  t = b'^2 + rr - L2 (1 - 2^-17),
                            //          b' = d.(o-c), L2 = |o-c|^2
                            //  expanded form (8 packed ops / pair):
                            //          t = (A - d.c)^2 - (oo' + w - 2 o.c),  A = d.o, oo' = |o|^2 (1 - 2^-16),
                            //          w = |c|^2 - rr - 2^-16 (|c|^2 + rr)  (host, rounded down)
                            v2f k0x, k0y, k0z, k1x, k1y, k1z, kA, kB;
                            if (EXPANDED) {
                                const float A = __builtin_fmaf(d.z, o.z, __builtin_fmaf(d.y, o.y, d.x * o.x));
                                const float oo =
                                    __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.y, o.y, o.x * o.x)) * (1.0f - 0x1p-16f);
                                k0x = v2f{-d.x, -d.x}; k0y = v2f{-d.y, -d.y}; k0z = v2f{-d.z, -d.z};
                                k1x = v2f{-2.0f * o.x, -2.0f * o.x}; k1y = v2f{-2.0f * o.y, -2.0f * o.y};
                                k1z = v2f{-2.0f * o.z, -2.0f * o.z};
                                kA = v2f{A, A};
                                kB = v2f{oo, oo};
                            } else {
                                k0x = v2f{o.x, o.x}; k0y = v2f{o.y, o.y}; k0z = v2f{o.z, o.z};
                                k1x = v2f{d.x, d.x}; k1y = v2f{d.y, d.y}; k1z = v2f{d.z, d.z};
                                kA = NKM;
                                kB = NKM;
                            }
                            for (uint32_t j = 0; j < cn; j += UNROLL) {
                                float t[UNROLL];
    #pragma unroll
                                for (int q = 0; q < UNROLL / 2; q++) {
                                    const float4 A4 = lgeom[j + 2 * q];
                                    const float4 B4 = lgeom[j + 2 * q + 1];
                                    const v2f cx = {A4.x, A4.y}, cy = {A4.z, A4.w}, cz = {B4.x, B4.y}, cw = {B4.z, B4.w};
                                    v2f tt;
                                    if (EXPANDED) {
                                        const v2f bb = pk_fma(k0x, cx, pk_fma(k0y, cy, pk_fma(k0z, cz, kA)));
                                        const v2f qq = pk_fma(k1x, cx, pk_fma(k1y, cy, pk_fma(k1z, cz, cw + kB)));
                                        tt = pk_fma(bb, bb, -qq);
                                    } else {
                                        const v2f ocx = k0x - cx, ocy = k0y - cy, ocz = k0z - cz;
                                        const v2f bb = pk_fma(k1z, ocz, pk_fma(k1y, ocy, k1x * ocx));
                                        const v2f l2 = pk_fma(ocz, ocz, pk_fma(ocy, ocy, ocx * ocx));
                                        tt = pk_fma(l2, kA, pk_fma(bb, bb, cw));
                                    }
                                    t[2 * q] = tt.x;
                                    t[2 * q + 1] = tt.y;
                                }
                                // max ignores NaN; a NaN t can only come from non-finite operands, for
                                // which the exact test reports a miss as well
                                const float m = __builtin_fmaxf(
                                    __builtin_fmaxf(__builtin_fmaxf(t[0], t[1]), __builtin_fmaxf(t[2], t[3])),
                                    __builtin_fmaxf(__builtin_fmaxf(t[4], t[5]), __builtin_fmaxf(t[6], t[7])));
                                if (!(m < 0.0f)) {
                                    WCOUNT(4);
                                    // one list entry per passing group of 8: (group << 8) | pass mask.  The mask
                                    // comes from the sign bits (t >= +0 or NaN-with-clear-sign => candidate).
                                    uint32_t neg = 0;
    #pragma unroll
                                    for (int q = 0; q < UNROLL; q++) neg |= (__float_as_uint(t[q]) >> 31) << q;
                                    const uint32_t pass8 = ~neg & 0xffu;
                                    if (cnt < (uint32_t)MAXC) lcand[cnt * BLOCK + tid] = (uint16_t)(((j >> 3) << 8) | pass8);
                                    cnt++;
                                    ncand_it += (uint32_t)__builtin_popcount(pass8);
                                }
                            }
                            if (!inline_chain) n_cand += ncand_it;
                        }
                        // ---- narrow phase: the reference's exact arithmetic, ascending index order.
                        // direct = every sphere of the chunk (exact-scan flag, or candidate list overflow)
                        const bool direct = exact_scan || cnt > (uint32_t)MAXC;
                        if (!exact_scan && direct && !inline_chain) n_fall++;
                        const uint32_t n_ent = direct ? (cn >> 3) : cnt;     // groups to visit
                        uint32_t ei = 0, grp = 0, bits = 0;
                        for (;;) {
                            if (bits == 0) {                                  // next list entry / next group
                                if (ei == n_ent) break;
                                if (direct) {
                                    grp = ei;
                                    bits = 0xffu;
                                } else {
                                    const uint32_t e = lcand[ei * BLOCK + tid];
                                    grp = e >> 8;
                                    bits = e & 0xffu;
                                }
                                ei++;
                                if (bits == 0) continue;
                            }
                            WCOUNT(5);
                            const uint32_t j = grp * 8 + (uint32_t)__builtin_ctz(bits);
                            bits &= bits - 1;
                            const uint32_t fo = (j >> 1) * 8 + (j & 1);                 // exact centre from the pair layout
                            const V3 cen = mk(lgeomf[fo], lgeomf[fo + 2], lgeomf[fo + 4]);
                            const float rr = EXPANDED ? lrr[j] : lgeomf[fo + 6];        // exact r^2
                            float t;
                            if (exact_sphere(o, td, cen, rr, p.t_min, p.t_max, t)) {
                                WCOUNT(6);
                                if (!use_bvh) {
                                    // index order = the order of `world`: the scan runs in primitive order, so with a
                                    // world_index an equal distance goes to the earlier world position explicitly
                                    if (p.world_rank) consider<1>(h, (int)(base + j), o, d, t, aux, p.bvh_nodes, p.world_rank);
                                    else consider<0>(h, (int)(base + j), o, d, t, aux, p.bvh_nodes, p.leaf_of);
                                } else if (inline_chain)
                                    consider<2>(h, (int)(base + j), o, d, t, aux, p.bvh_nodes, p.leaf_of);
                                else
                                    consider<1>(h, (int)(base + j), o, d, t, aux, p.bvh_nodes, p.leaf_of);
                            }
                        }
                    }
                }
                if (seg_active) {
                    // triangles: exact test against every triangle (after the spheres in index order)
                    for (uint32_t j = 0; j < p.n_tri; j++) {
                        // BVH semantics: a triangle the ray's own-leaf AABB test rejects was never returned by
                        // BVH::traverse, so the reference's exact slab test doubles as the broad phase (no margin
                        // needed: it IS the reference's candidate rule).  Linear semantics: exact test on all.
                        if (use_bvh && p.n_sph + p.n_tri > 1 &&
                            !intersects_aabb(o, aux, p.tri_box[2 * (size_t)j], p.tri_box[2 * (size_t)j + 1]))
                            continue;
                        float t;
                        if (exact_triangle(o, d, p.tri + 9 * (size_t)j, p.t_min, p.t_max, t)) {
                            if (!use_bvh) {
                                if (p.world_rank) consider<1>(h, (int)(p.n_sph + j), o, d, t, aux, p.bvh_nodes, p.world_rank);
                                else consider<0>(h, (int)(p.n_sph + j), o, d, t, aux, p.bvh_nodes, p.leaf_of);
                            } else if (inline_chain)
                                consider<2>(h, (int)(p.n_sph + j), o, d, t, aux, p.bvh_nodes, p.leaf_of);
                            else
                                consider<1>(h, (int)(p.n_sph + j), o, d, t, aux, p.bvh_nodes, p.leaf_of);
                        }
                    }
                }
                // would BVH::traverse have returned the winner?  (root-to-leaf AABB chain)
                bool redo = false;
                if (seg_active && use_bvh && !inline_chain && h.idx >= 0) {
                    WCOUNT(7);
                    redo = !bvh_reaches(p.bvh_nodes, p.leaf_of[h.idx], o, aux);
                }
                if (STREAMED) {
                    if (!__syncthreads_or(redo ? 1 : 0)) break;
                } else {
                    if (!redo) break;
                }
                seg_active = redo;
                inline_chain = true;
            }
        }
        if (active && !in_trav) {
            // ================= shade (main.rs:114-145) =================
            float term_r, term_g, term_b;
            bool finished;
            LCOUNT(7);
            // Both outcomes need one normalize_or_zero — of (P - centre) / the triangle's face vector for the normal of a hit
            // (sphere.rs:49-51, mesh.rs:163-165), of the direction for the sky (main.rs:136) — so the vector is chosen per
            // lane first and ONE copy of the normalisation runs at all the lanes (the two arms run at about half of them each).
            const bool hit = h.idx >= 0;
            float em = 0.f;
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
            V3 hp = o, nv = d;
            if (hit) {
                WCOUNT(8);
                LCOUNT(8);
                em = at32(p.emis, (uint32_t)h.idx);
                m = at32(p.mat, (uint32_t)h.idx);
                if (!(em > 0.0f)) {
                    hp = o + h.t * d;                                                  // Ray::at (ray.rs:147-149), as in consider
                    if ((uint32_t)h.idx < p.n_sph) {
                        float4 g = TRAVERSE ? RT_SPHERE_REC(p, (uint32_t)h.idx) : at32(p.geom, (uint32_t)h.idx);      // (asked for together with the material instead: no gain, measured in round 3)
                        nv = hp - mk(g.x, g.y, g.z);                                   // sphere.rs:49-51
                    } else {
                        const float* tv = p.tri + 9 * (size_t)(h.idx - p.n_sph);
                        V3 A = mk(tv[0], tv[1], tv[2]), B = mk(tv[3], tv[4], tv[5]), C = mk(tv[6], tv[7], tv[8]);
                        nv = cross(A - B, A - C);                                      // mesh.rs:163-165
                    }
                }
            }
            const V3 nn = normalize_or_zero(nv);
            if (hit) {
                if (em > 0.0f) {                              // main.rs:116-117
                    term_r = m.x * em;
                    term_g = m.y * em;
                    term_b = m.z * em;
                    finished = true;
                } else {
                    const V3 n = nn;
                    // push the hit on the path stack: albedo product is applied back-to-front
                    if (p.path32)
                        reinterpret_cast<uint32_t*>(lpath)[k * BLOCK + cold<QNODES>(tid)] = (uint32_t)h.idx;
                    else
                        reinterpret_cast<uint16_t*>(lpath)[k * BLOCK + cold<QNODES>(tid16)] = (uint16_t)h.idx;
                    k++;
                    depth_left--;
                    if (depth_left == 0) {
                        // the reference still draws UnitSphere before ray_color(.., 0) returns black
                        // (main.rs:119 then :109-111): advance the stream, the direction is never used
                        WCOUNT(9);
                        for (;;) {
                            WCOUNT(10);
                            const float y1 = uniform_m1_1(rng);
                            const float y2 = uniform_m1_1(rng);
                            if (!(y1 * y1 + y2 * y2 >= 1.0f)) break;
                        }
                    } else {
                        // scattered ray: generated at the top of the next round together with the camera rays
                        o = hp;                                                        // origin exactly P
                        bn = n;
                        brough = m.w;
                        bounce = true;
                        need_ray = true;
                    }
                    finished = (depth_left == 0);                                      // main.rs:109-111
                    term_r = term_g = term_b = 0.0f;
                }
            } else {
                WCOUNT(11);
                LCOUNT(9);
                // sky (main.rs:135-144)
                float t = nn.y * 0.5f + 1.0f;
                float omt = 1.0f - t;
                term_r = 1.0f * t + 0.3f * omt;
                term_g = 1.0f * t + 0.3f * omt;
                term_b = 1.0f * t + 0.8f * omt;
                finished = true;
            }
            if (finished) {
                WCOUNT(12);
                LCOUNT(10);
                // a1 (.) (a2 (.) ( ... (ak (.) terminal))) : right-to-left (main.rs:123)
                // two materials per round: the loads of a round are independent of each other (the indices come from LDS), so a
                // pair costs one trip to L2, not two; the products are taken in the same order (c5 +0.4 %, round 3)
                {
                    auto path_idx = [&](uint32_t i) -> uint32_t {
                        return p.path32 ? reinterpret_cast<uint32_t*>(lpath)[i * BLOCK + cold<QNODES>(tid)]
                                        : (uint32_t) reinterpret_cast<uint16_t*>(lpath)[i * BLOCK + cold<QNODES>(tid16)];
                    };
                    uint32_t i = k;
#pragma clang loop unroll(disable)
                    while (i >= 2) {
                        WCOUNT(13);
                        LCOUNT(11);
                        const float4 ma = at32(p.mat, path_idx(i - 1)), mb = at32(p.mat, path_idx(i - 2));
                        term_r = mb.x * (ma.x * term_r);
                        term_g = mb.y * (ma.y * term_g);
                        term_b = mb.z * (ma.z * term_b);
                        i -= 2;
                    }
                    if (i) {
                        WCOUNT(13);
                        LCOUNT(11);
                        const float4 ma = at32(p.mat, path_idx(0));
                        term_r = ma.x * term_r;
                        term_g = ma.y * term_g;
                        term_b = ma.z * term_b;
                    }
                }
                // ---- the sample's colour goes to the unit's record; the wave sums a pixel's colours in order when it commits the slot
                {
                    const uint32_t slot = useq >> 24;
                    float* r = ring + (__umul24(slot, p.slot_stride) + 1u + (useq & 0xffffffu)) * 3u;
                    r[0] = term_r;
                    r[1] = term_g;
                    r[2] = term_b;
                    // one unit less to wait for, and the segments it traced (k hits on the path stack; the last segment ended in the sky
                    // or on a light unless the depth ran out on a hit)
                    const uint32_t usegs = k + ((hit && !(em > 0.0f)) ? 0u : 1u);
                    (void)__hip_atomic_fetch_add(&wq.cnt[slot], (usegs << SLOT_UNIT_BITS) - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    have_unit = false;
                }
            }
        }
        TSTAMP(4);
    }

    if (p.strip_cost && lane < 2 && wq.cost_acc[lane])             // what the wave's running sums still hold
        atomicAdd(&p.strip_cost[(blockIdx.x % COST_COPIES) * MAX_BATCH + wq.cost_strip[lane]], (unsigned long long)wq.cost_acc[lane]);
    drain_counters();
    TFLUSH;
#undef wq
#undef wst
}

// The kernels are instantiated in rt_kernels_lin.hip / rt_kernels_trav.hip; the host side (rt_api.hip) gets them here.
using KernelFn = void (*)(const KParams);
KernelFn kernel_linear(bool streamed, bool expanded);
void sqrt_selftest_launch(uint32_t from, unsigned long long n, unsigned long long* d_bad, hipStream_t st);   // rt_kernels_trav.hip
KernelFn kernel_traverse(int variant, bool stats = false);   // 0: exact nodes, 1: quantised nodes, 2: quantised nodes with the capped LDS stack,
                                         // 3: exact nodes, whole tree resident in LDS (1024-thread workgroups); 4: the same, nearer child first, distance culling
                                         // 5: quantised nodes, nearer child first, distance culling (spheres only); 6: the same, capped LDS stack
                                         // 7: exact nodes, nearer child first, distance culling (spheres and triangles)
                                         // stats: the variant that also counts node visits (RT_FLAG_COUNT_STEPS)
constexpr int LTREE_BLOCK = 1024;

}  // namespace rtk
