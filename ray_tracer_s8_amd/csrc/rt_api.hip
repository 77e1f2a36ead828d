// rt_api.hip — host side of the C-ABI declared in include/rt_tile.h.
//
// Plays the roles the reference gives to the slave's worker (ray-tracer-slave/src/main.rs:32-106:
// take a RenderInfo, produce the strip's RGB8 bytes) and, in rt_render_frame, to the
// controller's dispatch + assembly (ray-tracer-controller/src/main.rs:47-75, 109-115).
// No torch types, no CPU fallback: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "rt_assign.h"
#include "rt_bvh.h"
#include "rt_kernel.hip.h"
#include "rt_tile.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// Nothing may unwind across the C boundary (rt_tile.h: "never throws or aborts"): every extern "C" entry point runs its
// body through this.  The bodies allocate (std::vector, std::string, std::thread); a failed allocation becomes
// RT_ERR_OOM, anything else RT_ERR_HIP with the exception's text.  Setting the message must not throw either.
void set_err_noexcept(const char* what) noexcept {
    try {
        g_err = what;
    } catch (...) {
    }
}
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_err_noexcept("host allocation failed");
        return RT_ERR_OOM;
    } catch (const std::exception& e) {
        set_err_noexcept(e.what());
        return RT_ERR_HIP;
    } catch (...) {
        set_err_noexcept("internal error");
        return RT_ERR_HIP;
    }
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" __FILE__ \
                                                                                        ":" +         \
                                        std::to_string(__LINE__) + ")");                              \
    } while (0)

struct DeviceCtx {
    int dev = -1;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // D2H of finished strips while the next launch of the batch runs
    int n_cu = 0;
    std::mutex mu;   // serialises synchronous calls on one device
};

// Test / A-B knobs of the launch path.  They used to be getenv() calls inside launch_batch and rt_render_frame — read per
// launch from several worker threads while tests flipped them with setenv (undefined behaviour), and two of them overrode
// explicit request flags.  Now: one process-level table of atomics, filled from the environment ONCE by the first rt_init,
// changed afterwards only through rt_debug_set (exported, deliberately not in rt_tile.h: tests and tools only), and an
// explicit RT_FLAG_* in the request always wins over a knob.
enum DebugKnob {
    DBG_LDS_TREE = 0,      // RT_LDS_TREE        0: never the LDS-resident tree engine                      (default 1)
    DBG_CULL_WALK,         // RT_CULL_WALK       0 / 1: culled walk off / on wherever it is valid; -1: host rule (default -1)
    DBG_NO_STAGE,          // RT_NO_STAGE        1: no LDS output staging                                   (default 0)
    DBG_SLOTS,             // RT_SLOTS           sample units: pixel slots per wave (<= 32); 0: host rule                  (default 0)
    DBG_FORCE_CAPPED,      // RT_FORCE_CAPPED    1: quantised walks take the capped-stack kernel             (default 0)
    DBG_STACK_LDS,         // RT_STACK_LDS       capped-stack kernel: stack entries per lane in LDS; 0: STACK_LDS_MAX
    DBG_COMPACT,           // RT_COMPACT         0: per-lane root tests in the exact-node L2 kernel          (default 1)
    DBG_REFILL_EIGHTHS,    // RT_REFILL_EIGHTHS  refill threshold of the walks; 0: host rule
    DBG_COMMIT_SLOTS,      // RT_COMMIT_SLOTS    sample units: complete slots a commit waits for; 0: host rule             (default 0)
    DBG_VERBOSE,           // RT_VERBOSE         engine / LDS plan of every launch on stderr                 (default 0)
    DBG_REORDER,           // RT_REORDER         0: primitive records stay in the caller's order (A/B)          (default 1)
    DBG_TAIL_TILES,        // RT_TAIL_TILES      tiles at the end of a launch's queue handed out in quarters; -1: host rule (default -1)
    DBG_STRIP_COST,        // RT_STRIP_COST      0: the kernels do not count per-strip ray segments for the frame context (A/B)   (default 1)
    DBG_N
};
std::atomic<int> g_dbg[DBG_N];
const struct { const char* env; int def; } g_dbg_spec[DBG_N] = {
    {"RT_LDS_TREE", 1}, {"RT_CULL_WALK", -1}, {"RT_NO_STAGE", 0}, {"RT_SLOTS", 0}, {"RT_FORCE_CAPPED", 0},
    {"RT_STACK_LDS", 0}, {"RT_COMPACT", 1}, {"RT_REFILL_EIGHTHS", 0}, {"RT_COMMIT_SLOTS", 0}, {"RT_VERBOSE", 0}, {"RT_REORDER", 1}, {"RT_TAIL_TILES", -1}, {"RT_STRIP_COST", 1}};
std::once_flag g_dbg_once;
void dbg_load_env() {
    std::call_once(g_dbg_once, [] {
        for (int k = 0; k < DBG_N; k++) {
            const char* e = getenv(g_dbg_spec[k].env);
            int v = g_dbg_spec[k].def;
            if (e) v = k == DBG_VERBOSE ? 1 : atoi(e);
            g_dbg[k].store(v, std::memory_order_relaxed);
        }
    });
}
inline int dbg(DebugKnob k) { return g_dbg[k].load(std::memory_order_relaxed); }

std::mutex g_mu;
bool g_init = false;
std::vector<DeviceCtx*> g_ctx;
std::atomic<int> g_live_scenes{0};   // rt_shutdown is refused while any scene is alive (scenes point at their DeviceCtx)

constexpr size_t LDS_LIMIT = 160 * 1024 - 3072;   // dynamic LDS budget; 3 KiB left for the kernels' static LDS (the queue words and
                                                  // one rtk::WaveQ per wave: 16 + 16 x 160 bytes in the 1024-thread kernel)
constexpr uint32_t RESIDENT_MAX = rtk::CHUNK;   // spheres kept wholly in LDS
constexpr uint32_t STREAM_CHUNK = 2048;         // chunk size when streaming through LDS
constexpr uint32_t STACK_LDS_MAX = 12;          // quantised-node kernel: stack entries per lane in LDS, deeper ones in HBM
constexpr uint32_t TRAVERSE_MIN_TRIS = 4;       // ... or above this many triangles (tools/crossover_tris.py: the LDS-tree walk wins from 8 triangles up)
constexpr uint32_t RT_QNODES_MIN_PRIMS = 4096;   // from here up the traversal walks the 32-byte quantised nodes (tools/crossover_q.py)
constexpr uint32_t REORDER_MIN_PRIMS = 64;      // from here up the primitive records are stored in the tree's depth-first leaf order
constexpr uint32_t DENSE_SCAN_MAX_PRIMS = 192;       // piles of up to this many spheres at a box density of at least ...
constexpr float DENSE_SCAN_MIN_DENSITY = 3.0f;       // ... this keep the linear scan (see `traverse`)
constexpr float LT_CULL_MIN_DENSITY = 2.2f;          // the LDS-resident tree is walked nearer child first, with distance culling, from this box density up
                                                     // (tests/test_gpu_engine_rules.py: at 1.1 ... 1.8 a helix, a lattice and a colonnade of 700 ... 960
                                                     // spheres lose 7 ... 17 % to the culled step, piles at 1.8 / 2.7 gain 2 / 38 %)
constexpr uint32_t TRAVERSE_MIN_PRIMS = 2;      // from this many primitives up the BVH-traversal engine is the default.  (Rounds 1-3: 32, with a density rule below it;
                                                // with the sample units the LDS-resident tree leads the scan on every scene of tools/small_scene_matrix.py —
                                                // 2 ... 32 spheres, five families, 1.02 ... 1.36 x — but one pile of 32, and on c2's 16-sphere room by 5 ... 8 %.)

}  // namespace

struct rt_scene {
    DeviceCtx* ctx = nullptr;
    uint32_t n_sph = 0, n_sph_pad = 0, n_tri = 0;
    float4* d_geom = nullptr;
    float4* d_geom_pk = nullptr;
    float4* d_geom_px = nullptr;   // expanded-form broad phase records
    bool expanded = false;         // host heuristic: expanded-form margin small against r^2
    float4* d_mat = nullptr;
    float* d_emis = nullptr;
    float* d_tri = nullptr;
    float4* d_tri_box = nullptr;
    float4* d_bvh = nullptr;       // rtbvh::FlatNode[]
    float4* d_trav = nullptr;      // rtbvh::TravNode[]
    uint32_t* d_stack_ovf = nullptr;   // quantised-node kernel: stack entries beyond the LDS part, [entry][thread]
    size_t stack_ovf_words = 0;
    hipEvent_t ovf_done = nullptr;     // end of the last launch that used d_stack_ovf: the area is one per scene, so such
                                       // launches are chained even when the caller spreads them over several streams
    uint4* d_travq = nullptr;      // rtbvh::QNode[] (quantised twin)
    float4* d_geom_r = nullptr;    // (cx,cy,cz,radius)
    uint32_t* d_big = nullptr;     // culled walk: spheres root-tested at query start (too large for its slack)
    uint32_t n_big = 0;
    float r_slack = 0.f;           //   largest radius among the other spheres
    bool cull_pays = false;        //   host heuristic: the scene is dense enough for the culled walk (build_host_scene)
    bool inverted_boxes = false;   // a sphere of negative radius: its AABB has lo > hi (sphere.rs:65-72), see launch_batch
    float tri_k = 0.f, tri_diag = 0.f, tri_es = 0.f, tri_e = 0.f;   // culled walk over the exact nodes: maxima over the triangles not in `big`
    bool xcull_pays = false;       //   host heuristic for scenes with triangles
    bool tri_ok = false;
    float cull_density = 0.f;      //   the box density behind it
    rtbvh::QGrid grid;
    float leaf_density = 0.f;      // sum of primitive box areas / scene box area (node-format heuristic)
    bool quant_ok = false;         // quantised walk usable and worthwhile (grid step small against the primitives)
    uint32_t root_ref = 0, bvh_depth = 0, n_internal = 0;
    uint32_t* d_leaf_of = nullptr;
    uint32_t* d_world_rank = nullptr;   // world_index of every primitive when the caller gave one (rt_tile.h "the world's order")
    bool has_order = false;
    float bvh_build_ms = 0.f;
    unsigned long long* d_counters = nullptr;   // [0..2] stats, [4 + slot] tile queues
    // Sample-unit rings (rt_kernel.hip.h "Sample units"): scratch of one launch, [waves][slots][16 B].  Launches on ONE stream
    // follow each other, so a ring belongs to the stream that used it last; a launch on another stream that has to share it
    // (more streams than rings) waits for that launch's end first.
    struct Ring {
        float* d = nullptr;
        size_t bytes = 0;
        hipStream_t last = nullptr;
        hipEvent_t done = nullptr;          // end of the last launch that used it
        uint64_t stamp = 0;
    };
    Ring rings[4];
    uint64_t ring_clock = 0;
    unsigned long long* d_cost = nullptr;       // per-strip ray segments of the host-buffer batched call (frame context)
    size_t d_cost_cap = 0;
    // staging for the host-buffer entry point (grown on demand)
    uint8_t* d_out = nullptr;
    size_t d_out_cap = 0;
    float* d_outf = nullptr;
    size_t d_outf_cap = 0;
    // HIP-event bookkeeping of launches not yet collected
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, free_ev;
    uint64_t primary_rays = 0;
    float h2d_ms = 0.f;
    uint32_t last_engine = 0, last_form = 0;
    std::mutex mu;
};

namespace {

int check_request(const rt_tile_request* rq) {
    if (!rq) return fail(RT_ERR_BAD_ARG, "request is NULL");
    if (rq->width == 0 || rq->height == 0 || rq->divisions == 0 || rq->spp == 0)
        return fail(RT_ERR_BAD_ARG, "width, height, divisions and spp must be non-zero");
    if (rq->division_no >= rq->divisions) return fail(RT_ERR_BAD_ARG, "division_no >= divisions");
    if (rq->height / rq->divisions == 0) return fail(RT_ERR_BAD_ARG, "height / divisions == 0 rows");
    if (rq->max_bounces > RT_MAX_BOUNCES) return fail(RT_ERR_LIMIT, "max_bounces > RT_MAX_BOUNCES");
    if (rq->spp > RT_MAX_SPP) return fail(RT_ERR_LIMIT, "spp > RT_MAX_SPP");
    if (rq->reserved != 0) return fail(RT_ERR_BAD_ARG, "reserved must be 0");
    if ((uint64_t)rq->width * rq->height > 0x7fffffffull) return fail(RT_ERR_LIMIT, "image too large");
    return RT_OK;
}

// Camera::new with the slave's arguments (main.rs:42-50 -> camera.rs:19-47)
void fill_camera(const rt_tile_request* rq, rtk::KParams& p) {
    const float origin[3] = {0.f, 0.f, 0.f};          // Point3::ZERO
    const float aspect_ratio = (float)rq->width / (float)rq->height;
    const float image_height = (float)rq->height;
    const float vh = 2.0f * std::tan(rq->fov / 2.0f);
    const float vw = aspect_ratio * vh;
    const float hor[3] = {vw, 0.f, 0.f}, ver[3] = {0.f, vh, 0.f};
    const float foc[3] = {0.f, 0.f, rq->focal_length};
    for (int i = 0; i < 3; i++) {
        p.org[i] = origin[i];
        p.hor[i] = hor[i];
        p.ver[i] = ver[i];
        // origin - horizontal / 2 - vertical / 2 - (0,0,focal_length)
        float v = origin[i] - hor[i] / 2.0f;
        v = v - ver[i] / 2.0f;
        v = v - foc[i];
        p.llc[i] = v;
    }
    p.lens_radius = rq->aperture / 2.0f;
    p.focus_distance = rq->focus_distance;
    p.u_den = aspect_ratio * image_height - 1.0f;      // camera.rs:116
    p.v_den = image_height - 1.0f;                     // camera.rs:117
}

struct EvPair {
    hipEvent_t a, b;
};

int get_events(rt_scene* sc, EvPair& ev) {
    if (!sc->free_ev.empty()) {
        ev.a = sc->free_ev.back().first;
        ev.b = sc->free_ev.back().second;
        sc->free_ev.pop_back();
        return RT_OK;
    }
    HIPCHK(hipEventCreate(&ev.a));
    HIPCHK(hipEventCreate(&ev.b));
    return RT_OK;
}

constexpr uint32_t QUEUE_SLOTS = 16384;          // uncollected launches per scene (one 8-byte queue head each)
#ifdef RT_PROFILE_TIME
constexpr uint32_t COUNTER_WORDS = 4 + QUEUE_SLOTS + 16 * 8192;      // (+ one block per wave: the phase clock, tools/phase_time.py)
#else
constexpr uint32_t COUNTER_WORDS = 4 + QUEUE_SLOTS;
#endif

bool same_frame(const rt_tile_request& a, const rt_tile_request& b) {
    return a.width == b.width && a.height == b.height && a.divisions == b.divisions && a.spp == b.spp &&
           a.max_bounces == b.max_bounces && a.aperture == b.aperture && a.focus_distance == b.focus_distance &&
           a.fov == b.fov && a.focal_length == b.focal_length && a.t_min == b.t_min && a.t_max == b.t_max &&
           a.flags == b.flags;
}

// Enqueue a batch of strips of one frame (<= MAX_BATCH) as ONE launch of persistent waves.
// Caller holds sc->mu and has the device current.
// d_strip_cost: optional device array of COST_COPIES x MAX_BATCH counters (zeroed by the caller, on `stream`): the kernels add the ray
// segments of strip i of the batch to [copy][i] (KParams::strip_cost); the caller sums the copies.
int launch_batch(rt_scene* sc, const rt_tile_request* rqs, uint32_t n, void* const* d_rgb, void* const* d_f32,
                 hipStream_t stream, unsigned long long* d_strip_cost = nullptr) {
    const rt_tile_request* rq = &rqs[0];
    rtk::KParams p;
    std::memset(&p, 0, sizeof p);
    p.W = rq->width;
    p.H = rq->height;
    p.Hs = rq->height / rq->divisions;
    p.spp = rq->spp;
    p.depth = rq->max_bounces + 1;
    p.n_sph = sc->n_sph;
    p.n_sph_pad = sc->n_sph_pad;
    p.n_tri = sc->n_tri;
    p.flags = rq->flags;
    // A sphere of negative radius has an AABB with lo > hi (Sphere::aabb = center -+ radius, sphere.rs:65-72): the reference's
    // sign-selected slab test rejects such a box for (almost) every ray, while the finite-direction shortcut of the kernels
    // (min / max of the two plane values, valid for lo <= hi) would enter it.  Such a scene is rendered with the crate's
    // literal test and the whole box chain throughout (the RT_FLAG_FULL_CHAIN path): slower, and exact.
    if (sc->inverted_boxes) p.flags |= RT_FLAG_FULL_CHAIN;
    // Engine choice.  BVH traversal reproduces reference semantics only, needs the tree to fit the traversal
    // stack, and pays off once the scene is larger than a couple of LDS chunks; RT_FLAG_BVH_TRAVERSE /
    // RT_FLAG_LINEAR_SCAN force either engine for A/B runs and tests.
    const uint32_t n_prims = sc->n_sph + sc->n_tri;
    const bool trav_ok = !(rq->flags & (RT_FLAG_EXACT_SCAN | RT_FLAG_NO_BVH_CULL | RT_FLAG_LINEAR_SCAN)) &&
                         sc->bvh_depth < (uint32_t)rtk::TRAV_STACK && n_prims > 0;   // LDS stack: (depth + 1) KiB per workgroup
    // (the linear engines test every triangle's box per segment: meshes switch to the tree much earlier)
    // (Small PILES of overlapping spheres keep the scan: at a box density of 3 and more a ray meets so many leaf boxes that up
    // to about 200 spheres the scan's 64 packed instructions per 8 spheres beat any walk — tools/dense_matrix.py, 48 ... 192 spheres at
    // density 3.3 ... 13: the culled LDS-tree walk of round 4 renders them at 0.67 ... 0.84 of the scan (the plain one: 0.59 ... 0.92), at
    // 256 it leads by 1.4 ... 1.5 x.  Round 3's two further pile rules — up to 384 spheres at densities 5 ... 12 to the scan, larger or
    // denser piles to the culled L2 walk although their tree fits LDS — are gone: the culled LDS-tree walk is the best engine in
    // every cell of tools/dense_mid_matrix.py, by 13 ... 30 %.)
    const bool dense_pile = sc->n_tri == 0 && n_prims <= DENSE_SCAN_MAX_PRIMS && sc->cull_density >= DENSE_SCAN_MIN_DENSITY;
    const bool traverse = trav_ok && ((rq->flags & RT_FLAG_BVH_TRAVERSE) || (n_prims >= TRAVERSE_MIN_PRIMS && !dense_pile) ||
                                      sc->n_tri > TRAVERSE_MIN_TRIS);
    // node format: from RT_QNODES_MIN_PRIMS primitives up the 32-byte quantised nodes (half the gather footprint, and an
    // LDS plan that keeps five workgroups per CU whatever the tree's depth): +14 % on sparse fields of every size, +17...29 %
    // on dense fields of 32 768+ spheres, within 2.5 % either way in between; below it the exact-node kernel's six
    // waves per SIMD win on the headline scene (c3 +1.5 %).  tools/crossover_q.py, DESIGN.md 4.7
    // (meshes keep the exact nodes: a quantised walk validates a triangle leaf by walking its box chain — two more gathers
    // per improving hit — and lost 2...16 % on the generated terrains of 7 200 and 100 352 triangles, tools/heuristics_matrix.py)
    // LDS-resident tree (engine 4, kernel variant 3): the exact 64-byte nodes of a small scene staged into LDS by one
    // 1024-thread workgroup per CU, 16-bit references / stack / leaf lists.  RT_FLAG_NO_LDS_TREE forces the
    // L2-gather kernel (A/B runs, tests).
    const bool ltree_env = dbg(DBG_LDS_TREE) != 0;
    bool ltree_fits = false;
    const size_t lt_lane = ((size_t)rtk::MAXL_LTREE + (size_t)(rq->max_bounces + 1) + (size_t)(sc->bvh_depth + 2)) * sizeof(uint16_t);
    if (traverse && ltree_env && !(rq->flags & RT_FLAG_NO_LDS_TREE) && sc->n_internal > 0 && n_prims <= 0x7fffu &&
        ((size_t)sc->n_internal + 2) * rtk::LNODE_DW < 0x8000u) {
        const size_t fixed = (((size_t)sc->n_internal + 2) * rtk::LNODE_DW + n_prims) * 4 + 16 + lt_lane * rtk::LTREE_BLOCK;   // + node DONE and the NaN field
        ltree_fits = fixed <= LDS_LIMIT;
    }
    // (Below the threshold a DENSE sphere scene whose tree does not fit LDS also takes the quantised nodes, for the culled
    // walk below: tools/cull_matrix_small.py, 2 000...3 500 overlapping spheres 1.55...1.95 x over the exact-node walk, fields of
    // box density 1...2 0.87...0.98 — hence the higher bar of 2.5 here.)
    // (round 4: piles whose tree FITS LDS no longer need a rule — the LDS-resident tree has its own culled walk, below)
    const bool dense_mid = sc->cull_pays && !(rq->flags & RT_FLAG_NO_CULL_WALK) && !ltree_fits && sc->cull_density >= 2.5f &&
                           n_prims >= 512;
    // (... and so does a sphere FIELD between the LDS tree's limit and that threshold: at box densities of 0.15 and more the quantised
    // walk leads the exact one by 17...24 % there — tools/qnodes_mid_matrix.py, 1200...4000 spheres; flat sheets of small spheres,
    // 0.03...0.1, are the scenes the exact nodes win by up to 12 %, and clusters fail quant_ok)
    const bool field_mid = !ltree_fits && sc->n_tri == 0 && n_prims < RT_QNODES_MIN_PRIMS && sc->cull_density >= 0.15f;
    const bool qnodes = traverse && sc->quant_ok && !(rq->flags & RT_FLAG_EXACT_NODES) &&
                        ((rq->flags & RT_FLAG_QUANT_NODES) || (n_prims >= RT_QNODES_MIN_PRIMS && sc->n_tri <= sc->n_sph) || dense_mid ||
                         field_mid);
    const bool ltree = ltree_fits && !qnodes;
    // Culled walk (engine 5, kernel variant 5): the quantised walk nearer child first, subtrees beyond the running closest hit
    // skipped.  Spheres only (the bound is derived from the sphere root test's error terms).
    // Default where the host heuristic says it pays (cull_pays: DESIGN.md 4.7); RT_FLAG_CULL_WALK / RT_FLAG_NO_CULL_WALK
    // force it on / off (A/B runs, tests), RT_CULL_WALK=0/1 likewise for a whole process.
    // (an explicit request flag wins over the process-level knob, the knob over the host rule)
    const int cull_env = dbg(DBG_CULL_WALK);
    const bool cull_want = (rq->flags & RT_FLAG_NO_CULL_WALK) ? false : (rq->flags & RT_FLAG_CULL_WALK) ? true
                           : cull_env >= 0 ? cull_env != 0 : sc->cull_pays;
    const bool cull = qnodes && cull_want && sc->n_tri == 0 && std::isfinite(sc->r_slack);
    // ... and over the exact nodes (kernel variant 7): scenes with triangles — the bound of cull_bound_tri — wherever the exact-node
    // L2 walk is the engine; default where the host heuristic says it pays (xcull_pays), forced by the same flags
    const bool xcull_want = (rq->flags & RT_FLAG_NO_CULL_WALK) ? false : (rq->flags & RT_FLAG_CULL_WALK) ? true
                            : cull_env >= 0 ? cull_env != 0 : sc->xcull_pays;
    const bool xcull = traverse && !qnodes && !ltree && xcull_want && sc->n_tri > 0 && sc->tri_ok && std::isfinite(sc->r_slack) &&
                       !sc->inverted_boxes;
    // ... and in the LDS-resident tree (kernel variant 4, engine 7; round 4): the same bound, so the same premises (a finite slack
    // radius, triangles within the K limit or in the `big` list, no inverted boxes); default from a box density of LT_CULL_MIN_DENSITY
    // up — below it the rays meet so few leaf boxes that ordering the children costs more than the skipped subtrees save
    // (tools/dense_matrix.py, tools/dense_mid_matrix.py, tests/test_gpu_engine_rules.py)
    const bool lt_cull_ok = ltree && !sc->inverted_boxes && std::isfinite(sc->r_slack) && (sc->n_tri == 0 || sc->tri_ok);
    const bool lt_cull_want = (rq->flags & RT_FLAG_NO_CULL_WALK) ? false : (rq->flags & RT_FLAG_CULL_WALK) ? true
                              : cull_env >= 0 ? cull_env != 0 : sc->cull_density >= LT_CULL_MIN_DENSITY;
    const bool ltcull = lt_cull_ok && lt_cull_want;
    const bool streamed = !traverse && sc->n_sph_pad > RESIDENT_MAX;
    p.chunk = traverse ? 0 : (streamed ? STREAM_CHUNK : sc->n_sph_pad);
    p.n_chunks = p.chunk ? (sc->n_sph_pad + p.chunk - 1) / p.chunk : 0;
    p.path32 = (sc->n_sph + sc->n_tri) > 65536u ? 1u : 0u;
    size_t geom_bytes = traverse ? 0 : (size_t)(p.chunk ? p.chunk : 1) * sizeof(float4);
    p.lds_cand_off = (uint32_t)geom_bytes;
    size_t path_bytes = (size_t)p.depth * rtk::BLOCK * (p.path32 ? 4 : 2);
    // ---- LDS plan of a traversal launch.  Occupancy is worth more than long leaf lists (c3: 6 workgroups per CU with
    // 7 slots +1.5 % over 5 with 8; c5: 5 with 5 slots +7.5 % over 4 with 8), and an uncapped stack more than either
    // (the HBM-overflow test on every push / pop costs 6...10 %).  So: the target number of workgroups per CU follows
    // from the kernel's registers (five waves per SIMD for both node formats).  The exact-node kernel has 7 slots, fixed; the quantised
    // kernel's lists shrink from MAXL down to MINL slots to reach its target, and a quantised walk whose whole stack still
    // does not fit takes the capped-stack kernel.
    // stack slots per lane: up to bvh_depth pending right children (+ 1 spare); the LDS-tree kernel's branch-free step
    // adds the DONE sentinel in slot 0 and needs the free slot its unconditional stores land in
    // output staging (one tile per wave): wherever the LDS plan has room for it
    const bool want_stage = dbg(DBG_NO_STAGE) == 0 && ((traverse && !ltree) || streamed);     // the kernels it is compiled into (see there)
    const bool list16 = traverse && !ltree && n_prims <= 65536u;          // 16-bit leaf-list entries: half the LDS
    const size_t stage_bytes_wg = (size_t)rtk::STAGE_TILES * rtk::STAGE_TILE_BYTES * ((ltree ? rtk::LTREE_BLOCK : rtk::BLOCK) / 64);
    const uint32_t stack_capped = sc->bvh_depth + 1;
    const uint32_t stack_need = sc->bvh_depth + (ltree ? 2u : 1u);
    uint32_t maxl = qnodes ? (uint32_t)rtk::MAXL : (uint32_t)rtk::MAXL_EXACT, stack_lds = stack_need;
    bool capped = false;
    if (traverse && qnodes) {
        const size_t per_wg = (160u * 1024u - 4096u) / 5u - 512u;     // 4 KiB of slack, 464 B static LDS
        const size_t slot = (size_t)rtk::BLOCK * (list16 ? sizeof(uint16_t) : sizeof(uint32_t));
        const size_t fixed = path_bytes + (size_t)stack_need * rtk::BLOCK * sizeof(uint32_t) + (want_stage ? stage_bytes_wg : 0);
        // (DBG_FORCE_CAPPED / DBG_STACK_LDS: tests drive the capped-stack kernel with small trees)
        const bool force_capped = dbg(DBG_FORCE_CAPPED) != 0;
        if (!force_capped && fixed + (size_t)rtk::MINL * slot <= per_wg) {
            maxl = (uint32_t)std::min<size_t>((size_t)rtk::MAXL, (per_wg - fixed) / slot);
        } else {
            const uint32_t cap = dbg(DBG_STACK_LDS) > 0 ? (uint32_t)dbg(DBG_STACK_LDS) : STACK_LDS_MAX;
            capped = stack_capped > cap;
            stack_lds = capped ? cap : stack_need;
        }
    }
    p.maxl = maxl;
    p.stack_lds = stack_lds;
    p.list16 = list16 ? 1u : 0u;
    size_t cand_bytes = traverse ? (size_t)maxl * rtk::BLOCK * (list16 ? sizeof(uint16_t) : sizeof(uint32_t))
                                 : (size_t)rtk::MAXC * rtk::BLOCK * sizeof(uint16_t);
    p.lds_path_off = (uint32_t)(geom_bytes + cand_bytes);
    const bool expanded = !traverse && sc->expanded && !(rq->flags & RT_FLAG_OC_BROAD_PHASE);
    p.lds_rr_off = (uint32_t)(geom_bytes + cand_bytes + path_bytes);
    size_t rr_bytes = expanded ? (size_t)(p.chunk ? p.chunk : 1) * sizeof(float) : 0;
    p.lds_stack_off = (uint32_t)(geom_bytes + cand_bytes + path_bytes + rr_bytes);
    size_t stack_bytes = traverse ? (size_t)stack_lds * rtk::BLOCK * sizeof(uint32_t) : 0;
    size_t lds = geom_bytes + cand_bytes + path_bytes + rr_bytes + stack_bytes;
    p.n_internal = sc->n_internal;
    p.lds_node_off = 0;
    const int bs = ltree ? rtk::LTREE_BLOCK : rtk::BLOCK;
    if (ltree) {
        // [nodes][leaf lists u16][path u16][stack u16]
        size_t off = ((((size_t)sc->n_internal + 2) * rtk::LNODE_DW + n_prims) * 4 + 15) & ~(size_t)15;    // nodes, DONE, NaN field of n_prims + 19 dwords
        p.lds_cand_off = (uint32_t)off;
        // leaf-list slots: MAXL_LTREE, and up to MAXL_LTREE_MAX where the tree leaves room (fewer flushes forced by a full list)
        uint32_t lt_maxl = rtk::MAXL_LTREE;
        while (lt_maxl < (uint32_t)rtk::MAXL_LTREE_MAX &&
               off + ((size_t)(lt_maxl + 1) + p.depth + stack_need) * bs * sizeof(uint16_t) <= LDS_LIMIT) lt_maxl++;
        p.maxl = lt_maxl;
        off += (size_t)lt_maxl * bs * sizeof(uint16_t);
        p.lds_path_off = (uint32_t)off;
        off += (size_t)p.depth * bs * sizeof(uint16_t);
        p.lds_stack_off = (uint32_t)off;
        off += (size_t)stack_need * bs * sizeof(uint16_t);
        lds = off;
    }
    // compacted root tests of the exact-node L2 kernel (1 KiB per wave; RT_COMPACT=0 keeps the per-lane flush for A/B runs)
    p.lds_cmp_off = 0xffffffffu;
    const bool compact_env = dbg(DBG_COMPACT) != 0;
    if (traverse && !qnodes && !ltree && compact_env && lds + 1024u * (rtk::BLOCK / 64) + 16 <= LDS_LIMIT) {
        lds = (lds + 15) & ~(size_t)15;
        p.lds_cmp_off = (uint32_t)lds;
        lds += 1024u * (rtk::BLOCK / 64);
    }
    p.lds_stage_off = 0xffffffffu;
    if (want_stage && lds + stage_bytes_wg <= LDS_LIMIT) {
        lds = (lds + 15) & ~(size_t)15;
        p.lds_stage_off = (uint32_t)lds;
        lds += stage_bytes_wg;
    }
    if (lds > LDS_LIMIT) return fail(RT_ERR_LIMIT, "LDS budget exceeded (scene chunk + path stack)");
    fill_camera(rq, p);
    p.t_min = rq->t_min;
    p.t_max = rq->t_max;
    p.spp_f = (float)rq->spp;
    p.spp_rcp = (rq->spp & (rq->spp - 1u)) == 0u ? 1.0f / (float)rq->spp : 0.0f;   // a power of two up to 2^31: exact in f32
    p.geom_pk = sc->d_geom_pk;
    p.geom_px = sc->d_geom_px;
    p.geom = sc->d_geom;
    p.mat = sc->d_mat;
    p.emis = sc->d_emis;
    p.tri = sc->d_tri;
    p.tri_box = sc->d_tri_box;
    p.bvh_nodes = sc->d_bvh;
    p.trav = sc->d_trav;
    p.travq = sc->d_travq;
    p.geom_r = sc->d_geom_r;
    for (int i = 0; i < 3; i++) {
        p.q_base[i] = sc->grid.base[i];
        p.q_step[i] = sc->grid.step[i];
        p.q_rstep[i] = 1.0f / sc->grid.step[i];
    }
    p.root_ref = sc->root_ref;
    if (ltree) p.root_ref = (p.root_ref & rtk::LEAF_BIT) ? (0x8000u | (p.root_ref & 0x7fffu)) : rtk::lt_r0(sc->n_internal) + p.root_ref * (uint32_t)rtk::LNODE_DW;
    {
        // refill threshold: long walks (large scenes) want finished lanes replaced sooner, short walks amortise the
        // per-round shading / ray-generation code over more finished lanes (tools/variants_q.sh sweeps)
        const int forced = dbg(DBG_REFILL_EIGHTHS);
        p.refill_eighths = forced > 0 ? (uint32_t)forced : (n_prims >= RT_QNODES_MIN_PRIMS ? 4u : 2u);
    }
    p.leaf_of = sc->d_leaf_of;
    p.world_rank = sc->has_order ? sc->d_world_rank : nullptr;
    p.n_strips = n;
    // Tile shape: 64x1 keeps each tile on whole 64-byte lines of the RGB8 strip (64 px * 3 B = 3 lines),
    // so one CU / one XCD L2 writes every byte of a line; 8x8 tiles split lines across XCDs and doubled the
    // HBM write traffic (profiles/r01_*).
    p.tiles_x = (p.W + 63u) / 64u;
    p.tiles_per_strip = p.tiles_x * p.Hs;
    // Sample units (rt_kernel.hip.h): pixel slots per wave, the commit threshold, the division by spp
    {
        p.grp = p.spp >= 8u ? 1u : (8u + p.spp - 1u) / p.spp;            // a slot is at least 8 units
        const uint64_t slot_units = (uint64_t)p.grp * p.spp;
        p.grp_magic = p.grp > 1u ? (uint32_t)((1ull << 32) / p.grp) + 1u : 0u;
        p.slot_stride = 1u + (uint32_t)slot_units;
        // enough slots for the pixels in flight (64 lanes' units, each pixel open as long as its longest path) plus the complete
        // ones a commit waits for.  The price of every slot is scratch that L2 has to keep between a sample's store and its pixel's
        // commit; what L2 does not keep goes out to HBM (c3, WRITE_SIZE per 23.7 MiB frame / Mrays/s: 32 slots 135 / 15 580, 24 slots
        // 88 / 15 330, 20 slots 40 / 15 270, 16 slots 33 / 14 300; c4 at 12 / 16 / 24 slots: 15 615 / 16 070 / 16 220 Mrays/s).  The
        // rate is what this path is measured by, HBM is idle either way (c3: 20 GB/s of 8 TB/s): 384 units per wave, at most 32 slots
        // — but never fewer than 16 pixels open while a pixel is at most 256 units: a slot is free again only when its LAST sample is in,
        // and with the 4 slots the 384 units gave the reference's literal 100 samples per pixel a wave stood still for want of a slot
        // (the mesh at 100 spp: 4 / 8 / 16 / 32 slots 6 990 / 7 250 / 7 340 / 7 370 Mrays/s); 8 up to 1 024 units, 4 beyond (scratch:
        // 12 bytes per unit and slot for every wave of the grid)
        const int forced = dbg(DBG_SLOTS);
        const uint64_t fewest = slot_units <= 256u ? 16u : slot_units <= 1024u ? 8u : 4u;
        p.n_slots = forced > 0 ? std::min<uint32_t>((uint32_t)forced, rtk::SLOTS_MAX)
                               : (uint32_t)std::min<uint64_t>(rtk::SLOTS_MAX, std::max<uint64_t>(fewest, 384u / slot_units));
        const uint32_t cs = dbg(DBG_COMMIT_SLOTS) > 0 ? (uint32_t)dbg(DBG_COMMIT_SLOTS) : std::max<uint32_t>(1u, p.n_slots / 4u);     // (c3: 4 ... 20 of 32 within 1 %)
        p.commit_slots = std::min<uint32_t>(cs, p.n_slots);
        // q / d == mulhi(q, floor(2^32 / d) + 1) whenever q * d < 2^32: q < 65 * spp with spp <= RT_MAX_SPP (4096)
        p.spp_magic = p.spp > 1u ? (uint32_t)((1ull << 32) / p.spp) + 1u : 0u;
        p.slotu_magic = (uint32_t)((1ull << 32) / slot_units) + 1u;
    }
    const uint64_t n_tiles = (uint64_t)p.tiles_per_strip * n;
    if (n_tiles > 0x1fffffffull) return fail(RT_ERR_LIMIT, "too many tiles in one launch");
    p.tiles_total = (uint32_t)n_tiles;
    p.n_tiles = (uint32_t)n_tiles;             // (queue entries: the split into whole tiles and quarters follows the grid below)
    p.tiles_big = (uint32_t)n_tiles;
    for (uint32_t i = 0; i < n; i++) {
        p.strips[i].seed = rqs[i].seed;
        p.strips[i].rgb = (uint8_t*)d_rgb[i];
        p.strips[i].f32 = d_f32 ? (float*)d_f32[i] : nullptr;
        p.strips[i].y0 = p.Hs * rqs[i].division_no;
    }
    const uint32_t slot = (uint32_t)sc->pending.size();
    if (slot >= QUEUE_SLOTS) return fail(RT_ERR_LIMIT, "too many uncollected launches: call rt_scene_collect()");
    p.counters = sc->d_counters;
    p.queue = sc->d_counters + 4 + slot;
    p.strip_cost = dbg(DBG_STRIP_COST) ? d_strip_cost : nullptr;

    // persistent grid: as many workgroups as the chip holds at this LDS/VGPR budget
    int per_cu = 0;
    const bool count_steps = traverse && (rq->flags & RT_FLAG_COUNT_STEPS);
    const bool cull_run = cull;
    p.big = sc->d_big;
    p.n_big = sc->n_big;
    p.r_slack = sc->r_slack;
    p.tri_k = sc->tri_k;
    p.tri_diag = sc->tri_diag;
    p.tri_es = sc->tri_es;
    p.tri_e = sc->tri_e;
    const rtk::KernelFn kern = traverse ? rtk::kernel_traverse(ltree ? (ltcull ? 4 : 3) : qnodes ? (cull_run ? (capped ? 6 : 5) : capped ? 2 : 1) : xcull ? 7 : 0, count_steps) : rtk::kernel_linear(streamed, expanded);
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, bs, lds));
    if (per_cu < 1) per_cu = 1;
    if (dbg(DBG_VERBOSE))
        fprintf(stderr, "[rt] engine %d%s  lds %zu B  workgroups/CU %d  leaf slots %u  bvh depth %u  leaf density %.3f  prims %u\n",
                traverse ? (ltree ? (ltcull ? 7 : 4) : qnodes ? (cull_run ? 5 : 3) : xcull ? 6 : 2) : (streamed ? 1 : 0), capped ? " (capped stack)" : "", lds, per_cu, maxl,
                sc->bvh_depth, sc->leaf_density, n_prims);
    uint32_t blocks = (uint32_t)sc->ctx->n_cu * (uint32_t)per_cu;
    const uint32_t waves_per_wg = (uint32_t)bs / 64u;
    const uint32_t useful = (p.n_tiles + waves_per_wg - 1) / waves_per_wg;   // a wave needs at least one tile
    if (blocks > useful) blocks = useful ? useful : 1;
    // Queue entries.  Whole tiles (64 pixels x spp units) first, and the LAST ones — two tiles per wave of the grid — in parts: quarters
    // (16 pixels), so that the launch's tail is one short entry long.  Two cases take parts for EVERY tile (round 4, measured on the
    // 100 352-triangle mesh: 1080p / 4 spp +13 %, 100 spp +19 %): a launch with fewer than 16 tiles per wave — its expensive tiles
    // (handed out first: the bottom rows) are still being worked on when the cheap ones at the end of the queue have long run out, and
    // an expensive whole tile is a large share of such a launch — and more than 16 samples per pixel, where a whole tile is thousands of
    // units; from 33 samples per pixel up the parts are sixteenths (4 pixels).  The price where it is not needed: 1-2 % (c2, c4).
    {
        const int forced = dbg(DBG_TAIL_TILES);
        const uint64_t waves = (uint64_t)blocks * waves_per_wg;
        const bool all_parts = p.tiles_total < 16ull * waves || p.spp > 16u;
        uint64_t conv = std::min<uint64_t>(p.tiles_total, forced >= 0 ? (uint64_t)forced : all_parts ? (uint64_t)p.tiles_total : 2ull * waves);
        p.sub_shift = p.spp > 32u ? 4u : 2u;
        if ((((uint64_t)p.tiles_total - conv) + (conv << p.sub_shift)) > 0x7fffffffull) p.sub_shift = 2u;       // (entry numbers are 31 bits)
        p.tiles_big = p.tiles_total - (uint32_t)conv;
        p.n_tiles = p.tiles_big + ((uint32_t)conv << p.sub_shift);
    }
    p.ovf_stride = blocks * (uint32_t)bs;
    p.stack_ovf = nullptr;
    if (traverse && capped) {
        const size_t words = (size_t)(stack_capped - stack_lds) * p.ovf_stride;
        if (words > sc->stack_ovf_words) {
            if (sc->d_stack_ovf) {
                if (sc->ovf_done) HIPCHK(hipEventSynchronize(sc->ovf_done));   // a launch in flight may still use the old area
                (void)hipFree(sc->d_stack_ovf);
                sc->d_stack_ovf = nullptr;
                sc->stack_ovf_words = 0;
            }
            HIPCHK(hipMalloc(&sc->d_stack_ovf, words * sizeof(uint32_t)));
            sc->stack_ovf_words = words;
        }
        p.stack_ovf = sc->d_stack_ovf;
    }
    // the launch's ring: the one this stream used last, else a free one, else the least recently used (after its last launch)
    rt_scene::Ring* rg = nullptr;
    {
        const size_t ring_bytes = (size_t)blocks * waves_per_wg * p.n_slots * p.slot_stride * 12u;
        for (auto& r : sc->rings)
            if (r.d && r.last == stream) { rg = &r; break; }
        if (!rg)
            for (auto& r : sc->rings)
                if (!r.d) { rg = &r; break; }
        if (!rg) {
            rg = &sc->rings[0];
            for (auto& r : sc->rings)
                if (r.stamp < rg->stamp) rg = &r;
        }
        if (rg->bytes < ring_bytes) {
            if (rg->d) {
                if (rg->done) HIPCHK(hipEventSynchronize(rg->done));     // a launch in flight may still use the old area
                (void)hipFree(rg->d);
                rg->d = nullptr;
                rg->bytes = 0;
            }
            HIPCHK(hipMalloc(&rg->d, ring_bytes));
            rg->bytes = ring_bytes;
        }
        if (!rg->done) HIPCHK(hipEventCreateWithFlags(&rg->done, hipEventDisableTiming));
        else if (rg->last != stream) HIPCHK(hipStreamWaitEvent(stream, rg->done, 0));
        rg->last = stream;
        rg->stamp = ++sc->ring_clock;
        p.ring = rg->d;
    }
    dim3 grid(blocks), block(bs);
    EvPair ev;
    int rc = get_events(sc, ev);
    if (rc) return rc;
    if (p.stack_ovf) {
        if (sc->ovf_done) HIPCHK(hipStreamWaitEvent(stream, sc->ovf_done, 0));
        else HIPCHK(hipEventCreateWithFlags(&sc->ovf_done, hipEventDisableTiming));
    }
    HIPCHK(hipMemsetAsync(p.queue, 0, sizeof(unsigned long long), stream));
    HIPCHK(hipEventRecord(ev.a, stream));
    sc->last_engine = traverse ? (ltree ? (ltcull ? 7u : 4u) : qnodes ? (cull_run ? 5u : 3u) : xcull ? 6u : 2u) : (streamed ? 1u : 0u);
    sc->last_form = expanded ? 1u : 0u;
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev.b, stream));
    HIPCHK(hipEventRecord(rg->done, stream));
    if (p.stack_ovf) HIPCHK(hipEventRecord(sc->ovf_done, stream));
    sc->pending.push_back({ev.a, ev.b});
    sc->primary_rays += (uint64_t)p.Hs * p.W * p.spp * n;
    return RT_OK;
}

int collect_locked(rt_scene* sc, rt_tile_stats* st) {
    float ms = 0.f;
    uint32_t n = 0;
    for (auto& pr : sc->pending) {
        HIPCHK(hipEventSynchronize(pr.second));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        n++;
        sc->free_ev.push_back(pr);
    }
    sc->pending.clear();
    unsigned long long c[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(c, sc->d_counters, sizeof c, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(sc->d_counters, 0, sizeof c));
    if (st) {
        st->ray_segments = c[0];
        st->broad_candidates = c[1];
        st->exact_fallbacks = c[2];
        st->primary_rays = sc->primary_rays;
        st->kernel_ms = ms;
        st->n_launches = n;
        st->h2d_ms = sc->h2d_ms;
        st->d2h_ms = 0.f;
        st->engine = sc->last_engine;
        st->broad_form = sc->last_form;
        st->node_steps = c[3];
    }
    sc->primary_rays = 0;
    sc->h2d_ms = 0.f;
    return RT_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

RT_API uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

RT_API const char* rt_strerror(int status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_BAD_ARG: return "bad argument";
        case RT_ERR_NOT_INITIALIZED: return "rt_init() has not succeeded";
        case RT_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case RT_ERR_BAD_DEVICE: return "device ordinal out of range";
        case RT_ERR_BUFFER_TOO_SMALL: return "output buffer smaller than (height/divisions)*width*3";
        case RT_ERR_FRAME_SIZE: return "height is not a multiple of divisions";
        case RT_ERR_HIP: return "HIP runtime error";
        case RT_ERR_LIMIT: return "limit exceeded";
        case RT_ERR_OOM: return "out of memory";
        default: return "unknown status";
    }
}

RT_API const char* rt_last_error(void) { return g_err.c_str(); }

RT_API void rt_tile_request_defaults(rt_tile_request* rq) {
    if (!rq) return;
    std::memset(rq, 0, sizeof *rq);
    rq->width = 1920;                 // controller main.rs:33-39
    rq->height = 1080;
    rq->divisions = 20;
    rq->division_no = 0;
    rq->spp = 100;                    // slave main.rs:51
    rq->max_bounces = 10;             // main.rs:39
    rq->aperture = 0.1f;              // main.rs:45
    rq->focus_distance = 1.0f;        // main.rs:46
    rq->fov = 3.14159265358979323846f / 2.0f;   // PI / 2f32, main.rs:47
    rq->focal_length = 1.0f;          // main.rs:48
    rq->t_min = 0.001f;               // shapes/mod.rs:12
    rq->t_max = 1000.0f;              // shapes/mod.rs:13
    rq->seed = 0;
    rq->flags = RT_FLAG_NONE;
}

RT_API size_t rt_tile_bytes(const rt_tile_request* rq) {
    if (!rq || rq->divisions == 0) return 0;
    return (size_t)(rq->height / rq->divisions) * rq->width * 3;   // main.rs:53-59
}

static int rt_init_impl(int* n_devices) {
    dbg_load_env();
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init) {
        if (n_devices) *n_devices = (int)g_ctx.size();
        return RT_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (n_devices) *n_devices = 0;
        return fail(RT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") +
                                          (e != hipSuccess ? hipGetErrorString(e) : "0 devices"));
    }
    // contexts are created lazily, on the first scene of a device: a rank of a multi-process job touches
    // only its own GPU
    for (int d = 0; d < n; d++) {
        DeviceCtx* c = new DeviceCtx;
        c->dev = d;
        g_ctx.push_back(c);
    }
    g_init = true;
    if (n_devices) *n_devices = n;
    return RT_OK;
}

// Create the device's stream and raise the kernels' dynamic-LDS limit (once per device).
static int ensure_ctx(DeviceCtx* c) {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->stream) return RT_OK;
    HIPCHK(hipSetDevice(c->dev));
    HIPCHK(hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, c->dev));
    for (int streamed = 0; streamed < 2; streamed++)
        for (int expanded = 0; expanded < 2; expanded++)
            HIPCHK(hipFuncSetAttribute((const void*)rtk::kernel_linear(streamed != 0, expanded != 0),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    for (int variant = 0; variant < 8; variant++)
        for (int stats = 0; stats < 2; stats++)
            HIPCHK(hipFuncSetAttribute((const void*)rtk::kernel_traverse(variant, stats != 0),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    hipStream_t st = nullptr, cs = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    c->copy_stream = cs;
    c->stream = st;
    return RT_OK;
}

static int rt_shutdown_impl(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_live_scenes.load() > 0)
        return fail(RT_ERR_BAD_ARG, "rt_shutdown() refused: destroy every rt_scene first (their device contexts stay valid)");
    for (DeviceCtx* c : g_ctx) {
        (void)hipSetDevice(c->dev);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        delete c;
    }
    g_ctx.clear();
    g_init = false;
    return RT_OK;
}

static int rt_scene_destroy_impl(rt_scene* sc);

// Everything rt_scene_create derives on the host from the primitive lists: the device-layout arrays and the reference's
// candidate-filter BVH.  rt_render_frame builds it ONCE per job and uploads it to every device.
struct HostScene {
    uint32_t ns = 0, nt = 0, n_sph_pad = 0;
    std::vector<float4> geom, geom_pk, geom_px, mat, tri_box, geom_r;
    std::vector<float> emis, tri;
    bool expanded = false, quant_ok = false;
    float leaf_density = 0.f, bvh_build_ms = 0.f;
    uint32_t n_internal = 0;         // internal nodes of the tree (bvh.trav may carry one placeholder)
    rtbvh::FlatBVH bvh;
    std::vector<uint32_t> big;       // culled walk: spheres far larger than the rest, and ...
    uint32_t n_big = 0;
    float r_slack = 0.f;             // ... the largest radius among the others
    bool cull_pays = false;          // enough of the rays hit something for nearer-first + culling to beat the plain walk
    bool inverted_boxes = false;     // some sphere has a negative radius
    float tri_k = 0.f, tri_diag = 0.f, tri_es = 0.f, tri_e = 0.f;
    bool xcull_pays = false;         // a scene with triangles that the culled walk over the exact nodes may take, and where it pays
    bool tri_ok = false;             //   ... may take at all (every triangle has a finite bound or a place in the list)
    float cull_density = 0.f;        // sum of the other spheres' box areas / area of the box around them
    std::vector<uint32_t> world_rank;   // the caller's world_index (one dummy entry when none came)
    bool has_order = false;
};

static int check_world(const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt, const uint32_t* world_index) {
    if ((ns && !sp) || (nt && !tr)) return fail(RT_ERR_BAD_ARG, "primitive pointer is NULL");
    // the kernels address nodes (64 B), geometry (16 B), materials (16 B) and triangles (36 B) with 32-bit byte offsets
    if ((uint64_t)ns + nt > RT_MAX_PRIMITIVES) return fail(RT_ERR_LIMIT, "too many primitives (RT_MAX_PRIMITIVES)");
    if (world_index) {                   // positions in RenderInfo.world: every one of 0 .. n - 1 exactly once
        const uint32_t np = ns + nt;
        std::vector<bool> seen(np, false);
        for (uint32_t i = 0; i < np; i++) {
            if (world_index[i] >= np || seen[world_index[i]])
                return fail(RT_ERR_BAD_ARG, "world_index is not a permutation of 0 .. n_spheres + n_triangles - 1");
            seen[world_index[i]] = true;
        }
    }
    return RT_OK;
}

static void build_host_scene(const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt, const uint32_t* world_index,
                             HostScene& hs) {
    hs.ns = ns;
    hs.nt = nt;
    hs.has_order = world_index != nullptr && ns + nt > 0;
    const uint32_t np = ns + nt;
    // the reference's candidate-filter BVH (slave main.rs:60), built once per scene instead of per strip
    std::vector<rtbvh::Box> boxes(np);
    for (uint32_t i = 0; i < ns; i++) {              // Sphere::aabb, sphere.rs:65-72
        const float c[3] = {sp[i].cx, sp[i].cy, sp[i].cz};
        for (int a = 0; a < 3; a++) {
            boxes[i].lo[a] = c[a] - sp[i].radius;
            boxes[i].hi[a] = c[a] + sp[i].radius;
        }
    }
    for (uint32_t i = 0; i < nt; i++) {              // Triangle::aabb, mesh.rs:46-96 (min_by / max_by order a,c,b)
        for (int a = 0; a < 3; a++) {
            const float va = tr[i].a[a], vb = tr[i].b[a], vc = tr[i].c[a];
            const float m1 = va > vc ? vc : va;      // min_by(a, c): a unless a > c
            boxes[ns + i].lo[a] = m1 > vb ? vb : m1;
            const float x1 = va > vc ? va : vc;      // max_by(a, c): c unless a > c
            boxes[ns + i].hi[a] = x1 > vb ? x1 : vb;
        }
    }
    auto tb0 = std::chrono::steady_clock::now();
    {
        // BVH::build(&mut req.world) (slave main.rs:60) numbers the shapes by their position in `world`: start the build
        // from the primitives in that order (rt_bvh.h); ties between equal distances then fall as in the reference
        std::vector<uint32_t> order;
        if (hs.has_order) {
            order.resize(np);
            for (uint32_t i = 0; i < np; i++) order[world_index[i]] = i;
        }
        hs.bvh = rtbvh::build(boxes, hs.has_order ? order.data() : nullptr);
    }
    rtbvh::FlatBVH& bvh = hs.bvh;
    hs.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tb0).count();
    hs.n_internal = (uint32_t)bvh.trav.size();        // before the placeholders below
    if (bvh.nodes.empty()) bvh.nodes.push_back(rtbvh::FlatNode{{0, 0, 0}, 0xffffffffu, {0, 0, 0}, 0});
    if (bvh.leaf_of.empty()) bvh.leaf_of.push_back(0);
    if (bvh.trav.empty()) bvh.trav.push_back(rtbvh::TravNode{});
    if (bvh.travq.empty()) bvh.travq.push_back(rtbvh::QNode{});
    // ---- Storage order (round 3).  The records the kernels fetch per primitive — sphere (centre, radius), material, emission, triangle
    // vertices — are laid out in the order in which the tree's depth-first walk meets the leaves, not in the caller's order: the
    // primitives a ray (and the rays of a wave) touch are then neighbours in memory, four sphere records to a 64-byte line, instead of
    // scattered over megabytes.  Only the library's INTERNAL primitive numbers change (spheres stay below n_sph, triangles above): leaf
    // references, leaf ranks, the `big` list and the world positions are renumbered with them, and every rule that looks at a
    // primitive's place in `world` (distance ties) goes through world_rank, which from here on always exists.
    std::vector<rt_sphere> sp_store;
    std::vector<rt_triangle> tr_store;
    std::vector<uint32_t> wi_store(np ? np : 1, 0u);
    for (uint32_t i = 0; i < np; i++) wi_store[i] = world_index ? world_index[i] : i;
    if (np >= REORDER_MIN_PRIMS && dbg(DBG_REORDER) != 0) {
        std::vector<uint32_t> old_of(np), new_of(np);
        for (uint32_t i = 0; i < np; i++) old_of[i] = i;
        std::stable_sort(old_of.begin(), old_of.begin() + ns, [&](uint32_t x, uint32_t y) { return bvh.leaf_of[x] < bvh.leaf_of[y]; });
        std::stable_sort(old_of.begin() + ns, old_of.end(), [&](uint32_t x, uint32_t y) { return bvh.leaf_of[x] < bvh.leaf_of[y]; });
        for (uint32_t i = 0; i < np; i++) new_of[old_of[i]] = i;
        sp_store.resize(ns);
        tr_store.resize(nt);
        std::vector<rtbvh::Box> boxes2(np);
        std::vector<uint32_t> leaf2(np), wi2(np);
        for (uint32_t i = 0; i < np; i++) {
            const uint32_t o = old_of[i];
            if (i < ns) sp_store[i] = sp[o];
            else tr_store[i - ns] = tr[o - ns];
            boxes2[i] = boxes[o];
            leaf2[i] = bvh.leaf_of[o];
            wi2[i] = wi_store[o];
        }
        boxes.swap(boxes2);
        bvh.leaf_of.swap(leaf2);
        wi_store.swap(wi2);
        auto remap = [&](uint32_t& ref) {
            if (ref & rtbvh::LEAF_BIT) ref = rtbvh::LEAF_BIT | new_of[ref & ~rtbvh::LEAF_BIT];
        };
        for (rtbvh::TravNode& t : bvh.trav) { remap(t.left); remap(t.right); }
        for (rtbvh::QNode& q : bvh.travq) { remap(q.left); remap(q.right); }
        remap(bvh.root_ref);
        sp = sp_store.data();
        tr = tr_store.data();
        hs.has_order = true;                 // (ties of the plain linear-scan semantics: by place in `world`, no longer by number)
    }
    if (hs.has_order) hs.world_rank.assign(wi_store.begin(), wi_store.begin() + np);
    else hs.world_rank.assign(1, 0u);
    hs.n_sph_pad = (ns + rtk::UNROLL - 1) / rtk::UNROLL * rtk::UNROLL;
    std::vector<float4>& geom = hs.geom;
    std::vector<float4>& mat = hs.mat;
    std::vector<float>& emis = hs.emis;
    std::vector<float>& tri = hs.tri;
    geom.assign(hs.n_sph_pad ? hs.n_sph_pad : 1, make_float4(0.f, 0.f, 0.f, 0.f));
    mat.assign(np ? np : 1, make_float4(0.f, 0.f, 0.f, 0.f));
    emis.assign(np ? np : 1, 0.f);
    tri.assign((size_t)nt * 9 + 1, 0.f);
    for (uint32_t i = 0; i < ns; i++) {
        // rr = radius.powi(2) (sphere.rs:45): one rounded multiply
        volatile float rr = sp[i].radius * sp[i].radius;
        geom[i] = make_float4(sp[i].cx, sp[i].cy, sp[i].cz, rr);
        mat[i] = make_float4(sp[i].albedo_r, sp[i].albedo_g, sp[i].albedo_b, sp[i].roughness);
        emis[i] = sp[i].emission;
    }
    // padding spheres can never pass either phase: rr = -inf makes every discriminant -inf
    for (uint32_t i = ns; i < hs.n_sph_pad; i++) geom[i] = make_float4(0.f, 0.f, 0.f, -INFINITY);
    // pair layout for the packed-FP32 broad phase: (c0x,c1x,c0y,c1y) (c0z,c1z,rr0,rr1)
    std::vector<float4>& geom_pk = hs.geom_pk;
    geom_pk.assign(geom.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t i = 0; i + 1 < hs.n_sph_pad; i += 2) {
        geom_pk[i] = make_float4(geom[i].x, geom[i + 1].x, geom[i].y, geom[i + 1].y);
        geom_pk[i + 1] = make_float4(geom[i].z, geom[i + 1].z, geom[i].w, geom[i + 1].w);
    }
    for (uint32_t i = 0; i < nt; i++) {
        std::memcpy(&tri[(size_t)i * 9], tr[i].a, 9 * sizeof(float));
        mat[ns + i] = make_float4(tr[i].albedo_r, tr[i].albedo_g, tr[i].albedo_b, tr[i].roughness);
        emis[ns + i] = tr[i].emission;
    }
    // expanded-form broad phase records: w = |c|^2 - rr - 2^-16 (|c|^2 + rr), evaluated in
    // double and rounded DOWN to f32 (conservative).
    std::vector<float4>& geom_px = hs.geom_px;
    geom_px.assign(geom.size(), make_float4(0.f, 0.f, 0.f, 0.f));
    {
        std::vector<float4> px(geom.size());
        std::vector<double> ratio;
        const double K = std::ldexp(1.0, -16);
        for (uint32_t i = 0; i < hs.n_sph_pad; i++) {
            if (i >= ns) {
                px[i] = make_float4(0.f, 0.f, 0.f, INFINITY);      // w = +inf: t = -inf, never a candidate
                continue;
            }
            const double cc = (double)sp[i].cx * sp[i].cx + (double)sp[i].cy * sp[i].cy + (double)sp[i].cz * sp[i].cz;
            const double rr = (double)geom[i].w;
            const double w = cc - rr - K * (cc + rr);
            float wf = (float)w;
            if ((double)wf > w) wf = std::nextafterf(wf, -INFINITY);
            px[i] = make_float4(sp[i].cx, sp[i].cy, sp[i].cz, wf);
            if (rr > 0) ratio.push_back(K * 2.0 * cc / rr);
        }
        for (uint32_t i = 0; i + 1 < hs.n_sph_pad; i += 2) {
            geom_px[i] = make_float4(px[i].x, px[i + 1].x, px[i].y, px[i + 1].y);
            geom_px[i + 1] = make_float4(px[i].z, px[i + 1].z, px[i].w, px[i + 1].w);
        }
        // heuristic: the expanded form's additive margin 2^-16 (|o|^2 + |c|^2 + rr) must stay small against rr
        // for the typical sphere, otherwise candidate lists blow up (c5-class scenes): then use the oc form.
        bool ok = !ratio.empty();
        if (ok) {
            std::nth_element(ratio.begin(), ratio.begin() + ratio.size() / 2, ratio.end());
            ok = ratio[ratio.size() / 2] < 0.5;
        }
        for (uint32_t i = 0; i < ns && ok; i++)
            ok = std::isfinite(px[i].x) && std::isfinite(px[i].y) && std::isfinite(px[i].z) && std::isfinite(px[i].w);
        hs.expanded = ok;
    }
    hs.tri_box.assign((size_t)nt * 2 + 1, make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t i = 0; i < nt; i++) {
        const rtbvh::Box& b = boxes[ns + i];
        hs.tri_box[2 * (size_t)i] = make_float4(b.lo[0], b.lo[1], b.lo[2], 0.f);
        hs.tri_box[2 * (size_t)i + 1] = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.f);
    }
    {
        // worthwhile only if the grid step is small against the primitives (else the rounded boxes admit crowds of
        // false leaves): median primitive box edge >= 8 steps on every axis
        bool ok = bvh.grid.ok && np > 1;
        if (ok) {
            std::vector<float> edge(np);
            for (uint32_t i = 0; i < np; i++) {
                float e = INFINITY;
                for (int a3 = 0; a3 < 3; a3++)
                    e = fminf(e, (boxes[i].hi[a3] - boxes[i].lo[a3]) / bvh.grid.step[a3]);
                edge[i] = e;
            }
            std::nth_element(edge.begin(), edge.begin() + np / 2, edge.end());
            ok = edge[np / 2] >= 8.0f;
        }
        hs.quant_ok = ok;
        // leaf density = sum of primitive box areas / area of the scene box ~ leaves a random ray reaches; above ~2
        // the walk is bound by the exact leaf tests, where the lighter exact-node kernel (5 waves/SIMD) wins
        double area = 0.0, root = 0.0;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < np; i++) {
            const double ex = (double)boxes[i].hi[0] - boxes[i].lo[0], ey = (double)boxes[i].hi[1] - boxes[i].lo[1],
                         ez = (double)boxes[i].hi[2] - boxes[i].lo[2];
            area += ex * ey + ey * ez + ez * ex;
            for (int a3 = 0; a3 < 3; a3++) {
                lo[a3] = fminf(lo[a3], boxes[i].lo[a3]);
                hi[a3] = fmaxf(hi[a3], boxes[i].hi[a3]);
            }
        }
        if (np) {
            const double ex = (double)hi[0] - lo[0], ey = (double)hi[1] - lo[1], ez = (double)hi[2] - lo[2];
            root = ex * ey + ey * ez + ez * ex;
        }
        hs.leaf_density = root > 0.0 ? (float)(area / root) : INFINITY;
    }
    hs.geom_r.assign(ns ? ns : 1, make_float4(0.f, 0.f, 0.f, 0.f));
    for (uint32_t i = 0; i < ns; i++) {
        hs.geom_r[i] = make_float4(sp[i].cx, sp[i].cy, sp[i].cz, sp[i].radius);
        if (sp[i].radius < 0.0f) hs.inverted_boxes = true;
    }
    // culled walk: its distance bound carries sqrt(2) * (largest radius) of slack, so the few spheres far larger than the
    // rest (a ground sphere) are listed apart and root-tested at every query start instead
    {
        std::vector<float> rad(ns);
        for (uint32_t i = 0; i < ns; i++) rad[i] = fabsf(sp[i].radius);
        float med = 0.f;
        if (ns) {
            std::vector<float> tmp(rad);
            std::nth_element(tmp.begin(), tmp.begin() + ns / 2, tmp.end());
            med = tmp[ns / 2];
        }
        std::vector<uint32_t> cand;
        for (uint32_t i = 0; i < ns; i++)
            if (rad[i] > 8.0f * med) cand.push_back(i);
        std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return rad[a] > rad[b] || (rad[a] == rad[b] && a < b); });
        if (cand.size() > 16) cand.resize(16);
        std::vector<char> is_big(ns ? ns : 1, 0);
        for (uint32_t i : cand) is_big[i] = 1;
        float rs = 0.f;
        for (uint32_t i = 0; i < ns; i++)
            if (!is_big[i] && rad[i] > rs) rs = rad[i];
        hs.big = cand;
        hs.n_big = (uint32_t)cand.size();
        hs.r_slack = rs;
        if (hs.big.empty()) hs.big.push_back(0);
        // Does it pay?  The ratio below is the expected number of (non-big) primitive boxes a random line through their
        // common box meets (Cauchy: box areas add up).  tools/cull_matrix.py, 2560x1440: sparse fields at 0.06...0.35 lose
        // 5...7 % to the ordering and the early root tests, c5 at 0.95 gains 12 %, fields / mixed radii / dense overlap at
        // 2...30 gain 1.35...3.5 x.  And the bound's slack (1.5 r_slack) must be small against the scene.
        double area = 0.0;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < ns; i++) {
            if (is_big[i]) continue;
            const double e = 2.0 * rad[i];
            area += 3.0 * e * e;
            const float c[3] = {sp[i].cx, sp[i].cy, sp[i].cz};
            for (int a3 = 0; a3 < 3; a3++) {
                lo[a3] = fminf(lo[a3], c[a3] - rad[i]);
                hi[a3] = fmaxf(hi[a3], c[a3] + rad[i]);
            }
        }
        const double ex = (double)hi[0] - lo[0], ey = (double)hi[1] - lo[1], ez = (double)hi[2] - lo[2];
        const double root = ex * ey + ey * ez + ez * ex, diag = std::sqrt(ex * ex + ey * ey + ez * ez);
        hs.cull_density = root > 0.0 ? (float)(area / root) : 0.f;
        hs.cull_pays = nt == 0 && root > 0.0 && std::isfinite(area / root) && area / root >= 0.7 && (double)rs <= 0.05 * diag;
        // (What the bounds claim — no accepted root of a primitive outside the `big` list enters its box beyond cull_bound /
        // cull_bound_tri of its compared distance — is tested by itself, on 1.8e7 seeded and adversarial (ray, primitive, box)
        // triples incl. K -> 0.25, |det| -> 1e-5, origins at 1e3 and tangent rays: tests/test_cull_lemma.py with the bounds
        // of csrc/rt_cull.h; reduced soaks against the oracle: tests/test_gpu_cull_soaks.py.)
        // Scenes with triangles (culled walk over the EXACT nodes, cull_bound_tri in rt_cull.h): its bound needs every
        // triangle outside the `big` list to have K = |e1||e2| <= 0.25 (with the reference's |det| >= 1e-5 that keeps the
        // computed determinant within 1.5 % of the true one) and carries the largest box diagonal as slack, so triangles with
        // a larger K, or a box diagonal of more than 8 x the median, join the list (16 entries with the spheres; more: no culling).
        if (nt > 0) {
            std::vector<float> dg(nt), kk(nt), es(nt), em(nt);
            for (uint32_t i = 0; i < nt; i++) {
                double e1 = 0, e2 = 0, e3 = 0, d2 = 0;
                for (int a3 = 0; a3 < 3; a3++) {
                    const double ab = (double)tr[i].b[a3] - tr[i].a[a3], ac = (double)tr[i].c[a3] - tr[i].a[a3], bc = (double)tr[i].c[a3] - tr[i].b[a3];
                    e1 += ab * ab; e2 += ac * ac; e3 += bc * bc;
                    const double ext = (double)boxes[ns + i].hi[a3] - boxes[ns + i].lo[a3];
                    d2 += ext * ext;
                }
                e1 = std::sqrt(e1); e2 = std::sqrt(e2); e3 = std::sqrt(e3);
                dg[i] = (float)(std::sqrt(d2) * 1.0001);
                kk[i] = (float)(e1 * e2 * 1.0001);
                es[i] = (float)((e1 + e2) * 1.0001);
                em[i] = (float)(std::max(e1, std::max(e2, e3)) * 1.0001);
            }
            std::vector<float> tmp(dg);
            std::nth_element(tmp.begin(), tmp.begin() + nt / 2, tmp.end());
            const float med_d = tmp[nt / 2];
            std::vector<uint32_t> bigt;
            bool ok = true;
            for (uint32_t i = 0; i < nt && ok; i++) {
                const bool fin = std::isfinite(dg[i]) && std::isfinite(kk[i]);
                if (!fin) ok = false;
                else if (kk[i] > 0.25f || dg[i] > 8.0f * med_d) bigt.push_back(ns + i);
                if (bigt.size() + hs.n_big > 16) ok = false;
            }
            if (ok) {
                std::vector<char> isb(nt, 0);
                for (uint32_t q : bigt) isb[q - ns] = 1;
                double tarea = 0.0;
                for (uint32_t i = 0; i < nt; i++) {
                    if (isb[i]) continue;
                    hs.tri_k = fmaxf(hs.tri_k, kk[i]);
                    hs.tri_diag = fmaxf(hs.tri_diag, dg[i]);
                    hs.tri_es = fmaxf(hs.tri_es, es[i]);
                    hs.tri_e = fmaxf(hs.tri_e, em[i]);
                    const rtbvh::Box& b = boxes[ns + i];
                    const double ex2 = (double)b.hi[0] - b.lo[0], ey2 = (double)b.hi[1] - b.lo[1], ez2 = (double)b.hi[2] - b.lo[2];
                    tarea += ex2 * ey2 + ey2 * ez2 + ez2 * ex2;
                    for (int a3 = 0; a3 < 3; a3++) {
                        lo[a3] = fminf(lo[a3], b.lo[a3]);
                        hi[a3] = fmaxf(hi[a3], b.hi[a3]);
                    }
                }
                const double fx = (double)hi[0] - lo[0], fy = (double)hi[1] - lo[1], fz = (double)hi[2] - lo[2];
                const double root2 = fx * fy + fy * fz + fz * fx, diag2 = std::sqrt(fx * fx + fy * fy + fz * fz);
                const double dens = root2 > 0.0 ? (area + tarea) / root2 : 0.0;
                if (hs.n_big) hs.big.resize(hs.n_big); else hs.big.clear();
                for (uint32_t q : bigt) hs.big.push_back(q);
                hs.n_big = (uint32_t)hs.big.size();
                if (hs.big.empty()) hs.big.push_back(0);
                hs.cull_density = (float)dens;
                hs.xcull_pays = std::isfinite(dens) && dens >= 0.7 && (double)rs <= 0.05 * diag2 && (double)hs.tri_diag <= 0.05 * diag2;
            }
            hs.tri_ok = ok;
        }
    }
}

// HostScene -> device: allocate, upload on the device's stream, hand back the handle
static int upload_scene(int device, const HostScene& hs, rt_scene** out) {
    const auto t_create0 = std::chrono::steady_clock::now();
    if (!g_init) return fail(RT_ERR_NOT_INITIALIZED, "call rt_init() first");
    if (device < 0 || device >= (int)g_ctx.size()) return fail(RT_ERR_BAD_DEVICE, "bad device ordinal");
    DeviceCtx* ctx = g_ctx[device];
    {
        int rc0 = ensure_ctx(ctx);
        if (rc0) return rc0;
    }
    HIPCHK(hipSetDevice(ctx->dev));
    rt_scene* sc = new (std::nothrow) rt_scene;
    if (!sc) return fail(RT_ERR_OOM, "host allocation failed");
    g_live_scenes.fetch_add(1);            // (rt_scene_destroy_impl, also on the error paths below, takes it back)
    struct SceneGuard {                    // an error return or an exception below releases everything made so far
        rt_scene* sc;
        ~SceneGuard() {
            if (sc) rt_scene_destroy_impl(sc);
        }
    } guard{sc};
    const rtbvh::FlatBVH& bvh = hs.bvh;
    sc->ctx = ctx;
    sc->n_sph = hs.ns;
    sc->n_tri = hs.nt;
    sc->n_sph_pad = hs.n_sph_pad;
    sc->expanded = hs.expanded;
    sc->bvh_build_ms = hs.bvh_build_ms;
    sc->root_ref = bvh.root_ref;
    sc->bvh_depth = bvh.depth;
    sc->n_internal = hs.n_internal;
    sc->grid = bvh.grid;
    sc->quant_ok = hs.quant_ok;
    sc->leaf_density = hs.leaf_density;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return fail(RT_ERR_HIP, "hipEventCreate failed");
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(RT_ERR_HIP, "hipEventCreate failed");
    }
#define SC_CHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            (void)hipEventDestroy(e0);                                                    \
            (void)hipEventDestroy(e1);                                                    \
            return fail(_e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(_e));               \
        }                                                                                 \
    } while (0)
#define SC_UP(dst, vec)                                                                                       \
    do {                                                                                                      \
        SC_CHK(hipMalloc(&sc->dst, (vec).size() * sizeof((vec)[0])));                                         \
        SC_CHK(hipMemcpyAsync(sc->dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice,  \
                              ctx->stream));                                                                  \
    } while (0)
    SC_CHK(hipEventRecord(e0, ctx->stream));
    SC_UP(d_geom, hs.geom);
    SC_UP(d_geom_pk, hs.geom_pk);
    SC_UP(d_geom_px, hs.geom_px);
    SC_UP(d_mat, hs.mat);
    SC_UP(d_emis, hs.emis);
    SC_UP(d_tri, hs.tri);
    SC_UP(d_tri_box, hs.tri_box);
    SC_UP(d_bvh, bvh.nodes);
    SC_UP(d_leaf_of, bvh.leaf_of);
    SC_UP(d_world_rank, hs.world_rank);
    sc->has_order = hs.has_order;
    SC_UP(d_trav, bvh.trav);
    SC_UP(d_travq, bvh.travq);
    SC_UP(d_geom_r, hs.geom_r);
    SC_UP(d_big, hs.big);
    sc->n_big = hs.n_big;
    sc->r_slack = hs.r_slack;
    sc->cull_pays = hs.cull_pays;
    sc->inverted_boxes = hs.inverted_boxes;
    sc->tri_k = hs.tri_k;
    sc->tri_diag = hs.tri_diag;
    sc->tri_es = hs.tri_es;
    sc->tri_e = hs.tri_e;
    sc->xcull_pays = hs.xcull_pays;
    sc->tri_ok = hs.tri_ok;
    sc->cull_density = hs.cull_density;
    SC_CHK(hipMalloc(&sc->d_counters, COUNTER_WORDS * sizeof(unsigned long long)));
    SC_CHK(hipMemsetAsync(sc->d_counters, 0, COUNTER_WORDS * sizeof(unsigned long long), ctx->stream));
    SC_CHK(hipEventRecord(e1, ctx->stream));
    SC_CHK(hipEventSynchronize(e1));
    SC_CHK(hipEventElapsedTime(&sc->h2d_ms, e0, e1));
#undef SC_UP
#undef SC_CHK
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (dbg(DBG_VERBOSE))
        fprintf(stderr, "[rt] scene: %u prims  culled walk: %u big spheres, slack radius %g, box density %.3f -> %s  bvh build %.2f ms (host)  uploads %.2f ms  upload total %.2f ms\n",
                hs.ns + hs.nt, hs.n_big, hs.r_slack, hs.cull_density, hs.cull_pays ? "default" : "off", sc->bvh_build_ms, sc->h2d_ms,
                std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_create0).count());
    guard.sc = nullptr;
    *out = sc;
    return RT_OK;
}

static int rt_scene_create_impl(int device, const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt,
                                const uint32_t* world_index, rt_scene** out) {
    if (!out) return fail(RT_ERR_BAD_ARG, "out_scene is NULL");
    *out = nullptr;
    int rc = check_world(sp, ns, tr, nt, world_index);
    if (rc) return rc;
    if (!g_init) return fail(RT_ERR_NOT_INITIALIZED, "call rt_init() first");
    if (device < 0 || device >= (int)g_ctx.size()) return fail(RT_ERR_BAD_DEVICE, "bad device ordinal");
    HostScene hs;
    build_host_scene(sp, ns, tr, nt, world_index, hs);
    return upload_scene(device, hs, out);
}

static int rt_scene_destroy_impl(rt_scene* sc) {
    if (!sc) return RT_OK;
    if (sc->ctx) (void)hipSetDevice(sc->ctx->dev);
    for (auto& pr : sc->pending) {
        (void)hipEventSynchronize(pr.second);
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto& pr : sc->free_ev) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    (void)hipFree(sc->d_geom);
    (void)hipFree(sc->d_geom_pk);
    (void)hipFree(sc->d_geom_px);
    (void)hipFree(sc->d_mat);
    (void)hipFree(sc->d_emis);
    (void)hipFree(sc->d_tri);
    (void)hipFree(sc->d_tri_box);
    (void)hipFree(sc->d_bvh);
    (void)hipFree(sc->d_trav);
    (void)hipFree(sc->d_travq);
    (void)hipFree(sc->d_stack_ovf);
    if (sc->ovf_done) (void)hipEventDestroy(sc->ovf_done);
    for (auto& r : sc->rings) {
        (void)hipFree(r.d);
        if (r.done) (void)hipEventDestroy(r.done);
    }
    (void)hipFree(sc->d_geom_r);
    (void)hipFree(sc->d_big);
    (void)hipFree(sc->d_leaf_of);
    (void)hipFree(sc->d_world_rank);
    (void)hipFree(sc->d_counters);
    (void)hipFree(sc->d_out);
    (void)hipFree(sc->d_outf);
    (void)hipFree(sc->d_cost);
    delete sc;
    g_live_scenes.fetch_sub(1);
    return RT_OK;
}

static int check_batch(const rt_tile_request* rqs, uint32_t n) {
    if (!rqs || n == 0) return fail(RT_ERR_BAD_ARG, "empty request batch");
    for (uint32_t i = 0; i < n; i++) {
        int rc = check_request(&rqs[i]);
        if (rc) return rc;
        if (!same_frame(rqs[0], rqs[i]))
            return fail(RT_ERR_BAD_ARG, "batched requests must agree on every field but division_no and seed");
    }
    return RT_OK;
}

static int rt_scene_render_tiles_device_impl(rt_scene* sc, const rt_tile_request* rqs, uint32_t n,
                                        void* const* d_out_rgb, size_t out_len_each, void* const* d_out_f32,
                                        void* hip_stream) {
    if (!sc) return fail(RT_ERR_BAD_ARG, "scene is NULL");
    int rc = check_batch(rqs, n);
    if (rc) return rc;
    if (!d_out_rgb) return fail(RT_ERR_BAD_ARG, "d_out_rgb is NULL");
    for (uint32_t i = 0; i < n; i++)
        if (!d_out_rgb[i]) return fail(RT_ERR_BAD_ARG, "d_out_rgb[i] is NULL");
    if (out_len_each < rt_tile_bytes(&rqs[0])) return fail(RT_ERR_BUFFER_TOO_SMALL, "out_len < (H/div)*W*3");
    std::lock_guard<std::mutex> lk(sc->mu);
    HIPCHK(hipSetDevice(sc->ctx->dev));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : sc->ctx->stream;
    for (uint32_t i0 = 0; i0 < n; i0 += rtk::MAX_BATCH) {
        uint32_t m = std::min<uint32_t>(rtk::MAX_BATCH, n - i0);
        rc = launch_batch(sc, rqs + i0, m, d_out_rgb + i0, d_out_f32 ? d_out_f32 + i0 : nullptr, st);
        if (rc) return rc;
    }
    return RT_OK;
}

static int rt_scene_render_tile_device_impl(rt_scene* sc, const rt_tile_request* rq, void* d_out_rgb, size_t out_len,
                                       void* d_out_f32, void* hip_stream) {
    void* rgb[1] = {d_out_rgb};
    void* f32[1] = {d_out_f32};
    return rt_scene_render_tiles_device_impl(sc, rq, 1, rgb, out_len, d_out_f32 ? f32 : nullptr, hip_stream);
}

static int rt_scene_collect_impl(rt_scene* sc, rt_tile_stats* st) {
    if (!sc) return fail(RT_ERR_BAD_ARG, "scene is NULL");
    std::lock_guard<std::mutex> lk(sc->mu);
    HIPCHK(hipSetDevice(sc->ctx->dev));
    return collect_locked(sc, st);
}

// strip_cost_out: optional host array of n counters: the ray segments of every strip (frame context)
static int rt_scene_render_tiles_impl(rt_scene* sc, const rt_tile_request* rqs, uint32_t n, uint8_t* const* out_rgb,
                                 size_t out_len_each, float* const* out_f32, rt_tile_stats* stats,
                                 unsigned long long* strip_cost_out = nullptr) {
    if (!sc) return fail(RT_ERR_BAD_ARG, "scene is NULL");
    int rc = check_batch(rqs, n);
    if (rc) return rc;
    if (!out_rgb) return fail(RT_ERR_BAD_ARG, "out_rgb is NULL");
    for (uint32_t i = 0; i < n; i++)
        if (!out_rgb[i]) return fail(RT_ERR_BAD_ARG, "out_rgb[i] is NULL");
    const size_t need = rt_tile_bytes(&rqs[0]);
    if (out_len_each < need) return fail(RT_ERR_BUFFER_TOO_SMALL, "out_len < (H/div)*W*3");
    bool want_f32 = false;
    if (out_f32)
        for (uint32_t i = 0; i < n; i++) want_f32 |= out_f32[i] != nullptr;
    std::lock_guard<std::mutex> dl(sc->ctx->mu);
    std::lock_guard<std::mutex> lk(sc->mu);
    HIPCHK(hipSetDevice(sc->ctx->dev));
    hipStream_t st = sc->ctx->stream;
    if (sc->d_out_cap < need * n) {
        (void)hipFree(sc->d_out);
        sc->d_out = nullptr;
        sc->d_out_cap = 0;
        hipError_t e = hipMalloc(&sc->d_out, need * n);
        if (e != hipSuccess) return fail(RT_ERR_OOM, "hipMalloc(strips) failed");
        sc->d_out_cap = need * n;
    }
    if (want_f32 && sc->d_outf_cap < need * n * sizeof(float)) {
        (void)hipFree(sc->d_outf);
        sc->d_outf = nullptr;
        sc->d_outf_cap = 0;
        hipError_t e = hipMalloc(&sc->d_outf, need * n * sizeof(float));
        if (e != hipSuccess) return fail(RT_ERR_OOM, "hipMalloc(strips f32) failed");
        sc->d_outf_cap = need * n * sizeof(float);
    }
    // per-strip costs: one block of COST_COPIES x MAX_BATCH counters per launch group of the call
    constexpr size_t COST_BLOCK = (size_t)rtk::COST_COPIES * rtk::MAX_BATCH;
    const size_t cost_blocks = strip_cost_out ? (n + rtk::MAX_BATCH - 1) / rtk::MAX_BATCH + 1 : 0;
    if (strip_cost_out) {
        if (sc->d_cost_cap < cost_blocks * COST_BLOCK) {
            (void)hipFree(sc->d_cost);
            sc->d_cost = nullptr;
            sc->d_cost_cap = 0;
            if (hipMalloc(&sc->d_cost, cost_blocks * COST_BLOCK * sizeof(unsigned long long)) != hipSuccess) return fail(RT_ERR_OOM, "hipMalloc(strip costs) failed");
            sc->d_cost_cap = cost_blocks * COST_BLOCK;
        }
        HIPCHK(hipMemsetAsync(sc->d_cost, 0, cost_blocks * COST_BLOCK * sizeof(unsigned long long), st));
    }
    // settle anything enqueued earlier so the stats of this call are its own
    rt_tile_stats prev;
    rc = collect_locked(sc, &prev);
    if (rc) return rc;
    sc->h2d_ms = prev.h2d_ms;
    std::vector<void*> drgb(n), df32(n);
    for (uint32_t i = 0; i < n; i++) {
        drgb[i] = sc->d_out + need * i;
        df32[i] = (want_f32 && out_f32[i]) ? (void*)(sc->d_outf + need * i) : nullptr;
    }
    // Launch groups: the last quarter of the strips goes out as its own launch, so that the D2H copies of the strips
    // before it (copy stream) run under it and only the last group's copies are exposed (d2h_ms = that exposed part).
    std::vector<std::pair<uint32_t, uint32_t>> groups;    // [first, count)
    {
        // ... when the downloads are worth hiding: a second launch costs its own tail (half a millisecond at c3, one at c4), a
        // frame's strips come down at about 50 GB/s, so below 64 MiB per call one launch and an exposed download are faster
        // (c3, 24.9 MB: 9.3 instead of 9.6 ms per frame; c4's 99.5 MB keeps the split)
        const uint32_t tail = (n >= 4 && (uint64_t)need * n > (64ull << 20)) ? std::max<uint32_t>(1, n / 4) : 0;
        for (uint32_t i0 = 0; i0 < n - tail; i0 += rtk::MAX_BATCH)
            groups.emplace_back(i0, std::min<uint32_t>(rtk::MAX_BATCH, n - tail - i0));
        for (uint32_t i0 = n - tail; i0 < n; i0 += rtk::MAX_BATCH)
            groups.emplace_back(i0, std::min<uint32_t>(rtk::MAX_BATCH, n - i0));
    }
    hipStream_t cs = sc->ctx->copy_stream;
    std::vector<EvPair> gev;
    gev.reserve(groups.size());
    // whatever happens below, the events go back to the scene's free list, and an error return waits for the work
    // already enqueued (it writes into caller memory)
    struct EvReturn {
        rt_scene* sc;
        std::vector<EvPair>& v;
        hipStream_t a, b;
        bool ok = false;
        ~EvReturn() {
            if (!ok) {
                (void)hipStreamSynchronize(a);
                (void)hipStreamSynchronize(b);
            }
            for (auto& e : v) sc->free_ev.push_back({e.a, e.b});
        }
    } ev_return{sc, gev, st, cs};
    for (size_t g = 0; g < groups.size(); g++) {
        EvPair e;
        rc = get_events(sc, e);
        if (rc) return rc;
        gev.push_back(e);
    }
    for (size_t g = 0; g < groups.size(); g++) {
        const uint32_t i0 = groups[g].first, m = groups[g].second;
        rc = launch_batch(sc, rqs + i0, m, drgb.data() + i0, want_f32 ? df32.data() + i0 : nullptr, st,
                          strip_cost_out ? sc->d_cost + g * COST_BLOCK : nullptr);
        if (rc) return rc;
        HIPCHK(hipEventRecord(gev[g].a, st));
    }
    for (size_t g = 0; g < groups.size(); g++) {
        HIPCHK(hipStreamWaitEvent(cs, gev[g].a, 0));
        for (uint32_t i = groups[g].first; i < groups[g].first + groups[g].second; i++) {
            HIPCHK(hipMemcpyAsync(out_rgb[i], drgb[i], need, hipMemcpyDeviceToHost, cs));
            if (df32[i]) HIPCHK(hipMemcpyAsync(out_f32[i], df32[i], need * sizeof(float), hipMemcpyDeviceToHost, cs));
        }
    }
    std::vector<unsigned long long> cost_raw;
    if (strip_cost_out) {
        cost_raw.resize(groups.size() * COST_BLOCK);
        HIPCHK(hipMemcpyAsync(cost_raw.data(), sc->d_cost, cost_raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
    }
    HIPCHK(hipEventRecord(gev.back().b, cs));
    HIPCHK(hipEventSynchronize(gev.back().b));
    if (strip_cost_out)
        for (size_t g = 0; g < groups.size(); g++)
            for (uint32_t i = 0; i < groups[g].second; i++) {
                unsigned long long sum = 0;
                for (uint32_t c = 0; c < rtk::COST_COPIES; c++) sum += cost_raw[g * COST_BLOCK + (size_t)c * rtk::MAX_BATCH + i];
                strip_cost_out[groups[g].first + i] = sum;
            }
    float d2h = 0.f;
    HIPCHK(hipEventElapsedTime(&d2h, gev.back().a, gev.back().b));   // last launch done -> last byte on the host
    ev_return.ok = true;
    rt_tile_stats s;
    rc = collect_locked(sc, &s);
    if (rc) return rc;
    s.d2h_ms = d2h;
    if (stats) *stats = s;
    return RT_OK;
}

static int rt_scene_render_tile_impl(rt_scene* sc, const rt_tile_request* rq, uint8_t* out_rgb, size_t out_len,
                                float* out_f32, rt_tile_stats* stats) {
    uint8_t* rgb[1] = {out_rgb};
    float* f32[1] = {out_f32};
    return rt_scene_render_tiles_impl(sc, rq, 1, rgb, out_len, out_f32 ? f32 : nullptr, stats);
}

// ---- Test / tool hooks.  NOT part of rt_tile.h and NOT in the product library: compiled only with -DRT_DEBUG_HOOKS, which
// build.py adds for lib/librt_s8_dbg.so (tests) and tools add to their instrumented variants.  tests/test_abi.py checks that
// librt_s8.so exports exactly the functions rt_tile.h declares.
#ifdef RT_DEBUG_HOOKS
// debug: raw read of the scene's device counter words (tools/phase_census.py)
extern "C" __attribute__((visibility("default"))) int rt_debug_read_counters(rt_scene* sc, uint32_t first, uint32_t n,
                                                                             unsigned long long* out) {
    return guarded([&]() -> int {
        if (!sc || !out || first > COUNTER_WORDS || n > COUNTER_WORDS - first) return fail(RT_ERR_BAD_ARG, "counter range");
        std::lock_guard<std::mutex> lk(sc->mu);
        HIPCHK(hipSetDevice(sc->ctx->dev));
        HIPCHK(hipMemcpy(out, sc->d_counters + first, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return RT_OK;
    });
}
// debug: sqrt_rn of the traversal kernels against the compiler's IEEE sequence on every f32 bit pattern in [from, from + n);
// *mismatches = patterns whose results differ.  Tests only; not part of rt_tile.h.
extern "C" __attribute__((visibility("default"))) int rt_debug_sqrt_selftest(int device, uint32_t from, unsigned long long n,
                                                                             unsigned long long* mismatches) {
    return guarded([&]() -> int {
        if (!mismatches || n > (1ull << 32) - from) return fail(RT_ERR_BAD_ARG, "sqrt self-test range");
        HIPCHK(hipSetDevice(device));
        unsigned long long* d_bad = nullptr;
        HIPCHK(hipMalloc(&d_bad, sizeof(unsigned long long)));
        hipError_t e = hipMemset(d_bad, 0, sizeof(unsigned long long));
        if (e == hipSuccess) {
            rtk::sqrt_selftest_launch(from, n, d_bad, 0);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpy(mismatches, d_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipFree(d_bad);
        if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
        return RT_OK;
    });
}
// debug: set one knob of the launch path (DebugKnob above) by its environment-variable name; returns the previous value,
// INT_MIN for an unknown name.  Tests and tools only; not part of rt_tile.h.
extern "C" __attribute__((visibility("default"))) int rt_debug_set(const char* name, int value) {
    dbg_load_env();
    if (!name) return INT32_MIN;
    for (int k = 0; k < DBG_N; k++)
        if (!strcmp(name, g_dbg_spec[k].env)) return g_dbg[k].exchange(value);
    return INT32_MIN;
}
// test hook (tests/test_abi.py): throw inside a guarded body; the status comes back, nothing unwinds
extern "C" __attribute__((visibility("default"))) int rt_debug_throw(int kind) {
    return guarded([&]() -> int {
        if (kind == 0) throw std::bad_alloc();
        if (kind == 1) throw std::runtime_error("rt_debug_throw");
        if (kind == 2) throw 42;
        if (kind == 3) { std::vector<char> v; v.reserve((size_t)-1 / 2); }      // a real failed allocation (length_error / bad_alloc)
        return RT_OK;
    });
}
#endif  // RT_DEBUG_HOOKS

static int rt_render_tile_impl(int device, const rt_tile_request* rq, const rt_sphere* sp, uint32_t ns,
                          const rt_triangle* tr, uint32_t nt, const uint32_t* world_index, uint8_t* out_rgb, size_t out_len,
                          float* out_f32, rt_tile_stats* stats) {
    int rc = check_request(rq);
    if (rc) return rc;
    if (!out_rgb) return fail(RT_ERR_BAD_ARG, "out_rgb is NULL");
    if (out_len < rt_tile_bytes(rq)) return fail(RT_ERR_BUFFER_TOO_SMALL, "out_len < (H/div)*W*3");
    rt_scene* sc = nullptr;
    rc = rt_scene_create_impl(device, sp, ns, tr, nt, world_index, &sc);
    if (rc) return rc;
    rc = rt_scene_render_tile_impl(sc, rq, out_rgb, out_len, out_f32, stats);
    std::string keep = g_err;
    rt_scene_destroy_impl(sc);
    if (rc) g_err = keep;
    return rc;
}

// =====================================================================================
// Frame context: the controller's dispatch + assembly (controller main.rs:47-75, 109-115) with everything a JOB needs kept
// alive between frames — one dispatcher thread per device entry, the world resident on every device, streams, strip
// buffers, and the page-locked registration of the caller's frame buffer.
// =====================================================================================
}  // extern "C"  (the struct below is a C++ type behind the opaque C handle)

struct FrameDev {
    int dev = 0;                        // device ordinal
    rt_scene* scene = nullptr;          // the job's world on this device (owned by the dispatcher thread)
    std::thread th;
    // strip-queue mode: two strips in flight, each on its own stream with its own device buffer (made on first use)
    hipStream_t qs[2] = {nullptr, nullptr};
    uint8_t* qd[2] = {nullptr, nullptr};
    size_t qcap = 0;
    // result of the last command
    int rc = RT_OK;
    std::string err;
    rt_tile_stats st;
    float busy_ms = 0.f;                // dispatcher wall time of the last render
};

struct rt_frame_ctx {
    std::vector<FrameDev> fd;
    std::mutex call_mu;                 // one API call at a time per context
    std::mutex mu;                      // command hand-over
    std::condition_variable cv_work, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    enum Cmd { CMD_NONE, CMD_UPLOAD, CMD_RENDER, CMD_STOP } cmd = CMD_NONE;
    // CMD_UPLOAD
    const HostScene* hs = nullptr;
    bool have_world = false;
    float scene_ms_pending = 0.f;       // duration of the last set_world, charged to the next frame's stats
    // CMD_RENDER
    rt_tile_request rq;
    uint8_t* out = nullptr;
    size_t strip = 0;
    bool use_queue = false;
    std::atomic<uint32_t> next_strip{0};
    // strip assignment (rt_assign.h): owner[k] = entry of strip k for THIS frame; strip_cost = ray segments per strip measured on the
    // job's last frame (valid while the world and the frame's geometry stay what they were)
    std::vector<uint32_t> owner;
    uint32_t assignment = 0;
    std::vector<double> strip_cost;
    rt_tile_request cost_rq;            // the frame the costs were measured on
    bool cost_valid = false;
    std::vector<unsigned long long> cost_now;   // filled by the dispatchers during a frame (each writes its own strips' entries)
    // the caller's frame buffer, page-locked once
    void* pinned_ptr = nullptr;
    size_t pinned_len = 0;
};

namespace {

int frame_dev_render(rt_frame_ctx* fc, int w) {
    FrameDev& d = fc->fd[w];
    const int nd = (int)fc->fd.size();
    const rt_tile_request& rq0 = fc->rq;
    const size_t strip = fc->strip;
    uint8_t* out_rgb = fc->out;
    std::memset(&d.st, 0, sizeof d.st);
    rt_scene* sc = d.scene;
    if (!sc) return fail(RT_ERR_BAD_ARG, "rt_frame_ctx_render before rt_frame_ctx_set_world");
    if (!fc->use_queue) {
        // strip k -> entry k % nd (controller main.rs:47-75 fires one request per division; Docker DNS round-robins them
        // over the slaves): all strips of this device go out as one batch (one launch per <= MAX_BATCH strips, the last
        // quarter as its own launch so that the downloads of the others run under it); stitch by division_no: strip k
        // lands at byte offset k * strip (controller main.rs:109-115)
        // (which strips: fc->owner, made by rt_frame_ctx_render — snake / longest-first by measured cost / k % nd, rt_assign.h)
        std::vector<rt_tile_request> rqs;
        std::vector<uint8_t*> outs;
        std::vector<uint32_t> mine;
        for (uint32_t k = 0; k < rq0.divisions; k++) {
            if (fc->owner[k] != (uint32_t)w) continue;
            rt_tile_request rq = rq0;
            rq.division_no = k;
            rqs.push_back(rq);
            outs.push_back(out_rgb + (size_t)k * strip);
            mine.push_back(k);
        }
        (void)nd;
        if (rqs.empty()) return RT_OK;
        std::vector<unsigned long long> cost(rqs.size(), 0ull);
        int r = rt_scene_render_tiles_impl(sc, rqs.data(), (uint32_t)rqs.size(), outs.data(), strip, nullptr, &d.st, cost.data());
        if (r) return r;
        for (size_t i = 0; i < mine.size(); i++) fc->cost_now[mine[i]] = cost[i];      // (disjoint entries per dispatcher)
        return RT_OK;
    }
    // ---- dynamic assignment: pull one strip at a time, the bottom of the frame first (its strips cost the most:
    // longest-first keeps the devices' finish times within one cheap strip of each other).  Two strips in flight per
    // entry, each on its own stream with its own device buffer: the launch tail and the download of one run under
    // the other.  Streams and buffers belong to the context: made on the first queue-mode frame, grown when a frame
    // has larger strips.
    DeviceCtx* ctx = sc->ctx;
    HIPCHK(hipSetDevice(ctx->dev));
    for (int i = 0; i < 2; i++)
        if (!d.qs[i]) HIPCHK(hipStreamCreateWithFlags(&d.qs[i], hipStreamNonBlocking));
    if (d.qcap < strip) {
        for (int i = 0; i < 2; i++) {
            (void)hipFree(d.qd[i]);
            d.qd[i] = nullptr;
        }
        d.qcap = 0;
        for (int i = 0; i < 2; i++)
            if (hipMalloc(&d.qd[i], strip) != hipSuccess) return fail(RT_ERR_OOM, "hipMalloc(strip) failed");
        d.qcap = strip;
    }
    bool busy[2] = {false, false};
    struct Settle {                       // an error return waits for what is already enqueued (it writes caller memory)
        FrameDev& d;
        bool* busy;
        ~Settle() {
            for (int i = 0; i < 2; i++)
                if (busy[i]) (void)hipStreamSynchronize(d.qs[i]);
        }
    } settle{d, busy};
    for (uint32_t turn = 0;; turn++) {
        const int sl = (int)(turn & 1u);
        if (busy[sl]) {
            HIPCHK(hipStreamSynchronize(d.qs[sl]));
            busy[sl] = false;
        }
        const uint32_t i = fc->next_strip.fetch_add(1);
        if (i >= rq0.divisions) break;
        rt_tile_request rq = rq0;
        rq.division_no = rq0.divisions - 1u - i;
        void* d1[1] = {d.qd[sl]};
        int r = rt_scene_render_tiles_device_impl(sc, &rq, 1, d1, strip, nullptr, d.qs[sl]);
        if (r) return r;
        HIPCHK(hipMemcpyAsync(out_rgb + (size_t)rq.division_no * strip, d.qd[sl], strip, hipMemcpyDeviceToHost, d.qs[sl]));
        busy[sl] = true;
    }
    for (int i = 0; i < 2; i++)
        if (busy[i]) {
            HIPCHK(hipStreamSynchronize(d.qs[i]));
            busy[i] = false;
        }
    return rt_scene_collect_impl(sc, &d.st);
}

void frame_dev_release(FrameDev& d) {
    if (d.scene) {
        rt_scene_destroy_impl(d.scene);       // (sets the scene's device current)
        d.scene = nullptr;
    } else {
        (void)hipSetDevice(d.dev);
    }
    for (int i = 0; i < 2; i++) {
        if (d.qs[i]) {
            (void)hipStreamSynchronize(d.qs[i]);
            (void)hipStreamDestroy(d.qs[i]);
            d.qs[i] = nullptr;
        }
        (void)hipFree(d.qd[i]);
        d.qd[i] = nullptr;
    }
    d.qcap = 0;
}

// dispatcher thread of entry w: sleeps on the context's condition variable between commands
void frame_dev_main(rt_frame_ctx* fc, int w) {
    FrameDev& d = fc->fd[w];
    uint64_t seen = 0;
    for (;;) {
        rt_frame_ctx::Cmd cmd;
        {
            std::unique_lock<std::mutex> lk(fc->mu);
            fc->cv_work.wait(lk, [&] { return fc->gen != seen; });
            seen = fc->gen;
            cmd = fc->cmd;
        }
        // an exception must not leave a thread function (std::terminate): same guard as the entry points
        const auto t0 = std::chrono::steady_clock::now();
        d.rc = guarded([&]() -> int {
            switch (cmd) {
                case rt_frame_ctx::CMD_UPLOAD: {
                    if (d.scene) {
                        rt_scene_destroy_impl(d.scene);
                        d.scene = nullptr;
                    }
                    return upload_scene(d.dev, *fc->hs, &d.scene);     // world uploaded once per device per job
                }
                case rt_frame_ctx::CMD_RENDER: return frame_dev_render(fc, w);
                case rt_frame_ctx::CMD_STOP: frame_dev_release(d); return RT_OK;
                default: return RT_OK;
            }
        });
        d.busy_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (d.rc) {
            try {
                d.err = g_err;
            } catch (...) {
            }
        }
        {
            std::lock_guard<std::mutex> lk(fc->mu);
            if (--fc->pending == 0) fc->cv_done.notify_all();
        }
        if (cmd == rt_frame_ctx::CMD_STOP) return;
    }
}

// hand one command to every dispatcher and wait for all of them; the first error (in entry order) is the call's
int frame_run(rt_frame_ctx* fc, rt_frame_ctx::Cmd cmd) {
    {
        std::lock_guard<std::mutex> lk(fc->mu);
        fc->cmd = cmd;
        fc->pending = (int)fc->fd.size();
        fc->gen++;
    }
    fc->cv_work.notify_all();
    {
        std::unique_lock<std::mutex> lk(fc->mu);
        fc->cv_done.wait(lk, [&] { return fc->pending == 0; });
    }
    for (FrameDev& d : fc->fd)
        if (d.rc) return fail(d.rc, d.err);
    return RT_OK;
}

void frame_unpin(rt_frame_ctx* fc) {
    if (fc->pinned_ptr) {
        if (!fc->fd.empty()) (void)hipSetDevice(fc->fd[0].dev);
        (void)hipHostUnregister(fc->pinned_ptr);
        fc->pinned_ptr = nullptr;
        fc->pinned_len = 0;
    }
}

int rt_frame_ctx_create_impl(const int* devices, int n_devices, rt_frame_ctx** out) {
    if (!out) return fail(RT_ERR_BAD_ARG, "out_ctx is NULL");
    *out = nullptr;
    if (!g_init) return fail(RT_ERR_NOT_INITIALIZED, "call rt_init() first");
    std::vector<int> devs;
    if (devices && n_devices > 0)
        devs.assign(devices, devices + n_devices);
    else
        for (size_t d = 0; d < g_ctx.size(); d++) devs.push_back((int)d);
    for (int d : devs)
        if (d < 0 || d >= (int)g_ctx.size()) return fail(RT_ERR_BAD_DEVICE, "bad device ordinal");
    rt_frame_ctx* fc = new rt_frame_ctx;
    fc->fd.resize(devs.size());
    std::memset(&fc->rq, 0, sizeof fc->rq);
    std::memset(&fc->cost_rq, 0, sizeof fc->cost_rq);
    size_t started = 0;
    try {
        for (; started < devs.size(); started++) {
            fc->fd[started].dev = devs[started];
            fc->fd[started].th = std::thread(frame_dev_main, fc, (int)started);
        }
    } catch (...) {
        // could not start every dispatcher: stop the ones that run (they wait for `pending` of their own count)
        fc->fd.resize(started);
        if (started) (void)frame_run(fc, rt_frame_ctx::CMD_STOP);
        for (FrameDev& d : fc->fd)
            if (d.th.joinable()) d.th.join();
        delete fc;
        throw;
    }
    *out = fc;
    return RT_OK;
}

int rt_frame_ctx_destroy_impl(rt_frame_ctx* fc) {
    if (!fc) return RT_OK;
    {
        std::lock_guard<std::mutex> call(fc->call_mu);
        (void)frame_run(fc, rt_frame_ctx::CMD_STOP);          // every dispatcher releases its world, streams and buffers
        for (FrameDev& d : fc->fd)
            if (d.th.joinable()) d.th.join();
        frame_unpin(fc);
    }
    delete fc;
    return RT_OK;
}

int rt_frame_ctx_set_world_impl(rt_frame_ctx* fc, const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt,
                                const uint32_t* world_index) {
    if (!fc) return fail(RT_ERR_BAD_ARG, "ctx is NULL");
    int rc = check_world(sp, ns, tr, nt, world_index);
    if (rc) return rc;
    if (!g_init) return fail(RT_ERR_NOT_INITIALIZED, "call rt_init() first");
    std::lock_guard<std::mutex> call(fc->call_mu);
    const auto t0 = std::chrono::steady_clock::now();
    // the world's host side once per job (the reference rebuilds the BVH per strip, slave main.rs:60)
    HostScene hs;
    build_host_scene(sp, ns, tr, nt, world_index, hs);
    fc->hs = &hs;
    fc->have_world = false;
    fc->cost_valid = false;            // another world: the strips' costs are to be measured again
    rc = frame_run(fc, rt_frame_ctx::CMD_UPLOAD);
    fc->hs = nullptr;
    if (rc) return rc;
    fc->have_world = true;
    fc->scene_ms_pending = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_frame_ctx_render_impl(rt_frame_ctx* fc, const rt_tile_request* rq_in, uint8_t* out_rgb, size_t out_len,
                             rt_frame_stats* stats) {
    if (!fc) return fail(RT_ERR_BAD_ARG, "ctx is NULL");
    if (!rq_in) return fail(RT_ERR_BAD_ARG, "request is NULL");
    rt_tile_request rq0 = *rq_in;
    rq0.division_no = 0;
    int rc = check_request(&rq0);
    if (rc) return rc;
    if (!out_rgb) return fail(RT_ERR_BAD_ARG, "out_rgb is NULL");
    // the controller's ImageBuffer::from_vec(width, height, ..).unwrap() (controller main.rs:117-119)
    // panics unless the strips tile the frame exactly
    if (rq0.height % rq0.divisions != 0) return fail(RT_ERR_FRAME_SIZE, "height % divisions != 0");
    const size_t strip = rt_tile_bytes(&rq0);
    const size_t frame_bytes = strip * rq0.divisions;
    if (out_len < frame_bytes) return fail(RT_ERR_BUFFER_TOO_SMALL, "out_len < H*W*3");
    std::lock_guard<std::mutex> call(fc->call_mu);
    if (!fc->have_world) return fail(RT_ERR_BAD_ARG, "rt_frame_ctx_render before rt_frame_ctx_set_world");
    const auto t0 = std::chrono::steady_clock::now();
    // page-lock the frame buffer ONCE: strip downloads become DMA that runs under the kernels.  The registration is kept
    // until another buffer comes (or release / destroy).  Best effort: a buffer the caller registered itself (or that
    // cannot be registered) is used as it is.
    float pin_ms = 0.f;
    const bool want_pin = !(rq0.flags & RT_FLAG_FRAME_NO_PIN);
    if (!want_pin || fc->pinned_ptr != (void*)out_rgb || fc->pinned_len < frame_bytes) {
        if (fc->pinned_ptr && !(want_pin && fc->pinned_ptr == (void*)out_rgb && fc->pinned_len >= frame_bytes)) frame_unpin(fc);
        if (want_pin) {
            (void)hipSetDevice(fc->fd[0].dev);                 // (not device 0 by accident: the context may exclude it)
            if (hipHostRegister(out_rgb, frame_bytes, hipHostRegisterPortable) == hipSuccess) {
                fc->pinned_ptr = out_rgb;
                fc->pinned_len = frame_bytes;
            } else {
                (void)hipGetLastError();
            }
            pin_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    rq0.flags &= ~(uint32_t)(RT_FLAG_FRAME_QUEUE | RT_FLAG_FRAME_NO_PIN | RT_FLAG_FRAME_STATIC);   // frame-level: not the kernels' business
    fc->use_queue = (rq_in->flags & RT_FLAG_FRAME_QUEUE) != 0;
    // The balanced assignments cut the frame into their OWN strips: at least six per entry, so that longest-first has something to
    // even out with (c5's 16 strips are two per entry at 8 devices: 3 % off the mean at best).  `divisions` is the reference's wire
    // format, not a property of the image: every sample's stream is keyed by its pixel's place in the FRAME (rt_tile.h `seed`), so
    // any cut into whole rows gives the same bytes.  The strip queue and RT_FLAG_FRAME_STATIC keep the request's strips.
    size_t istrip = strip;
    if (!fc->use_queue && !(rq_in->flags & RT_FLAG_FRAME_STATIC) && fc->fd.size() > 1) {
        const uint32_t want = std::max<uint32_t>(rq0.divisions, 6u * (uint32_t)fc->fd.size());
        for (uint32_t dv = want; dv <= std::min<uint32_t>(rq0.height, 4u * want); dv++)
            if (rq0.height % dv == 0) {
                rq0.divisions = dv;
                istrip = rt_tile_bytes(&rq0);
                break;
            }
    }
    fc->rq = rq0;
    fc->out = out_rgb;
    fc->strip = istrip;
    fc->next_strip.store(0);
    // which entry renders which strip (rt_assign.h).  The measured costs hold for the same world and the same frame geometry
    // (size, strips, samples, depth, camera, t window): the seed changes the paths, not where the expensive rows are.
    {
        rt_tile_request a = rq0, b = fc->cost_rq;
        a.seed = b.seed = 0;
        a.division_no = b.division_no = 0;
        a.flags = b.flags = 0;
        const bool usable = fc->cost_valid && same_frame(a, b) && fc->strip_cost.size() == rq0.divisions;
        const rtassign::Mode mode = fc->use_queue ? rtassign::QUEUE
                                    : (rq_in->flags & RT_FLAG_FRAME_STATIC) ? rtassign::STATIC_MOD
                                    : usable ? rtassign::BY_COST : rtassign::SNAKE;
        fc->assignment = (uint32_t)mode;
        rtassign::assign(rq0.divisions, (uint32_t)fc->fd.size(), usable ? fc->strip_cost.data() : nullptr,
                         mode == rtassign::QUEUE ? rtassign::STATIC_MOD : mode, fc->owner);
        fc->cost_now.assign(rq0.divisions, 0ull);
    }
    rc = frame_run(fc, rt_frame_ctx::CMD_RENDER);
    if (rc) return rc;
    if (!fc->use_queue) {
        // the strips' costs as this frame measured them: the next frame of the job is assigned by them
        fc->strip_cost.assign(fc->cost_now.begin(), fc->cost_now.end());
        fc->cost_rq = rq0;
        fc->cost_valid = true;
    }
    rt_frame_stats fs;
    std::memset(&fs, 0, sizeof fs);
    rt_tile_stats& tot = fs.totals;
    float last = -1.f;
    for (const FrameDev& d : fc->fd) {
        tot.ray_segments += d.st.ray_segments;
        tot.primary_rays += d.st.primary_rays;
        tot.broad_candidates += d.st.broad_candidates;
        tot.exact_fallbacks += d.st.exact_fallbacks;
        tot.node_steps += d.st.node_steps;
        tot.kernel_ms = std::max(tot.kernel_ms, d.st.kernel_ms);   // devices run concurrently
        tot.h2d_ms = std::max(tot.h2d_ms, d.st.h2d_ms);
        tot.d2h_ms = std::max(tot.d2h_ms, d.st.d2h_ms);
        tot.n_launches += d.st.n_launches;
        if (d.st.n_launches) {
            tot.engine = d.st.engine;
            tot.broad_form = d.st.broad_form;
        }
        if (d.busy_ms > last) {                                     // the entry that finished last
            last = d.busy_ms;
            // strip-queue mode keeps two launches in flight per entry, so their event times overlap and do not add up
            // to a duration: there kernel_ms is the dispatcher's wall time and nothing is booked as exposed download
            fs.kernel_ms = fc->use_queue ? d.busy_ms : d.st.kernel_ms;
            fs.d2h_exposed_ms = fc->use_queue ? 0.f : d.st.d2h_ms;
        }
    }
    fs.wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    fs.pin_ms = pin_ms;
    fs.scene_ms = fc->scene_ms_pending;
    fc->scene_ms_pending = 0.f;
    fs.host_ms = fs.wall_ms - fs.pin_ms - fs.kernel_ms - fs.d2h_exposed_ms;
    fs.n_devices = (uint32_t)fc->fd.size();
    fs.pinned = fc->pinned_ptr == (void*)out_rgb ? 1u : 0u;
    fs.assignment = fc->assignment;
    {
        unsigned long long mx = 0, tot = 0;
        for (size_t e = 0; e < fc->fd.size(); e++) {
            const unsigned long long sg = fc->fd[e].st.ray_segments;
            if (e < RT_FRAME_STATS_ENTRIES) fs.entry_segments[e] = sg;
            mx = std::max(mx, sg);
            tot += sg;
        }
        fs.balance_max_over_mean = tot ? (float)((double)mx * (double)fc->fd.size() / (double)tot) : 0.f;
    }
    if (stats) *stats = fs;
    return RT_OK;
}

int rt_frame_ctx_release_buffer_impl(rt_frame_ctx* fc) {
    if (!fc) return fail(RT_ERR_BAD_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> call(fc->call_mu);
    frame_unpin(fc);
    return RT_OK;
}

int rt_render_frame_impl(const int* devices, int n_devices, const rt_tile_request* rq_in, const rt_sphere* sp,
                         uint32_t ns, const rt_triangle* tr, uint32_t nt, const uint32_t* world_index, uint8_t* out_rgb,
                         size_t out_len, rt_tile_stats* stats) {
    // argument errors before any thread or upload
    if (!rq_in) return fail(RT_ERR_BAD_ARG, "request is NULL");
    rt_tile_request rq0 = *rq_in;
    rq0.division_no = 0;
    int rc = check_request(&rq0);
    if (rc) return rc;
    if (!out_rgb) return fail(RT_ERR_BAD_ARG, "out_rgb is NULL");
    rc = check_world(sp, ns, tr, nt, world_index);
    if (rc) return rc;
    if (!g_init) return fail(RT_ERR_NOT_INITIALIZED, "call rt_init() first");
    if (rq0.height % rq0.divisions != 0) return fail(RT_ERR_FRAME_SIZE, "height % divisions != 0");
    if (out_len < rt_tile_bytes(&rq0) * rq0.divisions) return fail(RT_ERR_BUFFER_TOO_SMALL, "out_len < H*W*3");
    rt_frame_ctx* fc = nullptr;
    rc = rt_frame_ctx_create_impl(devices, n_devices, &fc);
    if (rc) return rc;
    struct Destroy {
        rt_frame_ctx* fc;
        ~Destroy() {
            std::string keep = g_err;
            rt_frame_ctx_destroy_impl(fc);
            g_err = keep;
        }
    } destroy{fc};
    rc = rt_frame_ctx_set_world_impl(fc, sp, ns, tr, nt, world_index);
    if (rc) return rc;
    rt_frame_stats fs;
    rc = rt_frame_ctx_render_impl(fc, rq_in, out_rgb, out_len, &fs);
    if (rc) return rc;
    if (stats) *stats = fs.totals;
    return RT_OK;
}

}  // namespace

extern "C" {

// ---- the exported entry points: argument-for-argument the functions above, behind guarded() ----------------------
RT_API int rt_init(int* n_devices) { return guarded([&] { return rt_init_impl(n_devices); }); }
RT_API void rt_shutdown(void) { (void)guarded([&] { return rt_shutdown_impl(); }); }
RT_API int rt_scene_create(int device, const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt,
                           const uint32_t* world_index, rt_scene** out) {
    return guarded([&] { return rt_scene_create_impl(device, sp, ns, tr, nt, world_index, out); });
}
RT_API void rt_scene_destroy(rt_scene* sc) { (void)guarded([&] { return rt_scene_destroy_impl(sc); }); }
RT_API int rt_scene_render_tiles_device(rt_scene* sc, const rt_tile_request* rqs, uint32_t n, void* const* d_out_rgb,
                                        size_t out_len_each, void* const* d_out_f32, void* hip_stream) {
    return guarded([&] { return rt_scene_render_tiles_device_impl(sc, rqs, n, d_out_rgb, out_len_each, d_out_f32, hip_stream); });
}
RT_API int rt_scene_render_tile_device(rt_scene* sc, const rt_tile_request* rq, void* d_out_rgb, size_t out_len,
                                       void* d_out_f32, void* hip_stream) {
    return guarded([&] { return rt_scene_render_tile_device_impl(sc, rq, d_out_rgb, out_len, d_out_f32, hip_stream); });
}
RT_API int rt_scene_collect(rt_scene* sc, rt_tile_stats* st) { return guarded([&] { return rt_scene_collect_impl(sc, st); }); }
RT_API int rt_scene_render_tiles(rt_scene* sc, const rt_tile_request* rqs, uint32_t n, uint8_t* const* out_rgb,
                                 size_t out_len_each, float* const* out_f32, rt_tile_stats* stats) {
    return guarded([&] { return rt_scene_render_tiles_impl(sc, rqs, n, out_rgb, out_len_each, out_f32, stats); });
}
RT_API int rt_scene_render_tile(rt_scene* sc, const rt_tile_request* rq, uint8_t* out_rgb, size_t out_len, float* out_f32,
                                rt_tile_stats* stats) {
    return guarded([&] { return rt_scene_render_tile_impl(sc, rq, out_rgb, out_len, out_f32, stats); });
}
RT_API int rt_render_tile(int device, const rt_tile_request* rq, const rt_sphere* sp, uint32_t ns, const rt_triangle* tr,
                          uint32_t nt, const uint32_t* world_index, uint8_t* out_rgb, size_t out_len, float* out_f32,
                          rt_tile_stats* stats) {
    return guarded([&] { return rt_render_tile_impl(device, rq, sp, ns, tr, nt, world_index, out_rgb, out_len, out_f32, stats); });
}
RT_API int rt_render_frame(const int* devices, int n_devices, const rt_tile_request* rq, const rt_sphere* sp, uint32_t ns,
                           const rt_triangle* tr, uint32_t nt, const uint32_t* world_index, uint8_t* out_rgb, size_t out_len,
                           rt_tile_stats* stats) {
    return guarded([&] { return rt_render_frame_impl(devices, n_devices, rq, sp, ns, tr, nt, world_index, out_rgb, out_len, stats); });
}
RT_API int rt_frame_ctx_create(const int* devices, int n_devices, rt_frame_ctx** out) {
    return guarded([&] { return rt_frame_ctx_create_impl(devices, n_devices, out); });
}
RT_API int rt_frame_ctx_set_world(rt_frame_ctx* fc, const rt_sphere* sp, uint32_t ns, const rt_triangle* tr, uint32_t nt,
                                  const uint32_t* world_index) {
    return guarded([&] { return rt_frame_ctx_set_world_impl(fc, sp, ns, tr, nt, world_index); });
}
RT_API int rt_frame_ctx_render(rt_frame_ctx* fc, const rt_tile_request* rq, uint8_t* out_rgb, size_t out_len, rt_frame_stats* stats) {
    return guarded([&] { return rt_frame_ctx_render_impl(fc, rq, out_rgb, out_len, stats); });
}
RT_API int rt_frame_ctx_release_buffer(rt_frame_ctx* fc) { return guarded([&] { return rt_frame_ctx_release_buffer_impl(fc); }); }
RT_API void rt_frame_ctx_destroy(rt_frame_ctx* fc) { (void)guarded([&] { return rt_frame_ctx_destroy_impl(fc); }); }
// The file the library's HIP calls are bound to (rt_tile.h).  A process may hold two HIP runtimes — PyTorch wheels bundle their own
// libamdhip64.so (no SONAME), this library asks for ROCm's libamdhip64.so.7 — and the dynamic loader binds this library's hip*
// symbols to whichever came FIRST in the global scope (profiles/README.md, "two HIP runtimes in one process").  A host that makes
// its own streams or device buffers for the *_device entry points must make them with THIS runtime; the Python mirror uses it to
// refuse a process in which torch and the library would drive the device through different ones.
RT_API size_t rt_hip_runtime_path(char* buf, size_t cap) {
    Dl_info info;
    std::memset(&info, 0, sizeof info);
    if (!dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) || !info.dli_fname) return 0;
    const size_t n = std::strlen(info.dli_fname);
    if (buf && cap) {
        const size_t m = n < cap - 1 ? n : cap - 1;
        std::memcpy(buf, info.dli_fname, m);
        buf[m] = 0;
    }
    return n;
}
}  // extern "C"
