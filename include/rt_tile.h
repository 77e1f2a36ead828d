/*
 * rt_tile.h — C-ABI of the MI355X path-trace tile renderer.
 *
 * Drop-in boundary for ONE hot path of actuday6418/ray-tracer-s8: the body of the
 * slave's `worker()` NewJob arm (reference ray-tracer-slave/src/main.rs:37-83) and
 * everything it calls (`ray_color` main.rs:108-146, `Camera::get_ray` camera.rs:109-129,
 * `WorldRefList::intersect` shapes/mod.rs:158-191, Sphere/Triangle `get_roots`).
 *
 * The reference has no FFI; its seam is the controller->slave JSON `RenderInfo`
 * (ray-tracer-slave/src/lib.rs:10-15) answered by `ImageSlice` (lib.rs:17-22).  This
 * header carries exactly those fields as POD, plus the knobs the reference hard-codes
 * (main.rs:39,42-51; shapes/mod.rs:12-13) with the reference literals as defaults.
 *
 * Plain C: POD structs, pointers and sizes only.  A Rust `extern "C"` block, cgo or
 * ctypes can bind it verbatim.
 *
 * Every function returns an `rt_status` (0 = OK, negative = error) and never throws
 * or aborts across the boundary (the reference panics via `.unwrap()`).  Caller owns
 * every pointer before and after each call; the library copies in and retains nothing
 * outside an explicit `rt_scene` handle.
 */
#ifndef RT_TILE_H
#define RT_TILE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define RT_API
#else
#define RT_API __attribute__((visibility("default")))
#endif

#define RT_ABI_VERSION 4u      /* 2: rt_tile_stats.node_steps appended
                                  3: `world_index` (the position of every primitive in RenderInfo.world) on the entry
                                     points that take a world; the persistent frame context rt_frame_ctx_*;
                                     RT_FLAG_FRAME_QUEUE / RT_FLAG_FRAME_NO_PIN replace two environment variables
                                  4: the deterministic RNG is one stream per (pixel, SAMPLE) (see `seed` below: same seed,
                                     other images than ABI 3); RT_MAX_SPP; rt_hip_runtime_path(); the frame context assigns
                                     strips by measured cost and reports the balance (rt_frame_stats.balance_*,
                                     RT_FLAG_FRAME_STATIC) */

/* ---- status codes --------------------------------------------------------------- */
typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_BAD_ARG = -1,        /* null pointer, zero size, division_no >= divisions ...      */
    RT_ERR_NOT_INITIALIZED = -2,/* rt_init() not called / failed                              */
    RT_ERR_NO_DEVICE = -3,      /* no MI355X-class HIP device visible (no CPU fallback)       */
    RT_ERR_BAD_DEVICE = -4,     /* device ordinal out of range                                */
    RT_ERR_BUFFER_TOO_SMALL = -5,/* out_len < (H/div)*W*3                                     */
    RT_ERR_FRAME_SIZE = -6,     /* frame assembly asked with height % divisions != 0
                                   (reference controller panics: controller/src/main.rs:117-119) */
    RT_ERR_HIP = -7,            /* a HIP runtime call failed; see rt_last_error()             */
    RT_ERR_LIMIT = -8,          /* max_bounces > RT_MAX_BOUNCES, n_spheres too large ...      */
    RT_ERR_OOM = -9             /* host or device allocation failed                           */
} rt_status;

#define RT_MAX_BOUNCES 62u      /* path stack depth limit (reference literal is 10)           */
#define RT_MAX_SPP 4096u        /* samples per pixel limit (reference literal is 100)         */
#define RT_MAX_PRIMITIVES 0x3ffffffu /* spheres + triangles per scene: the kernels address nodes, geometry and
                                   materials with 32-bit byte offsets (64 B per tree node)    */

/* ---- scene primitives ----------------------------------------------------------- */

/* = reference `Sphere` (ray-tracer-slave/src/shapes/sphere.rs:12-20) minus `node_index`
 * (a BVH back-pointer the linear GPU scan does not use).  36 bytes, no padding. */
typedef struct rt_sphere {
    float cx, cy, cz;           /* center                                                     */
    float radius;
    float albedo_r, albedo_g, albedo_b; /* p_albedo_at                                        */
    float roughness;            /* p_roughness_at: 0 = Lambertian, 1 = mirror (main.rs:122)   */
    float emission;             /* p_emission_at: > 0 terminates the path (main.rs:116-117)   */
} rt_sphere;

/* = reference `Triangle` (ray-tracer-slave/src/shapes/mesh.rs:14-23) minus `node_index`.
 * 56 bytes, no padding. */
typedef struct rt_triangle {
    float a[3], b[3], c[3];
    float albedo_r, albedo_g, albedo_b;
    float roughness;
    float emission;
} rt_triangle;

/* ---- tile request = RenderMeta + division_no + the hard-coded knobs ------------- */

enum {
    RT_FLAG_NONE = 0u,
    /* Disable the conservative broad-phase filter: run the reference's exact root
     * computation against every primitive (slow; used by tests to prove the filter
     * never changes a result). */
    RT_FLAG_EXACT_SCAN = 1u << 0,
    /* Plain linear-scan semantics: do not apply the reference's BVH candidate culling
     * (bvh_impl.rs:373-398) to accepted hits and break distance ties by primitive index.
     * Default (flag clear) reproduces the reference: a hit counts only if BVH::traverse would
     * have returned the primitive, ties go to the earlier DFS leaf. */
    RT_FLAG_NO_BVH_CULL = 1u << 1,
    /* Force the 11-op "oc" broad phase even when the scene qualifies for the 8-op expanded
     * form (A/B testing; both are conservative and give identical images). */
    RT_FLAG_OC_BROAD_PHASE = 1u << 2,
    /* Walk the whole root-to-leaf AABB chain when validating a hit instead of using the
     * leaf-box monotonicity shortcut (A/B testing; identical images). */
    RT_FLAG_FULL_CHAIN = 1u << 3,
    /* Closest-hit engine.  Default: linear scan over the LDS-resident primitive list for scenes up to 32
     * spheres (and no mesh), per-lane traversal of the reference BVH above that.  Both give the
     * reference's result bit for bit; these force one or the other (A/B runs, tests). */
    RT_FLAG_BVH_TRAVERSE = 1u << 4,
    RT_FLAG_LINEAR_SCAN = 1u << 5,
    /* Traversal node format.  Default: the exact 64-byte nodes below 4096 primitives and for meshes, the
     * 32-byte conservatively quantised nodes (exact validation at the leaves) for larger sphere scenes.  Identical images;
     * these force one or the other (A/B runs, tests). */
    RT_FLAG_EXACT_NODES = 1u << 6,
    RT_FLAG_QUANT_NODES = 1u << 7,
    /* A tree whose exact nodes fit a CU's LDS (about 1000 primitives) is walked from an LDS-resident copy by default;
     * this flag keeps the nodes in HBM / L2 (A/B runs, tests).  Identical images. */
    RT_FLAG_NO_LDS_TREE = 1u << 8,
    /* Measurement aid (bench.py's roofline object): run the traversal kernel's counting twin, which also counts the
     * internal BVH nodes visited (rt_tile_stats.node_steps).  Same image; a few per cent slower, never the timed launch. */
    RT_FLAG_COUNT_STEPS = 1u << 9,
    /* Quantised walk, nearer child first, skipping every subtree whose box the ray enters beyond the running closest hit by
     * more than a proven slack (identical images).  Default for sphere scenes on the quantised nodes, and for scenes with
     * triangles on the exact nodes, that are dense enough for it to pay (c5, terrains); these force it on / off (A/B runs, tests). */
    RT_FLAG_CULL_WALK = 1u << 10,
    RT_FLAG_NO_CULL_WALK = 1u << 11,
    /* Frame-level flags (rt_render_frame / rt_frame_ctx_render only; the tile entry points ignore them).
     * FRAME_QUEUE: the devices pull strips one at a time, bottom of the frame (the expensive strips) first, from a shared
     * host-atomic queue, two launches in flight per device, instead of the static split strip k -> devices[k % n].
     * FRAME_NO_PIN: do not page-lock the caller's frame buffer (downloads then go through the runtime's staging). */
    RT_FLAG_FRAME_QUEUE = 1u << 12,
    RT_FLAG_FRAME_NO_PIN = 1u << 13,
    /* FRAME_STATIC: the plain split strip k -> devices[k % n] (rounds 1-3; A/B runs, tests) instead of the default, which
     * balances the devices by the strips' cost: see "Strip assignment" at rt_frame_ctx below. */
    RT_FLAG_FRAME_STATIC = 1u << 14
};

typedef struct rt_tile_request {
    /* RenderMeta (lib.rs:24-30); `id` (UUID) is opaque to the renderer, stays with caller */
    uint32_t width;             /* image width  W                                             */
    uint32_t height;            /* image height H                                             */
    uint32_t divisions;         /* number of horizontal strips; strip height Hs = H / div     */
    /* RenderInfo.division_no (lib.rs:14): strip index, 0 = TOP of the image (main.rs:66-71)  */
    uint32_t division_no;
    /* knobs the reference hard-codes; rt_tile_request_defaults() fills the literals         */
    uint32_t spp;               /* sample_count   = 100   (main.rs:51)                        */
    uint32_t max_bounces;       /* max_bounces    = 10    (main.rs:39); ray_color depth = +1  */
    float aperture;             /* 0.1            (main.rs:45)                                */
    float focus_distance;       /* 1.0            (main.rs:46)                                */
    float fov;                  /* PI/2 (f32)     (main.rs:47)                                */
    float focal_length;         /* 1.0            (main.rs:48)                                */
    float t_min;                /* 0.001          (shapes/mod.rs:12)                          */
    float t_max;                /* 1000.0         (shapes/mod.rs:13), half-open [t_min,t_max) */
    /* new: replaces `SmallRng::from_entropy()` per row (main.rs:69).  Sample s (0 .. spp-1) of the pixel in global row y,
     * column x of the frame draws from its own xoshiro256++ stream,
     *     SmallRng::seed_from_u64(seed + 4 * 0x9E3779B97F4A7C15 * ((y * width + x) * spp + s))      (wrapping u64)
     * i.e. the SplitMix64 sequence of `seed` cut into consecutive blocks of four outputs, one block per (pixel, sample);
     * the pixel is the f32 sum of its samples' colours in the order s = 0, 1, ... as main.rs:73-77 forms it.  Strips of a
     * frame rendered with one seed therefore stitch to exactly the single-strip frame.  DESIGN.md 3 "RNG";
     * tests/test_rng_distribution.py checks the images' distribution against the reference's stream-per-row structure. */
    uint64_t seed;
    uint32_t flags;             /* RT_FLAG_*                                                  */
    uint32_t reserved;          /* must be 0                                                  */
} rt_tile_request;

typedef struct rt_tile_stats {
    uint64_t ray_segments;      /* ray_color entries with depth > 0 (closest-hit queries)     */
    uint64_t primary_rays;      /* Hs * W * spp                                               */
    uint64_t broad_candidates;  /* primitives that passed the broad phase (0 in exact scan)   */
    uint64_t exact_fallbacks;   /* segments whose candidate list overflowed (exact rescans)   */
    float kernel_ms;            /* HIP-event time of the kernel(s) of this call               */
    float h2d_ms;               /* scene upload (0 when a resident rt_scene is used)          */
    float d2h_ms;               /* RGB8 strip download (0 for device output)                  */
    uint32_t n_launches;        /* kernel launches issued by this call                        */
    uint32_t engine;            /* closest-hit engine of the last launch: 0 linear scan (scene resident
                                   in LDS), 1 linear scan (scene streamed through LDS), 2 BVH traversal (exact nodes),
                                   3 BVH traversal (quantised nodes + exact leaf validation),
                                   4 BVH traversal, exact nodes resident in LDS,
                                   5 BVH traversal, quantised nodes, nearer child first with distance culling,
                                   6 BVH traversal, exact nodes, nearer child first with distance culling (scenes with triangles),
                                   7 BVH traversal, exact nodes resident in LDS, nearer child first with distance culling */
    uint32_t broad_form;        /* linear engines: 0 = oc form, 1 = expanded form   */
    uint64_t node_steps;        /* traversal engines under RT_FLAG_COUNT_STEPS: internal BVH nodes visited
                                   (each = two child-box slab tests); 0 otherwise.  broad_candidates = leaves
                                   reached = exact root tests for these engines */
} rt_tile_stats;

/* ---- lifecycle ------------------------------------------------------------------ */

/* Enumerate HIP devices, create one context (stream + events + counters) per device.
 * *n_devices may be NULL.  Returns RT_ERR_NO_DEVICE if none: there is NO CPU fallback. */
RT_API int rt_init(int* n_devices);
/* Refused (no effect; rt_last_error() says so) while any rt_scene is alive: destroy the scenes first. */
RT_API void rt_shutdown(void);
RT_API uint32_t rt_abi_version(void);
/* The libamdhip64 file this library's HIP calls are bound to (a process may map two HIP runtimes, e.g. a PyTorch wheel's
 * own next to ROCm's; the dynamic loader binds the library to whichever was loaded first).  A host that passes its own
 * hipStream_t / device pointers to the *_device entry points must make them with THIS runtime.  Copies at most cap - 1
 * characters and a terminator into buf (which may be NULL); returns the path's length, 0 if unknown. */
RT_API size_t rt_hip_runtime_path(char* buf, size_t cap);
RT_API const char* rt_strerror(int status);
/* Last error message of the calling thread ("" if none). */
RT_API const char* rt_last_error(void);

/* Fill the reference literals (main.rs:39-51, shapes/mod.rs:12-13, controller main.rs:33-39:
 * 1920x1080, 20 divisions); seed 0, flags 0. */
RT_API void rt_tile_request_defaults(rt_tile_request* req);

/* Bytes of one strip = (H / div) * W * 3  (main.rs:53-59).  0 on bad args. */
RT_API size_t rt_tile_bytes(const rt_tile_request* req);

/* ---- the world's order ------------------------------------------------------------ */
/* RenderInfo.world is ONE list, `Vec<Object>`, whose entries are spheres or triangles in any order
 * (ray-tracer-slave/src/lib.rs:11, shapes/mod.rs:23-27), and that order is observable: BVH::build numbers the shapes by
 * their position in it (bvh_impl.rs:421-427), so the halves of its `split_at(len / 2)` fallback (:277-291), the order of
 * the leaves BVH::traverse returns, and with it the winner among hits at exactly equal distance (`min_by` keeps the first,
 * shapes/mod.rs:177-182) all follow it.  The ABI carries the world as two typed arrays; `world_index` restores the order:
 * world_index[i] (i < n_spheres) is the position of spheres[i] in RenderInfo.world, world_index[n_spheres + j] that of
 * triangles[j]; it must be a permutation of 0 .. n_spheres + n_triangles - 1 (else RT_ERR_BAD_ARG).  NULL means the
 * spheres in their order followed by the triangles in theirs.  Copied during the call; the caller keeps ownership. */

/* ---- one strip, host buffers: replaces slave main.rs:53-83 ---------------------- */

/* Render strip `req->division_no` of the frame on `device` into out_rgb
 * (= ImageSlice.image: Hs*W*3 bytes, RGB8, row-major, top row of the strip first).
 * Synchronous.  Thread-safe across devices; calls on one device serialise.
 * Empty world (n_spheres + n_triangles == 0) renders the sky (the reference recurses
 * without bound in BVH::build, bvh_impl.rs:229-364).
 * Like the slave, accepts height % divisions != 0 and renders floor(H/div) rows.
 * out_f32 (optional, may be NULL): Hs*W*3 floats, post-gamma pre-quantise pixel values. */
RT_API int rt_render_tile(int device, const rt_tile_request* req,
                          const rt_sphere* spheres, uint32_t n_spheres,
                          const rt_triangle* triangles, uint32_t n_triangles,
                          const uint32_t* world_index,
                          uint8_t* out_rgb, size_t out_len,
                          float* out_f32, rt_tile_stats* stats);

/* ---- resident scene: upload the world once per device per job ------------------- */
/* (the reference re-sends and re-builds per strip: controller main.rs:58-62, slave main.rs:60) */

typedef struct rt_scene rt_scene;

RT_API int rt_scene_create(int device,
                           const rt_sphere* spheres, uint32_t n_spheres,
                           const rt_triangle* triangles, uint32_t n_triangles,
                           const uint32_t* world_index,
                           rt_scene** out_scene);
RT_API void rt_scene_destroy(rt_scene* scene);

/* Same as rt_render_tile, scene already in HBM. */
RT_API int rt_scene_render_tile(rt_scene* scene, const rt_tile_request* req,
                                uint8_t* out_rgb, size_t out_len,
                                float* out_f32, rt_tile_stats* stats);

/* Asynchronous, device-resident output: enqueue the strip on `hip_stream`
 * (a hipStream_t; NULL = the scene's own stream) writing RGB8 to device memory
 * d_out_rgb (>= rt_tile_bytes) and, if non-NULL, floats to d_out_f32.
 * Counters and HIP-event timings accumulate in the scene until rt_scene_collect().
 * Launches of one scene may be spread over several streams and overlap; the library chains only
 * those that share per-scene scratch (the capped-stack walk of trees deeper than the LDS stack). */
RT_API int rt_scene_render_tile_device(rt_scene* scene, const rt_tile_request* req,
                                       void* d_out_rgb, size_t out_len,
                                       void* d_out_f32, void* hip_stream);

/* Batched form: n strips of ONE frame (all fields equal except division_no and seed) are
 * rendered by a single launch of persistent waves that pull 64x1-pixel tiles of all n strips
 * from one queue — no per-strip launch tail.  d_out_rgb[i] receives strip i; d_out_f32 may
 * be NULL (or an array with NULL entries).  RT_ERR_BAD_ARG if the requests differ in any
 * frame-level field. */
RT_API int rt_scene_render_tiles_device(rt_scene* scene, const rt_tile_request* reqs, uint32_t n,
                                        void* const* d_out_rgb, size_t out_len_each,
                                        void* const* d_out_f32, void* hip_stream);

/* Host-buffer batched form (synchronous): one launch, then one D2H copy per strip. */
RT_API int rt_scene_render_tiles(rt_scene* scene, const rt_tile_request* reqs, uint32_t n,
                                 uint8_t* const* out_rgb, size_t out_len_each,
                                 float* const* out_f32, rt_tile_stats* stats);

/* Wait for all work enqueued on the scene, return accumulated counters / event time
 * since the previous collect, and reset them. */
RT_API int rt_scene_collect(rt_scene* scene, rt_tile_stats* stats);

/* ---- whole frame: replaces controller dispatch + assembly ----------------------- */
/* (controller main.rs:47-75 `for division_no in 0..divisions` and :109-115 stitch.)
 *
 * rt_frame_ctx is the controller's state for a JOB: a set of devices, each with one dispatcher thread, the job's world
 * resident in its HBM, its streams and its strip buffers, plus the page-locked registration of the caller's frame buffer.
 * Everything is created once — the context by rt_frame_ctx_create, the world by rt_frame_ctx_set_world, the registration
 * by the first rt_frame_ctx_render that sees a buffer — and reused: a second frame of the job (same world, same buffer)
 * registers, allocates, uploads and spawns nothing.  rt_render_frame is the one-shot wrapper (create, set world, render
 * one frame, destroy).
 *
 * Strip assignment.  Strips are not equally expensive (sky rows: one segment per sample; ground rows bounce), and a frame
 * is done when its slowest device is.  Default: the first frame of a job goes out in SNAKE order — strip k to entry k % n
 * in even rows of n strips, to n-1 - k % n in odd ones: every entry gets one strip of each row, which evens out any cost
 * profile that is close to linear in the strip's position — and the kernels count the ray segments of every strip; every
 * later frame of the job (same world, same frame geometry) is assigned LONGEST-FIRST by those counts, each strip to the
 * entry with the least load so far.  For these two assignments the context cuts the frame into its own strips — at least
 * six per entry, whole rows, the next count from max(divisions, 6 n) up that divides the height — since `divisions` is the
 * reference's wire format, not a property of the image: every cut into whole rows renders the same bytes (`seed` above).  All strips of an entry go out in one launch (above 64 MiB of pixels per device: two,
 * the last quarter of the strips running under the downloads of the rest).  RT_FLAG_FRAME_STATIC in req->flags: the plain
 * split strip k -> devices[k % n]; RT_FLAG_FRAME_QUEUE: the devices pull strips one at a time, bottom of the frame first,
 * two launches in flight per device.  The RGB8 strips are stitched by division_no into out_rgb (H*W*3).  Same bytes
 * whatever the assignment.  No collective, no peer traffic: strips are independent.
 * req->division_no is ignored.  height % divisions must be 0 (the controller's from_vec(..).unwrap() panics otherwise). */

typedef struct rt_frame_ctx rt_frame_ctx;
#define RT_FRAME_STATS_ENTRIES 16u

/* Where the wall time of one rt_frame_ctx_render call went (milliseconds).  Devices run concurrently: kernel_ms and
 * d2h_exposed_ms are those of the device that finished last. */
typedef struct rt_frame_stats {
    rt_tile_stats totals;       /* counters summed over the devices; kernel_ms / d2h_ms = the largest per-device value */
    float wall_ms;              /* the whole call, steady clock                                                        */
    float pin_ms;               /* page-locking out_rgb in this call (0 when the buffer was already registered)       */
    float scene_ms;             /* host-side world preparation + uploads charged to this frame: the duration of the
                                   rt_frame_ctx_set_world since the previous frame (0 for every later frame of the job) */
    float kernel_ms;            /* HIP-event time of the launches of the device that finished last                    */
    float d2h_exposed_ms;       /* last launch done -> last strip byte on the host, same device                       */
    float host_ms;              /* wall_ms - pin_ms - kernel_ms - d2h_exposed_ms: dispatch, thread wake-up, joins      */
    uint32_t n_devices;
    uint32_t pinned;            /* 1: out_rgb is page-locked (strip downloads are direct DMA)                          */
    uint32_t assignment;        /* 0: static k % n, 1: snake (no costs yet), 2: longest-first by the previous frame's
                                   per-strip ray segments, 3: strip queue                                              */
    float balance_max_over_mean;/* ray segments of the busiest entry / mean over the entries, THIS frame (1 = even)    */
    uint64_t entry_segments[RT_FRAME_STATS_ENTRIES];   /* ray segments per entry (the first RT_FRAME_STATS_ENTRIES)    */
} rt_frame_stats;

/* devices == NULL (or n_devices <= 0) means all devices.  A device may be listed more than once (several dispatcher
 * threads sharing it).  Starts one dispatcher thread per entry. */
RT_API int rt_frame_ctx_create(const int* devices, int n_devices, rt_frame_ctx** out_ctx);
/* Make this world the job's: host-side preparation once (device layouts, the candidate-filter BVH), one upload per
 * device.  Replaces the previous world of the context.  Must not race with rt_frame_ctx_render on the same context. */
RT_API int rt_frame_ctx_set_world(rt_frame_ctx* ctx,
                                  const rt_sphere* spheres, uint32_t n_spheres,
                                  const rt_triangle* triangles, uint32_t n_triangles,
                                  const uint32_t* world_index);
/* Render one frame of the job into out_rgb (>= H*W*3 bytes).  The context page-locks out_rgb (hipHostRegister) the
 * first time it sees it and keeps the registration until another buffer is passed or the context is destroyed — pass the
 * same buffer for every frame of a job and only the first pays pin_ms.  The caller must not free a buffer the context
 * still holds: call rt_frame_ctx_release_buffer (or destroy the context) first.  Synchronous; one call at a time per
 * context.  stats may be NULL. */
RT_API int rt_frame_ctx_render(rt_frame_ctx* ctx, const rt_tile_request* req,
                               uint8_t* out_rgb, size_t out_len, rt_frame_stats* stats);
/* Drop the page-locked registration of the last frame buffer (no-op if none). */
RT_API int rt_frame_ctx_release_buffer(rt_frame_ctx* ctx);
/* Stops the dispatcher threads, releases the worlds, buffers, streams and the registration. */
RT_API void rt_frame_ctx_destroy(rt_frame_ctx* ctx);

/* One-shot: a context over `devices`, this world, one frame, everything released again. */
RT_API int rt_render_frame(const int* devices, int n_devices, const rt_tile_request* req,
                           const rt_sphere* spheres, uint32_t n_spheres,
                           const rt_triangle* triangles, uint32_t n_triangles,
                           const uint32_t* world_index,
                           uint8_t* out_rgb, size_t out_len, rt_tile_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* RT_TILE_H */
