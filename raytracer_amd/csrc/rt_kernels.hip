// rt_kernels.hip — the wavefront path tracer for gfx950 (MI355X) and the
// device half of the C ABI.
//
// One stratum of one shard = one wavefront run:
//
//   (primary rays: camera + jitter, system.rs:237-242, 57-66, regenerated inside the depth-0 kernels)
//   repeat for depth = 0 .. max_depth-1 while rays remain:
//     k_intersect nearest hit over the object list except (system.rs:122-133, shapes/*)   rt_k_intersect.h
//                 the triangles; rays with no mesh to visit are shaded here (as k_shade), the others are queued
//                 as mesh records (ray + hit so far)
//     k_mesh_one  the mesh records against the triangles:  (mesh.rs, composite.rs)        rt_k_mesh_pool.h
//     | k_mesh    grid walk (one mesh job | general, or kd-tree), or one wave per ray for    rt_k_mesh.h,
//     | k_mesh_brute  short queues                                                            rt_k_mesh_grid_round.inc
//     k_shade     the mesh records: emit / scatter / terminate, (system.rs:112-120,135-142,  rt_k_shade.h
//                 survivors into the other ray queue            materials/*, texture.rs)     rt_k_material.h
//   k_tail        the last few thousand paths, all their remaining segments in one launch
// This file: the host side - scene upload, workspace, the pass loop, the C ABI entry points.
//
// Every path advances one segment per pass, so `depth` is a launch constant.
// Survivors are appended with one atomic per workgroup (ballot + popcount
// prefix), which keeps the queues dense as paths die at different depths.  A path
// contributes to its pixel exactly once, when it ends:
//   colour = a_0 * a_1 * ... * a_{k-1} * X,   X = background            (miss, depth cap)
//                                             X = emit + background     (light)
// because every reference material either scatters or emits, never both.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_api.h"
#include "../../include/rt_rng.h"
#include "rt_dev_shapes.h"
#include "rt_flatten.h"
#include "rt_shard.h"

namespace rt_host {
void set_last_error(const std::string &msg);
}

#include "rt_k_common.h"
#include "rt_k_intersect.h"
#include "rt_k_mesh.h"
#include "rt_k_mesh_pool.h"
#include "rt_k_shade.h"
#include "rt_k_tail.h"

// =====================================================================================
// host side
// =====================================================================================
static bool rt_debug() { static int v = -1; if (v < 0) v = getenv("RT_DEBUG") ? 1 : 0; return v == 1; }
#define RT_TRACE(...) do { if (rt_debug()) { fprintf(stderr, "[rt] " __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)
#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            rt_host::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
            return RT_ERR_DEVICE;                                                                                      \
        }                                                                                                              \
    } while (0)

// Wavefront state of one render call: everything the kernels write besides the caller's outputs.
// Either carved from one hipMalloc that the scene owns and grows on demand, or (rt_render_device_noalloc)
// from the static device array `rt_noalloc_scratch`.
struct Workspace {
    RayQueue q[2]; // ping-pong ray queues
    RayQueue mq;   // mesh record queue, ray part: the rays that have to visit a mesh this pass, by queue position
    HitBuf mqh;    // mesh record queue, hit part: their nearest hit so far (k_mesh merges the triangles' into it)
    double *mres_t = nullptr;     // k_mesh_one's result per mesh record: t in object space,
    uint32_t *mres_tri = nullptr; // triangle (~0: none; set by k_intersect with the record); k_shade merges it
    uint32_t *d_counts = nullptr;   // 2 x (depth_cap + 2) + 2: live rays per depth, mesh-queue lengths per depth, k_tail's lane totals
    uint32_t *d_rcnt = nullptr;     // the queues' region counters (QRegions, rt_k_common.h): [ray queues | mesh queue][region][depth], regions rstride words apart
    uint32_t rstride = 0;
    uint32_t *d_work = nullptr;     // k_mesh's block cursors: 2 structures x RT_MESH_CURSORS, 64 bytes apart; zeroed by every k_intersect
    uint32_t *d_guards = nullptr;   // 8
    size_t cap = 0;                 // ray slots
    uint32_t depth_cap = 0;         // max_depth the counters are sized for
};

// ---- rt_render_device_noalloc: wavefront state in a static device array -------------------------
// For callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc).
// Every GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends; the name below is only a host-side handle for hipGetSymbolAddress - host code uses the
// per-device pointer obtained at set-up (rt_scene_create), never the name.
// RT_NOALLOC_MAX_SLOTS, RT_NOALLOC_MAX_DEPTH: rt_api.h
#define RT_WS_BYTES_PER_SLOT 304u // carve_workspace: 34 f64 + 8 u32 arrays (two ray queues of 80 B, mesh records of 132 + 12 B)
#define RT_NOALLOC_BYTES ((size_t)RT_NOALLOC_MAX_SLOTS * RT_WS_BYTES_PER_SLOT + 8192u + 2u * RT_Q_REGIONS * 4096u) // 99.8 MiB
static __device__ __align__(256) unsigned char rt_noalloc_scratch[RT_NOALLOC_BYTES];
#define RT_MAX_DEVICES 64
static void *g_noalloc_ptr[RT_MAX_DEVICES]; // device address of rt_noalloc_scratch, per device
static std::atomic<int> g_noalloc_busy[RT_MAX_DEVICES]; // a call is using that device's array

// once per device, with that device current
static int noalloc_setup(int device)
{
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (device < 0 || device >= RT_MAX_DEVICES) return RT_OK; // such a device simply has no no-alloc entry point
    if (g_noalloc_ptr[device]) return RT_OK;
    void *p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(rt_noalloc_scratch)));
    g_noalloc_ptr[device] = p;
    return RT_OK;
}

// A host thread a scene keeps for rt_render_multi: it drives this scene's GPU while the calling thread drives the
// first one's.  Started on first use, joined when the scene is destroyed.
struct SceneWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool pending = false, finished = false, quit = false;
    ~SceneWorker();
    void submit(std::function<void()> f);
    void wait();
    void loop();
};

struct rt_scene {
    int device = 0;
    std::vector<void *> allocs; // scene data
    DevScene dev;
    // workspace, grown on demand
    void *ws = nullptr;
    Workspace own;
    bool has_csg = false, has_medium = false, has_mesh = false, has_torus = false;
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for this scene's LL variants of k_intersect / k_tail on its device
    // (to the device's limit, not to this scene's need: another scene may launch the same instantiation with more); a
    // refusal sends the scene (the tail) back to the scratch-memory lists
    bool ll_attr_set = false, ll_tail_attr_set = false, ll_tail_ok = true;
    int lds_limit = 65536;       // LDS a workgroup of this device may ask for
    uint32_t isect_ll_bytes = 0; // dynamic LDS of k_intersect<.., LL = true> (scene tables + every lane's interval lists); 0: the lists stay in scratch memory
    uint64_t n_mesh_tris = 0;
    uint32_t mesh_lds_bytes = 0; // dynamic LDS of k_mesh: occupancy words (+ cell starts)
    bool any_grid = false, any_kd = false; // (object, mesh) jobs walked through grids / through kd-trees
    uint32_t mesh_kd_lds_bytes = 0; // dynamic LDS of k_mesh<KD>: job records + stacks
    uint32_t mesh_pool_lds_bytes = 0; // dynamic LDS of k_mesh_one (tables + the waves' stacks); 0: several jobs or no room, k_mesh walks the grids
    uint32_t tail_grid_lds_bytes = 0; // dynamic LDS of k_tail_grid (scene tables + the one job's grid tables); 0: k_tail takes the tail
    // What the last stratum of the same shape did (pass_hint_key: image, tiles, shard, max_depth): rays per depth level
    // and where the tail took over.  With it the next stratum's passes are queued without waiting for a single
    // queue length to come back - the lengths only size grids, the kernels read the true ones on the device.
    uint64_t pass_hint_key[2] = {0, 0};
    std::vector<uint32_t> pass_hint_counts; // rays entering depth k (as far as the passes went)
    int pass_hint_tail_at = -1;             // depth at which the tail kernel took over, -1: it did not
    uint32_t host_depth_cap = 0;  // max_depth the pinned buffer and the events below are sized for
    uint32_t *h_pinned = nullptr; // readback slot
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> evp; // per-pass timing events: [0] before raygen, then 4 per depth
    std::vector<hipEvent_t> evr; // per depth: its queue length has been copied back
    std::vector<uint8_t> walked; // per depth: k_mesh (the grid walk) ran
    // rt_render / rt_render_multi: output buffers kept on the device between calls, grown on demand
    double *out_rgb = nullptr;
    int32_t *out_ids = nullptr;
    size_t out_cap = 0; // pixels
    // progress snapshots: tone-mapped RGB8 on the device and in pinned host memory
    uint8_t *snap_d = nullptr, *snap_h = nullptr;
    size_t snap_cap = 0; // bytes
    // rt_render_multi: this scene's stream, its "shard pushed to the root" event and its host thread, kept between calls
    hipStream_t multi_stream = nullptr;
    hipEvent_t multi_done = nullptr;
    SceneWorker *worker = nullptr;
};

template <typename T> static int upload(rt_scene *s, const std::vector<T> &v, const T **out)
{
    *out = nullptr;
    if (v.empty()) return RT_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    s->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)p;
    return RT_OK;
}

// Carves the wavefront state for `cap` ray slots and path depths up to `max_depth` out of `base`
// (256-byte aligned) and returns the bytes used; with base == nullptr it only measures.
static size_t carve_workspace(char *base, size_t cap, uint32_t max_depth, Workspace *w)
{
    const size_t n = (cap + 63) & ~(size_t)63; // every array starts 256-byte aligned
    size_t off = 0;
    auto take = [&](size_t count, size_t elem) {
        void *r = base ? (void *)(base + off) : nullptr;
        off += (count * elem + 255) & ~(size_t)255;
        return r;
    };
    Workspace tmp;
    Workspace &o = w ? *w : tmp;
    RayQueue *queues[3] = {&o.q[0], &o.q[1], &o.mq};
    for (RayQueue *rq : queues) {
        for (int c = 0; c < 3; ++c) rq->o[c] = (double *)take(n, 8);
        for (int c = 0; c < 3; ++c) rq->d[c] = (double *)take(n, 8);
        for (int c = 0; c < 3; ++c) rq->thr[c] = (double *)take(n, 8);
    }
    o.mqh.t = (double *)take(n, 8);
    for (int c = 0; c < 3; ++c) o.mqh.n[c] = (double *)take(n, 8);
    for (int c = 0; c < 2; ++c) o.mqh.uv[c] = (double *)take(n, 8);
    o.mres_t = (double *)take(n, 8);
    for (RayQueue *rq : queues) {
        rq->slot = (uint32_t *)take(n, 4);
        rq->pixel = (uint32_t *)take(n, 4);
        rq->primary = 0;
        rq->row_major = 0;
        rq->rank = rq->s_i = rq->s_j = 0u;
        rq->recip_sqrt_spp = 1.0;
        std::memset(&rq->g, 0, sizeof(rq->g));
    }
    o.mqh.obj = (int32_t *)take(n, 4);
    o.mres_tri = (uint32_t *)take(n, 4);
    o.d_counts = (uint32_t *)take(2 * ((size_t)max_depth + 2) + 2, 4);
    o.rstride = std::max<uint32_t>(1024u, (max_depth + 2u + 63u) & ~63u); // 4 KB between the counters of one queue
    o.d_rcnt = (uint32_t *)take(2 * (size_t)RT_Q_REGIONS * o.rstride, 4);
    o.d_work = (uint32_t *)take(2 * RT_MESH_CURSORS * 16, 4);
    o.d_guards = (uint32_t *)take(8, 4);
    o.cap = cap;
    o.depth_cap = max_depth;
    return off;
}

// RT_Q_SKEW=1 (a test knob, read when a scene is created): every append tries region 0 first
static std::atomic<int> g_q_skew{0};

// the counters of a queue: table 0 - the rays entering `depth`, table 1 - the mesh records of `depth`
static QRegions queue_regions(const Workspace &w, int table, uint32_t depth)
{
    QRegions r;
    r.cnt = w.d_rcnt + (size_t)table * RT_Q_REGIONS * w.rstride + depth;
    r.rc = (uint32_t)(((w.cap + 63) & ~(size_t)63) / RT_Q_REGIONS); // carve_workspace: the arrays hold RT_Q_REGIONS * rc entries
    r.stride = w.rstride;
    r.first_mask = g_q_skew.load(std::memory_order_relaxed) ? 0u : RT_Q_REGIONS - 1u;
    return r;
}

static void free_workspace(rt_scene *s)
{
    if (s->ws) (void)hipFree(s->ws);
    s->ws = nullptr;
    s->own = Workspace();
}

// pinned read-back words and events for path depths up to max_depth (host side of a render call)
static int ensure_host_side(rt_scene *s, uint32_t max_depth)
{
    if (max_depth > s->host_depth_cap || !s->h_pinned) {
        if (s->h_pinned) (void)hipHostFree(s->h_pinned);
        s->h_pinned = nullptr;
        HIP_TRY(hipHostMalloc((void **)&s->h_pinned, (3 * ((size_t)max_depth + 2) + 8) * sizeof(uint32_t), hipHostMallocDefault));
        s->host_depth_cap = max_depth;
    }
    while (s->evp.size() < 1 + 4 * (size_t)max_depth) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->evp.push_back(e);
    }
    while (s->evr.size() < (size_t)max_depth + 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->evr.push_back(e);
    }
    if (!s->ev0) {
        HIP_TRY(hipEventCreate(&s->ev0));
        HIP_TRY(hipEventCreate(&s->ev1));
    }
    if (s->walked.size() < (size_t)max_depth + 2) s->walked.resize((size_t)max_depth + 2);
    return RT_OK;
}

static int ensure_workspace(rt_scene *s, size_t cap, uint32_t max_depth)
{
    if (cap > s->own.cap || max_depth > s->own.depth_cap || !s->ws) {
        const size_t want_cap = cap > s->own.cap ? cap : s->own.cap;
        const uint32_t want_depth = max_depth > s->own.depth_cap ? max_depth : s->own.depth_cap;
        free_workspace(s);
        const size_t bytes = carve_workspace(nullptr, want_cap, want_depth, nullptr);
        HIP_TRY(hipMalloc(&s->ws, bytes));
        (void)carve_workspace((char *)s->ws, want_cap, want_depth, &s->own);
    }
    return ensure_host_side(s, max_depth);
}

extern "C" int rt_scene_create(const rt_scene_desc *desc, rt_scene **out)
{
    if (!desc || !out) {
        rt_host::set_last_error("rt_scene_create: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    g_q_skew.store(getenv("RT_Q_SKEW") && atoi(getenv("RT_Q_SKEW")) != 0 ? 1 : 0, std::memory_order_relaxed);
    rt_host::FlatScene flat;
    std::string err;
    int rc = rt_host::flatten_scene(desc, flat, err);
    if (rc != RT_OK) {
        rt_host::set_last_error("rt_scene_create: " + err);
        return rc;
    }
    RT_TRACE("flattened: %zu objects %zu leaves %zu grid words %zu cell refs", flat.objects.size(), flat.leaves.size(),
             flat.grid_bits.size(), flat.cell_tris.size() / 10);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        rt_host::set_last_error("rt_scene_create: no HIP device available (this library has no CPU path)");
        return RT_ERR_DEVICE;
    }
    rt_scene *s = new rt_scene();
    (void)hipGetDevice(&s->device);
    std::memset(&s->dev, 0, sizeof(s->dev));
    auto fail = [&](int code) {
        rt_scene_destroy(s);
        return code;
    };
    // set-up for rt_render_device_noalloc: this device's address of the static scratch array, and the
    // scene's pinned words and events for the deepest path that entry point accepts
    if ((rc = noalloc_setup(s->device))) return fail(rc);
    if ((rc = ensure_host_side(s, RT_NOALLOC_MAX_DEPTH))) return fail(rc);
    {
        // Structure per mesh, chosen by the flattener from how the mesh fills its grid (DevMesh::use_kd).
        // RT_ACCEL=grid|kd overrides it for every mesh (measurements, tests).
        if (const char *e = getenv("RT_ACCEL")) {
            const int force = !strcmp(e, "kd") ? 1 : (!strcmp(e, "grid") ? 0 : -1);
            if (force >= 0) {
                for (DevMesh &m : flat.meshes) m.use_kd = (uint32_t)force;
                for (DevMeshJob &j : flat.jobs) j.use_kd = (uint32_t)force;
            }
        }
        for (const DevMeshJob &j : flat.jobs) (j.use_kd ? s->any_kd : s->any_grid) = true;
        for (const DevMesh &m : flat.meshes)
            RT_TRACE("mesh structure: %s (a triangle shares its grid cell with %.1f on average)", m.use_kd ? "kd-tree" : "grid", (double)m.crowd);
    }
    if ((rc = upload(s, flat.objects, &s->dev.objects))) return fail(rc);
    if ((rc = upload(s, flat.leaves, &s->dev.leaves))) return fail(rc);
    if ((rc = upload(s, flat.prog, &s->dev.prog))) return fail(rc);
    if ((rc = upload(s, flat.meshes, &s->dev.meshes))) return fail(rc);
    if ((rc = upload(s, flat.grid_bits, &s->dev.grid_bits))) return fail(rc);
    if ((rc = upload(s, flat.cell_start, &s->dev.cell_start))) return fail(rc);
    s->dev.n_grid_words = (uint32_t)(flat.grid_bits.size() / 2);
    if ((rc = upload(s, flat.grid_dist, &s->dev.grid_dist))) return fail(rc);
    if ((rc = upload(s, flat.tri_nrm, &s->dev.tri_nrm))) return fail(rc);
    if ((rc = upload(s, flat.grp_box, &s->dev.grp_box))) return fail(rc);
    if ((rc = upload(s, flat.grp_tris, &s->dev.grp_tris))) return fail(rc);
    if ((rc = upload(s, flat.grp_idx, &s->dev.grp_idx))) return fail(rc);
    if ((rc = upload(s, flat.jobs, &s->dev.jobs))) return fail(rc);
    s->dev.n_cell_starts = (uint32_t)flat.cell_start.size();
    s->dev.n_dist_words = (uint32_t)flat.grid_dist.size();
    s->dev.n_jobs = (uint32_t)flat.jobs.size();
    if ((rc = upload(s, flat.kd, &s->dev.kd))) return fail(rc);
    if ((rc = upload(s, flat.leaf_tris, &s->dev.leaf_tris))) return fail(rc);
    s->mesh_kd_lds_bytes = (uint32_t)(std::min<size_t>(flat.jobs.size(), DEV_MESH_JOBS_LDS) * sizeof(DevMeshJob)) +
                           DEV_KD_STACK * RT_MESH_LANES * (uint32_t)sizeof(uint2);
    {
        // dynamic LDS of k_mesh: job records, occupancy words, distance nibbles and - when they still fit the
        // device's per-workgroup limit - the cell starts
        int lds_limit = 0;
        if (hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, s->device) != hipSuccess || lds_limit <= 0)
            lds_limit = 65536;
        const uint32_t base = (uint32_t)(std::min<size_t>(flat.jobs.size(), DEV_MESH_JOBS_LDS) * sizeof(DevMeshJob)) +
                              s->dev.n_grid_words * 8u + s->dev.n_dist_words * 4u;
        const uint32_t starts = s->dev.n_cell_starts * 4u;
        s->dev.starts_in_lds = (s->dev.n_cell_starts <= DEV_GRID_MAX_STARTS && base + starts <= (uint32_t)lds_limit) ? 1u : 0u;
        s->mesh_lds_bytes = base + (s->dev.starts_in_lds ? starts : 0u);
        if (s->mesh_lds_bytes > (uint32_t)lds_limit || s->mesh_kd_lds_bytes > (uint32_t)lds_limit) {
            rt_host::set_last_error("rt_scene_create: the mesh grid tables (" + std::to_string(s->mesh_lds_bytes) +
                                    " bytes) exceed the device's LDS per workgroup (" + std::to_string(lds_limit) + ")");
            return fail(RT_ERR_UNSUPPORTED);
        }
        {
            // k_mesh_one (a single grid job): the same tables and two stacks per wave
            const uint32_t want = ((s->mesh_lds_bytes + 15u) & ~15u) + (RT_POOL_LANES / 64) * RT_POOL_BYTES;
            const char *pool_env = getenv("RT_MESH_POOL"); // read per scene: "0" keeps k_mesh (measurements, tests)
            const bool pool_off = pool_env && !strcmp(pool_env, "0");
            s->mesh_pool_lds_bytes = (flat.jobs.size() == 1 && s->any_grid && want <= (uint32_t)lds_limit / 3u && !pool_off) ? want : 0u; // three workgroups per CU
        }
        RT_TRACE("k_mesh LDS: %u bytes (cell starts %s), with stacks %u, limit %d", s->mesh_lds_bytes,
                 s->dev.starts_in_lds ? "in LDS" : "in memory", s->mesh_pool_lds_bytes, lds_limit);
    }
    if ((rc = upload(s, flat.tris, &s->dev.tris))) return fail(rc);
    if ((rc = upload(s, flat.cell_tris, &s->dev.cell_tris))) return fail(rc);
    if ((rc = upload(s, flat.cell_pairs, &s->dev.cell_pairs))) return fail(rc);
    if ((rc = upload(s, flat.tri_n, &s->dev.tri_n))) return fail(rc);
    if ((rc = upload(s, flat.normals, &s->dev.normals))) return fail(rc);
    if ((rc = upload(s, flat.materials, &s->dev.materials))) return fail(rc);
    if ((rc = upload(s, flat.textures, &s->dev.textures))) return fail(rc);
    if ((rc = upload(s, flat.image_data, &s->dev.image_data))) return fail(rc);
    s->dev.n_objects = (int32_t)flat.objects.size();
    s->dev.n_leaves = (uint32_t)flat.leaves.size();
    s->dev.n_meshes = (uint32_t)flat.meshes.size();
    s->dev.n_materials = (uint32_t)flat.materials.size();
    s->dev.n_textures = (uint32_t)flat.textures.size();
    for (const DevObject &ob : flat.objects) {
        if (ob.kind == DO_CSG || (ob.kind == DO_MEDIUM && ob.prog_len > 0)) s->has_csg = true;
        if (ob.kind == DO_MEDIUM) s->has_medium = true;
        if (ob.kind == DO_MESH) s->has_mesh = true;
    }
    for (const DevLeaf &L : flat.leaves)
        if (L.kind == DL_TORUS) s->has_torus = true;
    for (const DevMesh &m : flat.meshes) s->n_mesh_tris += m.n_tris;
    s->dev.lds_list_cap = s->dev.lds_list_n = 0u;
    if (s->has_csg || s->has_medium) {
        // Interval lists in LDS (LdsLists, rt_dev_shapes.h) when the lists of two workgroups and their scene tables fit
        // a CU: `cap` intervals per list and `lists` lists per ray are the flattener's bounds over all the scene's
        // trees (a program needs one list per operand alive and one for a result; a fog boundary that is a single
        // primitive needs one).  12 bytes per end point and lane.
        int lds_limit = 0;
        if (hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, s->device) != hipSuccess || lds_limit <= 0)
            lds_limit = 65536;
        const uint32_t cap = (uint32_t)std::max(flat.max_intervals, 1), lists = s->has_csg ? (uint32_t)flat.max_stack + 1u : 1u;
        const uint64_t want = (uint64_t)((scene_lds_bytes(s->dev) + 15u) & ~15u) + (uint64_t)lists * 2u * cap * RT_ISECT_BLOCK * 12u;
        const char *e = getenv("RT_LDS_LISTS"); // "0": the lists stay in scratch memory (measurements, tests)
        const bool off = e && !strcmp(e, "0");
        // (the kernel's static LDS - the append's words - and the allocation granularity take a little more: 1 KB each)
        s->lds_limit = lds_limit;
        if (!off && scene_lds_bytes(s->dev) != 0u && cap <= DEV_LDS_LIST_MAX_INTERVALS && lists <= DEV_LDS_LIST_MAX_LISTS &&
            want + 1024u <= (uint64_t)lds_limit * RT_ISECT_BLOCK / (RT_LL_WAVES * 256u)) { // RT_LL_WAVES x 256 / RT_ISECT_BLOCK workgroups per CU
            s->isect_ll_bytes = (uint32_t)want;
            s->dev.lds_list_cap = cap;
            s->dev.lds_list_n = lists;
        }
        RT_TRACE("interval lists: at most %d intervals, %d lists at once -> %s (%u bytes of LDS per k_intersect workgroup)",
                 flat.max_intervals, flat.max_stack, s->isect_ll_bytes ? "in LDS" : "in scratch memory", s->isect_ll_bytes);
    }
    for (int k = 0; k < 3; ++k) s->dev.background[k] = flat.background[k];
    s->dev.cam_fov_factor = flat.cam_fov_factor;
    std::memcpy(s->dev.cam_c2w, flat.cam_c2w, sizeof(flat.cam_c2w));
    {
        // k_tail_grid (one lane per path, the grid walk inside): the same single grid job as k_mesh_one, with the
        // scene tables in LDS beside the grid tables
        int lds_limit = 0;
        if (hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, s->device) != hipSuccess || lds_limit <= 0)
            lds_limit = 65536;
        const uint32_t want = ((scene_lds_bytes(s->dev) + 15u) & ~15u) + s->mesh_lds_bytes;
        const char *tg_env = getenv("RT_TAIL_GRID"); // "0": k_tail takes every tail (measurements, tests)
        s->tail_grid_lds_bytes = (s->mesh_pool_lds_bytes != 0u && scene_lds_bytes(s->dev) != 0u && want <= (uint32_t)lds_limit &&
                                  !(tg_env && !strcmp(tg_env, "0"))) ? want : 0u;
        RT_TRACE("k_tail_grid LDS: %u bytes", s->tail_grid_lds_bytes);
    }
    *out = s;
    RT_TRACE("scene created");
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    delete s->worker; // joins the thread
    if (s->multi_stream) (void)hipStreamDestroy(s->multi_stream);
    if (s->multi_done) (void)hipEventDestroy(s->multi_done);
    for (void *p : s->allocs) (void)hipFree(p);
    free_workspace(s);
    if (s->h_pinned) (void)hipHostFree(s->h_pinned);
    if (s->out_rgb) (void)hipFree(s->out_rgb);
    if (s->out_ids) (void)hipFree(s->out_ids);
    if (s->snap_d) (void)hipFree(s->snap_d);
    if (s->snap_h) (void)hipHostFree(s->snap_h);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->evp) (void)hipEventDestroy(e);
    for (hipEvent_t e : s->evr) (void)hipEventDestroy(e);
    delete s;
}

static uint32_t grid_for(uint64_t n)
{
    uint64_t b = (n + RT_BLOCK - 1) / RT_BLOCK;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8; // 256 CUs x 8 blocks, grid-stride beyond that
    return (uint32_t)b;
}

static uint32_t grid_for_wide(uint64_t n)
{
    uint64_t b = (n + RT_BLOCK_WIDE - 1) / RT_BLOCK_WIDE;
    if (b < 1) b = 1;
    // what the chip holds at once of k_shade (162 registers: 3 waves per SIMD), grid-stride beyond: 4K mesh.scene, k_shade
    // per frame alone 0.273 ms against 0.289 with 2048 workgroups (RT_SHADE_WGS: tuning knob, not API)
    static const uint32_t shade_wgs = getenv("RT_SHADE_WGS") ? (uint32_t)atoi(getenv("RT_SHADE_WGS")) : 256u * 3u * (256u / RT_BLOCK_WIDE);
    if (b > shade_wgs) b = shade_wgs;
    return (uint32_t)b;
}

static int check_options(const rt_options *o)
{
    if (!o || o->width == 0 || o->height == 0 || o->samples == 0) {
        rt_host::set_last_error("rt_options: width, height and samples must be positive");
        return RT_ERR_INVALID;
    }
    uint32_t sc = o->shard_count ? o->shard_count : 1;
    if (o->shard_rank >= sc) {
        rt_host::set_last_error("rt_options: shard_rank out of range");
        return RT_ERR_INVALID;
    }
    if ((o->flags & RT_FLAG_ROW_MAJOR) && sc > 1u) {
        rt_host::set_last_error("rt_options: RT_FLAG_ROW_MAJOR needs shard_count 1");
        return RT_ERR_INVALID;
    }
    if ((uint64_t)o->width * o->height > 0xFFFFFFFFull || o->max_depth > 0xFFFFu) {
        rt_host::set_last_error("rt_options: image or depth too large for the random-number key");
        return RT_ERR_INVALID;
    }
    // the key holds 16 bits of stratum (the reference's sample count is a u16, main.rs:66-69): with more,
    // stratum 65536 + k would draw the numbers of stratum k
    if (o->samples >= 65536u * 2u || (uint64_t)std::sqrt((double)o->samples) * (uint64_t)std::sqrt((double)o->samples) > 65536ull) {
        rt_host::set_last_error("rt_options: more than 65536 strata do not fit the random-number key");
        return RT_ERR_INVALID;
    }
    return RT_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that wants more than the default 64 KB: as much as the device
// allows beside the kernel's own static LDS (not the asking scene's need: another scene may launch the same instantiation
// with more).  Some runtimes of this pool refuse limit itself when the kernel has static LDS as well, some the rest too:
// then what has been asked for so far, whichever scene asked (the attribute belongs to the function, not to a scene).
static bool allow_large_lds(const void *fn, int lds_limit, uint32_t need)
{
    static std::mutex m;
    static std::vector<std::pair<const void *, int>> granted; // per function: the largest size set so far
    std::lock_guard<std::mutex> lock(m);
    int have = 0;
    for (auto &g : granted)
        if (g.first == fn) have = g.second;
    if (have >= (int)need) return true;
    hipFuncAttributes fa;
    int stat = 0;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess)
        stat = (int)fa.sharedSizeBytes;
    else
        (void)hipGetLastError();
    const int tries[3] = {lds_limit - stat, lds_limit - stat - 1024, (int)need};
    for (int want : tries) {
        if (want < (int)need) continue;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess) {
            RT_TRACE("dynamic LDS of a kernel with %d static bytes: %d allowed (device limit %d, needed %u)", stat, want, lds_limit, need);
            bool found = false;
            for (auto &g : granted)
                if (g.first == fn) {
                    g.second = want;
                    found = true;
                }
            if (!found) granted.push_back({fn, want});
            return true;
        }
        (void)hipGetLastError();
    }
    return false;
}

#define RT_BRUTE_MAX_TESTS 40000000ull // rays x triangles below which the brute-force wave kernel wins (measured)

// One depth level for the rays of `q` (the shard's `slots_ptr[0]` slots, or the entries `q_regions` counts): k_intersect (nearest analytic hit; rays with no mesh to
// visit are shaded there and then), then for the mesh records it queued k_mesh | k_mesh_brute (the triangles)
// and k_shade.  Survivors of both go to `qout` / `count_out`, finished paths to `fb`.
// fused == false (rt_trace_rays): no shading; every ray's hit record ends up in `hit_all`.
// Returns 1 when the grid / kd walk (k_mesh) ran, 2 when k_mesh_one did, 0 otherwise.  Stage events (if any): ev_mid1 after k_intersect,
// ev_mid2 after the mesh kernel.
static int launch_intersect(rt_scene *s, const Workspace &w, hipStream_t st, uint32_t live, const RayQueue &q,
                            const uint32_t *slots_ptr, const QRegions &q_regions, const QRegions &mesh_regions,
                            const PassParams &pp, int32_t *ids, bool fused, const RayQueue &qout,
                            const QRegions &out_regions, double *fb, const HitBuf &hit_all, hipEvent_t ev_mid1, hipEvent_t ev_mid2)
{
    uint64_t gb = ((uint64_t)live + RT_ISECT_BLOCK - 1) / RT_ISECT_BLOCK;
    if (gb < 1) gb = 1;
    // As many workgroups as the chip holds at once (256 CUs x the waves per SIMD the variant is compiled for: a
    // workgroup is one wave on each of a CU's four SIMDs), grid-stride beyond that.  With 2048 workgroups of the plain
    // variant (768 resident) the last third of the launch ran on two thirds of the chip: 4K mesh.scene, k_intersect per
    // frame alone 0.970 ms at 2048 or 2304, 1.006 at 3072, 0.925 at 1536, 0.886 at 768.  RT_ISECT_WGS: tuning knob, not API.
    static const uint32_t isect_wgs = getenv("RT_ISECT_WGS") ? (uint32_t)atoi(getenv("RT_ISECT_WGS")) : 0u;
    dim3 grid((uint32_t)gb), block(RT_ISECT_BLOCK);
    static_assert(RT_ISECT_BLOCK == 128 || RT_ISECT_BLOCK == 256 || RT_ISECT_BLOCK == 512, "workgroups of k_intersect");
    // the scene's one grid job goes to k_mesh_one, which leaves its triangle beside the record for k_shade to merge
    // (only long queues: short ones go to k_mesh_brute, which writes the hit record itself)
    const bool one = s->has_mesh && s->any_grid && s->mesh_pool_lds_bytes != 0u &&
                     (uint64_t)live * s->n_mesh_tris > (uint64_t)RT_BRUTE_MAX_TESTS;
#define LAUNCH_ISP(C, M, T, L, F, LL, PQ)                                                                              \
    do {                                                                                                               \
        grid.x = std::min<uint32_t>((uint32_t)gb, isect_wgs ? isect_wgs : 256u * RT_ISECT_WAVES(C, M, T, LL, PQ) * 256u / RT_ISECT_BLOCK); \
        hipLaunchKernelGGL((k_intersect<C, M, T, L, F, LL, PQ>), grid, block, LL ? s->isect_ll_bytes : scene_lds_bytes(s->dev), st, s->dev, q, \
                           slots_ptr, q_regions, pp, ids, w.d_guards, w.mq, w.mqh, mesh_regions, qout, out_regions, fb, hit_all, w.d_work, \
                           one ? w.mres_tri : nullptr);                                                                \
    } while (0)
// the wavefront passes (fused) know which kind of queue they read: the kernel without the other kind's code
#define LAUNCH_IS(C, M, T, L, F, LL)                                                                                   \
    do {                                                                                                               \
        if (!F)                                                                                                        \
            LAUNCH_ISP(C, M, T, L, F, LL, (F ? 1 : 0));                                                                \
        else if (q.primary)                                                                                            \
            LAUNCH_ISP(C, M, T, L, F, LL, (F ? 1 : 0));                                                                \
        else                                                                                                           \
            LAUNCH_ISP(C, M, T, L, F, LL, (F ? 2 : 0));                                                                \
    } while (0)
// the variants that hold interval lists (CSG programs, fog boundaries) exist a second time with the lists in LDS, for
// the wavefront passes (fused) of scenes whose lists fit there
#define LAUNCH_I(C, M, T, L)                                                                                           \
    do {                                                                                                               \
        constexpr bool lists_ = (C || M) && L;                                                                         \
        if (fused) {                                                                                                   \
            if (lists_ && s->isect_ll_bytes && !s->ll_attr_set) {                                                      \
                const bool ok1_ = allow_large_lds(reinterpret_cast<const void *>(&k_intersect<C, M, T, L, true, lists_, 1>), s->lds_limit, s->isect_ll_bytes); \
                const bool ok2_ = allow_large_lds(reinterpret_cast<const void *>(&k_intersect<C, M, T, L, true, lists_, 2>), s->lds_limit, s->isect_ll_bytes); \
                if (!ok1_ || !ok2_) {                                                                                  \
                    RT_TRACE("hipFuncSetAttribute refused %u bytes of LDS: interval lists in scratch memory", s->isect_ll_bytes); \
                    s->isect_ll_bytes = 0u;                                                                            \
                    s->dev.lds_list_cap = s->dev.lds_list_n = 0u;                                                      \
                }                                                                                                      \
                s->ll_attr_set = true;                                                                                 \
            }                                                                                                          \
            if (lists_ && s->isect_ll_bytes)                                                                           \
                LAUNCH_IS(C, M, T, L, true, lists_);                                                                   \
            else                                                                                                       \
                LAUNCH_IS(C, M, T, L, true, false);                                                                    \
        } else {                                                                                                       \
            LAUNCH_IS(C, M, T, L, false, false);                                                                       \
        }                                                                                                              \
    } while (0)
    // scenes whose tables do not fit LDS (scene_lds_bytes == 0) run the one variant that reads them from memory,
    // which is the most general one
    const int variant = scene_lds_bytes(s->dev) == 0u ? 8 : (s->has_csg ? 4 : 0) | (s->has_medium ? 2 : 0) | (s->has_torus ? 1 : 0);
    switch (variant) {
    case 0: LAUNCH_I(false, false, false, true); break;
    case 1: LAUNCH_I(false, false, true, true); break;
    case 2: LAUNCH_I(false, true, false, true); break;
    case 3: LAUNCH_I(false, true, true, true); break;
    case 4: LAUNCH_I(true, false, false, true); break;
    case 5: LAUNCH_I(true, false, true, true); break;
    case 6: LAUNCH_I(true, true, false, true); break;
    case 7: LAUNCH_I(true, true, true, true); break;
    default: LAUNCH_I(true, true, true, false); break;
    }
#undef LAUNCH_I
#undef LAUNCH_IS
    if (ev_mid1) (void)hipEventRecord(ev_mid1, st);
    int walked = 0;
    if (s->has_mesh) {
        // short queues: one wave per ray over all triangles; long queues: one lane per ray through the mesh grid
        if ((uint64_t)live * s->n_mesh_tris <= (uint64_t)RT_BRUTE_MAX_TESTS) {
            uint64_t blocks = ((uint64_t)live * 64 + RT_BLOCK - 1) / RT_BLOCK;
            if (blocks > 256 * 8) blocks = 256 * 8;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(k_mesh_brute, dim3((uint32_t)blocks), dim3(RT_BLOCK), 0, st, s->dev, w.mq, mesh_regions, w.mqh, ids, pp);
        } else {
            // as many waves as the chip holds: wave w starts on block w of 64 records and takes further blocks from
            // the cursors (d_work) until the queue is empty; short queues: fewer lanes per wave (block < 64)
            static const uint32_t refill = getenv("RT_MESH_REFILL") ? (uint32_t)atoi(getenv("RT_MESH_REFILL")) : 16u;
            static const uint32_t max_wgs = getenv("RT_MESH_WGS") ? (uint32_t)atoi(getenv("RT_MESH_WGS")) : 256u * RT_MESH_WAVES; // = the waves the chip holds at once (they take blocks until the queue is empty); 4K mesh.scene: k_mesh 1.53 ms with 256 x 3, 1.58 with 256 x 6 and 256 x 12
            static const uint32_t block_env = getenv("RT_MESH_BLOCK") ? (uint32_t)atoi(getenv("RT_MESH_BLOCK")) : 0u;
            const uint64_t per_wg = RT_MESH_MIN_BLOCK * (RT_MESH_LANES / 64); // the kernel uses 4 to 64 lanes per wave
            uint64_t mb = ((uint64_t)live + per_wg - 1) / per_wg;
            if (mb < 1) mb = 1;
            if (mb > max_wgs) mb = max_wgs;
            const bool jobs_lds = s->dev.n_jobs <= DEV_MESH_JOBS_LDS;
#define LAUNCH_M(KD, JL, LDS_BYTES)                                                                                    \
    hipLaunchKernelGGL((k_mesh<KD, JL>), dim3((uint32_t)mb), dim3(RT_MESH_LANES), LDS_BYTES, st, s->dev, w.mq,        \
                       mesh_regions, w.d_work + (KD ? RT_MESH_CURSORS * 16 : 0), w.mqh, ids, refill, block_env, 256u * 4u * RT_MESH_WAVES, pp)
            // one launch per structure in use: each instantiation walks its own meshes and passes the others by
            // (the record carries the nearest hit from one launch to the next)
            if (one) {
                // the scene's one job: k_mesh_one (as many workgroups as the chip holds; they take blocks until the
                // queue is empty)
                const uint64_t per_pool_wg = RT_MESH_MIN_BLOCK * (RT_POOL_LANES / 64);
                uint64_t pb = ((uint64_t)live + per_pool_wg - 1) / per_pool_wg;
                if (pb < 1) pb = 1;
                if (pb > max_wgs) pb = max_wgs;
                hipLaunchKernelGGL(k_mesh_one, dim3((uint32_t)pb), dim3(RT_POOL_LANES), s->mesh_pool_lds_bytes, st, s->dev, w.mq,
                                   mesh_regions, w.d_work, w.mqh, w.mres_t, w.mres_tri, block_env, 256u * 4u * RT_MESH_WAVES, pp);
                walked = 2;
            } else if (s->any_grid) {
                if (jobs_lds)
                    LAUNCH_M(false, true, s->mesh_lds_bytes);
                else
                    LAUNCH_M(false, false, s->mesh_lds_bytes);
            }
            if (s->any_kd) {
                if (jobs_lds)
                    LAUNCH_M(true, true, s->mesh_kd_lds_bytes);
                else
                    LAUNCH_M(true, false, s->mesh_kd_lds_bytes);
            }
#undef LAUNCH_M
            if (!walked) walked = 1;
        }
    }
    if (ev_mid2) (void)hipEventRecord(ev_mid2, st);
    if (s->has_mesh) {
        if (fused) {
            // the mesh records: hit_color + scatter, survivors appended behind k_intersect's
            if (scene_lds_bytes(s->dev) != 0u)
                hipLaunchKernelGGL(k_shade<true>, dim3(grid_for_wide(live)), dim3(RT_BLOCK_WIDE), scene_lds_bytes(s->dev), st,
                                   s->dev, w.mq, mesh_regions, w.mqh, qout, out_regions, fb, pp, one ? w.mres_t : nullptr,
                                   one ? w.mres_tri : nullptr, ids);
            else
                hipLaunchKernelGGL(k_shade<false>, dim3(grid_for_wide(live)), dim3(RT_BLOCK_WIDE), 0, st, s->dev, w.mq,
                                   mesh_regions, w.mqh, qout, out_regions, fb, pp, one ? w.mres_t : nullptr, one ? w.mres_tri : nullptr, ids);
        } else {
            hipLaunchKernelGGL(k_copy_mesh_hits, dim3((live + 255) / 256), dim3(256), 0, st, s->dev, w.mq, w.mqh, mesh_regions, hit_all,
                               one ? w.mres_t : nullptr, one ? w.mres_tri : nullptr);
        }
    }
    return walked;
}

#ifndef RT_TAIL_PATHS
#define RT_TAIL_PATHS 140000u // k_tail_grid takes over when at most this many paths are left: 256 CUs x 4 SIMDs x 2 waves x 64 lanes, and a little
#endif
#define RT_TAIL_MAX_TESTS 36000000ull // paths x triangles below which k_tail takes the rest of the stratum (measured)

static void launch_tail(rt_scene *s, const Workspace &w, hipStream_t st, uint32_t live, const RayQueue &q, const PassParams &pp, double *d_rgb)
{
    const int variant = scene_lds_bytes(s->dev) == 0u ? 8 : (s->has_csg ? 4 : 0) | (s->has_medium ? 2 : 0) | (s->has_torus ? 1 : 0);
    uint32_t *lane_totals = w.d_counts + 2 * (pp.max_depth + 2);
    if (s->tail_grid_lds_bytes != 0u && variant != 8) {
        // one lane per path, the scene's one mesh through its grid: a workgroup per 256 paths (they are not all
        // resident at once when many paths are left; the ones that wait start as the short paths' waves finish)
        // ... but at least the waves the chip holds at once (256 CUs x 4 SIMDs x RT_TAILG_WAVES) when fewer paths are
        // left than fill them: the kernel then gives each wave 8 to 32 paths instead of 64
        const uint64_t resident = 256u * RT_TAILG_WAVES; // workgroups of 4 waves
        uint64_t blocks = ((uint64_t)live + RT_TAILG_LANES - 1) / RT_TAILG_LANES;
        if (blocks < resident) blocks = std::min<uint64_t>(resident, ((uint64_t)live + (RT_TAILG_LANES / 64) * RT_TAILG_MIN_LANES - 1) / ((RT_TAILG_LANES / 64) * RT_TAILG_MIN_LANES));
        if (blocks > 65535u * 16u) blocks = 65535u * 16u;
        if (blocks < 1) blocks = 1;
#define LAUNCH_TG(C, M, T)                                                                                             \
    hipLaunchKernelGGL((k_tail_grid<C, M, T>), dim3((uint32_t)blocks), dim3(RT_TAILG_LANES), s->tail_grid_lds_bytes, st, s->dev, q, \
                       queue_regions(w, 0, pp.depth), d_rgb, pp, w.d_guards, lane_totals)
        switch (variant) {
        case 0: LAUNCH_TG(false, false, false); break;
        case 1: LAUNCH_TG(false, false, true); break;
        case 2: LAUNCH_TG(false, true, false); break;
        case 3: LAUNCH_TG(false, true, true); break;
        case 4: LAUNCH_TG(true, false, false); break;
        case 5: LAUNCH_TG(true, false, true); break;
        case 6: LAUNCH_TG(true, true, false); break;
        default: LAUNCH_TG(true, true, true); break;
        }
#undef LAUNCH_TG
        return;
    }
    uint64_t blocks = ((uint64_t)live * (s->has_mesh ? 64 : 1) + RT_BLOCK - 1) / RT_BLOCK; // wave or lane per path
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
#define LAUNCH_TS(C, M, T, L, LL)                                                                                      \
    hipLaunchKernelGGL((k_tail<C, M, T, L, LL>), dim3((uint32_t)blocks), dim3(RT_BLOCK), LL ? s->isect_ll_bytes : scene_lds_bytes(s->dev), \
                       st, s->dev, q, queue_regions(w, 0, pp.depth), w.d_counts, d_rgb, pp, w.d_guards, s->has_mesh ? 1 : 0, lane_totals)
// (the variants with interval lists a second time with the lists in LDS, as k_intersect's)
#define LAUNCH_T(C, M, T, L)                                                                                           \
    do {                                                                                                               \
        constexpr bool lists_ = (C || M) && L && RT_BLOCK == RT_ISECT_BLOCK; /* one layout of the lists for both kernels */ \
        if (lists_ && s->isect_ll_bytes && !s->ll_tail_attr_set) {                                                     \
            if (!allow_large_lds(reinterpret_cast<const void *>(&k_tail<C, M, T, L, lists_>), s->lds_limit, s->isect_ll_bytes)) \
                s->ll_tail_ok = false;                                                                                 \
            s->ll_tail_attr_set = true;                                                                                \
        }                                                                                                              \
        if (lists_ && s->isect_ll_bytes && s->ll_tail_ok)                                                              \
            LAUNCH_TS(C, M, T, L, lists_);                                                                             \
        else                                                                                                           \
            LAUNCH_TS(C, M, T, L, false);                                                                              \
    } while (0)
    switch (variant) {
    case 0: LAUNCH_T(false, false, false, true); break;
    case 1: LAUNCH_T(false, false, true, true); break;
    case 2: LAUNCH_T(false, true, false, true); break;
    case 3: LAUNCH_T(false, true, true, true); break;
    case 4: LAUNCH_T(true, false, false, true); break;
    case 5: LAUNCH_T(true, false, true, true); break;
    case 6: LAUNCH_T(true, true, false, true); break;
    case 7: LAUNCH_T(true, true, true, true); break;
    default: LAUNCH_T(true, true, true, false); break; // tables in memory: the general variant
    }
#undef LAUNCH_T
#undef LAUNCH_TS
}

typedef int (*stratum_fn)(void *user, uint32_t done, uint32_t total);

// the wavefront loop for one shard; d_rgb (sum) and d_ids are in shard layout
// per-kernel events, recorded only when the caller asked for stage times (RT_FLAG_STAGE_TIMES)
#define STAGE_EV(CALL) do { if (stage_times) HIP_TRY(CALL); } while (0)
// `fixed`: a workspace the caller provides (rt_render_device_noalloc); nothing is allocated then, and sizes
// beyond what it and the scene's host side were set up for are refused.
static int render_strata(rt_scene *s, const rt_options *opts, double *d_rgb, int32_t *d_ids, rt_stats *stats,
                         hipStream_t st, stratum_fn per_stratum, void *user, const Workspace *fixed = nullptr,
                         bool call_back_for_empty_shards = false)
{
    int rc = check_options(opts);
    if (rc) return rc;
    if (!s || !d_rgb) {
        rt_host::set_last_error("rt_render: null scene or output");
        return RT_ERR_INVALID;
    }
    RT_TRACE("render_strata: device %d", s->device);
    HIP_TRY(hipSetDevice(s->device));
    rt_shard_geom g = rt_shard_geom_make(opts->width, opts->height, opts->tile_size, opts->shard_count);
    const uint32_t rank = opts->shard_rank;
    const uint64_t n_slots64 = (uint64_t)rt_shard_tiles(&g, rank) * g.tile_px;
    if (n_slots64 > 0xFFFFFFF0ull) {
        rt_host::set_last_error("rt_render: shard too large");
        return RT_ERR_INVALID;
    }
    const uint32_t n_slots = (uint32_t)n_slots64;
    RT_TRACE("slots %u", n_slots);
    if (fixed) {
        if (n_slots > fixed->cap || opts->max_depth > fixed->depth_cap || opts->max_depth > s->host_depth_cap) {
            rt_host::set_last_error("rt_render_device_noalloc: " + std::to_string(n_slots) + " ray slots, max_depth " +
                                    std::to_string(opts->max_depth) + " exceed the static scratch buffer (" +
                                    std::to_string(fixed->cap) + " slots, max_depth " +
                                    std::to_string(std::min(fixed->depth_cap, s->host_depth_cap)) + ")");
            return RT_ERR_UNSUPPORTED;
        }
    } else {
        rc = ensure_workspace(s, n_slots ? n_slots : 1, opts->max_depth);
        if (rc) return rc;
    }
    const Workspace &w = fixed ? *fixed : s->own;
    RT_TRACE("workspace ok");

    const uint32_t sqrt_spp = (uint32_t)std::sqrt((double)opts->samples); // system.rs:276
    const double recip = 1.0 / std::sqrt((double)opts->samples);          // system.rs:277
    const uint32_t strata = sqrt_spp * sqrt_spp;

    rt_stats local;
    std::memset(&local, 0, sizeof(local));
    local.strata = strata;

    RT_TRACE("ev0 %p st %p d_rgb %p", (void *)s->ev0, (void *)st, (void *)d_rgb);
    HIP_TRY(hipEventRecord(s->ev0, st));
    RT_TRACE("event recorded");
    const bool row_major = (opts->flags & RT_FLAG_ROW_MAJOR) != 0u; // check_options: only with a single shard
    const bool stage_times = (opts->flags & RT_FLAG_STAGE_TIMES) != 0u;
    const size_t n_out = row_major ? (size_t)opts->width * opts->height : (size_t)n_slots;
    HIP_TRY(hipMemsetAsync(d_rgb, 0, n_out * 3 * sizeof(double), st)); // alloc_render_buf
    if (d_ids) HIP_TRY(hipMemsetAsync(d_ids, 0xFF, n_out * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(w.d_guards, 0, 8 * sizeof(uint32_t), st));
    RT_TRACE("memsets queued");

    uint32_t done = 0;
    for (uint32_t s_i = 0; s_i < sqrt_spp && n_slots; ++s_i) {
        for (uint32_t s_j = 0; s_j < sqrt_spp; ++s_j) {
            PassParams pp;
            pp.seed = opts->seed;
            pp.stratum = s_i * sqrt_spp + s_j;
            pp.depth = 0;
            pp.max_depth = opts->max_depth;
            pp.force_uv = 0;
            pp.bias = opts->bias;
            if (opts->max_depth == 0) {
                hipLaunchKernelGGL(k_fill_background, dim3((n_slots + 255) / 256), dim3(256), 0, st, s->dev, g, rank,
                                   n_slots, d_rgb, row_major ? 1 : 0);
            } else {
                HIP_TRY(hipMemsetAsync(w.d_counts, 0, (2 * (opts->max_depth + 2) + 2) * sizeof(uint32_t), st));
                HIP_TRY(hipMemsetAsync(w.d_rcnt, 0, 2 * (size_t)RT_Q_REGIONS * w.rstride * sizeof(uint32_t), st));
                // depth 0 reads no ray queue: entry i is slot i and its camera ray is regenerated in place
                RayQueue q_primary = w.q[0];
                q_primary.primary = 1;
                q_primary.row_major = row_major ? 1 : 0;
                q_primary.rank = rank;
                q_primary.s_i = s_i;
                q_primary.s_j = s_j;
                q_primary.g = g;
                q_primary.recip_sqrt_spp = recip;
                HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)w.d_counts, (int)n_slots, 1, st));
                STAGE_EV(hipEventRecord(s->evp[0], st));
                // `live` bounds the rays of the next pass from above: the newest count that has come
                // back (counts never grow).  The kernels read the true count on the device; the bound
                // only sizes grids and picks between k_mesh and k_mesh_brute.
                uint32_t live = n_slots;
                // Paths left at which one launch takes the rest of the stratum.  k_tail_grid (one lane per path, the
                // mesh through its grid): as many as the chip holds at once at two waves per SIMD (RT_TAIL_PATHS;
                // passes that small have a floor of ~0.1 ms each whatever they hold), but not before the paths have
                // thinned to a sixth of the shard's slots.  Measured, 4K bench frame / its shard 0 of 8, one frame at
                // a time: take-over at 600 k paths 3.20 ms, at 131 k 2.67, at 65 k 2.75; shard at 75 k 0.87, at 16 k
                // 0.92.  k_tail: one wave per path with the triangles by brute force, or one lane per path in a scene
                // without meshes.
                const uint32_t tail_env = getenv("RT_TAIL_MAX") ? (uint32_t)atoi(getenv("RT_TAIL_MAX")) : 0xFFFFFFFFu;
                const uint32_t tail_paths = getenv("RT_TAIL_PATHS") ? (uint32_t)std::max(1, atoi(getenv("RT_TAIL_PATHS"))) : RT_TAIL_PATHS;
                uint32_t tail_max = s->has_mesh ? (uint32_t)(RT_TAIL_MAX_TESTS / (s->n_mesh_tris ? s->n_mesh_tris : 1u)) : 16384u;
                if (tail_max > 16384u) tail_max = 16384u;
                if (s->tail_grid_lds_bytes != 0u) tail_max = std::max(tail_max, std::min(n_slots / 6u, tail_paths));
                if (!s->has_mesh) {
                    // one lane per path and nothing shared between lanes: the launch lasts as long as its deepest path
                    // whether it holds 16 k paths or as many as the chip holds at once (the variants with CSG, fog or
                    // torus code are compiled for RT_TAIL_HEAVY_WAVES waves per SIMD: 65 k lanes each), while every pass it
                    // replaces has its floor
                    const uint32_t resident = (s->has_csg || s->has_medium || s->has_torus) ? 65536u * RT_TAIL_HEAVY_WAVES : tail_paths;
                    tail_max = std::max(tail_max, std::min(n_slots / 6u, std::min(resident, tail_paths)));
                }
                if (tail_env != 0xFFFFFFFFu) tail_max = tail_env;
                const uint32_t run_ahead = getenv("RT_RUN_AHEAD") ? (uint32_t)atoi(getenv("RT_RUN_AHEAD")) : 2u;
                // What the last stratum of this shape did: its passes are queued again without a single wait
                // (RT_PASS_HINT=0: always follow the counts).
                const bool hint_env = !(getenv("RT_PASS_HINT") && !strcmp(getenv("RT_PASS_HINT"), "0"));
                const uint64_t hint_key[2] = {((uint64_t)opts->width << 32) | opts->height,
                                              ((uint64_t)g.tile << 48) | ((uint64_t)(rank & 0xFFFFu) << 32) | ((uint64_t)(g.shard_count & 0xFFFFu) << 16) | (opts->max_depth & 0xFFFFu)};
                const bool hinted = hint_env && run_ahead != 0u && !s->pass_hint_counts.empty() && s->pass_hint_key[0] == hint_key[0] &&
                                    s->pass_hint_key[1] == hint_key[1];
                // bound on the rays entering depth k, from the hint (a margin: seeds differ from frame to frame)
                auto hint_bound = [&](uint32_t k) {
                    const uint64_t c = k < s->pass_hint_counts.size() ? s->pass_hint_counts[k] : 0u;
                    const uint64_t b = c + c / 8u + 1024u;
                    return (uint32_t)std::min<uint64_t>(b, n_slots);
                };
                int cur = 0;
                uint32_t passes = 0, totalled = 0; // passes launched; passes whose queue lengths k_queue_totals has summed
                auto totals = [&]() {
                    if (totalled < passes)
                        hipLaunchKernelGGL(k_queue_totals, dim3(1), dim3(64), 0, st, queue_regions(w, 0, 0), queue_regions(w, 1, 0),
                                           w.d_counts, w.d_counts + (opts->max_depth + 2), totalled, passes - totalled);
                    totalled = passes;
                };
                int tail_at = -1; // depth at which the tail kernel took over
                uint8_t *walked = s->walked.data(); // sized by ensure_host_side
                std::memset(walked, 0, opts->max_depth + 1);
                for (uint32_t depth = 0; depth < opts->max_depth; ++depth) {
                    pp.depth = depth;
                    int32_t *ids = (depth == 0 && pp.stratum == 0) ? d_ids : nullptr;
                    STAGE_EV(hipEventRecord(s->evp[1 + 4 * depth], st));
                    const RayQueue &q_in = depth == 0 ? q_primary : w.q[cur];
                    walked[depth] = (uint8_t)launch_intersect(s, w, st, live, q_in, w.d_counts, queue_regions(w, 0, depth),
                                                              queue_regions(w, 1, depth), pp, ids, true, w.q[1 - cur],
                                                              queue_regions(w, 0, depth + 1), d_rgb, w.mqh,
                                                              stage_times ? s->evp[2 + 4 * depth] : nullptr,
                                                              stage_times ? s->evp[3 + 4 * depth] : nullptr);
                    STAGE_EV(hipEventRecord(s->evp[4 + 4 * depth], st));
                    ++passes;
                    cur = 1 - cur;
                    // inside the hinted prefix: the next pass (or the tail) is queued at once
                    if (hinted && depth + 1 < opts->max_depth &&
                        ((int)depth + 1 == s->pass_hint_tail_at || depth + 1 < (uint32_t)s->pass_hint_counts.size())) {
                        live = hint_bound(depth + 1);
                        if ((int)depth + 1 == s->pass_hint_tail_at) {
                            pp.depth = depth + 1;
                            STAGE_EV(hipEventRecord(s->evp[1 + 4 * (depth + 1)], st));
                            launch_tail(s, w, st, live, w.q[cur], pp, d_rgb);
                            STAGE_EV(hipEventRecord(s->evp[2 + 4 * (depth + 1)], st));
                            tail_at = (int)depth + 1;
                            break;
                        }
                        continue;
                    }
                    // the lengths of the queues the passes so far appended to, as one word each (what the host reads)
                    totals();
                    // queue length of the next pass: copied back asynchronously after every pass.  The
                    // host waits for the copy of two passes ago only, so the device always has work
                    // queued across the copy's round trip; near the tail threshold (and on the last
                    // pass) it waits for the newest copy, because the switch needs the count.
                    uint32_t *h_live = s->h_pinned + 2 * (opts->max_depth + 2) + 8;
                    HIP_TRY(hipMemcpyAsync(h_live + depth + 1, w.d_counts + depth + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipEventRecord(s->evr[depth + 1], st));
                    const bool exact = live <= 8u * tail_max || depth + 1 == opts->max_depth || run_ahead == 0u || hinted;
                    int known = exact ? (int)depth + 1 : (int)depth + 1 - (int)run_ahead;
                    if (known >= 1) HIP_TRY(hipEventSynchronize(s->evr[known]));
                    for (int k = (int)depth + 1; k > known && k >= 1; --k)
                        if (hipEventQuery(s->evr[k]) == hipSuccess) {
                            known = k;
                            break;
                        }
                    if (known >= 1 && h_live[known] < live) live = h_live[known];
                    RT_TRACE("depth %u: count of depth %d known, bound %u", depth, known, live);
                    if (live == 0) break;
                    if (live <= tail_max && depth + 1 < opts->max_depth) {
                        // few paths left: one launch takes them all to their ends
                        pp.depth = depth + 1;
                        STAGE_EV(hipEventRecord(s->evp[1 + 4 * (depth + 1)], st));
                        launch_tail(s, w, st, live, w.q[cur], pp, d_rgb);
                        STAGE_EV(hipEventRecord(s->evp[2 + 4 * (depth + 1)], st));
                        tail_at = (int)depth + 1;
                        break;
                    }
                }
                totals(); // (the passes queued on the hint: all of them in one launch)
                HIP_TRY(hipMemcpyAsync(s->h_pinned, w.d_counts, (2 * (opts->max_depth + 2) + 2) * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, st)); // rays per depth, mesh-queue lengths per depth, lane totals
                HIP_TRY(hipStreamSynchronize(st));
                const uint32_t *mesh_counts = s->h_pinned + (opts->max_depth + 2);
                {
                    // the hint for the next stratum of this shape: the counts as far as the passes went, and the
                    // first level whose count is under the tail's threshold (one level later than this time when
                    // the tail took over above it: the levels below were not counted)
                    const uint32_t known_levels = tail_at >= 0 ? (uint32_t)tail_at + 1u : std::min(passes + 1u, opts->max_depth);
                    s->pass_hint_counts.assign(s->h_pinned, s->h_pinned + known_levels);
                    s->pass_hint_counts[0] = n_slots;
                    int next_tail = -1;
                    for (uint32_t k = 1; k < known_levels && k < opts->max_depth; ++k)
                        if (s->pass_hint_counts[k] <= tail_max) {
                            next_tail = s->pass_hint_counts[k] ? (int)k : -1; // nothing left there: the passes simply end
                            if (!s->pass_hint_counts[k]) s->pass_hint_counts.resize(k);
                            break;
                        }
                    if (next_tail < 0 && tail_at >= 0 && (uint32_t)tail_at + 1u < opts->max_depth && s->pass_hint_counts.size() == known_levels) {
                        next_tail = tail_at + 1;
                        s->pass_hint_counts.push_back(s->pass_hint_counts[tail_at]); // a bound
                    }
                    if (next_tail >= 0) s->pass_hint_counts.resize((size_t)next_tail + 1);
                    s->pass_hint_tail_at = next_tail;
                    s->pass_hint_key[0] = hint_key[0];
                    s->pass_hint_key[1] = hint_key[1];
                    RT_TRACE("hint: %zu levels, tail at %d (this stratum: %u passes, tail at %d, %s)", s->pass_hint_counts.size(),
                             next_tail, passes, tail_at, hinted ? "hinted" : "followed the counts");
                }
                if (rt_debug())
                    for (uint32_t k = 0; k < passes; ++k)
                        RT_TRACE("pass %u: rays %u mesh queue %u", k, s->h_pinned[k], mesh_counts[k]);
                {
                    // entry 0 counted the shard's slots; its rays are the pixels inside the image
                    uint64_t valid = 0;
                    const uint32_t n_local = rt_shard_tiles(&g, rank);
                    for (uint32_t lt = 0; lt < n_local; ++lt) {
                        const uint32_t kt = lt * g.shard_count + rank, tx = kt % g.tiles_x, ty = kt / g.tiles_x;
                        const uint32_t w = std::min(g.tile, g.width - tx * g.tile), h = std::min(g.tile, g.height - ty * g.tile);
                        valid += (uint64_t)w * h;
                    }
                    s->h_pinned[0] = (uint32_t)valid;
                }
                local.primary_rays += s->h_pinned[0];
                uint32_t levels = 0;
                for (uint32_t k = 0; k < opts->max_depth; ++k) {
                    local.rays += s->h_pinned[k];
                    if (s->h_pinned[k]) ++levels;
                }
                {
                    // a lane-per-path tail reports its segments as a total and its deepest level
                    const uint32_t *lane_totals = s->h_pinned + 2 * (opts->max_depth + 2);
                    local.rays += lane_totals[0];
                    if (lane_totals[0] && lane_totals[1] + 1 > levels) levels = lane_totals[1] + 1;
                }
                if (levels > local.bounce_passes) local.bounce_passes = levels; // (strata share the levels)
                // launches and rays per stage (the counts have come back anyway); times only with RT_FLAG_STAGE_TIMES
                for (uint32_t k = 0; k < passes; ++k) {
                    if (s->h_pinned[k]) {
                        local.n_intersect_launches++; // passes that had rays
                        local.intersect_rays += s->h_pinned[k];
                    }
                    if (walked[k]) {
                        local.n_mesh_launches++;
                        local.mesh_rays += mesh_counts[k];
                        if (walked[k] == 2) local.mesh_one_rays += mesh_counts[k];
                    }
                    local.mesh_records += mesh_counts[k];
                }
                if (stage_times) {
                    float ms_k = 0.f;
                    HIP_TRY(hipEventElapsedTime(&ms_k, s->evp[0], s->evp[1]));
                    local.ms_raygen += ms_k;
                    for (uint32_t k = 0; k < passes; ++k) {
                        HIP_TRY(hipEventElapsedTime(&ms_k, s->evp[1 + 4 * k], s->evp[2 + 4 * k]));
                        local.ms_intersect += ms_k;
                        HIP_TRY(hipEventElapsedTime(&ms_k, s->evp[2 + 4 * k], s->evp[3 + 4 * k]));
                        local.ms_mesh += ms_k;
                        HIP_TRY(hipEventElapsedTime(&ms_k, s->evp[3 + 4 * k], s->evp[4 + 4 * k]));
                        local.ms_shade += ms_k;
                    }
                    if (tail_at >= 0) {
                        HIP_TRY(hipEventElapsedTime(&ms_k, s->evp[1 + 4 * tail_at], s->evp[2 + 4 * tail_at]));
                        local.ms_tail += ms_k;
                    }
                }
            }
            HIP_TRY(hipGetLastError());
            ++done;
            if (per_stratum) {
                int stop = per_stratum(user, done, strata);
                if (stop < 0) {
                    (void)hipStreamSynchronize(st); // nothing stays queued behind a failed call
                    return stop;
                }
                if (stop > 0) goto finished;
            }
        }
    }
    if (!n_slots && call_back_for_empty_shards && per_stratum) {
        // a shard without tiles still meets the other devices between strata (rt_render_multi)
        for (uint32_t k = 1; k <= strata; ++k) {
            int stop = per_stratum(user, k, strata);
            if (stop < 0) return stop;
            if (stop > 0) break;
        }
    }
finished:
    HIP_TRY(hipMemcpyAsync(s->h_pinned, w.d_guards, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 5; ++k) local.guards[k] = s->h_pinned[k];
    local.interval_overflow = s->h_pinned[5];
    local.strata = n_slots ? done : strata; // fewer than floor(sqrt(samples))^2 when the callback stopped the render
#ifdef RT_TAILG_PROF
    {
        unsigned long long mx[64][4], sm[64][8];
        (void)hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_tg_max), sizeof(mx));
        (void)hipMemcpyFromSymbol(sm, HIP_SYMBOL(g_tg_sum), sizeof(sm));
        double tot = 0.0;
        for (int l = 0; l < 64 && sm[l][4]; ++l) {
            const double w_ = (double)sm[l][4];
            fprintf(stderr, "[k_tail_grid] level +%2d: waves %6llu lanes/wave %5.1f wanting %5.1f | slowest wave us: isect %6.2f grid %6.2f wave-brute %6.2f shade %6.2f | mean us: %5.2f %5.2f %5.2f %5.2f | grid rounds per wave %5.1f\n",
                    l, sm[l][4], sm[l][5] / w_, sm[l][6] / w_, mx[l][0] / 100.0, mx[l][1] / 100.0, mx[l][2] / 100.0, mx[l][3] / 100.0,
                    sm[l][0] / w_ / 100.0, sm[l][1] / w_ / 100.0, sm[l][2] / w_ / 100.0, sm[l][3] / w_ / 100.0, sm[l][7] / w_);
            tot += (mx[l][0] + mx[l][1] + mx[l][2] + mx[l][3]) / 100.0;
        }
        fprintf(stderr, "[k_tail_grid] sum of the slowest phases: %.1f us\n", tot);
        std::memset(mx, 0, sizeof(mx));
        std::memset(sm, 0, sizeof(sm));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tg_max), mx, sizeof(mx));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tg_sum), sm, sizeof(sm));
    }
#endif
#ifdef RT_MESH_PROF
    {
        unsigned long long pf[14];
        (void)hipMemcpyFromSymbol(pf, HIP_SYMBOL(g_pf), sizeof(pf));
        fprintf(stderr, "[k_mesh] waves %llu: cycles/wave service %.0f walk %.0f test %.0f; rounds/wave %.1f services/wave %.1f; lanes walking/round %.1f testing/round %.1f\n",
                pf[7], (double)pf[0] / pf[7], (double)pf[1] / pf[7], (double)pf[2] / pf[7], (double)pf[3] / pf[7],
                (double)pf[4] / pf[7], (double)pf[5] / (double)pf[3], (double)pf[6] / (double)pf[3]);
        fprintf(stderr, "[k_mesh] service split: reload+merge %.0f fetch %.0f advance %.0f write %.0f\n", (double)pf[8] / pf[7],
                (double)pf[9] / pf[7], (double)pf[10] / pf[7], (double)pf[11] / pf[7]);
        fprintf(stderr, "[k_mesh] triangle tests (lanes) %llu, hits %llu (%.1f %%)\n", pf[12], pf[13], 100.0 * (double)pf[13] / (double)(pf[12] ? pf[12] : 1));
        unsigned long long z[14] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pf), z, sizeof(z));
    }
#endif
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    local.ms_device = ms;
    if (stats) *stats = local;
    return RT_OK;
}

extern "C" int rt_render_device(rt_scene *scene, const rt_options *opts, double *d_rgb_sum, int32_t *d_primary_ids,
                                rt_stats *stats, void *stream)
{
    return render_strata(scene, opts, d_rgb_sum, d_primary_ids, stats, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int rt_render_device_noalloc(rt_scene *scene, const rt_options *opts, double *d_rgb_sum,
                                        int32_t *d_primary_ids, rt_stats *stats, void *stream)
{
    if (!scene) {
        rt_host::set_last_error("rt_render_device_noalloc: null scene");
        return RT_ERR_INVALID;
    }
    void *base = (scene->device >= 0 && scene->device < RT_MAX_DEVICES) ? g_noalloc_ptr[scene->device] : nullptr;
    Workspace w;
    if (!base || carve_workspace((char *)base, RT_NOALLOC_MAX_SLOTS, RT_NOALLOC_MAX_DEPTH, &w) > RT_NOALLOC_BYTES) {
        rt_host::set_last_error("rt_render_device_noalloc: no static scratch buffer on this device");
        return RT_ERR_DEVICE;
    }
    // every call on a GPU works in that GPU's one array: a second call while one is running is refused instead of
    // left to race with it
    std::atomic<int> &busy = g_noalloc_busy[scene->device];
    if (busy.exchange(1) != 0) {
        rt_host::set_last_error("rt_render_device_noalloc: another call is using this GPU's static scratch buffer (calls on one GPU must not overlap)");
        return RT_ERR_INVALID;
    }
    const int rc = render_strata(scene, opts, d_rgb_sum, d_primary_ids, stats, (hipStream_t)stream, nullptr, nullptr, &w);
    busy.store(0);
    return rc;
}

extern "C" int rt_unshard_device(const rt_options *opts, uint32_t shard_rank, const double *d_shard_rgb,
                                 double *d_full_rgb, const int32_t *d_shard_ids, int32_t *d_full_ids, void *stream)
{
    int rc = check_options(opts);
    if (rc) return rc;
    rt_shard_geom g = rt_shard_geom_make(opts->width, opts->height, opts->tile_size, opts->shard_count);
    if (shard_rank >= g.shard_count) {
        rt_host::set_last_error("rt_unshard_device: shard_rank out of range");
        return RT_ERR_INVALID;
    }
    uint32_t n_slots = rt_shard_tiles(&g, shard_rank) * g.tile_px;
    if (n_slots == 0) return RT_OK;
    hipLaunchKernelGGL(k_unshard, dim3((n_slots + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, shard_rank, n_slots,
                       d_shard_rgb, d_full_rgb, d_shard_ids, d_full_ids);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_unshard_rgb8_device(const rt_options *opts, uint32_t shard_rank, const uint8_t *d_shard_rgb8, uint8_t *d_full_rgb8,
                                      void *stream)
{
    int rc = check_options(opts);
    if (rc) return rc;
    if (!d_shard_rgb8 || !d_full_rgb8) {
        rt_host::set_last_error("rt_unshard_rgb8_device: null argument");
        return RT_ERR_INVALID;
    }
    rt_shard_geom g = rt_shard_geom_make(opts->width, opts->height, opts->tile_size, opts->shard_count);
    if (shard_rank >= g.shard_count) {
        rt_host::set_last_error("rt_unshard_rgb8_device: shard_rank out of range");
        return RT_ERR_INVALID;
    }
    uint32_t n_slots = rt_shard_tiles(&g, shard_rank) * g.tile_px;
    if (n_slots == 0) return RT_OK;
    hipLaunchKernelGGL(k_unshard_rgb8, dim3((n_slots + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, shard_rank, n_slots, d_shard_rgb8,
                       d_full_rgb8);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_tonemap_device(const double *d_rgb_sum, uint64_t n_pixels, double divisor, uint8_t *d_rgb8, void *stream)
{
    if (!d_rgb_sum || !d_rgb8) {
        rt_host::set_last_error("rt_tonemap_device: null argument");
        return RT_ERR_INVALID;
    }
    const uint64_t n_values = n_pixels * 3, threads = (n_values + 3) / 4;
    if (threads == 0) return RT_OK;
    if ((threads + 255) / 256 > 0x7FFFFFFFull) {
        rt_host::set_last_error("rt_tonemap_device: image too large for one launch");
        return RT_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_tonemap, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rgb_sum, n_values,
                       divisor, d_rgb8);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// ---- output buffers that rt_render / rt_render_multi keep on the device between calls ------------------
static int ensure_outputs(rt_scene *s, size_t n_px, bool want_ids)
{
    if (n_px > s->out_cap) {
        if (s->out_rgb) (void)hipFree(s->out_rgb);
        if (s->out_ids) (void)hipFree(s->out_ids);
        s->out_rgb = nullptr;
        s->out_ids = nullptr;
        s->out_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->out_rgb, n_px * 3 * sizeof(double)));
        s->out_cap = n_px;
    }
    if (want_ids && !s->out_ids) HIP_TRY(hipMalloc((void **)&s->out_ids, s->out_cap * sizeof(int32_t)));
    return RT_OK;
}

static int ensure_snapshot(rt_scene *s, size_t n_px)
{
    if (n_px * 3 > s->snap_cap) {
        if (s->snap_d) (void)hipFree(s->snap_d);
        if (s->snap_h) (void)hipHostFree(s->snap_h);
        s->snap_d = nullptr;
        s->snap_h = nullptr;
        s->snap_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->snap_d, n_px * 3));
        HIP_TRY(hipHostMalloc((void **)&s->snap_h, n_px * 3, hipHostMallocDefault));
        s->snap_cap = n_px * 3;
    }
    return RT_OK;
}

// RenderProgress over device buffers (rt_progress, rt_api.h): on_stratum after every stratum; on_snapshot at
// most every min_interval_ms and not after the last stratum, with the running sum tone-mapped on the device by
// strata_done + 1 (CliRenderProgress::sample_finished, main.rs:174-185) and 3 bytes per pixel copied to the host
struct ProgressCtx {
    const rt_progress *p;
    rt_scene *s;
    const double *d_rgb;
    size_t n_px;
    hipStream_t st;
    std::chrono::steady_clock::time_point last;
};

static int snapshot_to_host(rt_scene *s, const double *d_rgb, size_t n_px, uint32_t done, hipStream_t st)
{
    int rc = ensure_snapshot(s, n_px);
    if (rc) return rc;
    rc = rt_tonemap_device(d_rgb, n_px, (double)done + 1.0, s->snap_d, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(s->snap_h, s->snap_d, n_px * 3, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

static int progress_stratum(void *u, uint32_t done, uint32_t total)
{
    ProgressCtx *c = (ProgressCtx *)u;
    if (c->p->on_stratum && c->p->on_stratum(c->p->user, done, total)) return 1;
    if (c->p->on_snapshot && done < total) {
        const auto now = std::chrono::steady_clock::now();
        if (std::chrono::duration_cast<std::chrono::milliseconds>(now - c->last).count() >= (long long)c->p->min_interval_ms) {
            c->last = now;
            int rc = snapshot_to_host(c->s, c->d_rgb, c->n_px, done, c->st);
            if (rc) return rc;
            if (c->p->on_snapshot(c->p->user, done, total, c->s->snap_h, (uint64_t)c->n_px)) return 1;
        }
    }
    return 0;
}

extern "C" int rt_render_device_progress(rt_scene *scene, const rt_options *opts, double *d_rgb_sum, int32_t *d_primary_ids,
                                         rt_stats *stats, void *stream, const rt_progress *progress)
{
    if (!progress || (!progress->on_stratum && !progress->on_snapshot))
        return render_strata(scene, opts, d_rgb_sum, d_primary_ids, stats, (hipStream_t)stream, nullptr, nullptr);
    int rc = check_options(opts);
    if (rc) return rc;
    if (!scene) {
        rt_host::set_last_error("rt_render_device_progress: null scene");
        return RT_ERR_INVALID;
    }
    const size_t n_px = (opts->flags & RT_FLAG_ROW_MAJOR) ? (size_t)opts->width * opts->height : (size_t)rt_shard_pixels(opts);
    ProgressCtx ctx{progress, scene, d_rgb_sum, n_px, (hipStream_t)stream, std::chrono::steady_clock::now()};
    return render_strata(scene, opts, d_rgb_sum, d_primary_ids, stats, (hipStream_t)stream, progress_stratum, &ctx);
}

struct HostRenderCtx {
    const rt_options *opts;
    double *d_full_rgb; // row-major running sum on the device
    double *h_rgb;
    rt_progress_cb cb;
    void *user;
};

static int host_render_stratum(void *u, uint32_t done, uint32_t total)
{
    HostRenderCtx *c = (HostRenderCtx *)u;
    size_t n = (size_t)c->opts->width * c->opts->height;
    if (hipMemcpy(c->h_rgb, c->d_full_rgb, n * 3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
        rt_host::set_last_error("rt_render: copy to host failed");
        return RT_ERR_DEVICE;
    }
    return c->cb(c->user, done, total, c->h_rgb) ? 1 : 0;
}

extern "C" int rt_render(rt_scene *scene, const rt_options *opts_in, double *rgb_sum, int32_t *primary_ids,
                         rt_stats *stats, rt_progress_cb cb, void *user)
{
    int rc = check_options(opts_in);
    if (rc) return rc;
    if (!scene || !rgb_sum) {
        rt_host::set_last_error("rt_render: null scene or output");
        return RT_ERR_INVALID;
    }
    rt_options opts = *opts_in; // a whole image on this GPU
    opts.shard_rank = 0;
    opts.shard_count = 1;
    HIP_TRY(hipSetDevice(scene->device));
    opts.flags |= RT_FLAG_ROW_MAJOR; // the kernels write y * width + x directly
    const size_t n = (size_t)opts.width * opts.height;
    rc = ensure_outputs(scene, n, primary_ids != nullptr); // kept across calls, grown on demand
    if (rc) return rc;
    HostRenderCtx ctx{&opts, scene->out_rgb, rgb_sum, cb, user};
    // without a callback nothing leaves the device before the frame is done (the reference hands the running
    // sum to sample_finished after every stratum: 24 bytes per pixel each time)
    rc = render_strata(scene, &opts, scene->out_rgb, primary_ids ? scene->out_ids : nullptr, stats, nullptr,
                       cb ? host_render_stratum : nullptr, &ctx);
    if (rc == RT_OK) {
        HIP_TRY(hipMemcpy(rgb_sum, scene->out_rgb, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (primary_ids) HIP_TRY(hipMemcpy(primary_ids, scene->out_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return rc;
}

// system::render + write_render_result_to_file's pixel math (main.rs:106, 201-222) for a caller that wants the
// image, not the sums: the frame is tone-mapped where it lies and 3 bytes per pixel cross PCIe instead of 24.
extern "C" int rt_render_rgb8(rt_scene *scene, const rt_options *opts_in, uint8_t *rgb8, int32_t *primary_ids, rt_stats *stats,
                              const rt_progress *progress)
{
    int rc = check_options(opts_in);
    if (rc) return rc;
    if (!scene || !rgb8) {
        rt_host::set_last_error("rt_render_rgb8: null scene or output");
        return RT_ERR_INVALID;
    }
    rt_options opts = *opts_in; // a whole image on this GPU
    opts.shard_rank = 0;
    opts.shard_count = 1;
    opts.flags |= RT_FLAG_ROW_MAJOR;
    HIP_TRY(hipSetDevice(scene->device));
    const size_t n = (size_t)opts.width * opts.height;
    if ((rc = ensure_outputs(scene, n, primary_ids != nullptr))) return rc;
    if ((rc = ensure_snapshot(scene, n))) return rc;
    rt_stats local;
    std::memset(&local, 0, sizeof(local));
    const bool with_progress = progress && (progress->on_stratum || progress->on_snapshot);
    ProgressCtx ctx{progress, scene, scene->out_rgb, n, nullptr, std::chrono::steady_clock::now()};
    rc = render_strata(scene, &opts, scene->out_rgb, primary_ids ? scene->out_ids : nullptr, &local, nullptr,
                       with_progress ? progress_stratum : nullptr, &ctx);
    if (rc != RT_OK) return rc;
    if ((rc = rt_tonemap_device(scene->out_rgb, n, (double)local.strata + 1.0, scene->snap_d, nullptr))) return rc;
    HIP_TRY(hipMemcpy(rgb8, scene->snap_d, n * 3, hipMemcpyDeviceToHost));
    if (primary_ids) HIP_TRY(hipMemcpy(primary_ids, scene->out_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (stats) *stats = local;
    return RT_OK;
}

// Page-locked host memory for the outputs of rt_render / rt_render_rgb8 / rt_render_multi: the copy out of the GPU
// then runs at the link's speed (58 GB/s measured) instead of through a staging buffer.
extern "C" int rt_pinned_alloc(size_t bytes, void **out)
{
    if (!out) {
        rt_host::set_last_error("rt_pinned_alloc: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RT_OK;
}

extern "C" void rt_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// ---- one frame over several GPUs of this process ---------------------------------------------------------
// The reference's one parallel step spreads strata over threads and merges under a mutex (system.rs:288-299);
// here tile shards are spread over GPUs (pixels are independent, system.rs:244-253) and the finished tiles are
// gathered to the first scene's GPU over xGMI: every peer pushes its shard with hipMemcpyPeerAsync on ITS OWN
// stream into ITS OWN region of the root's buffer, so the copies run over the peers' links at the same time; the
// root waits for the peers' events and one k_unshard_all launch scatters all regions into the row-major image.
// One host thread per GPU drives that GPU's wavefront loop; threads, streams and events stay with the scenes
// between calls (SceneWorker).
SceneWorker::~SceneWorker()
{
    {
        std::lock_guard<std::mutex> lock(m);
        quit = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
}

void SceneWorker::submit(std::function<void()> f)
{
    std::unique_lock<std::mutex> lock(m);
    if (!th.joinable()) th = std::thread([this] { loop(); });
    job = std::move(f);
    pending = true;
    finished = false;
    cv.notify_all();
}

void SceneWorker::wait()
{
    std::unique_lock<std::mutex> lock(m);
    cv.wait(lock, [&] { return finished; });
}

void SceneWorker::loop()
{
    std::unique_lock<std::mutex> lock(m);
    for (;;) {
        cv.wait(lock, [&] { return pending || quit; });
        if (quit) return;
        pending = false;
        std::function<void()> f = std::move(job);
        lock.unlock();
        f();
        lock.lock();
        finished = true;
        cv.notify_all();
    }
}

// stream, event and worker thread a scene keeps for rt_render_multi (made on first use, with its device current)
static int ensure_multi_side(rt_scene *s)
{
    HIP_TRY(hipSetDevice(s->device));
    if (!s->multi_stream) HIP_TRY(hipStreamCreateWithFlags(&s->multi_stream, hipStreamNonBlocking));
    if (!s->multi_done) HIP_TRY(hipEventCreateWithFlags(&s->multi_done, hipEventDisableTiming));
    if (!s->worker) s->worker = new SceneWorker();
    return RT_OK;
}

namespace {
struct MultiBarrier { // all device threads meet between strata when progress is reported
    std::mutex m;
    std::condition_variable cv;
    uint32_t n = 0, waiting = 0, generation = 0;
    bool aborted = false; // a device failed: nobody waits any more
    void wait()
    {
        std::unique_lock<std::mutex> lock(m);
        if (aborted) return;
        const uint32_t gen = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return gen != generation || aborted; });
        }
    }
    void abort()
    {
        std::lock_guard<std::mutex> lock(m);
        aborted = true;
        cv.notify_all();
    }
};

struct MultiShared {
    const rt_progress *progress = nullptr;
    rt_scene *const *scenes = nullptr;
    uint32_t n = 0;
    rt_options opts;
    MultiBarrier barrier;
    std::atomic<int> stop{0};
    std::atomic<int> error{0};
    std::chrono::steady_clock::time_point last;
    std::vector<uint8_t> image;           // row-major RGB8 snapshot
    std::vector<std::vector<uint32_t>> maps; // per shard: slot -> pixel
    std::vector<std::string> messages;
    std::vector<int> rcs;
    std::mutex fail_mutex;
    // a device's failure: its code and message stay, everybody else stops at the next stratum
    void fail(uint32_t k, int code, const std::string &msg)
    {
        {
            std::lock_guard<std::mutex> lock(fail_mutex);
            if (rcs[k] == RT_OK) {
                rcs[k] = code;
                messages[k] = msg;
            }
        }
        error.store(1);
        barrier.abort();
    }
};

struct MultiThreadCtx {
    MultiShared *sh;
    uint32_t k;
    double *d_shard;
    size_t n_slots;
    hipStream_t st;
};

// per stratum, on every device thread; thread 0 talks to the caller
int multi_stratum(void *u, uint32_t done, uint32_t total)
{
    MultiThreadCtx *c = (MultiThreadCtx *)u;
    MultiShared *sh = c->sh;
    const rt_progress *p = sh->progress;
    if (!p) return sh->error.load() ? 1 : 0;
    bool snap = false;
    if (c->k == 0) {
        if (p->on_stratum && p->on_stratum(p->user, done, total)) sh->stop.store(1);
        if (p->on_snapshot && done < total) {
            const auto now = std::chrono::steady_clock::now();
            if (std::chrono::duration_cast<std::chrono::milliseconds>(now - sh->last).count() >= (long long)p->min_interval_ms) {
                sh->last = now;
                snap = true;
            }
        }
        sh->stop.fetch_or(snap ? 2 : 0);
    }
    sh->barrier.wait(); // every device has finished stratum `done`; thread 0's decisions are visible
    const int flags = sh->stop.load();
    if (flags & 2) {
        rt_scene *s = sh->scenes[c->k];
        if (c->n_slots) {
            // a snapshot that fails is this device's failure: the call returns its code, not a shortened render
            const int rc = snapshot_to_host(s, c->d_shard, c->n_slots, done, c->st);
            if (rc != RT_OK) sh->fail(c->k, rc, std::string("progress snapshot: ") + rt_last_error());
        }
        if (!sh->error.load()) {
            const std::vector<uint32_t> &map = sh->maps[c->k];
            for (size_t i = 0; i < c->n_slots; ++i)
                if (map[i] != 0xFFFFFFFFu) std::memcpy(&sh->image[(size_t)map[i] * 3], s->snap_h + i * 3, 3);
        }
        sh->barrier.wait(); // the image is complete
        if (c->k == 0) {
            if (!sh->error.load() && p->on_snapshot(p->user, done, total, sh->image.data(), (uint64_t)sh->opts.width * sh->opts.height))
                sh->stop.fetch_or(1);
            sh->stop.fetch_and(~2);
        }
        sh->barrier.wait();
    }
    return (sh->stop.load() & 1) || sh->error.load() ? 1 : 0;
}
} // namespace

// rgb_sum (f64 sums) or rgb8 (tone-mapped by strata + 1, main.rs:201-222): exactly one is non-null
static int render_multi_impl(rt_scene *const *scenes, uint32_t n_devices, const rt_options *opts_in, double *rgb_sum, uint8_t *rgb8,
                             int32_t *primary_ids, rt_stats *stats, const rt_progress *progress)
{
    int rc = check_options(opts_in);
    if (rc) return rc;
    if (!scenes || n_devices == 0 || (!rgb_sum && !rgb8)) {
        rt_host::set_last_error("rt_render_multi: null scenes or output, or no device");
        return RT_ERR_INVALID;
    }
    for (uint32_t k = 0; k < n_devices; ++k) {
        if (!scenes[k]) {
            rt_host::set_last_error("rt_render_multi: null scene");
            return RT_ERR_INVALID;
        }
        // one scene serves one render at a time (its workspace, pinned words and events): the same GPU may appear
        // more than once, but through scenes of its own
        for (uint32_t j = 0; j < k; ++j)
            if (scenes[j] == scenes[k]) {
                rt_host::set_last_error("rt_render_multi: scenes[" + std::to_string(j) + "] and scenes[" + std::to_string(k) +
                                        "] are the same rt_scene; create one scene per entry");
                return RT_ERR_INVALID;
            }
    }
    const bool with_progress = progress && (progress->on_stratum || progress->on_snapshot);
    MultiShared sh;
    sh.progress = with_progress ? progress : nullptr;
    sh.scenes = scenes;
    sh.n = n_devices;
    sh.opts = *opts_in;
    sh.opts.shard_count = n_devices;
    sh.opts.flags &= ~RT_FLAG_ROW_MAJOR;
    sh.barrier.n = n_devices;
    sh.last = std::chrono::steady_clock::now();
    sh.messages.resize(n_devices);
    sh.rcs.assign(n_devices, RT_OK);
    const size_t n_px = (size_t)opts_in->width * opts_in->height;
    std::vector<size_t> n_slots(n_devices), first_slot(n_devices);
    size_t total_slots = 0;
    for (uint32_t k = 0; k < n_devices; ++k) {
        rt_options o = sh.opts;
        o.shard_rank = k;
        n_slots[k] = (size_t)rt_shard_pixels(&o);
        first_slot[k] = total_slots; // regions in rank order: what k_unshard_all assumes
        total_slots += n_slots[k];
    }
    if (total_slots > 0xFFFFFFF0ull) {
        rt_host::set_last_error("rt_render_multi: image too large");
        return RT_ERR_INVALID;
    }
    if (with_progress && progress->on_snapshot) {
        sh.image.assign(n_px * 3, 0);
        sh.maps.resize(n_devices);
        for (uint32_t k = 0; k < n_devices; ++k) {
            rt_options o = sh.opts;
            o.shard_rank = k;
            sh.maps[k].resize(n_slots[k]);
            if (n_slots[k] && (rc = rt_shard_map(&o, sh.maps[k].data()))) return rc;
        }
    }
    // Buffers (kept by the scenes).  Root: [full image | region of shard 0 | region of shard 1 | ...]; a scene on
    // the root's GPU renders straight into its region; a scene on another GPU renders into a buffer of its own
    // and pushes it into its region over its link.
    rt_scene *root = scenes[0];
    for (uint32_t k = 0; k < n_devices; ++k) {
        if ((rc = ensure_multi_side(scenes[k]))) return rc;
        const bool local = scenes[k]->device == root->device;
        const size_t need = k == 0 ? n_px + total_slots : (local ? 1 : n_slots[k]);
        if ((rc = ensure_outputs(scenes[k], need ? need : 1, primary_ids != nullptr))) return rc;
        if (!local) {
            // the push is issued on the peer's stream into the root's memory
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, scenes[k]->device, root->device));
            if (can) {
                hipError_t e = hipDeviceEnablePeerAccess(root->device, 0); // current device: scenes[k]'s (ensure_multi_side)
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
                (void)hipGetLastError();
            }
        }
    }
    if (rgb8) {
        // the picture: every GPU tone-maps its own tiles where they lie (the pixel math of main.rs:201-217 is per pixel,
        // so it commutes with the gather) and 3 bytes per slot cross the links instead of 24.  The root's byte buffer:
        // [row-major image | region of shard 0 | region of shard 1 | ...]; a peer on another GPU tone-maps into its own.
        for (uint32_t k = 0; k < n_devices; ++k) {
            HIP_TRY(hipSetDevice(scenes[k]->device));
            const size_t need8 = k == 0 ? n_px + total_slots : (scenes[k]->device == root->device ? 0 : n_slots[k]);
            if (need8 && (rc = ensure_snapshot(scenes[k], need8))) return rc;
        }
    }
    auto region_rgb = [&](uint32_t k) { return root->out_rgb + (n_px + first_slot[k]) * 3; };
    auto region_ids = [&](uint32_t k) { return primary_ids ? root->out_ids + n_px + first_slot[k] : (int32_t *)nullptr; };
    auto is_local = [&](uint32_t k) { return scenes[k]->device == root->device; };
    auto shard_rgb = [&](uint32_t k) { return is_local(k) ? region_rgb(k) : scenes[k]->out_rgb; };
    auto shard_ids = [&](uint32_t k) { return !primary_ids ? (int32_t *)nullptr : (is_local(k) ? region_ids(k) : scenes[k]->out_ids); };
    std::vector<rt_stats> st(n_devices);
    std::vector<MultiThreadCtx> ctxs(n_devices);
    auto worker = [&](uint32_t k) {
        rt_scene *s = scenes[k];
        rt_options o = sh.opts;
        o.shard_rank = k;
        ctxs[k] = MultiThreadCtx{&sh, k, shard_rgb(k), n_slots[k], s->multi_stream};
        // with progress every thread must pass the same barriers: an empty shard still walks the strata
        int r = render_strata(s, &o, shard_rgb(k), shard_ids(k), &st[k], s->multi_stream, with_progress ? multi_stratum : nullptr,
                              &ctxs[k], nullptr, /*call_back_for_empty_shards=*/true);
        if (r == RT_OK && n_slots[k] && !sh.error.load()) {
            // the one exchange step, this GPU's half: push the shard into its region of the root's buffer
            hipError_t e = hipSuccess;
            if (rgb8) {
                uint8_t *region8 = root->snap_d + (n_px + first_slot[k]) * 3;
                uint8_t *mine8 = is_local(k) ? region8 : s->snap_d;
                r = rt_tonemap_device(shard_rgb(k), n_slots[k], (double)st[k].strata + 1.0, mine8, s->multi_stream);
                if (r == RT_OK && !is_local(k))
                    e = hipMemcpyPeerAsync(region8, root->device, mine8, s->device, n_slots[k] * 3, s->multi_stream);
            } else if (!is_local(k)) {
                e = hipMemcpyPeerAsync(region_rgb(k), root->device, s->out_rgb, s->device, n_slots[k] * 3 * sizeof(double), s->multi_stream);
            }
            if (e == hipSuccess && primary_ids && !is_local(k))
                e = hipMemcpyPeerAsync(region_ids(k), root->device, s->out_ids, s->device, n_slots[k] * sizeof(int32_t), s->multi_stream);
            if (e == hipSuccess) e = hipEventRecord(s->multi_done, s->multi_stream);
            if (e != hipSuccess) {
                r = RT_ERR_DEVICE;
                rt_host::set_last_error(std::string("tile gather: ") + hipGetErrorString(e));
            }
        }
        if (r != RT_OK) sh.fail(k, r, rt_last_error());
    };
    if (n_devices == 1) {
        worker(0);
    } else {
        for (uint32_t k = 1; k < n_devices; ++k) scenes[k]->worker->submit([&worker, k] { worker(k); });
        worker(0); // the calling thread drives the root's GPU (and talks to the caller's callbacks)
        for (uint32_t k = 1; k < n_devices; ++k) scenes[k]->worker->wait();
    }
    rc = RT_OK;
    for (uint32_t k = 0; k < n_devices; ++k)
        if (sh.rcs[k] != RT_OK && rc == RT_OK) {
            rc = sh.rcs[k];
            rt_host::set_last_error("rt_render_multi: device " + std::to_string(k) + ": " + sh.messages[k]);
        }
    if (rc != RT_OK) {
        // leave nothing queued behind a failed call
        for (uint32_t k = 0; k < n_devices; ++k)
            if (hipSetDevice(scenes[k]->device) == hipSuccess) (void)hipStreamSynchronize(scenes[k]->multi_stream);
        return rc;
    }
    {
        HIP_TRY(hipSetDevice(root->device));
        hipStream_t s0 = root->multi_stream;
        for (uint32_t k = 1; k < n_devices; ++k)
            if (n_slots[k]) HIP_TRY(hipStreamWaitEvent(s0, scenes[k]->multi_done, 0));
        const rt_shard_geom g = rt_shard_geom_make(sh.opts.width, sh.opts.height, sh.opts.tile_size, n_devices);
        if (rgb8) {
            hipLaunchKernelGGL(k_unshard_all_rgb8, dim3((uint32_t)((total_slots + 255) / 256)), dim3(256), 0, s0, g, (uint32_t)total_slots,
                               root->snap_d + n_px * 3, root->snap_d);
            if (primary_ids)
                hipLaunchKernelGGL(k_unshard_all, dim3((uint32_t)((total_slots + 255) / 256)), dim3(256), 0, s0, g, (uint32_t)total_slots,
                                   (const double *)nullptr, (double *)nullptr, root->out_ids + n_px, root->out_ids);
        } else {
            hipLaunchKernelGGL(k_unshard_all, dim3((uint32_t)((total_slots + 255) / 256)), dim3(256), 0, s0, g, (uint32_t)total_slots,
                               root->out_rgb + n_px * 3, root->out_rgb, primary_ids ? root->out_ids + n_px : nullptr,
                               primary_ids ? root->out_ids : nullptr);
        }
        HIP_TRY(hipGetLastError());
        if (rgb8) {
            HIP_TRY(hipMemcpyAsync(rgb8, root->snap_d, n_px * 3, hipMemcpyDeviceToHost, s0));
        } else {
            HIP_TRY(hipMemcpyAsync(rgb_sum, root->out_rgb, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost, s0));
        }
        if (primary_ids) HIP_TRY(hipMemcpyAsync(primary_ids, root->out_ids, n_px * sizeof(int32_t), hipMemcpyDeviceToHost, s0));
        HIP_TRY(hipStreamSynchronize(s0));
    }
    if (stats) {
        *stats = st[0];
        for (uint32_t k = 1; k < n_devices; ++k) {
            stats->rays += st[k].rays;
            stats->primary_rays += st[k].primary_rays;
            stats->mesh_rays += st[k].mesh_rays;
            stats->intersect_rays += st[k].intersect_rays;
            stats->mesh_records += st[k].mesh_records;
            stats->mesh_one_rays += st[k].mesh_one_rays;
            stats->interval_overflow += st[k].interval_overflow;
            for (int g = 0; g < 5; ++g) stats->guards[g] += st[k].guards[g];
            if (st[k].bounce_passes > stats->bounce_passes) stats->bounce_passes = st[k].bounce_passes;
            if (st[k].strata > stats->strata) stats->strata = st[k].strata;
            if (st[k].ms_device > stats->ms_device) stats->ms_device = st[k].ms_device; // the slowest device
        }
    }
    return RT_OK;
}

extern "C" int rt_render_multi(rt_scene *const *scenes, uint32_t n_devices, const rt_options *opts_in, double *rgb_sum,
                               int32_t *primary_ids, rt_stats *stats, const rt_progress *progress)
{
    if (!rgb_sum) {
        rt_host::set_last_error("rt_render_multi: null output");
        return RT_ERR_INVALID;
    }
    return render_multi_impl(scenes, n_devices, opts_in, rgb_sum, nullptr, primary_ids, stats, progress);
}

extern "C" int rt_render_multi_rgb8(rt_scene *const *scenes, uint32_t n_devices, const rt_options *opts_in, uint8_t *rgb8,
                                    int32_t *primary_ids, rt_stats *stats, const rt_progress *progress)
{
    if (!rgb8) {
        rt_host::set_last_error("rt_render_multi_rgb8: null output");
        return RT_ERR_INVALID;
    }
    return render_multi_impl(scenes, n_devices, opts_in, nullptr, rgb8, primary_ids, stats, progress);
}

extern "C" int rt_trace_rays(rt_scene *s, uint64_t seed, uint64_t n, const double *origins, const double *directions,
                             rt_hit *hits, rt_stats *stats)
{
    if (!s || !origins || !directions || !hits) {
        rt_host::set_last_error("rt_trace_rays: null argument");
        return RT_ERR_INVALID;
    }
    if (n == 0) return RT_OK;
    if (n > 0x7FFFFFFFull) {
        rt_host::set_last_error("rt_trace_rays: too many rays in one call");
        return RT_ERR_INVALID;
    }
    for (uint64_t i = 0; i < n; ++i) {
        const double *dv = directions + 3 * i;
        const double l2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
        if (!(std::fabs(l2 - 1.0) <= 2e-9)) {
            rt_host::set_last_error("rt_trace_rays: direction " + std::to_string(i) + " does not have unit length");
            return RT_ERR_INVALID;
        }
    }
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_workspace(s, (size_t)n, 1);
    if (rc) return rc;
    const Workspace &w = s->own;
    // The caller's interleaved rays go up in two copies and are de-interleaved on the device; the hits come back
    // interleaved in one.  Staging lives in arrays of the workspace this call does not use otherwise - the
    // attenuation arrays of the second ray queue and of the mesh queue (three consecutive arrays each, see
    // carve_workspace) going up, the first ray queue's nine arrays, free again after the trace, coming back.
    double *stage_o = w.q[1].thr[0], *stage_d = w.mq.thr[0];
    rt_hit *stage_hits = reinterpret_cast<rt_hit *>(w.q[0].o[0]);
    if ((size_t)(w.q[1].thr[2] - w.q[1].thr[0]) < 2 * (size_t)n || (size_t)(w.mq.thr[2] - w.mq.thr[0]) < 2 * (size_t)n ||
        (size_t)(w.q[0].thr[2] - w.q[0].o[0]) < 8 * (size_t)n || sizeof(rt_hit) != 7 * sizeof(double)) {
        rt_host::set_last_error("rt_trace_rays: workspace layout does not hold the staging arrays");
        return RT_ERR_DEVICE;
    }
    HIP_TRY(hipMemcpy(stage_o, origins, n * 3 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(stage_d, directions, n * 3 * sizeof(double), hipMemcpyHostToDevice));
    uint32_t n32 = (uint32_t)n;
    hipLaunchKernelGGL(k_rays_in, dim3((n32 + 255) / 256), dim3(256), 0, nullptr, n32, stage_o, stage_d, w.q[0]);
    // the rays lie at positions 0 .. n-1: a queue whose first region holds them all
    HIP_TRY(hipMemset(w.d_rcnt, 0, 2 * (size_t)RT_Q_REGIONS * w.rstride * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(w.d_rcnt, &n32, sizeof(uint32_t), hipMemcpyHostToDevice));
    QRegions by_position = queue_regions(w, 0, 0);
    by_position.rc = (uint32_t)((w.cap + 63) & ~(size_t)63);
    HIP_TRY(hipMemset(w.d_guards, 0, 8 * sizeof(uint32_t)));
    PassParams pp;
    pp.seed = seed;
    pp.stratum = 0;
    pp.depth = 0;
    pp.max_depth = 1;
    pp.force_uv = 1;
    pp.bias = 0.0;
    // the per-ray hit records live in the arrays of the second ray queue, which this call does not use otherwise
    HitBuf hit;
    hit.t = w.q[1].o[0];
    hit.n[0] = w.q[1].o[1];
    hit.n[1] = w.q[1].o[2];
    hit.n[2] = w.q[1].d[0];
    hit.uv[0] = w.q[1].d[1];
    hit.uv[1] = w.q[1].d[2];
    hit.obj = (int32_t *)w.q[1].slot;
    (void)launch_intersect(s, w, nullptr, n32, w.q[0], w.d_counts, by_position, queue_regions(w, 1, 0), pp, nullptr, false, w.q[1],
                           queue_regions(w, 0, 1), nullptr, hit, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_hits_out, dim3((n32 + 255) / 256), dim3(256), 0, nullptr, n32, hit, stage_hits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(hits, stage_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        uint32_t g[6];
        HIP_TRY(hipMemcpy(g, w.d_guards, sizeof(g), hipMemcpyDeviceToHost));
        stats->rays = n;
        for (int k = 0; k < 5; ++k) stats->guards[k] = g[k];
        stats->interval_overflow = g[5];
    }
    return RT_OK;
}
