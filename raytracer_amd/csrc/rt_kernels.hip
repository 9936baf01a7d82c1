// rt_kernels.hip — the wavefront path tracer for gfx950 (MI355X) and the
// device half of the C ABI.
//
// One stratum of one shard = one wavefront run:
//
//   (primary rays: camera + jitter, system.rs:237-242, 57-66, regenerated inside the depth-0 kernels)
//   repeat for depth = 0 .. max_depth-1 while rays remain:
//     k_intersect nearest hit over the object list        (system.rs:122-133, shapes/*)
//     k_shade     emit / scatter / terminate, compacting  (system.rs:112-120,135-142,
//                 survivors into the other ray queue       materials/*, texture.rs)
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
#include <string>
#include <vector>

#include "../../include/rt_api.h"
#include "../../include/rt_rng.h"
#include "rt_dev_shapes.h"
#include "rt_flatten.h"
#include "rt_shard.h"

namespace rt_host {
void set_last_error(const std::string &msg);
}

// ------------------------------------------------------------------ device buffers
struct RayQueue {
    double *o[3];
    double *d[3];
    double *thr[3];
    uint32_t *slot;  // index into the shard's frame buffer
    uint32_t *pixel; // y * width + x, the random-number key
    // primary != 0: nothing above is read.  Entry i is slot i of the shard and its ray is that
    // pixel's camera ray, regenerated wherever it is needed (a jitter draw and a normalise) instead
    // of being written by one kernel and read back by three: 1.8 GB less traffic per 4K frame.
    int primary;
    int row_major; // outputs are indexed by pixel instead of by shard slot (RT_FLAG_ROW_MAJOR)
    uint32_t rank, s_i, s_j;
    rt_shard_geom g;
    double recip_sqrt_spp;
};

struct HitBuf {
    double *t;
    double *n[3];
    double *uv[2];
    int32_t *obj;
};

#ifndef RT_MESH_BATCH
#define RT_MESH_BATCH 2
#endif
#ifndef RT_MESH_WAVES
#define RT_MESH_WAVES 3
#endif
#define RT_MESH_OCC __attribute__((amdgpu_waves_per_eu(RT_MESH_WAVES, RT_MESH_WAVES)))

// k_intersect streams rays in and hit records out: without CSG it is worth 4 waves per SIMD (a few
// spilled registers against twice the loads in flight: 1.06 -> 0.87 ms per 4K frame, measured)
#define RT_ISECT_OCC __attribute__((amdgpu_waves_per_eu((CSG || TORUS) ? 1 : 4)))

struct PassParams {
    uint64_t seed;
    uint32_t stratum;
    uint32_t depth;
    uint32_t max_depth;
    uint32_t force_uv;
    double bias;
};

#define RT_BLOCK 256

RT_DEV uint32_t lane_id() { return threadIdx.x & 63u; }

// Append with one atomic per workgroup: a single counter word takes about 90 atomics per
// microsecond on this chip, so a per-wave atomic caps a pass at ~6 G appended rays/s; with
// 1024-thread workgroups the counter is out of the picture.  Called by every thread of the
// block, in block-uniform control flow.  `lds` holds blockDim.x / 64 + 1 words.
RT_DEV uint32_t block_append(bool alive, uint32_t *counter, uint32_t *lds)
{
    const unsigned long long mask = __ballot(alive);
    const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if (lane_id() == 0) lds[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; ++w) {
            uint32_t c = lds[w];
            lds[w] = total;
            total += c;
        }
        lds[n_waves] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t at = lds[n_waves] + lds[wave] + (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
    __syncthreads();
    return at;
}

#define RT_BLOCK_WIDE 1024 // ray generation and shading: light on registers, heavy on queue appends

// The scene's small tables (objects, leaves, meshes, materials, textures) are read by every ray
// through chains of dependent loads: objects[k] -> leaves[leaf0] -> parameters.  From memory each
// link costs a cache round trip of several hundred cycles - measured, the object loop of
// k_intersect waited 27 000 cycles per 64 rays on them.  Each workgroup therefore copies the
// tables into LDS once and the kernel works on a DevScene whose pointers lead there.  Scenes whose
// tables exceed RT_SCENE_LDS_MAX keep reading memory.  Ends with a barrier.
#define RT_SCENE_LDS_MAX 32768u
static inline __host__ __device__ uint32_t scene_lds_bytes(const DevScene &sc)
{
    const uint32_t b = (((uint32_t)sc.n_objects * (uint32_t)sizeof(DevObject) + 15u) & ~15u) +
                       ((sc.n_leaves * (uint32_t)sizeof(DevLeaf) + 15u) & ~15u) +
                       ((sc.n_meshes * (uint32_t)sizeof(DevMesh) + 15u) & ~15u) +
                       ((sc.n_materials * (uint32_t)sizeof(DevMaterial) + 15u) & ~15u) +
                       ((sc.n_textures * (uint32_t)sizeof(DevTexture) + 15u) & ~15u);
    return b <= RT_SCENE_LDS_MAX ? b : 0u;
}

RT_DEV DevScene scene_in_lds(const DevScene &sc, unsigned char *lds)
{
    DevScene v = sc;
    if (scene_lds_bytes(sc) != 0u) {
        unsigned char *p = lds;
#define RT_STAGE(FIELD, TYPE, COUNT)                                                                                   \
    do {                                                                                                               \
        const uint32_t words_ = (uint32_t)(COUNT) * (uint32_t)(sizeof(TYPE) / 4);                                      \
        const uint32_t *src_ = reinterpret_cast<const uint32_t *>(sc.FIELD);                                           \
        uint32_t *dst_ = reinterpret_cast<uint32_t *>(p);                                                              \
        for (uint32_t i_ = threadIdx.x; i_ < words_; i_ += blockDim.x) dst_[i_] = src_[i_];                            \
        v.FIELD = reinterpret_cast<const TYPE *>(p);                                                                   \
        p += (words_ * 4u + 15u) & ~15u;                                                                               \
    } while (0)
        RT_STAGE(objects, DevObject, sc.n_objects);
        RT_STAGE(leaves, DevLeaf, sc.n_leaves);
        RT_STAGE(meshes, DevMesh, sc.n_meshes);
        RT_STAGE(materials, DevMaterial, sc.n_materials);
        RT_STAGE(textures, DevTexture, sc.n_textures);
#undef RT_STAGE
    }
    __syncthreads();
    return v;
}

// ------------------------------------------------------------------ ray generation
// get_stratified_ray + Camera::pixel_ray (system.rs:237-242, 57-66) for slot `slot` of the shard;
// false for the padding slots beyond the image edge
RT_DEV bool primary_ray(const DevScene &sc, const RayQueue &q, const PassParams &pp, uint32_t slot, d3 &origin, d3 &dir,
                        uint32_t &pixel)
{
    uint32_t x = 0, y = 0;
    if (!rt_shard_decode(&q.g, q.rank, slot, &x, &y)) return false;
    pixel = y * q.g.width + x;
    rt_rng_block blk = rt_rng_draw(pp.seed, pixel, pp.stratum, 0u, RT_RNG_SITE_JITTER, 0u);
    double sx = (((double)q.s_i + rt_rng_u0(&blk)) * q.recip_sqrt_spp) - 0.5;
    double sy = (((double)q.s_j + rt_rng_u1(&blk)) * q.recip_sqrt_spp) - 0.5;
    double px = (double)x + sx, py = (double)y + sy;
    double w = (double)q.g.width, h = (double)q.g.height;
    double aspect = w / h;
    double ndcx = px / w, ndcy = py / h;
    double cx = (2.0 * ndcx - 1.0) * sc.cam_fov_factor * aspect;
    double cy = (1.0 - 2.0 * ndcy) * sc.cam_fov_factor;
    const double *m = sc.cam_c2w;
    origin = mk3(m[12], m[13], m[14]);
    d3 dp = mk3(cx * m[0] + cy * m[4] + -1.0 * m[8] + m[12], cx * m[1] + cy * m[5] + -1.0 * m[9] + m[13],
                cx * m[2] + cy * m[6] + -1.0 * m[10] + m[14]);
    dir = normalize3(dp - origin);
    return true;
}

// origin, direction and random-number key of queue entry i; false for a padding slot
RT_DEV bool fetch_ray(const DevScene &sc, const RayQueue &q, const PassParams &pp, uint32_t i, d3 &o, d3 &d, uint32_t &pixel)
{
    if (q.primary) return primary_ray(sc, q, pp, i, o, d, pixel);
    o = mk3(q.o[0][i], q.o[1][i], q.o[2][i]);
    d = mk3(q.d[0][i], q.d[1][i], q.d[2][i]);
    pixel = q.pixel[i];
    return true;
}

// index of the ray's pixel in the output buffers
RT_DEV uint32_t ray_slot(const RayQueue &q, uint32_t i)
{
    if (!q.primary) return q.slot[i];
    if (!q.row_major) return i;
    uint32_t x = 0, y = 0;
    (void)rt_shard_decode(&q.g, q.rank, i, &x, &y);
    return y * q.g.width + x;
}

#define RT_HIT_PADDING (-2) // hit.obj of a padding slot: no ray, nothing to shade

// ------------------------------------------------------------------ nearest hit
struct Candidate {
    double t;
    uint32_t aux;
    bool ok;
};

// first_positive_intersection / first_intersection (shapes/mod.rs:45-55) over a list
RT_DEV Candidate pick_from_list(const IvList &l, int mode)
{
    Candidate c;
    c.ok = false;
    c.t = 0.0;
    c.aux = 0u;
    if (l.n == 0) return c;
    if (mode == DM_FIRST_ENTRY) {
        c.t = l.t[0];
        c.aux = l.tag[0];
        c.ok = true;
        return c;
    }
    for (int i = 0; i < 2 * DEV_MAX_INTERVALS; ++i) {
        if (i >= 2 * l.n) break;
        if (!(l.t[i] < 0.0)) {
            c.t = l.t[i];
            c.aux = l.tag[i];
            c.ok = true;
            return c;
        }
    }
    return c;
}

template <bool TORUS>
RT_DEV Candidate eval_leaf_object(const DevScene &sc, const DevObject &ob, d3 o, d3 d, Guards &gd)
{
    const DevLeaf &L = sc.leaves[ob.leaf0];
    LeafRay ry = to_leaf(L, o, d);
    double t[4];
    uint32_t sub[4];
    int n = leaf_ends<TORUS>(L, ry, t, sub, gd);
    Candidate c;
    c.ok = false;
    c.t = 0.0;
    c.aux = 0u;
    int k = -1;
    if (n > 0) {
        if (ob.mode == DM_FIRST_ENTRY) {
            k = 0;
        } else {
            for (int i = 0; i < 4; ++i) {
                if (i >= n) break;
                if (!(t[i] < 0.0)) {
                    k = i;
                    break;
                }
            }
        }
    }
    if (k >= 0) {
        double tk = k == 0 ? t[0] : (k == 1 ? t[1] : (k == 2 ? t[2] : t[3]));
        uint32_t sk = k == 0 ? sub[0] : (k == 1 ? sub[1] : (k == 2 ? sub[2] : sub[3]));
        c.t = leaf_t_to_world(L, o, ry, tk);
        c.aux = sk;
        c.ok = true;
    }
    return c;
}

__device__ __noinline__ Candidate eval_csg_object(const DevScene &sc, const DevObject &ob, d3 o, d3 d, Guards &gd)
{
    IvList res;
    res.n = 0;
    Candidate c;
    c.ok = false;
    c.t = 0.0;
    c.aux = 0u;
    if (!csg_eval(sc, ob, o, d, res, gd)) return c;
    return pick_from_list(res, ob.mode);
}

// Does the ray need the walk through a mesh of this object?  It must pass the Mesh::new box
// (mesh.rs:32-42,117: no box hit, no intersection) and enter the triangles' own bounds before
// the nearest hit found among the other objects (`have_hit`, `best_t`; slack keeps ties in play).
RT_DEV bool mesh_object_wanted(const DevScene &sc, const DevObject &ob, d3 o, d3 d, bool have_hit, double best_t)
{
    const DevLeaf &F = sc.leaves[ob.frame];
    LeafRay ry = to_leaf(F, o, d);
    d3 inv = mk3(1.0 / ry.d.x, 1.0 / ry.d.y, 1.0 / ry.d.z);
    // object-space length per unit of world length (d need not have unit length: rt_trace_rays)
    const double scale = F.identity ? 1.0 : len3(xf_dir(F.W, d)) / len3(d);
    const double bound = have_hit ? best_t * scale * (1.0 + 1e-7) + 1e-7 : 1.0e300;
    bool any = false;
    for (int mi = 0; mi < ob.n_meshes; ++mi) {
        const DevMesh &m = sc.meshes[ob.mesh0 + mi];
        double t0, t1;
        any = any || (mesh_box_hit(m, ry.o, inv) && mesh_clip(m, ry.o, inv, t0, t1) && t0 <= bound);
    }
    return any;
}

// HomogenousMedium::intersect (homogenous_medium.rs:29-57)
template <bool CSG, bool TORUS>
__device__ __noinline__ Candidate eval_medium_object(const DevScene &sc, const DevObject &ob, uint32_t obj_index, d3 o,
                                                     d3 d, const PassParams &pp, uint32_t pixel, Guards &gd)
{
    Candidate c;
    c.ok = false;
    c.t = 0.0;
    c.aux = 0u;
    const DevLeaf &F = sc.leaves[ob.frame];
    LeafRay ry = to_leaf(F, o, d);
    // a world-space distance along the ray is `scale` times longer in the medium's space
    double scale = F.identity ? 1.0 : len3(xf_dir(F.W, d));
    IvList res;
    res.n = 0;
    if (ob.prog_len == 0) {
        leaf_world_intervals<TORUS>(sc.leaves[ob.leaf0], 0u, o, d, res, gd);
    } else if (!CSG || !csg_eval(sc, ob, o, d, res, gd)) {
        return c;
    }
    if (res.n == 0) return c;
    int first = 0;
    while (first < res.n && res.t[2 * first] < 0.0 && res.t[2 * first + 1] < 0.0) ++first; // skip_negative_intervals
    double dlen = len3(ry.d);
    uint32_t k = 0;
    for (int i = first; i < res.n; ++i) {
        double at = fmax(res.t[2 * i] * scale, 0.0), bt = res.t[2 * i + 1] * scale;
        double distance = len3(ry.d * (bt - at));
        rt_rng_block blk = rt_rng_draw(pp.seed, pixel, pp.stratum, pp.depth, RT_RNG_SITE_MEDIUM + k, obj_index);
        ++k;
        double hit_distance = -(1.0 / ob.density) * log(rt_rng_u0(&blk));
        if (hit_distance < distance) {
            double tm = at + hit_distance / dlen;
            c.t = leaf_t_to_world(F, o, ry, tm);
            c.ok = true;
            return c;
        }
    }
    return c;
}

// Pass 1 of the nearest-hit search: every object except the triangle work.
// A ray that passes the bounding box of a mesh object is appended to
// `mesh_queue`; k_mesh then walks the mesh grids and merges its result into the
// hit record.  CSG / MEDIUM prune code (and registers) the
// scene does not need.
struct HitRec {
    double t; // -1: nothing hit
    d3 n;
    double u, v;
    int obj;
};

// Ray::trace (system.rs:122-133) over every object except the triangle work; returns whether
// the ray still has to visit a mesh.
template <bool CSG, bool MEDIUM, bool TORUS>
RT_DEV bool intersect_analytic(const DevScene &sc, d3 o, d3 d, uint32_t pixel, const PassParams &pp, HitRec &h, Guards &gd)
{
    bool wants_mesh = false;
    // first object wins ties, t must be < f64::MAX
    double best_t = DBL_MAX;
    int best_obj = -1;
    uint32_t best_aux = 0u;
    const d3 inv_w = (CSG || MEDIUM || TORUS) ? mk3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z) : mk3(0, 0, 0);
    for (int k = 0; k < sc.n_objects; ++k) {
        const DevObject &ob = sc.objects[k];
        Candidate c;
        c.ok = false;
        c.t = 0.0;
        c.aux = 0u;
        if ((CSG || MEDIUM || TORUS) && ob.has_bounds) {
            // the expensive objects (CSG programs, fog boundaries, tori) carry the world bounds of all
            // their leaves: a ray that misses those, or has them behind it, meets no leaf at t >= 0
            double t0 = 0.0, t1 = 1.0e300; // NaNs (0 * inf) drop out of fmin / fmax: the box then counts as met
            const double ax = (ob.wmin[0] - o.x) * inv_w.x, bx = (ob.wmax[0] - o.x) * inv_w.x;
            t0 = fmax(t0, fmin(ax, bx));
            t1 = fmin(t1, fmax(ax, bx));
            const double ay = (ob.wmin[1] - o.y) * inv_w.y, by = (ob.wmax[1] - o.y) * inv_w.y;
            t0 = fmax(t0, fmin(ay, by));
            t1 = fmin(t1, fmax(ay, by));
            const double az = (ob.wmin[2] - o.z) * inv_w.z, bz = (ob.wmax[2] - o.z) * inv_w.z;
            t0 = fmax(t0, fmin(az, bz));
            t1 = fmin(t1, fmax(az, bz));
            if (t0 > t1 + 1e-9 * (fabs(t1) + 1.0)) continue;
        }
        switch (ob.kind) {
        case DO_LEAF: c = eval_leaf_object<TORUS>(sc, ob, o, d, gd); break;
        case DO_CSG:
            if (CSG) c = eval_csg_object(sc, ob, o, d, gd);
            break;
        case DO_MESH: wants_mesh = true; break; // decided below, once the other objects' nearest hit is known
        default:
            if (MEDIUM) c = eval_medium_object<CSG, TORUS>(sc, ob, (uint32_t)k, o, d, pp, pixel, gd);
            break;
        }
        // Object::intersect keeps t >= 0 (object.rs:50-54)
        if (c.ok && c.t >= 0.0 && c.t < DBL_MAX && (best_obj < 0 || c.t < best_t)) {
            best_t = c.t;
            best_obj = k;
            best_aux = c.aux;
        }
    }

    if (wants_mesh) {
        wants_mesh = false;
        for (int k = 0; k < sc.n_objects; ++k) {
            const DevObject &ob = sc.objects[k];
            if (ob.kind == DO_MESH) wants_mesh = wants_mesh || mesh_object_wanted(sc, ob, o, d, best_obj >= 0, best_t);
        }
    }

    d3 n = mk3(0.0, 0.0, 0.0);
    double u = 0.0, v = 0.0;
    if (best_obj >= 0) {
        const DevObject &ob = sc.objects[best_obj];
        const bool need_uv = ob.needs_uv || pp.force_uv;
        d3 pw = o + d * best_t;
        if (ob.kind == DO_LEAF || ob.kind == DO_CSG) {
            const DevLeaf &L = sc.leaves[ob.leaf0 + (best_aux & 0xFFu)];
            LeafRay ry = to_leaf(L, o, d);
            d3 ql = L.identity ? pw : xf_point(L.W, pw);
            d3 nl;
            leaf_surface(L, ry, ql, best_aux, need_uv, nl, u, v);
            n = L.identity ? nl : normalize3(xf_normal(L.W, nl));
            if (best_aux & TAG_NEG) n = -n;
        } else {
            const DevLeaf &F = sc.leaves[ob.frame];
            n = F.identity ? mk3(1.0, 0.0, 0.0) : normalize3(xf_normal(F.W, mk3(1.0, 0.0, 0.0)));
        }
    }
    h.t = best_obj >= 0 ? best_t : -1.0;
    h.n = n;
    h.u = u;
    h.v = v;
    h.obj = best_obj;
    return wants_mesh;
}

template <bool CSG, bool MEDIUM, bool TORUS>
__global__ void __launch_bounds__(RT_BLOCK) RT_ISECT_OCC k_intersect(DevScene sc_mem, RayQueue q, const uint32_t *count_ptr, HitBuf hit,
                                                         PassParams pp, int32_t *primary_ids, uint32_t *guards_out,
                                                         uint32_t *mesh_queue, uint32_t *mesh_count)
{
    extern __shared__ double4 scene_lds[];
    const DevScene sc = scene_in_lds(sc_mem, reinterpret_cast<unsigned char *>(scene_lds));
    __shared__ uint32_t append_lds[RT_BLOCK / 64 + 1];
    const uint32_t count = *count_ptr;
    const uint32_t stride = gridDim.x * blockDim.x;
    Guards gd;
    for (int k = 0; k < 5; ++k) gd.g[k] = 0u;
    gd.overflow = 0u;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {
        const uint32_t i = base + threadIdx.x;
        bool wants_mesh = false;
        d3 o, d;
        uint32_t pixel = 0u;
        if (i < count && !fetch_ray(sc, q, pp, i, o, d, pixel)) {
            hit.obj[i] = RT_HIT_PADDING;
        } else if (i < count) {
            HitRec h;
            wants_mesh = intersect_analytic<CSG, MEDIUM, TORUS>(sc, o, d, pixel, pp, h, gd);
            hit.t[i] = h.t;
            hit.n[0][i] = h.n.x;
            hit.n[1][i] = h.n.y;
            hit.n[2][i] = h.n.z;
            hit.uv[0][i] = h.u;
            hit.uv[1][i] = h.v;
            hit.obj[i] = h.obj;
            if (primary_ids) primary_ids[ray_slot(q, i)] = h.obj;
        }
        uint32_t at = block_append(wants_mesh, mesh_count, append_lds);
        if (wants_mesh) mesh_queue[at] = i;
    }
    for (int k = 0; k < 5; ++k)
        if (gd.g[k]) atomicAdd(&guards_out[k], gd.g[k]);
    if (gd.overflow) atomicAdd(&guards_out[5], gd.overflow);
}

// normal / uv of a mesh hit (mesh.rs:97-112 + the Composite's to_world)
RT_DEV void mesh_hit_surface(const DevScene &sc, d3 o, d3 d, int best_obj, uint32_t best_aux, d3 &n, double &u, double &v)
{
    const DevObject &ob = sc.objects[best_obj];
    const DevLeaf &F = sc.leaves[ob.frame];
    const DevMesh &m = sc.meshes[ob.mesh0 + (best_aux >> 24)];
    LeafRay ry = to_leaf(F, o, d);
    uint32_t tri = best_aux & 0xFFFFFFu;
    double tt = 0.0;
    u = 0.0;
    v = 0.0;
    tri_hit(sc.tris + ((size_t)m.tri0 + tri) * 9, ry.o, ry.d, tt, u, v);
    const uint32_t *tn = sc.tri_n + ((size_t)m.tri0 + tri) * 3;
    const double *nb = sc.normals + (size_t)m.normal0 * 3;
    d3 n0 = mk3(nb[3 * tn[0]], nb[3 * tn[0] + 1], nb[3 * tn[0] + 2]);
    d3 n1 = mk3(nb[3 * tn[1]], nb[3 * tn[1] + 1], nb[3 * tn[1] + 2]);
    d3 n2 = mk3(nb[3 * tn[2]], nb[3 * tn[2] + 1], nb[3 * tn[2] + 2]);
    // mesh.rs:102-106
    d3 nl = m.smooth ? normalize3((1.0 - u - v) * n0 + u * n1 + v * n2) : normalize3(n0 + n1 + n2);
    n = F.identity ? nl : normalize3(xf_normal(F.W, nl));
}

RT_DEV void write_mesh_hit(const DevScene &sc, const RayQueue &q, const HitBuf &hit, int32_t *primary_ids, uint32_t i,
                           d3 o, d3 d, double best_t, int best_obj, uint32_t best_aux)
{
    d3 n;
    double u, v;
    mesh_hit_surface(sc, o, d, best_obj, best_aux, n, u, v);
    hit.t[i] = best_t;
    hit.n[0][i] = n.x;
    hit.n[1][i] = n.y;
    hit.n[2][i] = n.z;
    hit.uv[0][i] = u;
    hit.uv[1][i] = v;
    hit.obj[i] = best_obj;
    if (primary_ids) primary_ids[ray_slot(q, i)] = best_obj;
}

// Pass 2: the rays queued by k_intersect against the mesh objects
// (Composite::intersect over Mesh children, composite.rs:20-34, mesh.rs:116-128).
// A mesh hit replaces the recorded hit when it is nearer, or equally near and
// earlier in the object list - the order Ray::trace's min_by would have seen.
//
// Each lane walks its ray through the mesh's uniform grid.  The grid's occupancy
// words, distance nibbles and (when they fit) cell starts sit in LDS: a cell at
// Chebyshev distance k from the nearest occupied cell lets the ray jump k-1 cells
// ahead, cells next to occupied ones are crossed one boundary at a time (3D-DDA),
// and only an occupied cell goes to memory, for its run of triangle records.  The
// walk ends when the nearest hit so far lies before the exit of the current cell:
// every triangle with a nearer hit touches a cell already visited (cells are padded
// by far more than the rounding of the walk).
//
// The walk is done in grid coordinates (cells are unit cubes, the ray parameter is
// unchanged): p_g(t) = og + dg * t.
//
// Waves take blocks of 64 queue entries, interleaved; with more than one block per
// wave (large queues) a wave refills its finished lanes once `refill` of them are idle.
#ifdef RT_MESH_PROF
__device__ unsigned long long g_pf[12]; // cycles: service, walk, test; rounds, services, lanes walking, lanes testing, waves
#endif
// The walk state is kept in plain scalars (not an aggregate): selecting a field by axis
// would otherwise be compiled into an indexed access to a copy in scratch memory.
#define WALK_LOCATE_AXIS(OG, DG, IG, N, C, TM)                                                                         \
    do {                                                                                                               \
        double f_ = floor((OG) + (DG) * w_tc);                                                                         \
        const double top_ = (double)((N) - 1u);                                                                        \
        f_ = f_ > 0.0 ? f_ : 0.0; /* also sends a NaN to cell 0 */                                                     \
        f_ = f_ > top_ ? top_ : f_;                                                                                    \
        (C) = (int)f_;                                                                                                 \
        const double edge_ = (double)((C) + ((DG) > 0.0 ? 1 : 0));                                                     \
        (TM) = (DG) != 0.0 ? (edge_ - (OG)) * (IG) : 1.0e300; /* parallel to the planes: never crosses one */          \
    } while (0)
// cell and boundary parameters of the point at ray parameter w_tc
#define WALK_LOCATE()                                                                                                  \
    do {                                                                                                               \
        WALK_LOCATE_AXIS(ogx, dgx, igx, nx, cx, tmx);                                                                  \
        WALK_LOCATE_AXIS(ogy, dgy, igy, ny, cy, tmy);                                                                  \
        WALK_LOCATE_AXIS(ogz, dgz, igz, nz, cz, tmz);                                                                  \
    } while (0)
// cross the nearest cell boundary; OK = false when that leaves the grid or passes t_end
#define WALK_STEP(OK)                                                                                                  \
    do {                                                                                                               \
        const bool ax_ = tmx <= tmy && tmx <= tmz;                                                                     \
        const bool ay_ = !ax_ && tmy <= tmz;                                                                           \
        const bool az_ = !ax_ && !ay_;                                                                                 \
        const double t_exit_ = ax_ ? tmx : (ay_ ? tmy : tmz);                                                          \
        const double ia_ = ax_ ? igx : (ay_ ? igy : igz);                                                              \
        const int ca_ = (ax_ ? cx : (ay_ ? cy : cz)) + (ia_ > 0.0 ? 1 : -1);                                           \
        const uint32_t na_ = ax_ ? nx : (ay_ ? ny : nz);                                                               \
        (OK) = !(ca_ < 0 || ca_ >= (int)na_ || t_exit_ > t_end);                                                       \
        if (OK) {                                                                                                      \
            w_tc = t_exit_;                                                                                            \
            const double step_ = fabs(ia_);                                                                            \
            cx = ax_ ? ca_ : cx;                                                                                       \
            cy = ay_ ? ca_ : cy;                                                                                       \
            cz = az_ ? ca_ : cz;                                                                                       \
            tmx = ax_ ? tmx + step_ : tmx;                                                                             \
            tmy = ay_ ? tmy + step_ : tmy;                                                                             \
            tmz = az_ ? tmz + step_ : tmz;                                                                             \
        }                                                                                                              \
    } while (0)

//
// KD selects the other acceleration structure, a kd-tree per mesh: 16-byte nodes read with one
// load, a front-to-back walk with the deferred far children on a per-lane stack in LDS (8-byte
// entries: node and the far bound of its cell as an f32 rounded up; cells come in ray order, so
// the near bound of a popped cell is the far bound of the leaf just finished), leaf triangles as
// contiguous 80-byte records.  Everything around the walk - queue blocks, refill, job records,
// the merge - is shared.  rt_scene_create picks the structure per scene.
template <bool KD>
__global__ void __launch_bounds__(RT_MESH_LANES) RT_MESH_OCC k_mesh(DevScene sc, RayQueue q, const uint32_t *mesh_queue,
                                                      const uint32_t *mesh_count, HitBuf hit, int32_t *primary_ids,
                                                      uint32_t refill, uint32_t block_arg, uint32_t resident_waves,
                                                      PassParams pp)
{
    extern __shared__ double4 mesh_lds[]; // job records; then the grid tables, or the kd stacks
    const uint32_t n_jobs_lds = sc.n_jobs < DEV_MESH_JOBS_LDS ? sc.n_jobs : DEV_MESH_JOBS_LDS;
    DevMeshJob *jobs_lds = reinterpret_cast<DevMeshJob *>(mesh_lds);
    uint2 *grid_lds = reinterpret_cast<uint2 *>(jobs_lds + n_jobs_lds);
    uint32_t *dist_lds = reinterpret_cast<uint32_t *>(grid_lds + (KD ? 0u : sc.n_grid_words));
    uint32_t *starts_lds = dist_lds + (KD ? 0u : sc.n_dist_words);
    const bool starts_in_lds = !KD && sc.n_cell_starts <= DEV_GRID_MAX_STARTS;
    uint2 *stk = grid_lds + threadIdx.x; // KD: this lane's stack column, entry e at stk[e * RT_MESH_LANES]
    {
        const uint32_t *jw = reinterpret_cast<const uint32_t *>(sc.jobs);
        uint32_t *jl = reinterpret_cast<uint32_t *>(jobs_lds);
        for (uint32_t i = threadIdx.x; i < n_jobs_lds * (uint32_t)(sizeof(DevMeshJob) / 4); i += blockDim.x) jl[i] = jw[i];
        if (!KD) {
            const uint2 *gw = reinterpret_cast<const uint2 *>(sc.grid_bits);
            for (uint32_t i = threadIdx.x; i < sc.n_grid_words; i += blockDim.x) grid_lds[i] = gw[i];
            for (uint32_t i = threadIdx.x; i < sc.n_dist_words; i += blockDim.x) dist_lds[i] = sc.grid_dist[i];
            if (starts_in_lds)
                for (uint32_t i = threadIdx.x; i < sc.n_cell_starts; i += blockDim.x) starts_lds[i] = sc.cell_start[i];
        }
    }
    __syncthreads();
// job record `J` (an index): from LDS when it is one of the first DEV_MESH_JOBS_LDS
#define JOB(J) (*((J) < (int)n_jobs_lds ? (const DevMeshJob *)(jobs_lds + (J)) : (sc.jobs + (J))))

    // every wave takes blocks of 64 queue entries, interleaved with the other waves: neighbouring
    // rays cost about the same, so contiguous chunks would leave whole waves with the expensive ones
    const uint32_t count = *mesh_count;
    const uint32_t n_waves = gridDim.x * (RT_MESH_LANES / 64);
    const uint32_t wave = blockIdx.x * (RT_MESH_LANES / 64) + (threadIdx.x >> 6);
    const uint64_t lane_lt = (1ull << lane_id()) - 1ull;
    // short queues: fewer rays per wave (16 or 32 lanes used), so that the slowest wave of the pass is
    // done sooner; the queue length is only known here, the host sized the grid from an upper bound
    uint32_t block = block_arg;
    if (block == 0u) {
        const uint32_t spread = n_waves < resident_waves ? n_waves : resident_waves;
        block = 16u;
        while (block < 64u && (uint64_t)block * spread < count) block <<= 1;
    }
    // a wave with a single block has nothing to refill from: it services its lanes twice, to load
    // the block and to store the results once every lane is done
    if ((uint64_t)n_waves * block >= count) refill = 65u;
    uint64_t cursor = (uint64_t)wave * block; // next queue entry of this wave
    uint64_t block_end = cursor + block;      // end of its current block

    // lane state
    bool active = false, have_ray = false;
    uint32_t ri = 0;
    d3 o = mk3(0, 0, 0), d = mk3(0, 0, 1); // ray in the current object's space
    int jb = -1;                           // current job = (object, mesh) pair
    int best_job = -1;                     // job of the nearest mesh hit
    double best_t = 0.0, wbound = 0.0; // nearest world hit so far; the same as a bound in object space
    int best_obj = -1;
    uint32_t best_aux = 0u;
    bool replaced = false;
    double obj_t = 0.0;
    uint32_t obj_aux = 0u;
    int obj_job = -1;
    bool obj_found = false;
    // walk state: grid-space origin, direction, 1 / direction; ray parameter at the next cell
    // boundary per axis; where the current cell was entered; where the walk ends
    double ogx = 0, ogy = 0, ogz = 0, dgx = 0, dgy = 0, dgz = 0, igx = 0, igy = 0, igz = 0;
    double tmx = 0, tmy = 0, tmz = 0, w_tc = 0, t_end = 0;
    int cx = 0, cy = 0, cz = 0;
    uint32_t nx = 1, ny = 1, nz = 1, bits0 = 0, cell0 = 0, dist0 = 0;
    // KD walk state: current node, stack height, the ray parameter range of the current cell; the tree's
    // nodes and records (igx, igy, igz hold 1 / direction then)
    uint32_t kd_node = 0u, kd0 = 0u, kidx0 = 0u;
    int kd_sp = 0;
    double kd_tmin = 0.0, kd_tmax = 0.0;
    double mt = 1.0e300;
    uint32_t mtri = 0xFFFFFFFFu;

#ifdef RT_MESH_PROF
    long long pf_t = clock64(), pf_service = 0, pf_walk = 0, pf_test = 0, pf_n = 0, pf_a = 0, pf_b = 0, pf_c = 0, pf_d = 0;
    unsigned long long pf_rounds = 0, pf_services = 0, pf_lanes_walk = 0, pf_lanes_test = 0, pf_tests = 0;
#define PF_MARK(ACC) do { long long n_ = clock64(); ACC += n_ - pf_t; pf_t = n_; } while (0)
#else
#define PF_MARK(ACC) do { } while (0)
#endif
    for (;;) {
        // ---------------------------------------------------------------- service finished lanes
        const bool wants = !active && (have_ray || (cursor < count && lane_id() < block));
        const uint32_t n_wants = (uint32_t)__popcll(__ballot(wants));
        const bool none_active = __ballot(active) == 0ull;
        if (none_active && n_wants == 0u) break;
        if (n_wants >= refill || none_active) {
            d3 wo = mk3(0, 0, 0), wd = mk3(0, 0, 1);
            bool advance = false;
            if (!active && have_ray) {
                uint32_t px_;
                (void)fetch_ray(sc, q, pp, ri, wo, wd, px_);
                // result of the walk that just ended: min_by keeps the first minimum
                if (mtri != 0xFFFFFFFFu && (!obj_found || mt < obj_t)) {
                    obj_t = mt;
                    obj_aux = mtri;
                    obj_job = jb;
                    obj_found = true;
                }
                advance = true;
            }
            PF_MARK(pf_a);
            const bool takes = !active && !have_ray && lane_id() < block;
            const uint64_t empty = __ballot(takes);
            if (takes) {
                const uint64_t j = cursor + (uint64_t)__popcll(empty & lane_lt);
                if (j < block_end && j < count) {
                    ri = mesh_queue[j];
                    uint32_t px_;
                    (void)fetch_ray(sc, q, pp, ri, wo, wd, px_);
                    best_obj = hit.obj[ri];
                    best_t = best_obj >= 0 ? hit.t[ri] : DBL_MAX;
                    best_aux = 0u;
                    replaced = false;
                    have_ray = true;
                    jb = -1;
                    obj_found = false;
                    advance = true;
                }
            }
            cursor += (uint64_t)__popcll(empty);
            if (cursor >= block_end) { // block used up (lanes left without a ray take from the next one)
                cursor = block_end + (uint64_t)(n_waves - 1u) * block;
                block_end = cursor + block;
            }
            PF_MARK(pf_b);
            if (advance) {
                // next (object, mesh) job whose boxes the ray passes; leaving an object merges its
                // nearest triangle into the ray's hit
                bool found = false;
                for (int guard = 0; guard < 1 << 20; ++guard) {
                    const int prev_obj = jb >= 0 ? JOB(jb).object : -1;
                    ++jb;
                    const int next_obj = jb < (int)sc.n_jobs ? JOB(jb).object : -2;
                    if (next_obj != prev_obj) {
                        if (prev_obj >= 0 && obj_found) {
                            const DevMeshJob &P = JOB(jb - 1);
                            double tw = obj_t; // Intersection::to_world's t (system.rs:190-195)
                            if (!P.identity) tw = signum(obj_t) * len3(xf_point(P.O, o + d * obj_t) - wo);
                            if (tw >= 0.0 && tw < DBL_MAX &&
                                (best_obj < 0 || tw < best_t || (tw == best_t && prev_obj < best_obj))) {
                                best_t = tw;
                                best_obj = prev_obj;
                                best_aux = obj_aux;
                                best_job = obj_job;
                                replaced = true;
                            }
                        }
                        if (next_obj < 0) break;
                        const DevMeshJob &J = JOB(jb);
                        // Ray::transform (system.rs:145-152); a hit farther than the nearest one so far
                        // cannot win: object-space distance = world distance * |W d|; slack keeps ties
                        double scale = 1.0;
                        if (J.identity) {
                            o = wo;
                            d = wd;
                        } else {
                            o = xf_point(J.W, wo);
                            const d3 dl = xf_dir(J.W, wd);
                            scale = len3(dl) / len3(wd); // per unit of world length: wd need not be a unit vector
                            d = normalize3(dl);
                        }
                        wbound = best_obj >= 0 ? best_t * scale * (1.0 + 1e-7) + 1e-7 : 1.0e300;
                        obj_found = false;
                    }
                    const DevMeshJob &J = JOB(jb);
                    const d3 inv = mk3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                    if (!box_hit(J.qmin, J.qmax, o, inv)) continue;
                    double t0, t1;
                    if (!box_clip(J.bmin, J.bmax, o, inv, t0, t1)) continue;
                    if (wbound < t1) t1 = wbound;
                    if (obj_found && obj_t < t1) t1 = obj_t; // a later mesh must be strictly nearer
                    if (t0 > t1) continue;
                    if (KD) {
                        igx = inv.x;
                        igy = inv.y;
                        igz = inv.z;
                        kd0 = J.kd0;
                        kidx0 = J.kd_index0;
                        kd_node = 0u; // node indices are mesh-relative
                        kd_sp = 0;
                        kd_tmin = t0;
                        kd_tmax = t1;
                        mt = 1.0e300;
                        mtri = 0xFFFFFFFFu;
                        found = true;
                        break;
                    }
                    // enter the grid at t0
                    ogx = (o.x - J.bmin[0]) * J.inv_cell[0];
                    ogy = (o.y - J.bmin[1]) * J.inv_cell[1];
                    ogz = (o.z - J.bmin[2]) * J.inv_cell[2];
                    dgx = d.x * J.inv_cell[0];
                    dgy = d.y * J.inv_cell[1];
                    dgz = d.z * J.inv_cell[2];
                    igx = 1.0 / dgx;
                    igy = 1.0 / dgy;
                    igz = 1.0 / dgz;
                    nx = J.n[0];
                    ny = J.n[1];
                    nz = J.n[2];
                    w_tc = t0;
                    t_end = t1;
                    bits0 = J.bits0;
                    cell0 = J.cell0;
                    dist0 = J.dist0;
                    WALK_LOCATE();
                    mt = 1.0e300;
                    mtri = 0xFFFFFFFFu;
                    found = true;
                    break;
                }
                PF_MARK(pf_c);
                if (found) {
                    active = true;
                } else {
                    if (replaced) {
                        // normal / uv of the hit (mesh.rs:97-112 + the Composite's to_world)
                        const DevMeshJob &B = JOB(best_job);
                        d3 bo = wo, bd = wd;
                        if (!B.identity) {
                            bo = xf_point(B.W, wo);
                            bd = normalize3(xf_dir(B.W, wd));
                        }
                        const size_t tr = (size_t)B.tri0 + best_aux;
                        const double *nb = sc.tri_nrm + tr * 9;
                        const d3 n0 = mk3(nb[0], nb[1], nb[2]), n1 = mk3(nb[3], nb[4], nb[5]), n2 = mk3(nb[6], nb[7], nb[8]);
                        double tt = 0.0, u = 0.0, v = 0.0;
                        tri_hit(sc.tris + tr * 9, bo, bd, tt, u, v);
                        d3 nl = B.smooth ? normalize3((1.0 - u - v) * n0 + u * n1 + v * n2) : normalize3(n0 + n1 + n2);
                        d3 n = B.identity ? nl : normalize3(xf_normal(B.W, nl));
                        hit.t[ri] = best_t;
                        hit.n[0][ri] = n.x;
                        hit.n[1][ri] = n.y;
                        hit.n[2][ri] = n.z;
                        hit.uv[0][ri] = u;
                        hit.uv[1][ri] = v;
                        hit.obj[ri] = best_obj;
                        if (primary_ids) primary_ids[ray_slot(q, ri)] = best_obj;
                    }
                    have_ray = false;
                }
            }
            PF_MARK(pf_d);
#ifdef RT_MESH_PROF
            ++pf_services;
#endif
            continue; // re-evaluate: lanes whose new ray needed no walk can be filled again
        }
#ifdef RT_MESH_PROF
        ++pf_rounds;
        pf_lanes_walk += __popcll(__ballot(active));
#endif

        if (KD) {
            // ------------------------------------------------------------ kd-tree: down to the next leaf, test it, pop
            // "while-while": all lanes first walk down to their next leaf (cheap steps), then the wave
            // tests triangles together (the expensive part).
            if (active) {
                const uint4 *__restrict__ kd = reinterpret_cast<const uint4 *>(sc.kd + kd0); // one 16-byte load per node
                uint4 raw = kd[kd_node];
                for (int step = 0; step < 64 && (raw.z & 3u) != 3u; ++step) { // depth <= DEV_KD_STACK
                    const uint32_t axis = raw.z & 3u, right = raw.z >> 2;
                    const double split = __hiloint2double((int)raw.y, (int)raw.x);
                    const double oa = axis == 0 ? o.x : (axis == 1 ? o.y : o.z);
                    const double ia = axis == 0 ? igx : (axis == 1 ? igy : igz);
                    const double da = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
                    const double tplane = (split - oa) * ia;
                    const bool below_first = (oa < split) || (oa == split && da <= 0.0);
                    const uint32_t first = below_first ? kd_node + 1u : right;
                    const uint32_t second = below_first ? right : kd_node + 1u;
                    // slack: a hit within rounding distance of the plane is searched on both sides
                    const double eps = 1e-9 * (fabs(tplane) + 1.0);
                    const bool only_first = (tplane > kd_tmax + eps) || (tplane <= 0.0);
                    const bool only_second = !only_first && (tplane < kd_tmin - eps);
                    if (!only_first && !only_second) {
                        if (kd_sp < DEV_KD_STACK) {
                            stk[kd_sp * RT_MESH_LANES] = make_uint2(second, __float_as_uint(__double2float_ru(kd_tmax)));
                            ++kd_sp;
                        }
                        kd_tmax = tplane + eps;
                    }
                    kd_node = only_second ? second : first;
                    raw = kd[kd_node];
                }
                PF_MARK(pf_walk);
                if ((raw.z & 3u) == 3u) {
                    const uint32_t first = kidx0 + (raw.z >> 2), last = first + raw.w;
                    const double2 *__restrict__ rec = reinterpret_cast<const double2 *>(sc.leaf_tris + (size_t)first * 10);
                    for (uint32_t e = first; e < last; e += RT_MESH_BATCH, rec += 5 * RT_MESH_BATCH) {
                        double2 r[RT_MESH_BATCH][5];
#pragma unroll
                        for (int b = 0; b < RT_MESH_BATCH; ++b) {
                            const double2 *src = (e + (uint32_t)b < last) ? rec + 5 * b : rec; // past the end: repeat the first
#pragma unroll
                            for (int c = 0; c < 5; ++c) r[b][c] = src[c];
                        }
#pragma unroll
                        for (int b = 0; b < RT_MESH_BATCH; ++b) {
                            const d3 v0 = mk3(r[b][0].x, r[b][0].y, r[b][1].x), e1 = mk3(r[b][1].y, r[b][2].x, r[b][2].y),
                                     e2 = mk3(r[b][3].x, r[b][3].y, r[b][4].x);
                            const uint32_t ti = (uint32_t)__double2loint(r[b][4].y);
                            double t, u, v;
                            if (tri_hit_v(v0, e1, e2, o, d, t, u, v)) {
                                if (t < mt || (t == mt && ti < mtri)) { // a repeated record cannot beat itself
                                    mt = t;
                                    mtri = ti;
                                }
                            }
                        }
                    }
                }
                if (kd_sp == 0) {
                    active = false;
                } else {
                    --kd_sp;
                    const uint2 e = stk[kd_sp * RT_MESH_LANES];
                    kd_node = e.x;
                    // the stored bounds are f32 round-ups: step back by more than that rounding
                    kd_tmin = kd_tmax - (2.4e-7 * fabs(kd_tmax) + 1e-9);
                    kd_tmax = (double)__uint_as_float(e.y);
                    if (mt < kd_tmin) active = false; // every remaining cell lies beyond the hit
                }
            }
            PF_MARK(pf_test);
            continue;
        }
        // ---------------------------------------------------------------- walk to the next occupied cell
        // "while-while": all lanes first move to their next occupied cell (LDS only), then the wave
        // tests triangles together (the expensive part).
        if (active) {
            uint32_t cell = 0u;
            bool occupied = false;
            for (int step = 0; step < 3 * DEV_GRID_MAX_AXIS + 3; ++step) {
                cell = ((uint32_t)cz * ny + (uint32_t)cy) * nx + (uint32_t)cx;
                const uint32_t kd = (dist_lds[dist0 + (cell >> 3)] >> ((cell & 7u) * 4u)) & 15u;
                if (kd == 0u) {
                    occupied = true;
                    break;
                }
                if (kd >= 2u) {
                    // nothing within kd-1 cells of this one along any axis
                    const double dmin = fmin(fmin(fabs(igx), fabs(igy)), fabs(igz));
                    w_tc += ((double)kd - 1.01) * dmin;
                    if (w_tc > t_end) {
                        active = false;
                        break;
                    }
                    WALK_LOCATE();
                } else {
                    bool ok;
                    WALK_STEP(ok);
                    if (!ok) {
                        active = false;
                        break;
                    }
                }
            }
            PF_MARK(pf_walk);
#ifdef RT_MESH_PROF
            pf_lanes_test += __popcll(__ballot(active && occupied));
#endif
            if (active && occupied) {
                const uint2 gw = grid_lds[bits0 + (cell >> 5)];
#ifdef RT_MESH_PROF
                pf_tests += 0; // counted below per lane
#endif
                const uint32_t r = cell0 + gw.y + (uint32_t)__popc(gw.x & ((1u << (cell & 31u)) - 1u));
                uint32_t first, last;
                if (starts_in_lds) {
                    first = starts_lds[r];
                    last = starts_lds[r + 1u];
                } else {
                    first = sc.cell_start[r];
                    last = sc.cell_start[r + 1u];
                }
                // RT_MESH_BATCH records per step, all loaded before the first test: a step costs one
                // round trip to memory (~2000 cycles for 64 lanes' scattered records) whatever it tests
                const double2 *__restrict__ rec = reinterpret_cast<const double2 *>(sc.cell_tris + (size_t)first * 10);
                for (uint32_t e = first; e < last; e += RT_MESH_BATCH, rec += 5 * RT_MESH_BATCH) {
                    double2 r[RT_MESH_BATCH][5];
#pragma unroll
                    for (int b = 0; b < RT_MESH_BATCH; ++b) {
                        const double2 *src = (e + (uint32_t)b < last) ? rec + 5 * b : rec; // past the end: repeat the first
#pragma unroll
                        for (int c = 0; c < 5; ++c) r[b][c] = src[c];
                    }
#pragma unroll
                    for (int b = 0; b < RT_MESH_BATCH; ++b) {
                        const d3 v0 = mk3(r[b][0].x, r[b][0].y, r[b][1].x), e1 = mk3(r[b][1].y, r[b][2].x, r[b][2].y),
                                 e2 = mk3(r[b][3].x, r[b][3].y, r[b][4].x);
                        const uint32_t ti = (uint32_t)__double2loint(r[b][4].y);
                        double t, u, v;
                        if (tri_hit_v(v0, e1, e2, o, d, t, u, v)) {
                            if (t < mt || (t == mt && ti < mtri)) { // a repeated record cannot beat itself
                                mt = t;
                                mtri = ti;
                            }
                        }
                    }
                }
                // leave the cell, unless the hit found lies before its far side (slack: rounding)
                const double t_exit = fmin(fmin(tmx, tmy), tmz);
                bool ok = !(mt < t_exit - 1e-9 * (fabs(t_exit) + 1.0));
                if (ok) WALK_STEP(ok);
                if (!ok) active = false;
            }
        }
        PF_MARK(pf_test);
    }
#ifdef RT_MESH_PROF
    if (lane_id() == 0) {
        atomicAdd(&g_pf[0], (unsigned long long)(pf_a + pf_b + pf_c + pf_d));
        atomicAdd(&g_pf[8], (unsigned long long)pf_a);
        atomicAdd(&g_pf[9], (unsigned long long)pf_b);
        atomicAdd(&g_pf[10], (unsigned long long)pf_c);
        atomicAdd(&g_pf[11], (unsigned long long)pf_d);
        atomicAdd(&g_pf[1], (unsigned long long)pf_walk);
        atomicAdd(&g_pf[2], (unsigned long long)pf_test);
        atomicAdd(&g_pf[3], pf_rounds);
        atomicAdd(&g_pf[4], pf_services);
        atomicAdd(&g_pf[5], pf_lanes_walk);
        atomicAdd(&g_pf[6], pf_lanes_test);
        atomicAdd(&g_pf[7], 1ull);
    }
#endif
}

// The same merge for a short queue: one wave per ray (the reference's brute force, mesh.rs:54-64,
// spread over 64 lanes and culled by group bounds), then a wave-wide lexicographic min over
// (t, triangle index).  A lone walking lane takes ~100 microseconds over a grazing ray; a wave
// on one ray takes a few, which is what the late, nearly empty bounce passes need.
RT_DEV void wave_min_hit(double &t, uint32_t &tri)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ot = __shfl_xor(t, off);
        uint32_t oi = __shfl_xor(tri, off);
        if (ot < t || (ot == t && oi < tri)) {
            t = ot;
            tri = oi;
        }
    }
}

// All 64 lanes of the wave hold the same ray.  Returns whether a mesh replaces the hit so far.
RT_DEV bool mesh_brute_wave(const DevScene &sc, d3 o, d3 d, int &best_obj, double &best_t, uint32_t &best_aux)
{
    const uint32_t lane = lane_id();
    bool replaced = false;
    for (int k = 0; k < sc.n_objects; ++k) {
        const DevObject &ob = sc.objects[k];
        if (ob.kind != DO_MESH) continue;
        const DevLeaf &F = sc.leaves[ob.frame];
        LeafRay ry = to_leaf(F, o, d);
        d3 inv = mk3(1.0 / ry.d.x, 1.0 / ry.d.y, 1.0 / ry.d.z);
        double obj_t = 0.0;
        uint32_t obj_aux = 0u;
        bool obj_found = false;
        for (int mi = 0; mi < ob.n_meshes; ++mi) {
            const DevMesh &m = sc.meshes[ob.mesh0 + mi];
            if (!mesh_box_hit(m, ry.o, inv)) continue;
            double lt = 1.0e300;
            uint32_t ltri = 0xFFFFFFFFu;
            // group boxes first, one per lane; then the triangles of the groups the ray meets, one per lane
            for (uint32_t g0 = 0; g0 < m.n_groups; g0 += 64u) {
                bool meets = false;
                if (g0 + lane < m.n_groups) {
                    const double *bx = sc.grp_box + (size_t)(m.grp0 + g0 + lane) * 6;
                    double t0 = 0.0, t1 = 1.0e300; // NaNs (0 * inf) drop out of fmin / fmax: the box then counts as met
                    const double ax = (bx[0] - ry.o.x) * inv.x, bxx = (bx[3] - ry.o.x) * inv.x;
                    t0 = fmax(t0, fmin(ax, bxx));
                    t1 = fmin(t1, fmax(ax, bxx));
                    const double ay = (bx[1] - ry.o.y) * inv.y, by = (bx[4] - ry.o.y) * inv.y;
                    t0 = fmax(t0, fmin(ay, by));
                    t1 = fmin(t1, fmax(ay, by));
                    const double az = (bx[2] - ry.o.z) * inv.z, bz = (bx[5] - ry.o.z) * inv.z;
                    t0 = fmax(t0, fmin(az, bz));
                    t1 = fmin(t1, fmax(az, bz));
                    meets = !(t0 > t1 + 1e-9 * (fabs(t1) + 1.0));
                }
                unsigned long long mask = __ballot(meets);
                while (mask) {
                    // two groups per step, loaded before the first test
                    const uint32_t ga = g0 + (uint32_t)__ffsll((long long)mask) - 1u;
                    mask &= mask - 1ull;
                    const uint32_t gb = mask ? g0 + (uint32_t)__ffsll((long long)mask) - 1u : ga;
                    mask &= mask - 1ull;
                    const size_t sa = (size_t)(m.grp0 + ga) * 64u + lane, sb = (size_t)(m.grp0 + gb) * 64u + lane;
                    const uint32_t ia = sc.grp_idx[sa], ib = gb != ga ? sc.grp_idx[sb] : 0xFFFFFFFFu;
                    const double *a = sc.grp_tris + sa * 9, *b = sc.grp_tris + sb * 9;
                    const d3 a0 = mk3(a[0], a[1], a[2]), a1 = mk3(a[3], a[4], a[5]), a2 = mk3(a[6], a[7], a[8]);
                    const d3 b0 = mk3(b[0], b[1], b[2]), b1 = mk3(b[3], b[4], b[5]), b2 = mk3(b[6], b[7], b[8]);
                    double t, u, v;
                    if (ia != 0xFFFFFFFFu && tri_hit_v(a0, a1, a2, ry.o, ry.d, t, u, v) && (t < lt || (t == lt && ia < ltri))) {
                        lt = t;
                        ltri = ia;
                    }
                    if (ib != 0xFFFFFFFFu && tri_hit_v(b0, b1, b2, ry.o, ry.d, t, u, v) && (t < lt || (t == lt && ib < ltri))) {
                        lt = t;
                        ltri = ib;
                    }
                }
            }
            wave_min_hit(lt, ltri);
            if (ltri != 0xFFFFFFFFu && (!obj_found || lt < obj_t)) { // min_by keeps the first minimum
                obj_t = lt;
                obj_aux = ((uint32_t)mi << 24) | (ltri & 0xFFFFFFu);
                obj_found = true;
            }
        }
        if (obj_found) {
            double tw = leaf_t_to_world(F, o, ry, obj_t);
            if (tw >= 0.0 && tw < DBL_MAX && (best_obj < 0 || tw < best_t || (tw == best_t && k < best_obj))) {
                best_t = tw;
                best_obj = k;
                best_aux = obj_aux;
                replaced = true;
            }
        }
    }
    return replaced;
}

__global__ void __launch_bounds__(RT_BLOCK) k_mesh_brute(DevScene sc, RayQueue q, const uint32_t *mesh_queue,
                                                          const uint32_t *mesh_count, HitBuf hit, int32_t *primary_ids,
                                                          PassParams pp)
{
    const uint32_t count = *mesh_count;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t j = wave; j < count; j += n_waves) {
        const uint32_t i = mesh_queue[j];
        d3 o, d;
        uint32_t px_;
        (void)fetch_ray(sc, q, pp, i, o, d, px_);
        int best_obj = hit.obj[i];
        double best_t = best_obj >= 0 ? hit.t[i] : DBL_MAX;
        uint32_t best_aux = 0u;
        if (!mesh_brute_wave(sc, o, d, best_obj, best_t, best_aux) || lane_id() != 0u) continue;
        write_mesh_hit(sc, q, hit, primary_ids, i, o, d, best_t, best_obj, best_aux);
    }
}

// ------------------------------------------------------------------ shading
// direction.rs:29-39
RT_DEV d3 uniform_sphere(const PassParams &pp, uint32_t pixel)
{
    rt_rng_block blk = rt_rng_draw(pp.seed, pixel, pp.stratum, pp.depth, RT_RNG_SITE_SPHERE, 0u);
    double theta = 2.0 * RT_PI * rt_rng_u0(&blk);
    double phi = acos(1.0 - 2.0 * rt_rng_u1(&blk));
    double sp, cp, st, ct;
    sincos(phi, &sp, &cp);
    sincos(theta, &st, &ct);
    return mk3(sp * ct, sp * st, cp);
}

RT_DEV uint32_t sat_u32(double v)
{
    if (!(v > 0.0)) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

// texture.rs:54-94
RT_DEV d3 texture_color(const DevScene &sc, int tex, double u, double v)
{
    const DevTexture &t = sc.textures[tex];
    if (t.kind == RT_TEX_SOLID) return mk3(t.c1[0], t.c1[1], t.c1[2]);
    if (t.kind == RT_TEX_CHECKERBOARD) {
        double sx = u * t.scale, sy = v * t.scale;
        double fx = sx - trunc(sx), fy = sy - trunc(sy);
        bool xp = (fabs(fx) > 0.5) != (fx < 0.0);
        bool yp = (fabs(fy) > 0.5) != (fy < 0.0);
        double p = (xp != yp) ? 1.0 : 0.0;
        return mk3(t.c1[0], t.c1[1], t.c1[2]) * (1.0 - p) + mk3(t.c2[0], t.c2[1], t.c2[2]) * p;
    }
    double max_x = (double)(t.width - 1), max_y = (double)(t.height - 1);
    uint32_t x = sat_u32(u * t.scale * max_x) % (uint32_t)t.width;
    uint32_t y = sat_u32(v * t.scale * max_y) % (uint32_t)t.height;
    const uint8_t *c = sc.image_data + t.offset + ((size_t)y * (size_t)t.width + x) * 3;
    return mk3((double)c[0] / 255.0, (double)c[1] / 255.0, (double)c[2] / 255.0);
}

RT_DEV double clampd(double lo, double hi, double v) { return fmax(lo, fmin(hi, v)); }

// dielectric.rs:89-108
RT_DEV double fresnel(d3 inc, d3 nrm, double ior)
{
    double cos_i = clampd(-1.0, 1.0, dot3(inc, nrm));
    double eta_i = 1.0, eta_t = ior;
    if (cos_i > 0.0) {
        eta_i = ior;
        eta_t = 1.0;
    }
    double sin_t = eta_i / eta_t * sqrt(fmax(1.0 - cos_i * cos_i, 0.0));
    if (sin_t >= 1.0) return 1.0;
    double cos_t = sqrt(fmax(1.0 - sin_t * sin_t, 0.0));
    cos_i = fabs(cos_i);
    double r_s = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    double r_p = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (r_s * r_s + r_p * r_p) / 2.0;
}

// dielectric.rs:68-86
RT_DEV d3 refract(d3 inc, d3 nrm, double ior)
{
    double cos_i = clampd(-1.0, 1.0, dot3(inc, nrm));
    double eta_i = 1.0, eta_t = ior;
    d3 n = nrm;
    if (cos_i < 0.0) {
        cos_i = -cos_i;
    } else {
        eta_i = ior;
        eta_t = 1.0;
        n = -nrm;
    }
    double eta = eta_i / eta_t;
    double k = 1.0 - eta * eta * (1.0 - cos_i * cos_i);
    if (k < 0.0) return mk3(0.0, 0.0, 0.0);
    return inc * eta + n * (eta * cos_i - sqrt(k));
}

RT_DEV d3 reflect3(d3 d, d3 n) { return d - (n * 2.0) * dot3(d, n); } // direction.rs:75-77

// One path segment's end (system.rs:112-120,135-142): either the path goes on (returns true; `o`,
// `d`, `thr` become the scattered ray and the updated attenuation) or it ends seeing `X`.
RT_DEV bool shade_one(const DevScene &sc, d3 &o, d3 &d, d3 &thr, uint32_t pixel, const HitRec &h, const PassParams &pp,
                      d3 bg, d3 &X)
{
    X = bg; // what the path sees where it ends
    if (h.obj < 0) return false;
    const d3 n = h.n;
    const double u = h.u, v = h.v;
    const DevMaterial &m = sc.materials[sc.objects[h.obj].material];
    d3 p = o + d * h.t; // RayHit::point (system.rs:173-175)
    d3 att = mk3(1.0, 1.0, 1.0);
    d3 no = mk3(0, 0, 0), nd = mk3(0, 0, 0);
    switch (m.kind) {
    case RT_MAT_LAMBERTIAN: {
        // lambertian.rs:20-31
        no = p + n * pp.bias;
        d3 target = p + n + uniform_sphere(pp, pixel);
        nd = normalize3(target - p);
        att = texture_color(sc, m.texture, u, v);
        break;
    }
    case RT_MAT_METAL: {
        // metal.rs:21-32
        d3 refl = normalize3(reflect3(d, n));
        d3 fz = m.fuzz * uniform_sphere(pp, pixel);
        no = p + n * pp.bias;
        nd = normalize3(refl + fz);
        att = texture_color(sc, m.texture, u, v);
        break;
    }
    case RT_MAT_DIELECTRIC: {
        // dielectric.rs:25-53
        bool outside = dot3(d, n) < 0.0;
        d3 b = n * pp.bias;
        double kr = fresnel(d, n, m.ior);
        rt_rng_block blk = rt_rng_draw(pp.seed, pixel, pp.stratum, pp.depth, RT_RNG_SITE_DIELECTRIC, 0u);
        d3 fz = m.fuzz * uniform_sphere(pp, pixel);
        if (rt_rng_u0(&blk) < kr) {
            nd = normalize3(reflect3(d, n) + fz);
            no = outside ? p + b : p - b;
        } else {
            nd = normalize3(refract(d, n, m.ior) + fz);
            no = outside ? p - b : p + b;
        }
        break;
    }
    case RT_MAT_DIFFUSE_LIGHT: {
        // diffuse_light.rs:19-26 + system.rs:139: emit + background
        X = m.intensity * texture_color(sc, m.texture, u, v) + bg;
        return false;
    }
    default: {
        // isotropic.rs:19-26
        no = p;
        nd = uniform_sphere(pp, pixel);
        att = texture_color(sc, m.texture, u, v);
        break;
    }
    }
    thr = mk3(thr.x * att.x, thr.y * att.y, thr.z * att.z);
    o = no;
    d = nd;
    // the scattered ray has depth + 1; Ray::cast returns the background at max_depth
    return pp.depth + 1u < pp.max_depth;
}

__global__ void __launch_bounds__(RT_BLOCK_WIDE) k_shade(DevScene sc_mem, RayQueue qin, const uint32_t *count_in_ptr, HitBuf hit,
                                                     RayQueue qout, uint32_t *count_out, double *fb, PassParams pp)
{
    extern __shared__ double4 scene_lds[];
    const DevScene sc = scene_in_lds(sc_mem, reinterpret_cast<unsigned char *>(scene_lds));
    __shared__ uint32_t append_lds[RT_BLOCK_WIDE / 64 + 1];
    const uint32_t count = *count_in_ptr;
    const uint32_t stride = gridDim.x * blockDim.x;
    const d3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        d3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), thr = mk3(0, 0, 0);
        uint32_t slot = 0, pixel = 0;
        HitRec h;
        h.obj = i < count ? hit.obj[i] : RT_HIT_PADDING;
        if (h.obj != RT_HIT_PADDING) {
            h.t = 0.0;
            h.n = mk3(0, 0, 0);
            h.u = h.v = 0.0;
            if (qin.primary) {
                thr = mk3(1.0, 1.0, 1.0);
                (void)primary_ray(sc, qin, pp, i, o, d, pixel);
                slot = qin.row_major ? pixel : i;
            } else {
                slot = qin.slot[i];
                pixel = qin.pixel[i];
                thr = mk3(qin.thr[0][i], qin.thr[1][i], qin.thr[2][i]);
                if (h.obj >= 0) {
                    o = mk3(qin.o[0][i], qin.o[1][i], qin.o[2][i]);
                    d = mk3(qin.d[0][i], qin.d[1][i], qin.d[2][i]);
                }
            }
            if (h.obj >= 0) {
                h.t = hit.t[i];
                h.n = mk3(hit.n[0][i], hit.n[1][i], hit.n[2][i]);
                h.u = hit.uv[0][i];
                h.v = hit.uv[1][i];
            }
            d3 X;
            alive = shade_one(sc, o, d, thr, pixel, h, pp, bg, X);
            if (!alive) {
                double *px = fb + (size_t)slot * 3;
                px[0] += thr.x * X.x;
                px[1] += thr.y * X.y;
                px[2] += thr.z * X.z;
            }
        }
        uint32_t at = block_append(alive, count_out, append_lds);
        if (alive) {
            qout.o[0][at] = o.x;
            qout.o[1][at] = o.y;
            qout.o[2][at] = o.z;
            qout.d[0][at] = d.x;
            qout.d[1][at] = d.y;
            qout.d[2][at] = d.z;
            qout.thr[0][at] = thr.x;
            qout.thr[1][at] = thr.y;
            qout.thr[2][at] = thr.z;
            qout.slot[at] = slot;
            qout.pixel[at] = pixel;
        }
    }
}

// The tail of a stratum: once few paths are left, one wave takes one path through all of its
// remaining segments - nearest hit over the analytic objects (every lane computes the same),
// the meshes by brute force across the 64 lanes, the material - with no launch per bounce.
// The late passes hold a fraction of a percent of the rays but were a sixth of the frame time
// as separate launches.  counts[k] gains one for every ray traced at depth k.
template <bool CSG, bool MEDIUM, bool TORUS>
__global__ void __launch_bounds__(RT_BLOCK) k_tail(DevScene sc_mem, RayQueue q, const uint32_t *counts, uint32_t *counts_out,
                                                    double *fb, PassParams pp, uint32_t *guards_out, int has_mesh,
                                                    uint32_t *lane_totals)
{
    extern __shared__ double4 scene_lds[];
    const DevScene sc = scene_in_lds(sc_mem, reinterpret_cast<unsigned char *>(scene_lds));
    const uint32_t count = counts[pp.depth];
    // a scene with meshes: one wave per path (the 64 lanes share the mesh test); without: one lane per path
    const bool per_lane = !has_mesh;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, n_threads = gridDim.x * blockDim.x;
    const uint32_t unit = per_lane ? tid : tid >> 6, n_units = per_lane ? n_threads : n_threads >> 6;
    const bool writer = per_lane || lane_id() == 0u;
    const d3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
    Guards gd;
    for (int k = 0; k < 5; ++k) gd.g[k] = 0u;
    gd.overflow = 0u;
    uint32_t my_rays = 0u, my_depth = 0u; // per_lane: segments traced beyond the first, deepest level reached
    for (uint32_t i = unit; i < count; i += n_units) {
        d3 o = mk3(q.o[0][i], q.o[1][i], q.o[2][i]);
        d3 d = mk3(q.d[0][i], q.d[1][i], q.d[2][i]);
        d3 thr = mk3(q.thr[0][i], q.thr[1][i], q.thr[2][i]);
        const uint32_t slot = q.slot[i], pixel = q.pixel[i];
        PassParams p = pp;
        for (;;) {
            HitRec h;
            bool wants_mesh = intersect_analytic<CSG, MEDIUM, TORUS>(sc, o, d, pixel, p, h, gd);
            if (has_mesh && wants_mesh) {
                int best_obj = h.obj;
                double best_t = best_obj >= 0 ? h.t : DBL_MAX;
                uint32_t best_aux = 0u;
                if (mesh_brute_wave(sc, o, d, best_obj, best_t, best_aux)) {
                    h.t = best_t;
                    h.obj = best_obj;
                    mesh_hit_surface(sc, o, d, best_obj, best_aux, h.n, h.u, h.v);
                }
            }
            d3 X;
            const bool alive = shade_one(sc, o, d, thr, pixel, h, p, bg, X);
            if (!alive) {
                if (writer) {
                    double *px = fb + (size_t)slot * 3;
                    px[0] += thr.x * X.x;
                    px[1] += thr.y * X.y;
                    px[2] += thr.z * X.z;
                }
                break;
            }
            ++p.depth;
            // rays per depth level: one atomic per segment is fine for a wave per path; lanes
            // would hammer a few words, so they keep a total and the deepest level instead
            if (per_lane) {
                ++my_rays;
                my_depth = p.depth > my_depth ? p.depth : my_depth;
            } else if (writer) {
                atomicAdd(&counts_out[p.depth], 1u);
            }
        }
    }
    if (per_lane) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            my_rays += __shfl_xor(my_rays, off);
            const uint32_t od = __shfl_xor(my_depth, off);
            my_depth = od > my_depth ? od : my_depth;
            for (int k = 0; k < 5; ++k) gd.g[k] += __shfl_xor(gd.g[k], off);
            gd.overflow += __shfl_xor(gd.overflow, off);
        }
        if (lane_id() == 0u) {
            if (my_rays) atomicAdd(&lane_totals[0], my_rays);
            if (my_depth) atomicMax(&lane_totals[1], my_depth);
        }
    }
    if (lane_id() == 0u) { // wave per path: every lane counted the same events; lane per path: summed above
        for (int k = 0; k < 5; ++k)
            if (gd.g[k]) atomicAdd(&guards_out[k], gd.g[k]);
        if (gd.overflow) atomicAdd(&guards_out[5], gd.overflow);
    }
}

// max_depth == 0: every primary ray returns the background (system.rs:113-114)
__global__ void k_fill_background(DevScene sc, rt_shard_geom g, uint32_t rank, uint32_t n_slots, double *fb, int row_major)
{
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x, y;
    if (slot < n_slots && rt_shard_decode(&g, rank, slot, &x, &y)) {
        if (row_major) slot = y * g.width + x;
        fb[(size_t)slot * 3 + 0] += sc.background[0];
        fb[(size_t)slot * 3 + 1] += sc.background[1];
        fb[(size_t)slot * 3 + 2] += sc.background[2];
    }
}

__global__ void k_unshard(rt_shard_geom g, uint32_t rank, uint32_t n_slots, const double *shard_rgb, double *full_rgb,
                          const int32_t *shard_ids, int32_t *full_ids)
{
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x, y;
    if (slot < n_slots && rt_shard_decode(&g, rank, slot, &x, &y)) {
        size_t p = (size_t)y * g.width + x;
        if (shard_rgb && full_rgb) {
            full_rgb[p * 3 + 0] = shard_rgb[(size_t)slot * 3 + 0];
            full_rgb[p * 3 + 1] = shard_rgb[(size_t)slot * 3 + 1];
            full_rgb[p * 3 + 2] = shard_rgb[(size_t)slot * 3 + 2];
        }
        if (shard_ids && full_ids) full_ids[p] = shard_ids[slot];
    }
}

__global__ void k_iota(uint32_t n, uint32_t *slot, uint32_t *pixel, double *t0, double *t1, double *t2)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        slot[i] = i;
        pixel[i] = i;
        t0[i] = t1[i] = t2[i] = 1.0;
    }
}

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

struct rt_scene {
    int device = 0;
    std::vector<void *> allocs; // scene data
    DevScene dev;
    // workspace, grown on demand
    size_t cap = 0;
    void *ws = nullptr;
    RayQueue q[2];
    HitBuf hit;
    bool has_csg = false, has_medium = false, has_mesh = false, has_torus = false;
    uint64_t n_mesh_tris = 0;
    uint32_t mesh_lds_bytes = 0; // dynamic LDS of k_mesh: occupancy words (+ cell starts)
    bool mesh_kd = false;           // walk kd-trees instead of grids (chosen in rt_scene_create)
    uint32_t mesh_kd_lds_bytes = 0; // dynamic LDS of k_mesh<KD>: job records + stacks
    uint32_t *mesh_queue = nullptr; // ray indices that need the mesh walk this pass
    uint32_t *d_counts = nullptr; // 2 x (max_depth + 2): live rays per depth, then mesh-queue lengths per depth
    uint32_t counts_cap = 0;
    uint32_t *d_guards = nullptr; // 6
    uint32_t *h_pinned = nullptr; // readback slot
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> evp; // per-pass timing events: [0] before raygen, then 4 per depth
    std::vector<hipEvent_t> evr; // per depth: its queue length has been copied back
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

static void free_workspace(rt_scene *s)
{
    if (s->ws) (void)hipFree(s->ws);
    s->ws = nullptr;
    s->cap = 0;
}

static int ensure_workspace(rt_scene *s, size_t cap, uint32_t max_depth)
{
    if (cap > s->cap) {
        free_workspace(s);
        // 2 queues x (9 f64 + 2 u32) + hits (6 f64 + 1 i32), each array 256-B aligned
        size_t n = (cap + 63) & ~(size_t)63;
        size_t bytes = n * (2 * (9 * 8 + 2 * 4) + 6 * 8 + 4 + 4);
        HIP_TRY(hipMalloc(&s->ws, bytes));
        char *p = (char *)s->ws;
        auto take = [&](size_t elem) {
            void *r = p;
            p += n * elem;
            return r;
        };
        for (int k = 0; k < 2; ++k) {
            for (int c = 0; c < 3; ++c) s->q[k].o[c] = (double *)take(8);
            for (int c = 0; c < 3; ++c) s->q[k].d[c] = (double *)take(8);
            for (int c = 0; c < 3; ++c) s->q[k].thr[c] = (double *)take(8);
        }
        s->hit.t = (double *)take(8);
        for (int c = 0; c < 3; ++c) s->hit.n[c] = (double *)take(8);
        for (int c = 0; c < 2; ++c) s->hit.uv[c] = (double *)take(8);
        for (int k = 0; k < 2; ++k) {
            s->q[k].slot = (uint32_t *)take(4);
            s->q[k].pixel = (uint32_t *)take(4);
            s->q[k].primary = 0;
            s->q[k].row_major = 0;
            s->q[k].rank = s->q[k].s_i = s->q[k].s_j = 0u;
            s->q[k].recip_sqrt_spp = 1.0;
            std::memset(&s->q[k].g, 0, sizeof(s->q[k].g));
        }
        s->hit.obj = (int32_t *)take(4);
        s->mesh_queue = (uint32_t *)take(4);
        s->cap = cap;
    }
    if (max_depth + 2 > s->counts_cap) {
        if (s->d_counts) (void)hipFree(s->d_counts);
        s->d_counts = nullptr;
        HIP_TRY(hipMalloc((void **)&s->d_counts, (2 * (max_depth + 2) + 2) * sizeof(uint32_t))); // + k_tail's lane totals
        if (s->h_pinned) (void)hipHostFree(s->h_pinned);
        s->h_pinned = nullptr;
        HIP_TRY(hipHostMalloc((void **)&s->h_pinned, (3 * (max_depth + 2) + 8) * sizeof(uint32_t), hipHostMallocDefault));
        s->counts_cap = max_depth + 2;
    }
    if (!s->d_guards) HIP_TRY(hipMalloc((void **)&s->d_guards, 8 * sizeof(uint32_t)));
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
    return RT_OK;
}

extern "C" int rt_scene_create(const rt_scene_desc *desc, rt_scene **out)
{
    if (!desc || !out) {
        rt_host::set_last_error("rt_scene_create: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
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
        // Structure per scene.  The grid is the default (same speed on LinkedTorus.obj, no stack, a
        // trivial builder); it degrades when a mesh crowds many triangles into few cells, which the
        // tree's SAH splits handle.  RT_ACCEL=grid|kd overrides (measurements, tests).
        const double crowd = flat.grid_crowd; // triangles in the cell of an average (cell, triangle) reference, worst mesh
        s->mesh_kd = crowd > 24.0;            // LinkedTorus.obj: 7; Monkey.obj: 5
        if (const char *e = getenv("RT_ACCEL")) {
            if (!strcmp(e, "kd")) s->mesh_kd = true;
            if (!strcmp(e, "grid")) s->mesh_kd = false;
        }
        RT_TRACE("mesh structure: %s (a triangle shares its grid cell with %.1f on average)", s->mesh_kd ? "kd-tree" : "grid", crowd);
    }
    s->mesh_lds_bytes = (uint32_t)(std::min<size_t>(flat.jobs.size(), DEV_MESH_JOBS_LDS) * sizeof(DevMeshJob)) +
                        s->dev.n_grid_words * 8u + s->dev.n_dist_words * 4u +
                        (s->dev.n_cell_starts <= DEV_GRID_MAX_STARTS ? s->dev.n_cell_starts * 4u : 0u);
    if ((rc = upload(s, flat.tris, &s->dev.tris))) return fail(rc);
    if ((rc = upload(s, flat.cell_tris, &s->dev.cell_tris))) return fail(rc);
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
    for (int k = 0; k < 3; ++k) s->dev.background[k] = flat.background[k];
    s->dev.cam_fov_factor = flat.cam_fov_factor;
    std::memcpy(s->dev.cam_c2w, flat.cam_c2w, sizeof(flat.cam_c2w));
    *out = s;
    RT_TRACE("scene created");
    return RT_OK;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
    if (!s) return;
    for (void *p : s->allocs) (void)hipFree(p);
    free_workspace(s);
    if (s->d_counts) (void)hipFree(s->d_counts);
    if (s->d_guards) (void)hipFree(s->d_guards);
    if (s->h_pinned) (void)hipHostFree(s->h_pinned);
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
    if (b > 256 * 2) b = 256 * 2;
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
    return RT_OK;
}

#define RT_BRUTE_MAX_TESTS 40000000ull // rays x triangles below which the brute-force wave kernel wins (measured)

// nearest hit for the `count_ptr[0]` rays of `q`: the analytic pass, then the mesh pass
// returns 1 when the grid walk (k_mesh) ran, 0 otherwise; `mid` is recorded between the two stages
static int launch_intersect(rt_scene *s, hipStream_t st, uint32_t live, const RayQueue &q, const uint32_t *count_ptr,
                            uint32_t *mesh_count, const PassParams &pp, int32_t *ids, hipEvent_t mid)
{
    dim3 grid(grid_for(live)), block(RT_BLOCK);
#define LAUNCH_I(C, M, T)                                                                                              \
    hipLaunchKernelGGL((k_intersect<C, M, T>), grid, block, scene_lds_bytes(s->dev), st, s->dev, q, count_ptr, s->hit, pp, ids, s->d_guards, \
                       s->mesh_queue, mesh_count)
    const int variant = (s->has_csg ? 4 : 0) | (s->has_medium ? 2 : 0) | (s->has_torus ? 1 : 0);
    switch (variant) {
    case 0: LAUNCH_I(false, false, false); break;
    case 1: LAUNCH_I(false, false, true); break;
    case 2: LAUNCH_I(false, true, false); break;
    case 3: LAUNCH_I(false, true, true); break;
    case 4: LAUNCH_I(true, false, false); break;
    case 5: LAUNCH_I(true, false, true); break;
    case 6: LAUNCH_I(true, true, false); break;
    default: LAUNCH_I(true, true, true); break;
    }
#undef LAUNCH_I
    (void)hipEventRecord(mid, st);
    int walked = 0;
    if (s->has_mesh) {
        // short queues: one wave per ray over all triangles; long queues: one lane per ray through the mesh grid
        if ((uint64_t)live * s->n_mesh_tris <= (uint64_t)RT_BRUTE_MAX_TESTS) {
            uint64_t blocks = ((uint64_t)live * 64 + RT_BLOCK - 1) / RT_BLOCK;
            if (blocks > 256 * 8) blocks = 256 * 8;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(k_mesh_brute, dim3((uint32_t)blocks), block, 0, st, s->dev, q, s->mesh_queue, mesh_count,
                               s->hit, ids, pp);
        } else {
            // one 64-ray block per wave while the waves fit the chip a few times over; beyond that the
            // waves stay and refill their idle lanes from further blocks
            static const uint32_t refill = getenv("RT_MESH_REFILL") ? (uint32_t)atoi(getenv("RT_MESH_REFILL")) : 16u;
            static const uint32_t max_wgs = getenv("RT_MESH_WGS") ? (uint32_t)atoi(getenv("RT_MESH_WGS")) : 256u * 6u;
            static const uint32_t block_env = getenv("RT_MESH_BLOCK") ? (uint32_t)atoi(getenv("RT_MESH_BLOCK")) : 0u;
            const uint64_t per_wg = 16u * (RT_MESH_LANES / 64); // the kernel uses 16, 32 or 64 lanes per wave
            uint64_t mb = ((uint64_t)live + per_wg - 1) / per_wg;
            if (mb < 1) mb = 1;
            if (mb > max_wgs) mb = max_wgs;
            if (s->mesh_kd)
                hipLaunchKernelGGL(k_mesh<true>, dim3((uint32_t)mb), dim3(RT_MESH_LANES), s->mesh_kd_lds_bytes, st, s->dev, q,
                                   s->mesh_queue, mesh_count, s->hit, ids, refill, block_env, 256u * 4u * RT_MESH_WAVES, pp);
            else
                hipLaunchKernelGGL(k_mesh<false>, dim3((uint32_t)mb), dim3(RT_MESH_LANES), s->mesh_lds_bytes, st, s->dev, q,
                                   s->mesh_queue, mesh_count, s->hit, ids, refill, block_env, 256u * 4u * RT_MESH_WAVES, pp);
            walked = 1;
        }
    }
    return walked;
}

#define RT_TAIL_MAX_TESTS 36000000ull // paths x triangles below which k_tail takes the rest of the stratum (measured)

static void launch_tail(rt_scene *s, hipStream_t st, uint32_t live, const RayQueue &q, const PassParams &pp, double *d_rgb)
{
    uint64_t blocks = ((uint64_t)live * (s->has_mesh ? 64 : 1) + RT_BLOCK - 1) / RT_BLOCK; // wave or lane per path
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
#define LAUNCH_T(C, M, T)                                                                                              \
    hipLaunchKernelGGL((k_tail<C, M, T>), dim3((uint32_t)blocks), dim3(RT_BLOCK), scene_lds_bytes(s->dev), st, s->dev, q, s->d_counts,       \
                       s->d_counts, d_rgb, pp, s->d_guards, s->has_mesh ? 1 : 0, s->d_counts + 2 * (pp.max_depth + 2))
    const int variant = (s->has_csg ? 4 : 0) | (s->has_medium ? 2 : 0) | (s->has_torus ? 1 : 0);
    switch (variant) {
    case 0: LAUNCH_T(false, false, false); break;
    case 1: LAUNCH_T(false, false, true); break;
    case 2: LAUNCH_T(false, true, false); break;
    case 3: LAUNCH_T(false, true, true); break;
    case 4: LAUNCH_T(true, false, false); break;
    case 5: LAUNCH_T(true, false, true); break;
    case 6: LAUNCH_T(true, true, false); break;
    default: LAUNCH_T(true, true, true); break;
    }
#undef LAUNCH_T
}

typedef int (*stratum_fn)(void *user, uint32_t done, uint32_t total);

// the wavefront loop for one shard; d_rgb (sum) and d_ids are in shard layout
static int render_strata(rt_scene *s, const rt_options *opts, double *d_rgb, int32_t *d_ids, rt_stats *stats,
                         hipStream_t st, stratum_fn per_stratum, void *user)
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
    rc = ensure_workspace(s, n_slots ? n_slots : 1, opts->max_depth);
    if (rc) return rc;
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
    const size_t n_out = row_major ? (size_t)opts->width * opts->height : (size_t)n_slots;
    HIP_TRY(hipMemsetAsync(d_rgb, 0, n_out * 3 * sizeof(double), st)); // alloc_render_buf
    if (d_ids) HIP_TRY(hipMemsetAsync(d_ids, 0xFF, n_out * sizeof(int32_t), st));
    HIP_TRY(hipMemsetAsync(s->d_guards, 0, 8 * sizeof(uint32_t), st));
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
                HIP_TRY(hipMemsetAsync(s->d_counts, 0, (2 * (opts->max_depth + 2) + 2) * sizeof(uint32_t), st));
                // depth 0 reads no ray queue: entry i is slot i and its camera ray is regenerated in place
                RayQueue q_primary = s->q[0];
                q_primary.primary = 1;
                q_primary.row_major = row_major ? 1 : 0;
                q_primary.rank = rank;
                q_primary.s_i = s_i;
                q_primary.s_j = s_j;
                q_primary.g = g;
                q_primary.recip_sqrt_spp = recip;
                HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)s->d_counts, (int)n_slots, 1, st));
                HIP_TRY(hipEventRecord(s->evp[0], st));
                // `live` bounds the rays of the next pass from above: the newest count that has come
                // back (counts never grow).  The kernels read the true count on the device; the bound
                // only sizes grids and picks between k_mesh and k_mesh_brute.
                uint32_t live = n_slots;
                // paths left at which the tail kernel takes over (one wave per path, meshes by brute force)
                static const uint32_t tail_env = getenv("RT_TAIL_MAX") ? (uint32_t)atoi(getenv("RT_TAIL_MAX")) : 0xFFFFFFFFu;
                uint32_t tail_max = s->has_mesh ? (uint32_t)(RT_TAIL_MAX_TESTS / (s->n_mesh_tris ? s->n_mesh_tris : 1u)) : 16384u;
                if (tail_max > 16384u) tail_max = 16384u;
                if (tail_env != 0xFFFFFFFFu) tail_max = tail_env;
                static const uint32_t run_ahead = getenv("RT_RUN_AHEAD") ? (uint32_t)atoi(getenv("RT_RUN_AHEAD")) : 2u;
                int cur = 0;
                uint32_t passes = 0;
                int tail_at = -1; // depth at which k_tail took over
                std::vector<uint8_t> walked(opts->max_depth + 1, 0);
                for (uint32_t depth = 0; depth < opts->max_depth; ++depth) {
                    pp.depth = depth;
                    int32_t *ids = (depth == 0 && pp.stratum == 0) ? d_ids : nullptr;
                    HIP_TRY(hipEventRecord(s->evp[1 + 4 * depth], st));
                    const RayQueue &q_in = depth == 0 ? q_primary : s->q[cur];
                    walked[depth] = (uint8_t)launch_intersect(s, st, live, q_in, s->d_counts + depth,
                                                              s->d_counts + (opts->max_depth + 2) + depth, pp, ids,
                                                              s->evp[2 + 4 * depth]);
                    HIP_TRY(hipEventRecord(s->evp[3 + 4 * depth], st));
                    hipLaunchKernelGGL(k_shade, dim3(grid_for_wide(live)), dim3(RT_BLOCK_WIDE), scene_lds_bytes(s->dev), st, s->dev, q_in,
                                       s->d_counts + depth, s->hit, s->q[1 - cur], s->d_counts + depth + 1, d_rgb, pp);
                    HIP_TRY(hipEventRecord(s->evp[4 + 4 * depth], st));
                    ++passes;
                    cur = 1 - cur;
                    // queue length of the next pass: copied back asynchronously after every pass.  The
                    // host waits for the copy of two passes ago only, so the device always has work
                    // queued across the copy's round trip; near the tail threshold (and on the last
                    // pass) it waits for the newest copy, because the switch needs the count.
                    uint32_t *h_live = s->h_pinned + 2 * (opts->max_depth + 2) + 8;
                    HIP_TRY(hipMemcpyAsync(h_live + depth + 1, s->d_counts + depth + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipEventRecord(s->evr[depth + 1], st));
                    const bool exact = live <= 8u * tail_max || depth + 1 == opts->max_depth || run_ahead == 0u;
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
                        HIP_TRY(hipEventRecord(s->evp[1 + 4 * (depth + 1)], st));
                        launch_tail(s, st, live, s->q[cur], pp, d_rgb);
                        HIP_TRY(hipEventRecord(s->evp[2 + 4 * (depth + 1)], st));
                        tail_at = (int)depth + 1;
                        break;
                    }
                }
                HIP_TRY(hipMemcpyAsync(s->h_pinned, s->d_counts, (2 * (opts->max_depth + 2) + 2) * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, st)); // rays per depth, mesh-queue lengths per depth, lane totals
                HIP_TRY(hipStreamSynchronize(st));
                std::vector<uint32_t> mesh_counts(opts->max_depth + 1, 0u);
                for (uint32_t k = 0; k <= opts->max_depth; ++k) mesh_counts[k] = s->h_pinned[opts->max_depth + 2 + k];
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
                {
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
                        if (s->h_pinned[k]) {
                            local.n_intersect_launches++; // passes that had rays
                            local.intersect_rays += s->h_pinned[k];
                        }
                        if (walked[k]) {
                            local.n_mesh_launches++;
                            local.mesh_rays += mesh_counts[k];
                        }
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
                if (stop < 0) return stop;
                if (stop > 0) goto finished;
            }
        }
    }
finished:
    HIP_TRY(hipMemcpyAsync(s->h_pinned, s->d_guards, 6 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 5; ++k) local.guards[k] = s->h_pinned[k];
    local.interval_overflow = s->h_pinned[5];
#ifdef RT_MESH_PROF
    {
        unsigned long long pf[12];
        (void)hipMemcpyFromSymbol(pf, HIP_SYMBOL(g_pf), sizeof(pf));
        fprintf(stderr, "[k_mesh] waves %llu: cycles/wave service %.0f walk %.0f test %.0f; rounds/wave %.1f services/wave %.1f; lanes walking/round %.1f testing/round %.1f\n",
                pf[7], (double)pf[0] / pf[7], (double)pf[1] / pf[7], (double)pf[2] / pf[7], (double)pf[3] / pf[7],
                (double)pf[4] / pf[7], (double)pf[5] / (double)pf[3], (double)pf[6] / (double)pf[3]);
        fprintf(stderr, "[k_mesh] service split: reload+merge %.0f fetch %.0f advance %.0f write %.0f\n", (double)pf[8] / pf[7],
                (double)pf[9] / pf[7], (double)pf[10] / pf[7], (double)pf[11] / pf[7]);
        unsigned long long z[12] = {0};
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
    if (!c->cb) return 0;
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
    double *d_full = nullptr;
    int32_t *d_ids_full = nullptr;
    auto cleanup = [&]() {
        if (d_full) (void)hipFree(d_full);
        if (d_ids_full) (void)hipFree(d_ids_full);
    };
#define TRY_OR_CLEAN(expr)                                                                                             \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            rt_host::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
            cleanup();                                                                                                 \
            return RT_ERR_DEVICE;                                                                                      \
        }                                                                                                              \
    } while (0)
    TRY_OR_CLEAN(hipMalloc((void **)&d_full, n * 3 * sizeof(double)));
    if (primary_ids) TRY_OR_CLEAN(hipMalloc((void **)&d_ids_full, n * sizeof(int32_t)));
    HostRenderCtx ctx{&opts, d_full, rgb_sum, cb, user};
    rc = render_strata(scene, &opts, d_full, d_ids_full, stats, nullptr, host_render_stratum, &ctx);
    if (rc == RT_OK) {
        TRY_OR_CLEAN(hipMemcpy(rgb_sum, d_full, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
        if (primary_ids) TRY_OR_CLEAN(hipMemcpy(primary_ids, d_ids_full, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    cleanup();
    return rc;
#undef TRY_OR_CLEAN
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
    // de-interleave on the host into the SoA queue
    std::vector<double> col(n);
    for (int c = 0; c < 3; ++c) {
        for (uint64_t i = 0; i < n; ++i) col[i] = origins[3 * i + c];
        HIP_TRY(hipMemcpy(s->q[0].o[c], col.data(), n * sizeof(double), hipMemcpyHostToDevice));
        for (uint64_t i = 0; i < n; ++i) col[i] = directions[3 * i + c];
        HIP_TRY(hipMemcpy(s->q[0].d[c], col.data(), n * sizeof(double), hipMemcpyHostToDevice));
    }
    uint32_t n32 = (uint32_t)n;
    hipLaunchKernelGGL(k_iota, dim3((n32 + 255) / 256), dim3(256), 0, nullptr, n32, s->q[0].slot, s->q[0].pixel,
                       s->q[0].thr[0], s->q[0].thr[1], s->q[0].thr[2]);
    HIP_TRY(hipMemcpy(s->d_counts, &n32, sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(s->d_guards, 0, 8 * sizeof(uint32_t)));
    PassParams pp;
    pp.seed = seed;
    pp.stratum = 0;
    pp.depth = 0;
    pp.max_depth = 1;
    pp.force_uv = 1;
    pp.bias = 0.0;
    HIP_TRY(hipMemset(s->d_counts + 3, 0, sizeof(uint32_t)));
    (void)launch_intersect(s, nullptr, n32, s->q[0], s->d_counts, s->d_counts + 3, pp, nullptr, s->evp[2]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    std::vector<int32_t> obj(n);
    std::vector<double> cols[6];
    double *src[6] = {s->hit.t, s->hit.n[0], s->hit.n[1], s->hit.n[2], s->hit.uv[0], s->hit.uv[1]};
    for (int c = 0; c < 6; ++c) {
        cols[c].resize(n);
        HIP_TRY(hipMemcpy(cols[c].data(), src[c], n * sizeof(double), hipMemcpyDeviceToHost));
    }
    HIP_TRY(hipMemcpy(obj.data(), s->hit.obj, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        hits[i].t = cols[0][i];
        hits[i].n[0] = cols[1][i];
        hits[i].n[1] = cols[2][i];
        hits[i].n[2] = cols[3][i];
        hits[i].uv[0] = cols[4][i];
        hits[i].uv[1] = cols[5][i];
        hits[i].object = obj[i];
        hits[i].pad = 0;
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        uint32_t g[6];
        HIP_TRY(hipMemcpy(g, s->d_guards, sizeof(g), hipMemcpyDeviceToHost));
        stats->rays = n;
        for (int k = 0; k < 5; ++k) stats->guards[k] = g[k];
        stats->interval_overflow = g[5];
    }
    return RT_OK;
}
