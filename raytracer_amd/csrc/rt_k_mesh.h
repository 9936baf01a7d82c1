// rt_k_mesh.h — k_mesh (grid or kd-tree walk, one lane per ray) and k_mesh_brute (one wave per ray)
// (device code of librt_amd.so; included by rt_kernels.hip, which holds the host side of the C ABI)
#pragma once

#include "rt_k_common.h"

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

// Pass 2: the mesh records queued by k_intersect (`q` + `hit`, both indexed by queue position: ray, pixel
// keys and the nearest hit among the other objects) against the mesh objects
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
// Waves take blocks of 64 queue entries: wave w starts on block w, and a wave that has used its block up takes
// the next one nobody has taken yet (`work`, one atomic per wave and block); with more than one block per wave
// (large queues) a wave refills its finished lanes once `refill` of them are idle.
#ifdef RT_MESH_PROF
__device__ unsigned long long g_pf[14]; // cycles: service, walk, test; rounds, services, lanes walking, lanes testing, waves
#endif
// The walk state is kept in plain scalars (not an aggregate): selecting a field by axis
// would otherwise be compiled into an indexed access to a copy in scratch memory.
#define WALK_LOCATE_AXIS(OG, DG, IG, N, C, TM)                                                                         \
    do {                                                                                                               \
        double f_ = floor((OG) + (DG) * w_tc);                                                                         \
        f_ = f_ > 0.0 ? f_ : 0.0;         /* also sends a NaN to cell 0 */                                             \
        f_ = f_ > 1.0e9 ? 1.0e9 : f_;     /* fits an int; the clamp to the last cell is an integer one (no double */   \
        (C) = (int)f_;                    /* per axis to keep in registers) */                                         \
        (C) = (C) > (int)((N) - 1u) ? (int)((N) - 1u) : (C);                                                           \
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
// the merge - is shared.  The flattener picks the structure per mesh (DevMeshJob::use_kd).
// JOBS_LDS: every job record is in LDS (at most DEV_MESH_JOBS_LDS of them).  A compile-time fact on purpose:
// a run-time choice between the LDS copy and memory makes each field access a flat load.
typedef float v2f __attribute__((ext_vector_type(2)));
RT_DEV v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

template <bool KD, bool JOBS_LDS>
__global__ void __launch_bounds__(RT_MESH_LANES) RT_MESH_OCC k_mesh(DevScene sc, RayQueue q,
                                                      QRegions mesh_regions, uint32_t *work, HitBuf hit,
                                                      int32_t *primary_ids, uint32_t refill, uint32_t block_arg, uint32_t resident_waves,
                                                      PassParams pp)
{
    extern __shared__ double4 mesh_lds[]; // job records; then the grid tables, or the kd stacks
    const uint32_t n_jobs_lds = sc.n_jobs < DEV_MESH_JOBS_LDS ? sc.n_jobs : DEV_MESH_JOBS_LDS;
    DevMeshJob *jobs_lds = reinterpret_cast<DevMeshJob *>(mesh_lds);
    uint2 *grid_lds = reinterpret_cast<uint2 *>(jobs_lds + n_jobs_lds);
    uint32_t *dist_lds = reinterpret_cast<uint32_t *>(grid_lds + (KD ? 0u : sc.n_grid_words));
    uint32_t *starts_lds = dist_lds + (KD ? 0u : sc.n_dist_words);
    const bool starts_in_lds = !KD && sc.starts_in_lds != 0u; // decided by rt_scene_create against the device's LDS limit
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
#define JOB(J) (*((JOBS_LDS || (J) < (int)n_jobs_lds) ? (const DevMeshJob *)(jobs_lds + (J)) : (sc.jobs + (J))))

    // every wave takes blocks of 64 queue entries: its own number first, then whichever comes next (neighbouring
    // rays cost about the same, so contiguous chunks would leave whole waves with the expensive ones)
    const QMap qm = qmap_load(mesh_regions); // record number -> position
    const uint32_t count = qm.total();
    const uint32_t n_waves = gridDim.x * (RT_MESH_LANES / 64);
    const uint32_t wave = blockIdx.x * (RT_MESH_LANES / 64) + (threadIdx.x >> 6);
    const uint64_t lane_lt = (1ull << lane_id()) - 1ull;
    // short queues: fewer rays per wave (4 to 32 lanes used), so that the slowest wave of the pass is
    // done sooner; the queue length is only known here, the host sized the grid from an upper bound
    uint32_t block = block_arg;
    if (block == 0u) {
        const uint32_t spread = n_waves < resident_waves ? n_waves : resident_waves;
        block = RT_MESH_MIN_BLOCK;
        while (block < 64u && (uint64_t)block * spread < count) block <<= 1;
    }
    // a wave with a single block has nothing to refill from: it services its lanes twice, to load
    // the block and to store the results once every lane is done
    const bool single_block = (uint64_t)n_waves * block >= count;
    if (single_block) refill = 65u;
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
    unsigned long long pf_rounds = 0, pf_services = 0, pf_lanes_walk = 0, pf_lanes_test = 0, pf_tests = 0, pf_hits = 0;
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
                    ri = qm.pos((uint32_t)cursor, (uint32_t)j); // the queue holds records, read by position
                    uint32_t px_;
                    (void)fetch_ray(sc, q, pp, ri, wo, wd, px_);
                    best_obj = hit.obj[ri];
                    const double recorded_t = hit.t[ri]; // loaded whatever obj says: one round trip, not two
                    best_t = best_obj >= 0 ? recorded_t : DBL_MAX;
                    best_aux = 0u;
                    replaced = false;
                    have_ray = true;
                    jb = -1;
                    obj_found = false;
                    advance = true;
                }
            }
            if (cursor < count) {
                cursor += (uint64_t)__popcll(empty);
                if (cursor >= block_end) { // block used up (lanes left without a ray take from the next one)
                    if (single_block) {
                        cursor = count;
                    } else {
                        // whichever wave gets here first takes the next block: the waves stay filled until the queue is
                        // empty, whatever their rays cost (a fixed share per wave left the waves with the cheap rays
                        // idle through the second half of the launch)
                        const uint32_t n_part = n_waves < RT_MESH_CURSORS ? n_waves : RT_MESH_CURSORS, part = wave % n_part;
                        uint32_t nb = 0u;
                        if (lane_id() == 0u) nb = atomicAdd(work + part * 16u, 1u);
                        nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb); // a scalar: the cursor stays the same in every lane
                        cursor = ((uint64_t)n_waves + (uint64_t)nb * n_part + part) * block;
                        block_end = cursor + block;
                    }
                }
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
                    if ((J.use_kd != 0u) != KD) continue; // the other instantiation's mesh (structure per mesh)
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
#include "rt_k_mesh_grid_round.inc"
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
    {
        // lane-level: triangle tests made, of which hits
        unsigned long long tt = pf_tests, hh = pf_hits;
        for (int off = 32; off >= 1; off >>= 1) {
            tt += __shfl_xor(tt, off);
            hh += __shfl_xor(hh, off);
        }
        if (lane_id() == 0) {
            atomicAdd(&g_pf[12], tt);
            atomicAdd(&g_pf[13], hh);
        }
    }
#endif
}

#undef JOB
#undef PF_MARK

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

__global__ void __launch_bounds__(RT_BLOCK) k_mesh_brute(DevScene sc, RayQueue q,
                                                          QRegions mesh_regions, HitBuf hit, int32_t *primary_ids,
                                                          PassParams pp)
{
    const QMap qm = qmap_load(mesh_regions);
    const uint32_t count = qm.total();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t idx = wave; idx < count; idx += n_waves) { // mesh records, by position
        const uint32_t i = qm.pos(idx);
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
