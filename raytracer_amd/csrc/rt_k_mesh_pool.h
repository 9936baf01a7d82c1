// rt_k_mesh_pool.h — k_mesh_one: k_mesh's grid walk for a scene with a single (mesh object, mesh) job, with the
// per-ray work around the walk done for whole waves at a time (device code of librt_amd.so; included by
// rt_kernels.hip)
#pragma once

#include "rt_k_mesh.h"

// k_mesh gives a lane that has finished its ray the next record, runs the job loop for it (frame maps, box
// tests, grid entry) and, before that, writes the finished ray's hit (normal, uv): all of it with the few lanes
// that happen to be idle (it waits for 16), each time with two or three dependent trips to memory, and with the
// walk state of the other lanes held in registers meanwhile.  With one job in the scene (one mesh object made of
// one mesh: every mesh scene the reference ships) there is no job loop and no merge over objects to carry along,
// the job's constants are the same for every lane (scalar registers), and the two halves leave the walk:
//
//   PREPARE  a batch of its own, all 64 lanes on up to 64 new records: ray, recorded hit, frame map, box tests ->
//            the wave's `ready` stack in LDS (record number, ray in object space, entry and exit parameter, walk
//            steps per cell: 92 bytes)
//   pop      an idle lane takes the top of `ready` (LDS reads and the grid cell of the entry point) and walks
//   round    rt_k_mesh_grid_round.inc, as in k_mesh
//   walk end a ray that met a triangle stores (t in object space, triangle) at its record's place: 12 bytes
//   finish   in k_shade, which holds the record's ray and hit in registers anyway (mesh_one_finish, rt_k_shade.h):
//            Intersection::to_world's t, the comparison with the recorded hit, normal, uv
//
// Same arithmetic per ray as k_mesh (the same expressions on the same inputs), so the two kernels lead to identical
// images (test_mesh_kernels_agree); scenes with several jobs, with a kd-tree, or whose grid tables leave no room
// for the stacks run k_mesh.
#define RT_POOL_LANES 256
#define RT_POOL_READY 80u // entries; a PREPARE batch is min(64, RT_POOL_READY - entries in the stack) records
#define RT_POOL_READY_BYTES (RT_POOL_READY * 92u)
#define RT_POOL_BYTES RT_POOL_READY_BYTES // per wave
#ifndef RT_POOL_POP_MIN
#define RT_POOL_POP_MIN 8u // idle lanes before the wave stops to refill them
#endif
#ifndef RT_POOL_WAVES
#define RT_POOL_WAVES 3
#endif
#define RT_POOL_OCC __attribute__((amdgpu_waves_per_eu(RT_POOL_WAVES, RT_POOL_WAVES)))

RT_DEV uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
RT_DEV double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__global__ void __launch_bounds__(RT_POOL_LANES) RT_POOL_OCC k_mesh_one(DevScene sc, RayQueue q, QRegions mesh_regions,
                                                                        uint32_t *work, HitBuf hit, double *res_t, uint32_t *res_tri,
                                                                        uint32_t block_arg, uint32_t resident_waves, PassParams pp)
{
    extern __shared__ double4 mesh_lds[]; // the job record, grid tables, then the waves' stacks
    DevMeshJob *jobs_lds = reinterpret_cast<DevMeshJob *>(mesh_lds);
    uint2 *grid_lds = reinterpret_cast<uint2 *>(jobs_lds + 1);
    uint32_t *dist_lds = reinterpret_cast<uint32_t *>(grid_lds + sc.n_grid_words);
    uint32_t *starts_lds = dist_lds + sc.n_dist_words;
    const bool starts_in_lds = sc.starts_in_lds != 0u;
    {
        const uint32_t *jw = reinterpret_cast<const uint32_t *>(sc.jobs);
        uint32_t *jl = reinterpret_cast<uint32_t *>(jobs_lds);
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(DevMeshJob) / 4); i += blockDim.x) jl[i] = jw[i];
        const uint2 *gw = reinterpret_cast<const uint2 *>(sc.grid_bits);
        for (uint32_t i = threadIdx.x; i < sc.n_grid_words; i += blockDim.x) grid_lds[i] = gw[i];
        for (uint32_t i = threadIdx.x; i < sc.n_dist_words; i += blockDim.x) dist_lds[i] = sc.grid_dist[i];
        if (starts_in_lds)
            for (uint32_t i = threadIdx.x; i < sc.n_cell_starts; i += blockDim.x) starts_lds[i] = sc.cell_start[i];
    }
    __syncthreads();
#define JOB(J) (*(const DevMeshJob *)jobs_lds) // the one job
    // this wave's stacks (structure of arrays: lanes touch neighbouring words)
    const uint32_t tables_bytes = (uint32_t)sizeof(DevMeshJob) + sc.n_grid_words * 8u + sc.n_dist_words * 4u +
                                  (starts_in_lds ? sc.n_cell_starts * 4u : 0u);
    unsigned char *pool = reinterpret_cast<unsigned char *>(mesh_lds) + ((tables_bytes + 15u) & ~15u) +
                          (size_t)(threadIdx.x >> 6) * RT_POOL_BYTES;
    double *r_f = reinterpret_cast<double *>(pool);                        // o[3], d[3], t0, t1, ig[3]: 11 x RT_POOL_READY
    uint32_t *r_u = reinterpret_cast<uint32_t *>(r_f + 11 * RT_POOL_READY); // record number

    const QMap qm = qmap_load(mesh_regions); // record number -> position
    const uint32_t count = qm.total();
    const uint32_t n_waves = gridDim.x * (RT_POOL_LANES / 64);
    const uint32_t wave = uniform_u32(blockIdx.x * (RT_POOL_LANES / 64) + (threadIdx.x >> 6));
    const uint64_t lane_lt = (1ull << lane_id()) - 1ull;
    // short queues: fewer rays per wave, so that the slowest wave of the pass is done sooner (as k_mesh)
    uint32_t block = block_arg;
    if (block == 0u) {
        const uint32_t spread = n_waves < resident_waves ? n_waves : resident_waves;
        block = RT_MESH_MIN_BLOCK;
        while (block < 64u && (uint64_t)block * spread < count) block <<= 1;
    }
    const bool single_block = (uint64_t)n_waves * block >= count;
    uint64_t cursor = (uint64_t)wave * block; // next queue entry of this wave
    uint64_t block_end = cursor + block;      // end of its current block
    if (cursor > count) cursor = count;
    uint32_t nr = 0u; // entries in the stack (the same in every lane)

    // the job's constants: the same in every lane
    const uint32_t nx = uniform_u32(JOB(0).n[0]), ny = uniform_u32(JOB(0).n[1]), nz = uniform_u32(JOB(0).n[2]);
    const uint32_t bits0 = uniform_u32(JOB(0).bits0), cell0 = uniform_u32(JOB(0).cell0), dist0 = uniform_u32(JOB(0).dist0);
    const bool identity = uniform_u32((uint32_t)JOB(0).identity) != 0u;
    const double bmin_x = uniform_f64(JOB(0).bmin[0]), bmin_y = uniform_f64(JOB(0).bmin[1]), bmin_z = uniform_f64(JOB(0).bmin[2]);
    const double ic_x = uniform_f64(JOB(0).inv_cell[0]), ic_y = uniform_f64(JOB(0).inv_cell[1]), ic_z = uniform_f64(JOB(0).inv_cell[2]);
    const int jb = 0;
    (void)jb;

// the ray in grid coordinates (k_mesh keeps these six in registers; here they are two operations on scalars away)
#define ogx ((o.x - bmin_x) * ic_x)
#define ogy ((o.y - bmin_y) * ic_y)
#define ogz ((o.z - bmin_z) * ic_z)
#define dgx (d.x * ic_x)
#define dgy (d.y * ic_y)
#define dgz (d.z * ic_z)

    // lane state: the ray being walked (object space), its walk, its nearest triangle so far
    bool active = false;
    uint32_t ri = 0;
    d3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
    double igx = 0, igy = 0, igz = 0; // 1 / (d * inv_cell)
    double tmx = 0, tmy = 0, tmz = 0, w_tc = 0, t_end = 0;
    int cx = 0, cy = 0, cz = 0;
    double mt = 1.0e300;
    uint32_t mtri = 0xFFFFFFFFu;
    bool walked = false; // a walk has ended in this lane and its result has not been pushed yet
#ifdef RT_MESH_PROF
    // reported in k_mesh's format: "reload+merge" = walk end, "fetch" = PREPARE, "advance" = pop, "write" = -;
    // services = PREPARE batches
    long long pf_t = clock64(), pf_walk = 0, pf_test = 0, pf_a = 0, pf_b = 0, pf_c = 0, pf_d = 0;
    unsigned long long pf_rounds = 0, pf_services = 0, pf_lanes_walk = 0, pf_lanes_test = 0, pf_tests = 0, pf_hits = 0;
#define PF_MARK(ACC) do { long long n_ = clock64(); ACC += n_ - pf_t; pf_t = n_; } while (0)
#else
#define PF_MARK(ACC) do { } while (0)
#endif

    for (;;) {
        // ---------------------------------------------------------------- walk end: the nearest triangle, for k_shade
        if (walked && !active) {
            if (mtri != 0xFFFFFFFFu) {
                res_t[ri] = mt; // object-space t; k_shade (mesh_one_finish) merges it with the recorded hit
                res_tri[ri] = mtri;
            }
            walked = false;
        }
        const uint64_t idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        PF_MARK(pf_a);

        // ---------------------------------------------------------------- PREPARE + pop: refill the idle lanes
        if (n_idle >= RT_POOL_POP_MIN || n_idle == 64u) {
            while (nr < n_idle && cursor < count) {
                uint32_t batch = RT_POOL_READY - nr;
                if (batch > 64u) batch = 64u;
                if ((uint64_t)batch > block_end - cursor) batch = (uint32_t)(block_end - cursor);
                uint64_t j = cursor + lane_id();
                bool found = false;
                d3 po = mk3(0, 0, 0), pd = mk3(0, 0, 1);
                double t0 = 0.0, t1 = 0.0;
                if (lane_id() < batch && j < count) {
                    j = qm.pos((uint32_t)cursor, (uint32_t)j);
                    d3 wo, wd;
                    uint32_t px_;
                    (void)fetch_ray(sc, q, pp, (uint32_t)j, wo, wd, px_);
                    const int rec_obj = hit.obj[j];
                    const double recorded_t = hit.t[j]; // loaded whatever obj says: one round trip, not two
                    const DevMeshJob &J = JOB(0);
                    // Ray::transform (system.rs:145-152); a hit farther than the recorded one cannot win:
                    // object-space distance = world distance * |W d|; slack keeps ties (k_mesh's job loop)
                    double scale = 1.0;
                    if (identity) {
                        po = wo;
                        pd = wd;
                    } else {
                        po = xf_point(J.W, wo);
                        const d3 dl = xf_dir(J.W, wd);
                        scale = len3(dl) / len3(wd);
                        pd = normalize3(dl);
                    }
                    const double wbound = rec_obj >= 0 ? recorded_t * scale * (1.0 + 1e-7) + 1e-7 : 1.0e300;
                    const d3 inv = mk3(1.0 / pd.x, 1.0 / pd.y, 1.0 / pd.z);
                    found = box_hit(J.qmin, J.qmax, po, inv) && box_clip(J.bmin, J.bmax, po, inv, t0, t1);
                    if (wbound < t1) t1 = wbound;
                    if (t0 > t1) found = false;
                }
                const uint64_t pushing = __ballot(found);
                if (found) {
                    const uint32_t e = nr + (uint32_t)__popcll(pushing & lane_lt);
                    r_f[e] = po.x;
                    r_f[RT_POOL_READY + e] = po.y;
                    r_f[2 * RT_POOL_READY + e] = po.z;
                    r_f[3 * RT_POOL_READY + e] = pd.x;
                    r_f[4 * RT_POOL_READY + e] = pd.y;
                    r_f[5 * RT_POOL_READY + e] = pd.z;
                    r_f[6 * RT_POOL_READY + e] = t0;
                    r_f[7 * RT_POOL_READY + e] = t1;
                    r_f[8 * RT_POOL_READY + e] = 1.0 / (pd.x * ic_x); // the walk's step per cell (k_mesh: igx = 1 / dgx)
                    r_f[9 * RT_POOL_READY + e] = 1.0 / (pd.y * ic_y);
                    r_f[10 * RT_POOL_READY + e] = 1.0 / (pd.z * ic_z);
                    r_u[e] = (uint32_t)j;
                }
                nr += (uint32_t)__popcll(pushing);
                cursor += batch;
                if (cursor >= block_end) {
                    if (single_block) {
                        cursor = count;
                    } else {
                        // whichever wave gets here first takes the next block (k_mesh's cursors)
                        const uint32_t n_part = n_waves < RT_MESH_CURSORS ? n_waves : RT_MESH_CURSORS, part = wave % n_part;
                        uint32_t nb = 0u;
                        if (lane_id() == 0u) nb = atomicAdd(work + part * 16u, 1u);
                        nb = uniform_u32(nb);
                        cursor = ((uint64_t)n_waves + (uint64_t)nb * n_part + part) * block;
                        block_end = cursor + block;
                    }
                }
                if (cursor > count) cursor = count;
#ifdef RT_MESH_PROF
                ++pf_services;
#endif
            }
            PF_MARK(pf_b);
            // pop: idle lane number r takes entry nr - 1 - r
            const uint32_t r = (uint32_t)__popcll(idle & lane_lt);
            if (!active && r < nr) {
                const uint32_t e = nr - 1u - r;
                o = mk3(r_f[e], r_f[RT_POOL_READY + e], r_f[2 * RT_POOL_READY + e]);
                d = mk3(r_f[3 * RT_POOL_READY + e], r_f[4 * RT_POOL_READY + e], r_f[5 * RT_POOL_READY + e]);
                w_tc = r_f[6 * RT_POOL_READY + e];
                t_end = r_f[7 * RT_POOL_READY + e];
                ri = r_u[e];
                // enter the grid at w_tc
                igx = r_f[8 * RT_POOL_READY + e];
                igy = r_f[9 * RT_POOL_READY + e];
                igz = r_f[10 * RT_POOL_READY + e];
                WALK_LOCATE();
                mt = 1.0e300;
                mtri = 0xFFFFFFFFu;
                active = true;
                walked = true;
            }
            nr -= nr < n_idle ? nr : n_idle;
            PF_MARK(pf_c);
        }

        const bool nothing_left = __ballot(active) == 0ull && nr == 0u && cursor >= count;
        PF_MARK(pf_d);
        if (nothing_left) break;
#ifdef RT_MESH_PROF
        ++pf_rounds;
        pf_lanes_walk += __popcll(__ballot(active));
#endif

        // ---------------------------------------------------------------- one round of the walk
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
#undef PF_MARK
#undef JOB
#undef ogx
#undef ogy
#undef ogz
#undef dgx
#undef dgy
#undef dgz
}
