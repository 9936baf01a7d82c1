// rt_k_tail.h — k_tail_grid: the rest of a stratum in one launch, one lane per path, for scenes with a single
// (mesh object, mesh) grid job (device code of librt_amd.so; included by rt_kernels.hip)
#pragma once

#include "rt_k_mesh_pool.h"
#include "rt_k_shade.h"

// Ray::cast's recursion (system.rs:112-142) unrolled on the device for the paths that are left once the bounce
// passes have thinned out.  The passes are launched per depth level with queues in memory between them; their
// cost has a floor - a pass of a few ten thousand rays lasts as long as its slowest ray's grid walk, about 100
// microseconds, plus the host's read-back of the queue length - and the last twenty levels of a 4K mesh.scene
// frame hold 0.3 % of its rays.  k_tail (rt_k_shade.h) takes those levels in one launch with one WAVE per path
// and the triangles by brute force, which is right for a few thousand paths; this kernel takes over much
// earlier (RT_TAIL_PATHS: as many paths as the chip holds at once, 131 k of a 4K frame's 8.3 M): a wave
// owns 64 paths, one per lane, and loops intersect -> mesh -> shade until they have all ended.  Nothing is
// exchanged between waves - no queues, no appends, no counters, no barriers after the tables are staged - so a
// wave's time is its own paths' and the launch lasts as long as the wave with the deepest path.  Per level:
//   intersect   intersect_analytic, as k_intersect (one lane per path)
//   mesh        more than RT_TAILG_BRUTE_MAX lanes want the mesh: each walks its ray through the grid (k_mesh_one's
//               PREPARE expressions, the rounds of rt_k_mesh_grid_round.inc on tables in LDS, mesh_one_finish);
//               fewer: the wave takes them one at a time, 64 lanes on one ray (mesh_brute_wave, as k_tail) - a
//               lone walking lane takes tens of microseconds over a grazing ray, the wave a few
//   shade       shade_one; a path that ends adds its colour to its pixel (fb_add)
// Same expressions on the same inputs as the passes, so frames are bit-identical wherever the switch happens
// (test_tail_takeover_is_invisible).  Rays are reported as k_tail's lane-per-path mode does: a total and the
// deepest level (lane_totals).
#define RT_TAILG_LANES 256
#ifndef RT_TAILG_WAVES
#define RT_TAILG_WAVES 2
#endif
#ifndef RT_TAILG_STEPWISE
#define RT_TAILG_STEPWISE 0 // 1: the walk in steps at each lane's own pace (rt_k_mesh_grid_step.inc) instead of rounds. Measured on the
                            // 4K bench frame (131 k paths taken over): 0.553-0.560 ms against 0.493-0.501 with rounds - with the chip full
                            // of these waves a level is bound by instruction issue, and a wave that executes the walk half and the
                            // test half in every iteration issues more
#endif
#ifndef RT_TAILG_MIN_LANES
#define RT_TAILG_MIN_LANES 8u // fewest paths a wave takes at a time when few are left
#endif
#ifndef RT_TAILG_BRUTE_MAX
#define RT_TAILG_BRUTE_MAX 6u // lanes wanting the mesh up to which the wave takes them one by one
#endif

#ifdef RT_TAILG_PROF
// per level (relative to the take-over) and phase (intersect, mesh by grid, mesh by the wave, shade): the slowest
// wave's time and the sum over waves, in ticks of the 100 MHz wall clock; [4] = waves that ran the level, [5] = their live lanes,
// [6] = lanes wanting the mesh, [7] = rounds of the grid walk
__device__ unsigned long long g_tg_max[64][4];
__device__ unsigned long long g_tg_sum[64][8];
#define TG_MARK(PHASE)                                                                                                 \
    do {                                                                                                               \
        const unsigned long long n_ = wall_clock64();                                                                  \
        if (lane_id() == 0u && tg_level < 64u) {                                                                       \
            atomicMax(&g_tg_max[tg_level][PHASE], n_ - tg_t);                                                          \
            atomicAdd(&g_tg_sum[tg_level][PHASE], n_ - tg_t);                                                          \
        }                                                                                                              \
        tg_t = n_;                                                                                                     \
    } while (0)
#else
#define TG_MARK(PHASE) do { } while (0)
#endif

RT_DEV double bcast_f64(double v, int lane)
{
    return __hiloint2double(__shfl(__double2hiint(v), lane), __shfl(__double2loint(v), lane));
}
RT_DEV d3 bcast_d3(d3 v, int lane) { return mk3(bcast_f64(v.x, lane), bcast_f64(v.y, lane), bcast_f64(v.z, lane)); }

// dynamic LDS: [scene tables (scene_lds_bytes, 16-byte aligned) | job record | grid tables]
template <bool CSG, bool MEDIUM, bool TORUS>
__global__ void __launch_bounds__(RT_TAILG_LANES) __attribute__((amdgpu_waves_per_eu(RT_TAILG_WAVES, RT_TAILG_WAVES)))
k_tail_grid(DevScene sc_mem, RayQueue q, QRegions q_regions, double *fb, PassParams pp, uint32_t *guards_out,
            uint32_t *lane_totals)
{
    extern __shared__ double4 tail_lds[];
    unsigned char *lds = reinterpret_cast<unsigned char *>(tail_lds);
    DevMeshJob *jobs_lds = reinterpret_cast<DevMeshJob *>(lds + ((scene_lds_bytes(sc_mem) + 15u) & ~15u));
    uint2 *grid_lds = reinterpret_cast<uint2 *>(jobs_lds + 1);
    uint32_t *dist_lds = reinterpret_cast<uint32_t *>(grid_lds + sc_mem.n_grid_words);
    uint32_t *starts_lds = dist_lds + sc_mem.n_dist_words;
    const bool starts_in_lds = sc_mem.starts_in_lds != 0u;
    {
        const uint32_t *jw = reinterpret_cast<const uint32_t *>(sc_mem.jobs);
        uint32_t *jl = reinterpret_cast<uint32_t *>(jobs_lds);
        for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(DevMeshJob) / 4); i += blockDim.x) jl[i] = jw[i];
        const uint2 *gw = reinterpret_cast<const uint2 *>(sc_mem.grid_bits);
        for (uint32_t i = threadIdx.x; i < sc_mem.n_grid_words; i += blockDim.x) grid_lds[i] = gw[i];
        for (uint32_t i = threadIdx.x; i < sc_mem.n_dist_words; i += blockDim.x) dist_lds[i] = sc_mem.grid_dist[i];
        if (starts_in_lds)
            for (uint32_t i = threadIdx.x; i < sc_mem.n_cell_starts; i += blockDim.x) starts_lds[i] = sc_mem.cell_start[i];
    }
    const DevScene sc = scene_in_lds<true>(sc_mem, lds); // ends with the barrier that covers the copies above too
#define JOB(J) (*(const DevMeshJob *)jobs_lds) // the one job

    // the job's constants: the same in every lane (k_mesh_one)
    const uint32_t nx = uniform_u32(JOB(0).n[0]), ny = uniform_u32(JOB(0).n[1]), nz = uniform_u32(JOB(0).n[2]);
    const uint32_t bits0 = uniform_u32(JOB(0).bits0), cell0 = uniform_u32(JOB(0).cell0), dist0 = uniform_u32(JOB(0).dist0);
    const bool identity = uniform_u32((uint32_t)JOB(0).identity) != 0u;
    const double bmin_x = uniform_f64(JOB(0).bmin[0]), bmin_y = uniform_f64(JOB(0).bmin[1]), bmin_z = uniform_f64(JOB(0).bmin[2]);
    const double ic_x = uniform_f64(JOB(0).inv_cell[0]), ic_y = uniform_f64(JOB(0).inv_cell[1]), ic_z = uniform_f64(JOB(0).inv_cell[2]);
    const int jb = 0;
    (void)jb;
// the ray in grid coordinates, two operations on scalars away (k_mesh_one)
#define ogx ((o.x - bmin_x) * ic_x)
#define ogy ((o.y - bmin_y) * ic_y)
#define ogz ((o.z - bmin_z) * ic_z)
#define dgx (d.x * ic_x)
#define dgy (d.y * ic_y)
#define dgz (d.z * ic_z)
#define PF_MARK(ACC) do { } while (0)

    const QMap qm = qmap_load(q_regions);
    const uint32_t count = qm.total();
    const uint32_t n_waves = gridDim.x * (RT_TAILG_LANES / 64);
    const uint32_t wave = uniform_u32(blockIdx.x * (RT_TAILG_LANES / 64) + (threadIdx.x >> 6));
    const d3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
    Guards gd;
    for (int k = 0; k < 5; ++k) gd.g[k] = 0u;
    gd.overflow = 0u;
    uint32_t my_rays = 0u, my_depth = 0u; // segments traced beyond each path's first, deepest level reached

    // Few paths: fewer of them per wave, over all the waves the launch has (the host sizes it for the chip) - a wave's
    // level lasts as long as its slowest lane's walk, and its rounds as long as the longest run of records among its lanes
    uint32_t per_wave = RT_TAILG_MIN_LANES;
    while (per_wave < 64u && (uint64_t)per_wave * n_waves < count) per_wave <<= 1;
    for (uint64_t base = (uint64_t)wave * per_wave; base < count; base += (uint64_t)n_waves * per_wave) {
        bool alive = lane_id() < per_wave && base + lane_id() < count;
        d3 wo = mk3(0, 0, 0), wd = mk3(0, 0, 1), thr = mk3(0, 0, 0);
        uint32_t slot = 0u, pixel = 0u;
        if (alive) {
            const uint32_t i = qm.pos((uint32_t)base, (uint32_t)(base + lane_id()));
            wo = mk3(q.o[0][i], q.o[1][i], q.o[2][i]);
            wd = mk3(q.d[0][i], q.d[1][i], q.d[2][i]);
            thr = mk3(q.thr[0][i], q.thr[1][i], q.thr[2][i]);
            slot = q.slot[i];
            pixel = q.pixel[i];
        }
        PassParams p = pp; // the wave's paths are all at the same depth: each level advances every live one
        while (__ballot(alive) != 0ull) {
#ifdef RT_TAILG_PROF
            const uint32_t tg_level = p.depth - pp.depth;
            unsigned long long tg_t = wall_clock64();
            if (lane_id() == 0u && tg_level < 64u) {
                atomicAdd(&g_tg_sum[tg_level][4], 1ull);
                atomicAdd(&g_tg_sum[tg_level][5], (unsigned long long)__popcll(__ballot(alive)));
            }
#endif
            HitRec h;
            h.t = -1.0;
            h.n = mk3(0, 0, 0);
            h.u = h.v = 0.0;
            h.obj = -1;
            bool wants = false;
            if (alive) wants = intersect_analytic<CSG, MEDIUM, TORUS>(sc, wo, wd, pixel, p, h, gd);
            const uint64_t want_mask = __ballot(wants);
            TG_MARK(0);
#ifdef RT_TAILG_PROF
            if (lane_id() == 0u && tg_level < 64u) atomicAdd(&g_tg_sum[tg_level][6], (unsigned long long)__popcll(want_mask));
#endif
            if ((uint32_t)__popcll(want_mask) > RT_TAILG_BRUTE_MAX) {
                // ------------------------------------------------------------ one lane per ray through the grid
                // k_mesh_one's PREPARE: Ray::transform (system.rs:145-152), the Mesh::new box, the grid's bounds; a
                // hit farther than the one found among the other objects cannot win (slack keeps ties)
                bool active = false;
                d3 o = wo, d = wd;
                double igx = 0, igy = 0, igz = 0, tmx = 0, tmy = 0, tmz = 0, w_tc = 0, t_end = 0;
                int cx = 0, cy = 0, cz = 0;
                double mt = 1.0e300;
                uint32_t mtri = 0xFFFFFFFFu;
#if RT_TAILG_STEPWISE
                bool in_cell = false; // the lane is inside a cell's run of records (rt_k_mesh_grid_step.inc)
                uint32_t lpos = 0u, llast = 0u;
                float fox = 0.f, foy = 0.f, foz = 0.f;
#endif
                if (wants) {
                    const DevMeshJob &J = JOB(0);
                    double scale = 1.0;
                    if (!identity) {
                        o = xf_point(J.W, wo);
                        const d3 dl = xf_dir(J.W, wd);
                        scale = len3(dl) / len3(wd);
                        d = normalize3(dl);
                    }
                    const double wbound = h.obj >= 0 ? h.t * scale * (1.0 + 1e-7) + 1e-7 : 1.0e300;
                    const d3 inv = mk3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                    double t0 = 0.0, t1 = 0.0;
                    bool found = box_hit(J.qmin, J.qmax, o, inv) && box_clip(J.bmin, J.bmax, o, inv, t0, t1);
                    if (wbound < t1) t1 = wbound;
                    if (t0 > t1) found = false;
                    if (found) {
                        w_tc = t0;
                        t_end = t1;
                        igx = 1.0 / (d.x * ic_x); // the walk's step per cell
                        igy = 1.0 / (d.y * ic_y);
                        igz = 1.0 / (d.z * ic_z);
                        WALK_LOCATE();
                        active = true;
                    }
                }
                while (__ballot(active) != 0ull) {
#ifdef RT_TAILG_PROF
                    if (lane_id() == 0u && tg_level < 64u) atomicAdd(&g_tg_sum[tg_level][7], 1ull);
#endif
#if RT_TAILG_STEPWISE
#include "rt_k_mesh_grid_step.inc"
#else
#include "rt_k_mesh_grid_round.inc"
#endif
                }
                if (wants && mtri != 0xFFFFFFFFu) (void)mesh_one_finish(sc, wo, wd, mt, mtri, h);
                TG_MARK(1);
            } else {
                // ------------------------------------------------------------ the wave on one ray at a time
                uint64_t m = want_mask;
                while (m) {
                    const int src = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const d3 bo = bcast_d3(wo, src), bd = bcast_d3(wd, src);
                    int best_obj = __shfl(h.obj, src);
                    const double rec_t = bcast_f64(h.t, src);
                    double best_t = best_obj >= 0 ? rec_t : DBL_MAX;
                    uint32_t best_aux = 0u;
                    const bool replaced = mesh_brute_wave(sc, bo, bd, best_obj, best_t, best_aux);
                    if (replaced && (int)lane_id() == src) {
                        h.t = best_t;
                        h.obj = best_obj;
                        mesh_hit_surface(sc, wo, wd, best_obj, best_aux, h.n, h.u, h.v);
                    }
                }
                TG_MARK(2);
            }
            if (alive) {
                d3 X;
                if (shade_one(sc, wo, wd, thr, pixel, h, p, bg, X)) {
                    ++my_rays;
                    my_depth = p.depth + 1u > my_depth ? p.depth + 1u : my_depth;
                } else {
                    fb_add(fb + (size_t)slot * 3, thr.x * X.x, thr.y * X.y, thr.z * X.z);
                    alive = false;
                }
            }
            TG_MARK(3);
            ++p.depth;
        }
    }
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
        for (int k = 0; k < 5; ++k)
            if (gd.g[k]) atomicAdd(&guards_out[k], gd.g[k]);
        if (gd.overflow) atomicAdd(&guards_out[5], gd.overflow);
    }
#undef PF_MARK
#undef JOB
#undef ogx
#undef ogy
#undef ogz
#undef dgx
#undef dgy
#undef dgz
}
