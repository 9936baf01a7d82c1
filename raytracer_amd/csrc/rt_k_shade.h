// rt_k_shade.h — k_shade, k_tail and the small frame-buffer kernels
// (device code of librt_amd.so; included by rt_kernels.hip, which holds the host side of the C ABI)
#pragma once

#include "rt_k_intersect.h"
#include "rt_k_material.h"
#include "rt_k_mesh.h"

// k_mesh_one's result for a mesh record - the nearest triangle of the scene's one mesh job, t in object space - merged
// into the record's hit `h` (the nearest hit among the other objects) for the world ray (wo, wd): Intersection::to_world's
// t (system.rs:190-195), min_by's order (nearer, or equally near and earlier in the object list), then normal and uv of
// the hit (mesh.rs:97-112 + the Composite's to_world).  The expressions are k_mesh's (its job loop and its write-back),
// on the same inputs.  Returns whether the mesh replaced the hit.
RT_DEV bool mesh_one_finish(const DevScene &sc, d3 wo, d3 wd, double obj_t, uint32_t tri, HitRec &h)
{
    const DevMeshJob &B = sc.jobs[0];
    d3 bo = wo, bd = wd;
    double tw = obj_t;
    if (!B.identity) {
        bo = xf_point(B.W, wo);
        bd = normalize3(xf_dir(B.W, wd));
        tw = signum(obj_t) * len3(xf_point(B.O, bo + bd * obj_t) - wo);
    }
    if (!(tw >= 0.0 && tw < DBL_MAX && (h.obj < 0 || tw < h.t || (tw == h.t && B.object < h.obj)))) return false;
    const size_t tr = (size_t)B.tri0 + tri;
    const double *nb = sc.tri_nrm + tr * 9;
    const d3 n0 = mk3(nb[0], nb[1], nb[2]), n1 = mk3(nb[3], nb[4], nb[5]), n2 = mk3(nb[6], nb[7], nb[8]);
    double tt = 0.0, u = 0.0, v = 0.0;
    tri_hit(sc.tris + tr * 9, bo, bd, tt, u, v);
    d3 nl = B.smooth ? normalize3((1.0 - u - v) * n0 + u * n1 + v * n2) : normalize3(n0 + n1 + n2);
    h.n = B.identity ? nl : normalize3(xf_normal(B.W, nl));
    h.t = tw;
    h.u = u;
    h.v = v;
    h.obj = B.object;
    return true;
}

// rt_trace_rays: the merged hit of mesh record j back to its ray's slot in the per-ray hit buffer (res_tri != null:
// k_mesh_one walked the records and left the merge to mesh_one_finish)
__global__ void k_copy_mesh_hits(DevScene sc, RayQueue mq, HitBuf mqh, QRegions mesh_regions, HitBuf hit, const double *res_t,
                                 const uint32_t *res_tri)
{
    const QMap qm = qmap_load(mesh_regions);
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= qm.total()) return;
    const uint32_t j = qm.pos(idx);
    const uint32_t i = mq.slot[j];
    HitRec h;
    h.t = mqh.t[j];
    h.n = mk3(mqh.n[0][j], mqh.n[1][j], mqh.n[2][j]);
    h.u = mqh.uv[0][j];
    h.v = mqh.uv[1][j];
    h.obj = mqh.obj[j];
    if (res_tri) {
        const uint32_t tri = res_tri[j];
        if (tri != 0xFFFFFFFFu)
            (void)mesh_one_finish(sc, mk3(mq.o[0][j], mq.o[1][j], mq.o[2][j]), mk3(mq.d[0][j], mq.d[1][j], mq.d[2][j]), res_t[j], tri, h);
    }
    hit.t[i] = h.t;
    hit.n[0][i] = h.n.x;
    hit.n[1][i] = h.n.y;
    hit.n[2][i] = h.n.z;
    hit.uv[0][i] = h.u;
    hit.uv[1][i] = h.v;
    hit.obj[i] = h.obj;
}

// The mesh records after the mesh kernel: hit_color + scatter, survivors appended behind k_intersect's.  res_tri != null:
// the mesh kernel was k_mesh_one, which leaves the merge of its triangle into the record's hit to this kernel
// (mesh_one_finish); primary_ids (depth 0 of stratum 0 only) then takes the mesh object's id here.
template <bool SLDS>
__global__ void __launch_bounds__(RT_BLOCK_WIDE) k_shade(DevScene sc_mem, RayQueue qin, QRegions in_regions, HitBuf hit,
                                                     RayQueue qout, QRegions out_regions, double *fb, PassParams pp,
                                                     const double *res_t, const uint32_t *res_tri, int32_t *primary_ids)
{
    extern __shared__ double4 scene_lds[];
    const DevScene sc = scene_in_lds<SLDS>(sc_mem, reinterpret_cast<unsigned char *>(scene_lds));
    __shared__ uint32_t append_lds[RT_APPEND_LDS(RT_BLOCK_WIDE)];
    const QMap qm = qmap_load(in_regions);
    const uint32_t count = qm.total();
    const uint32_t stride = gridDim.x * blockDim.x;
    const d3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
    uint32_t round = blockIdx.x; // the region the workgroup's append goes to first: the next one every time
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride, ++round) {
        const bool have = base + threadIdx.x < count;
        const uint32_t i = have ? qm.pos(base + (threadIdx.x & ~63u), base + threadIdx.x) : 0u; // the record's position
        bool alive = false;
        d3 o = mk3(0, 0, 0), d = mk3(0, 0, 0), thr = mk3(0, 0, 0);
        uint32_t slot = 0, pixel = 0;
        HitRec h;
        h.obj = have ? hit.obj[i] : RT_HIT_PADDING;
        const uint32_t tri = (res_tri && have) ? res_tri[i] : 0xFFFFFFFFu;
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
                if (h.obj >= 0 || tri != 0xFFFFFFFFu) {
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
            if (tri != 0xFFFFFFFFu && mesh_one_finish(sc, o, d, res_t[i], tri, h) && primary_ids) primary_ids[slot] = h.obj;
            d3 X;
            alive = shade_one(sc, o, d, thr, pixel, h, pp, bg, X);
            if (!alive) {
                fb_add(fb + (size_t)slot * 3, thr.x * X.x, thr.y * X.y, thr.z * X.z);
            }
        }
        uint32_t at = block_append(alive, out_regions, round, append_lds);
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
// LL: the interval lists in LDS behind the scene tables, as in k_intersect (the same layout: RT_BLOCK == RT_ISECT_BLOCK)
template <bool CSG, bool MEDIUM, bool TORUS, bool SLDS, bool LL>
__global__ void __launch_bounds__(RT_BLOCK) RT_TAIL_OCC k_tail(DevScene sc_mem, RayQueue q, QRegions q_regions, uint32_t *counts_out,
                                                    double *fb, PassParams pp, uint32_t *guards_out, int has_mesh,
                                                    uint32_t *lane_totals)
{
    static_assert(!LL || RT_BLOCK == RT_ISECT_BLOCK, "one layout of the LDS interval lists for both kernels");
    extern __shared__ double4 scene_lds[];
    const DevScene sc = scene_in_lds<SLDS>(sc_mem, reinterpret_cast<unsigned char *>(scene_lds));
    LdsLists<RT_BLOCK> ll;
    if (LL) {
        unsigned char *lists_lds = reinterpret_cast<unsigned char *>(scene_lds) + ((scene_lds_bytes(sc_mem) + 15u) & ~15u);
        ll.cap_ = (int)sc_mem.lds_list_cap;
        ll.n_lists = (int)sc_mem.lds_list_n;
        ll.t = reinterpret_cast<double *>(lists_lds) + threadIdx.x;
        ll.tag = reinterpret_cast<uint32_t *>(lists_lds + (size_t)ll.n_lists * 2 * ll.cap_ * RT_BLOCK * 8) + threadIdx.x;
        ll.counts = 0u;
    }
    const QMap qm = qmap_load(q_regions);
    const uint32_t count = qm.total();
    // a scene with meshes: one wave per path (the 64 lanes share the mesh test); without: one lane per path
    // (one lane per path, 64 paths to a wave: spreading few paths over all resident waves, 8 to 32 to a wave, as k_tail_grid
    // does, measured the same here - csg / torus 4K, fog_light 4K x 4, cornell_box 1080p within 1 % either way: without a
    // grid walk a level lasts as long as one lane's own chain of instructions)
    const bool per_lane = !has_mesh;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, n_threads = gridDim.x * blockDim.x;
    const uint32_t unit = per_lane ? tid : tid >> 6, n_units = per_lane ? n_threads : n_threads >> 6;
    const bool writer = per_lane || lane_id() == 0u;
    const d3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
    Guards gd;
    for (int k = 0; k < 5; ++k) gd.g[k] = 0u;
    gd.overflow = 0u;
    uint32_t my_rays = 0u, my_depth = 0u; // per_lane: segments traced beyond the first, deepest level reached
    for (uint32_t idx = unit; idx < count; idx += n_units) {
        const uint32_t i = qm.pos(idx);
        d3 o = mk3(q.o[0][i], q.o[1][i], q.o[2][i]);
        d3 d = mk3(q.d[0][i], q.d[1][i], q.d[2][i]);
        d3 thr = mk3(q.thr[0][i], q.thr[1][i], q.thr[2][i]);
        const uint32_t slot = q.slot[i], pixel = q.pixel[i];
        PassParams p = pp;
        for (;;) {
            HitRec h;
            bool wants_mesh;
            if constexpr (LL) {
                wants_mesh = intersect_analytic<CSG, MEDIUM, TORUS, true>(sc, o, d, pixel, p, h, gd, ll);
            } else {
                wants_mesh = intersect_analytic<CSG, MEDIUM, TORUS>(sc, o, d, pixel, p, h, gd);
            }
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
                    fb_add(fb + (size_t)slot * 3, thr.x * X.x, thr.y * X.y, thr.z * X.z);
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

// The lengths of the queues that passes first .. first + n - 1 appended to, as one word each: the rays entering the
// next depth and the pass's mesh records (`rays`, `mesh`: the counters of depth 0; depth k's are k words on)
__global__ void k_queue_totals(QRegions rays, QRegions mesh, uint32_t *n_rays, uint32_t *n_mesh, uint32_t first, uint32_t n)
{
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        const uint32_t depth = first + k;
        QRegions r = rays, m = mesh;
        r.cnt += depth + 1u;
        m.cnt += depth;
        n_rays[depth + 1u] = qmap_load(r).total();
        n_mesh[depth] = qmap_load(m).total();
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

// main.rs:201-217 per channel: (sum / divisor).sqrt() * 255, min 255, `as u8` (saturating; NaN -> 255 through
// f64::min, then 255).  Four channels per thread, one 32-bit store; the last thread takes what is left.
__global__ void k_tonemap(const double *__restrict__ rgb_sum, uint64_t n_values, double divisor, uint8_t *__restrict__ rgb8)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n_values) return;
    uint32_t packed = 0u;
    const uint32_t n = (uint32_t)((n_values - i) < 4 ? (n_values - i) : 4);
    for (uint32_t k = 0; k < 4; ++k) {
        if (k >= n) break;
        double v = sqrt(rgb_sum[i + k] / divisor) * 255.0;
        v = (v != v) ? 255.0 : (v < 255.0 ? v : 255.0);
        const uint32_t b = v <= 0.0 ? 0u : (uint32_t)v;
        packed |= b << (8 * k);
    }
    if (n == 4 && ((uintptr_t)(rgb8 + i) & 3u) == 0) {
        *(uint32_t *)(rgb8 + i) = packed;
    } else {
        for (uint32_t k = 0; k < n; ++k) rgb8[i + k] = (uint8_t)(packed >> (8 * k));
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

// ... and of a shard that has been tone-mapped where it was rendered (3 bytes per slot instead of 24 across the links)
__global__ void k_unshard_rgb8(rt_shard_geom g, uint32_t rank, uint32_t n_slots, const uint8_t *__restrict__ shard_rgb8, uint8_t *__restrict__ full_rgb8)
{
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x, y;
    if (slot < n_slots && rt_shard_decode(&g, rank, slot, &x, &y)) {
        const size_t p = (size_t)y * g.width + x;
        full_rgb8[p * 3 + 0] = shard_rgb8[(size_t)slot * 3 + 0];
        full_rgb8[p * 3 + 1] = shard_rgb8[(size_t)slot * 3 + 1];
        full_rgb8[p * 3 + 2] = shard_rgb8[(size_t)slot * 3 + 2];
    }
}

// The same for all shards of a frame at once (rt_render_multi): their buffers lie one after the other in rank
// order, `shards_*` is the first.  Tiles are dealt round-robin, so the first n_tiles % N shards own one tile more
// than the others and a slot's shard follows from its position.
__global__ void k_unshard_all(rt_shard_geom g, uint32_t total_slots, const double *shards_rgb, double *full_rgb,
                              const int32_t *shards_ids, int32_t *full_ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_slots) return;
    const uint32_t gt = i / g.tile_px, j = i - gt * g.tile_px;
    const uint32_t base = g.n_tiles / g.shard_count, rem = g.n_tiles % g.shard_count;
    uint32_t rank, local_tile;
    if (gt < rem * (base + 1u)) {
        rank = gt / (base + 1u);
        local_tile = gt - rank * (base + 1u);
    } else { // base >= 1 here: a shard without tiles has no slots
        const uint32_t r = gt - rem * (base + 1u);
        rank = rem + r / base;
        local_tile = r % base;
    }
    uint32_t x, y;
    if (rt_shard_decode(&g, rank, local_tile * g.tile_px + j, &x, &y)) {
        const size_t p = (size_t)y * g.width + x;
        if (shards_rgb && full_rgb) {
            full_rgb[p * 3 + 0] = shards_rgb[(size_t)i * 3 + 0];
            full_rgb[p * 3 + 1] = shards_rgb[(size_t)i * 3 + 1];
            full_rgb[p * 3 + 2] = shards_rgb[(size_t)i * 3 + 2];
        }
        if (shards_ids && full_ids) full_ids[p] = shards_ids[i];
    }
}

// ... and for shards that were tone-mapped where they were rendered (rt_render_multi_rgb8)
__global__ void k_unshard_all_rgb8(rt_shard_geom g, uint32_t total_slots, const uint8_t *__restrict__ shards_rgb8, uint8_t *__restrict__ full_rgb8)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_slots) return;
    const uint32_t gt = i / g.tile_px, j = i - gt * g.tile_px;
    const uint32_t base = g.n_tiles / g.shard_count, rem = g.n_tiles % g.shard_count;
    uint32_t rank, local_tile;
    if (gt < rem * (base + 1u)) {
        rank = gt / (base + 1u);
        local_tile = gt - rank * (base + 1u);
    } else {
        const uint32_t r = gt - rem * (base + 1u);
        rank = rem + r / base;
        local_tile = r % base;
    }
    uint32_t x, y;
    if (rt_shard_decode(&g, rank, local_tile * g.tile_px + j, &x, &y)) {
        const size_t p = (size_t)y * g.width + x;
        full_rgb8[p * 3 + 0] = shards_rgb8[(size_t)i * 3 + 0];
        full_rgb8[p * 3 + 1] = shards_rgb8[(size_t)i * 3 + 1];
        full_rgb8[p * 3 + 2] = shards_rgb8[(size_t)i * 3 + 2];
    }
}

// rt_trace_rays: the caller's interleaved rays (xyz, xyz, ...) into queue entries 0 .. n-1, ray i keyed as pixel i
__global__ void k_rays_in(uint32_t n, const double *__restrict__ origins, const double *__restrict__ directions, RayQueue q)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < 3; ++c) {
        q.o[c][i] = origins[(size_t)i * 3 + c];
        q.d[c][i] = directions[(size_t)i * 3 + c];
        q.thr[c][i] = 1.0;
    }
    q.slot[i] = i;
    q.pixel[i] = i;
}

// ... and the per-ray hit records out as rt_hit structs
__global__ void k_hits_out(uint32_t n, HitBuf hit, rt_hit *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rt_hit h;
    h.t = hit.t[i];
    h.n[0] = hit.n[0][i];
    h.n[1] = hit.n[1][i];
    h.n[2] = hit.n[2][i];
    h.uv[0] = hit.uv[0][i];
    h.uv[1] = hit.uv[1][i];
    h.object = hit.obj[i];
    h.pad = 0;
    out[i] = h;
}
