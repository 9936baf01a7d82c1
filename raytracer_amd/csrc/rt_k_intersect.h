// rt_k_intersect.h — k_intersect: nearest hit over every object except the triangle work
// (device code of librt_amd.so; included by rt_kernels.hip, which holds the host side of the C ABI)
#pragma once

#include "rt_k_common.h"
#include "rt_k_material.h"

// ------------------------------------------------------------------ nearest hit
struct Candidate {
    double t;
    uint32_t aux;
    bool ok;
};

// first_positive_intersection / first_intersection (shapes/mod.rs:45-55) over a list
template <class LST>
RT_DEV Candidate pick_from_list(const LST &l, int mode)
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
    for (int i = 0; i < 2 * l.cap(); ++i) {
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

// LL: the ray's interval lists are in LDS (`ll`: LdsLists, rt_dev_shapes.h), not in scratch memory
template <bool TORUS, bool LL, class LISTS>
RT_CSG_FN Candidate eval_csg_object(const DevScene &sc, const DevObject &ob, d3 o, d3 d, Guards &gd, LISTS &ll)
{
    Candidate c;
    c.ok = false;
    c.t = 0.0;
    c.aux = 0u;
    if constexpr (LL) {
        ll.counts = 0u;
        const int r = csg_eval<TORUS>(sc, ob, o, d, ll, ll.n_lists - 1, gd);
        if (r < 0) return c;
        auto &&res = ll[r];
        return pick_from_list(res, ob.mode);
    } else {
        IvListT<DEV_MAX_INTERVALS> lists[DEV_CSG_STACK + 1];
        const int r = csg_eval<TORUS>(sc, ob, o, d, lists, DEV_CSG_STACK, gd);
        if (r < 0) return c;
        return pick_from_list(lists[r], ob.mode);
    }
}

// Does the ray need the walk through a mesh of this object?  Only a conservative answer is wanted here (k_mesh
// applies the reference's own tests, mesh.rs:32-42,117, to every ray it is handed): the ray must enter the
// padded bounds of the mesh's triangles at t >= 0 and before the nearest hit found among the other objects
// (`have_hit`, `best_t`; slack keeps ties in play) - and, where the flattener found them worth having, one of the
// boxes that hold the triangles between them (DevMesh::part).  Those bounds lie inside the Mesh::new box, so a ray
// that misses them meets no triangle.  The test is made in world space, on the boxes' world bounds (DevMesh::wbox,
// wpart): the ray as it is, no transform, no square root, the slab parameters the world ray's own.
// (This test took 18 % of k_intersect's wave time with exact divisions and box_clip's branches,
// profiles/r03e_isect_wave_time.txt: being conservative it needs neither - the reciprocals are the hardware's
// approximation refined once, 2^-40 or better, the slabs are min / max chains in which a NaN (0 x inf: the ray runs
// inside a face's plane) drops out, and the world boxes are padded by 1e-7 of their size, rt_flatten.cpp.)
RT_DEV double approx_recip(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    const double r1 = fma(r, fma(-x, r, 1.0), r);
    return (r1 == r1 && fabs(r1) <= 1.0e300) ? r1 : r; // x = 0, denormal or huge: the hardware's own answer (inf, 0)
}
RT_DEV bool slab_enters(const double *b, d3 o, d3 inv, double bound)
{
    double t0 = 0.0, t1 = 1.0e300;
    const double ax = (b[0] - o.x) * inv.x, bx = (b[3] - o.x) * inv.x;
    t0 = fmax(t0, fmin(ax, bx));
    t1 = fmin(t1, fmax(ax, bx));
    const double ay = (b[1] - o.y) * inv.y, by = (b[4] - o.y) * inv.y;
    t0 = fmax(t0, fmin(ay, by));
    t1 = fmin(t1, fmax(ay, by));
    const double az = (b[2] - o.z) * inv.z, bz = (b[5] - o.z) * inv.z;
    t0 = fmax(t0, fmin(az, bz));
    t1 = fmin(t1, fmax(az, bz));
    return !(t0 > t1) && t0 <= bound + 1e-9 * (fabs(t1) + 1.0);
}
RT_DEV bool mesh_object_wanted(const DevScene &sc, const DevObject &ob, d3 o, d3 d, bool have_hit, double best_t)
{
    const double bound = have_hit ? best_t * (1.0 + 1e-7) + 1e-7 : 1.0e300;
    const d3 inv = mk3(approx_recip(d.x), approx_recip(d.y), approx_recip(d.z));
    bool any = false;
    for (int mi = 0; mi < ob.n_meshes; ++mi) {
        const DevMesh &m = sc.meshes[ob.mesh0 + mi];
        bool met = slab_enters(m.wbox, o, inv, bound);
        if (met && m.n_parts) {
            met = false;
            for (uint32_t k = 0; k < DEV_MESH_PARTS; ++k) {
                if (k >= m.n_parts) break;
                met = met || slab_enters(m.wpart[k], o, inv, bound);
            }
        }
        any = any || met;
    }
    return any;
}

// HomogenousMedium::intersect (homogenous_medium.rs:29-57) on the lists `stack` (stk + 1 of them)
template <bool CSG, bool TORUS, class STACK>
RT_CSG_FN Candidate medium_on_lists(const DevScene &sc, const DevObject &ob, uint32_t obj_index, d3 o, d3 d, const PassParams &pp,
                                    uint32_t pixel, Guards &gd, STACK &stack, const int stk)
{
    Candidate c;
    c.ok = false;
    c.t = 0.0;
    c.aux = 0u;
    const DevLeaf &F = sc.leaves[ob.frame];
    LeafRay ry = to_leaf(F, o, d);
    // a world-space distance along the ray is `scale` times longer in the medium's space
    double scale = F.identity ? 1.0 : len3(xf_dir(F.W, d));
    int r = 0;
    if (ob.prog_len == 0) {
        auto &&l0 = stack[0];
        l0.n = 0;
        leaf_world_intervals<TORUS>(sc.leaves[ob.leaf0], 0u, o, d, l0, gd);
    } else {
        r = CSG ? csg_eval<TORUS>(sc, ob, o, d, stack, stk, gd) : -1;
        if (r < 0) return c;
    }
    auto &&res = stack[r];
    const int res_n = res.n;
    if (res_n == 0) return c;
    int first = 0;
    while (first < res_n && res.t[2 * first] < 0.0 && res.t[2 * first + 1] < 0.0) ++first; // skip_negative_intervals
    double dlen = len3(ry.d);
    uint32_t k = 0;
    for (int i = first; i < res_n; ++i) {
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

template <bool CSG, bool TORUS, bool LL, class LISTS>
RT_CSG_FN Candidate eval_medium_object(const DevScene &sc, const DevObject &ob, uint32_t obj_index, d3 o,
                                                     d3 d, const PassParams &pp, uint32_t pixel, Guards &gd, LISTS &ll)
{
    if constexpr (LL) {
        ll.counts = 0u;
        return medium_on_lists<CSG, TORUS>(sc, ob, obj_index, o, d, pp, pixel, gd, ll, ll.n_lists - 1);
    } else {
        IvListT<DEV_MAX_INTERVALS> lists[CSG ? DEV_CSG_STACK + 1 : 1];
        return medium_on_lists<CSG, TORUS>(sc, ob, obj_index, o, d, pp, pixel, gd, lists, CSG ? DEV_CSG_STACK : 0);
    }
}

// Pass 1 of the nearest-hit search: every object except the triangle work.
// A ray that passes the bounding box of a mesh object is appended to
// `mesh_queue`; k_mesh then walks the mesh grids and merges its result into the
// hit record.  CSG / MEDIUM prune code (and registers) the
// scene does not need.
// Ray::trace (system.rs:122-133) over every object except the triangle work; returns whether
// the ray still has to visit a mesh.
// LL: the interval lists are in LDS (`ll`), else `ll` is a NoLdsLists and the lists are in scratch memory.
struct NoLdsLists {};
template <bool CSG, bool MEDIUM, bool TORUS, bool LL, class LISTS>
RT_DEV bool intersect_analytic(const DevScene &sc, d3 o, d3 d, uint32_t pixel, const PassParams &pp, HitRec &h, Guards &gd, LISTS &ll)
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
            // ... and one whose bounds begin behind the nearest hit so far cannot replace it (a later object must be
            // strictly nearer, system.rs:125-129; the bounds are padded far beyond the rounding of either side) - bit 1
            // of has_bounds: unless a skipped to_world may bring its hits nearer than its primitives (rt_flatten.cpp)
            if ((ob.has_bounds & 2) && best_obj >= 0 && t0 > best_t) continue;
        }
        switch (ob.kind) {
        case DO_LEAF: c = eval_leaf_object<TORUS>(sc, ob, o, d, gd); break;
        case DO_CSG:
            if (CSG) c = eval_csg_object<TORUS, LL>(sc, ob, o, d, gd, ll);
            break;
        case DO_MESH: wants_mesh = true; break; // decided below, once the other objects' nearest hit is known
        default:
            if (MEDIUM) c = eval_medium_object<CSG, TORUS, LL>(sc, ob, (uint32_t)k, o, d, pp, pixel, gd, ll);
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
            const uint32_t skips = CSG ? (best_aux >> TAG_SKIP_SHIFT) : 0u;
            if (CSG && skips) {
                // the hit's t carries the scale of every to_world the reference skipped (DEV_PROG_SKIP);
                // the surface point is where the ray really meets the leaf
                double t_surface = best_t;
                for (int li = 0; li < ob.n_leaves; ++li) {
                    const DevLeaf &Q = sc.leaves[ob.leaf0 + li];
                    if (Q.kind == DL_NONE && Q.skip_ord > 0 && ((skips >> (Q.skip_ord - 1)) & 1u)) t_surface /= skip_scale(Q, d);
                }
                pw = o + d * t_surface;
            }
            LeafRay ry = to_leaf(L, o, d);
            d3 ql = L.identity ? pw : xf_point(L.W, pw);
            d3 nl;
            leaf_surface(L, ry, ql, best_aux, need_uv, nl, u, v);
            // system.rs:196 normalises every world normal; only a plane's own normal can arrive
            // un-normalised (`normal <0, 2, 0>`), every other shape hands over a unit vector
            if (L.kind == DL_PLANE) {
                // ... which for a plane is one of two constants (plane_constants in rt_flatten.cpp: the same operations)
                const double *wn = L.p + ((dot3(ry.d, mk3(L.p[3], L.p[4], L.p[5])) > 0.0) ? 21 : 6);
                n = mk3(wn[0], wn[1], wn[2]);
            } else {
                n = L.identity ? nl : normalize3(xf_normal(L.W, nl));
            }
            if (CSG && skips) {
                // ... and the normal misses the skipped nodes' maps, innermost first
                for (int li = 0; li < ob.n_leaves; ++li) {
                    const DevLeaf &Q = sc.leaves[ob.leaf0 + li];
                    if (Q.kind == DL_NONE && Q.skip_ord > 0 && ((skips >> (Q.skip_ord - 1)) & 1u)) n = normalize3(xf_normal(Q.O, n));
                }
            }
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

// ... with the lists in scratch memory
template <bool CSG, bool MEDIUM, bool TORUS>
RT_DEV bool intersect_analytic(const DevScene &sc, d3 o, d3 d, uint32_t pixel, const PassParams &pp, HitRec &h, Guards &gd)
{
    NoLdsLists none;
    return intersect_analytic<CSG, MEDIUM, TORUS, false>(sc, o, d, pixel, pp, h, gd, none);
}

// One path segment per ray: nearest hit over the analytic objects (Ray::trace, system.rs:122-133), then
//  - a ray that still has to visit a mesh is appended to the mesh record queue (`mq` + `mqh`, indexed by queue
//    position): origin, direction, attenuation so far, pixel keys and the hit found so far.  k_mesh reads the
//    records by position (no gather), merges the triangles' nearest hit into them, k_shade finishes them;
//  - every other ray is finished here (FUSED): hit_color + Material::scatter (system.rs:135-142) straight from
//    registers - its hit record never goes to memory - and the scattered ray is appended to `qout`, or the
//    path's colour is added to its pixel.
// !FUSED (rt_trace_rays): no shading; the hit record of every ray i goes to `hit[i]`, and mesh records carry i
// in their slot field so that k_copy_mesh_hits can put the merged hit back.
// LL: the interval lists of the CSG / fog variants in LDS behind the scene tables (sc_mem.lds_list_cap intervals per
// list, sc_mem.lds_list_n lists per ray: the launch code has checked that two workgroups per CU find room)
// PQ: 0 - whether the queue is the primary one (camera rays regenerated from the slot number) is read from the queue at
// run time; 1 / 2 - it is known to be the primary one / a ray queue, and the other half's code, pointers and registers
// are not in the kernel (the wavefront loop launches these; 2 may also run at another occupancy, RT_QUEUE_WAVES)
template <bool CSG, bool MEDIUM, bool TORUS, bool SLDS, bool FUSED, bool LL, int PQ>
__global__ void __launch_bounds__(RT_ISECT_BLOCK) RT_ISECT_OCC k_intersect(DevScene sc_mem, RayQueue q, const uint32_t *slots_ptr, QRegions q_regions,
                                                         PassParams pp, int32_t *primary_ids, uint32_t *guards_out,
                                                         RayQueue mq, HitBuf mqh, QRegions mesh_regions, RayQueue qout,
                                                         QRegions out_regions, double *fb, HitBuf hit, uint32_t *mesh_work, uint32_t *mesh_res_tri)
{
    // k_mesh's block cursors start every pass at zero (this kernel precedes it on the stream)
    if (blockIdx.x == 0 && threadIdx.x < 2 * RT_MESH_CURSORS) mesh_work[threadIdx.x * 16] = 0u;
    extern __shared__ double4 scene_lds[];
    const DevScene sc = scene_in_lds<SLDS>(sc_mem, reinterpret_cast<unsigned char *>(scene_lds));
    __shared__ uint32_t append_lds[RT_APPEND_LDS(RT_ISECT_BLOCK)];
    LdsLists<RT_ISECT_BLOCK> ll;
    if (LL) {
        unsigned char *lists_lds = reinterpret_cast<unsigned char *>(scene_lds) + ((scene_lds_bytes(sc_mem) + 15u) & ~15u);
        ll.cap_ = (int)sc_mem.lds_list_cap;
        ll.n_lists = (int)sc_mem.lds_list_n;
        ll.t = reinterpret_cast<double *>(lists_lds) + threadIdx.x;
        ll.tag = reinterpret_cast<uint32_t *>(lists_lds + (size_t)ll.n_lists * 2 * ll.cap_ * RT_ISECT_BLOCK * 8) + threadIdx.x;
        ll.counts = 0u;
    }
    // the bounce passes append per wave (wave_append, rt_k_common.h), the few entries of a round through LDS unless
    // that holds interval lists; the camera pass, in which most rays go on, appends per workgroup
#ifndef RT_LL_WAVE_APPEND
#define RT_LL_WAVE_APPEND 1
#endif
    constexpr bool WAVE_APPEND = FUSED && PQ == 2 && (!LL || RT_LL_WAVE_APPEND);
    constexpr bool STAGE = WAVE_APPEND && !LL && RT_STAGE_CAP > 0;
    constexpr int SC = STAGE ? RT_STAGE_CAP : 0;
    __shared__ double stage_d[STAGE ? (RT_ISECT_BLOCK / 64) * (15 + 9) * SC : 1];
    __shared__ uint32_t stage_u[STAGE ? (RT_ISECT_BLOCK / 64) * (3 + 2) * SC : 1];
    WaveStage<SC, 15, 3> stage_m;
    WaveStage<SC, 9, 2> stage_q;
    stage_m.fill = stage_q.fill = 0u;
    // where this wave's next append goes first
    uint32_t region_m = (blockIdx.x * (RT_ISECT_BLOCK / 64) + (threadIdx.x >> 6)) & (RT_Q_REGIONS - 1u), region_q = region_m;
    if (STAGE) {
        const uint32_t wv = threadIdx.x >> 6;
        stage_m.d = stage_d + wv * (15 + 9) * SC;
        stage_q.d = stage_m.d + 15 * SC;
        stage_m.u = stage_u + wv * (3 + 2) * SC;
        stage_q.u = stage_m.u + 3 * SC;
    }
    // entries 0..n-1 of a wave's staged list go to the queue
    auto flush_m = [&](uint32_t n) {
        const uint32_t l = lane_id(), at = wave_reserve(mesh_regions, region_m, n, l);
        if (l < n) {
            const double *e = stage_m.d + l;
            mq.o[0][at] = e[0 * SC];
            mq.o[1][at] = e[1 * SC];
            mq.o[2][at] = e[2 * SC];
            mq.d[0][at] = e[3 * SC];
            mq.d[1][at] = e[4 * SC];
            mq.d[2][at] = e[5 * SC];
            mq.thr[0][at] = e[6 * SC];
            mq.thr[1][at] = e[7 * SC];
            mq.thr[2][at] = e[8 * SC];
            mqh.t[at] = e[9 * SC];
            mqh.n[0][at] = e[10 * SC];
            mqh.n[1][at] = e[11 * SC];
            mqh.n[2][at] = e[12 * SC];
            mqh.uv[0][at] = e[13 * SC];
            mqh.uv[1][at] = e[14 * SC];
            mq.slot[at] = stage_m.u[0 * SC + l];
            mq.pixel[at] = stage_m.u[1 * SC + l];
            mqh.obj[at] = (int32_t)stage_m.u[2 * SC + l];
            if (mesh_res_tri) mesh_res_tri[at] = 0xFFFFFFFFu;
        }
    };
    auto flush_q = [&](uint32_t n) {
        const uint32_t l = lane_id(), at = wave_reserve(out_regions, region_q, n, l);
        if (l < n) {
            const double *e = stage_q.d + l;
            qout.o[0][at] = e[0 * SC];
            qout.o[1][at] = e[1 * SC];
            qout.o[2][at] = e[2 * SC];
            qout.d[0][at] = e[3 * SC];
            qout.d[1][at] = e[4 * SC];
            qout.d[2][at] = e[5 * SC];
            qout.thr[0][at] = e[6 * SC];
            qout.thr[1][at] = e[7 * SC];
            qout.thr[2][at] = e[8 * SC];
            qout.slot[at] = stage_q.u[0 * SC + l];
            qout.pixel[at] = stage_q.u[1 * SC + l];
        }
    };
    // the camera pass goes over the shard's slots; a ray queue is read through the map of its regions
    const bool primary = PQ == 0 ? q.primary != 0 : PQ == 1;
    QMap qm;
    if (PQ != 1 && !primary) qm = qmap_load(q_regions);
    const uint32_t count = primary ? *slots_ptr : qm.total();
    const uint32_t stride = gridDim.x * blockDim.x;
    const d3 bg = mk3(sc.background[0], sc.background[1], sc.background[2]);
    Guards gd;
    for (int k = 0; k < 5; ++k) gd.g[k] = 0u;
    gd.overflow = 0u;
    // The region a workgroup's append goes to first: the next one every time; in the camera pass the one that the
    // slots' position in the shard names (slots k * count / 8 .. fill region k: count <= 8 * rc, so none overflows, and the
    // queues keep the order of the image.  The workgroups running at one time then share a counter word, which the camera
    // pass's one atomic per 256 slots does not load; 4K bench frame, four in flight: 1 % better than going round)
    uint32_t round = blockIdx.x;
    const uint32_t per_region = (count + RT_Q_REGIONS - 1u) / RT_Q_REGIONS;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride, ++round) {
        uint32_t i = base + threadIdx.x;
        bool wants_mesh = false, alive = false;
        d3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), thr = mk3(1.0, 1.0, 1.0);
        uint32_t pixel = 0u, slot = i;
        HitRec h;
        h.t = -1.0;
        h.n = mk3(0, 0, 0);
        h.u = h.v = 0.0;
        h.obj = -1;
        bool have;
        if (primary) {
            have = i < count && primary_ray(sc, q, pp, i, o, d, pixel);
        } else {
            have = i < count;
            if (have) {
                i = qm.pos(base + (threadIdx.x & ~63u), i); // from here on: the entry's position
                if (!FUSED) slot = i;
                o = mk3(q.o[0][i], q.o[1][i], q.o[2][i]);
                d = mk3(q.d[0][i], q.d[1][i], q.d[2][i]);
                pixel = q.pixel[i];
            }
        }
        if (have) {
            if (FUSED) {
                if (primary) {
                    slot = q.row_major ? pixel : i;
                } else {
                    slot = q.slot[i];
                    thr = mk3(q.thr[0][i], q.thr[1][i], q.thr[2][i]);
                }
            }
            if constexpr (LL) {
                wants_mesh = intersect_analytic<CSG, MEDIUM, TORUS, true>(sc, o, d, pixel, pp, h, gd, ll);
            } else {
                wants_mesh = intersect_analytic<CSG, MEDIUM, TORUS>(sc, o, d, pixel, pp, h, gd);
            }
            if (primary_ids && (PQ != 2 || !FUSED)) primary_ids[FUSED ? slot : ray_slot(q, i)] = h.obj;
            if (!FUSED) {
                hit.t[i] = h.t;
                hit.n[0][i] = h.n.x;
                hit.n[1][i] = h.n.y;
                hit.n[2][i] = h.n.z;
                hit.uv[0][i] = h.u;
                hit.uv[1][i] = h.v;
                hit.obj[i] = h.obj;
            } else if (!wants_mesh) {
                d3 X;
                alive = shade_one(sc, o, d, thr, pixel, h, pp, bg, X);
                if (!alive) {
                    fb_add(fb + (size_t)slot * 3, thr.x * X.x, thr.y * X.y, thr.z * X.z);
                }
            }
        }
        uint32_t at_m = 0u, at_q = 0u;
        if constexpr (WAVE_APPEND) {
            // (from here on wants_mesh / alive: the lane stores its entry itself)
            const double vm[15] = {o.x, o.y, o.z, d.x, d.y, d.z, thr.x, thr.y, thr.z, h.t, h.n.x, h.n.y, h.n.z, h.u, h.v};
            const uint32_t um[3] = {slot, pixel, (uint32_t)h.obj};
            wants_mesh = wave_append(stage_m, wants_mesh, vm, um, mesh_regions, region_m, at_m, flush_m);
            const double vq[9] = {o.x, o.y, o.z, d.x, d.y, d.z, thr.x, thr.y, thr.z};
            const uint32_t uq[2] = {slot, pixel};
            alive = wave_append(stage_q, alive, vq, uq, out_regions, region_q, at_q, flush_q);
        } else {
            const uint32_t r0 = PQ == 1 ? base / per_region : round;
            block_append2(wants_mesh, alive, mesh_regions, out_regions, r0, append_lds, at_m, at_q);
        }
        if (wants_mesh) {
            mq.o[0][at_m] = o.x;
            mq.o[1][at_m] = o.y;
            mq.o[2][at_m] = o.z;
            mq.d[0][at_m] = d.x;
            mq.d[1][at_m] = d.y;
            mq.d[2][at_m] = d.z;
            if (FUSED) {
                mq.thr[0][at_m] = thr.x;
                mq.thr[1][at_m] = thr.y;
                mq.thr[2][at_m] = thr.z;
            }
            mq.slot[at_m] = slot;
            mq.pixel[at_m] = pixel;
            mqh.t[at_m] = h.t;
            mqh.n[0][at_m] = h.n.x;
            mqh.n[1][at_m] = h.n.y;
            mqh.n[2][at_m] = h.n.z;
            mqh.uv[0][at_m] = h.u;
            mqh.uv[1][at_m] = h.v;
            mqh.obj[at_m] = h.obj;
            if (mesh_res_tri) mesh_res_tri[at_m] = 0xFFFFFFFFu; // k_mesh_one's result for this record: none yet
        }
        if (FUSED && alive) {
            qout.o[0][at_q] = o.x;
            qout.o[1][at_q] = o.y;
            qout.o[2][at_q] = o.z;
            qout.d[0][at_q] = d.x;
            qout.d[1][at_q] = d.y;
            qout.d[2][at_q] = d.z;
            qout.thr[0][at_q] = thr.x;
            qout.thr[1][at_q] = thr.y;
            qout.thr[2][at_q] = thr.z;
            qout.slot[at_q] = slot;
            qout.pixel[at_q] = pixel;
        }
    }
    if constexpr (STAGE) {
        wave_lds_fence();
        if (stage_m.fill) flush_m(stage_m.fill);
        if (stage_q.fill) flush_q(stage_q.fill);
    }
    for (int k = 0; k < 5; ++k)
        if (gd.g[k]) atomicAdd(&guards_out[k], gd.g[k]);
    if (gd.overflow) atomicAdd(&guards_out[5], gd.overflow);
}