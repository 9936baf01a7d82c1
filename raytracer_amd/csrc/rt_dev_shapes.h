// rt_dev_shapes.h — per-ray shape evaluation on gfx950.
//
// Reference semantics (src/shapes/*.rs) with a different evaluation plan:
//   * a primitive is visited once, in its own space, through composed maps;
//   * interval end points are (world t, tag) pairs - the CSG interval algebra of
//     csg.rs only ever compares t (system.rs:202-206) and negates normals, which
//     a tag bit records;
//   * normal and uv are derived afterwards, for the nearest hit only.
// Situations in which the reference panics are counted and treated as a miss.
#pragma once
#include "rt_dev_math.h"

// end-point tag: bits 0-7 leaf (object-relative), 8-10 part, 11 normal negated,
// 12-31 to_world steps the reference skipped on the way up (DEV_PROG_SKIP, one bit per skip_ord)
#define TAG_SKIP_SHIFT 12
#define TAG_PART_SHIFT 8
#define TAG_NEG (1u << 11)
#define PART_CYL_SIDE 0u
#define PART_CYL_TOP 1u
#define PART_CYL_BOTTOM 2u

struct Guards {
    uint32_t g[5];
    uint32_t overflow;
};

struct LeafRay {
    d3 o, d;
};

RT_DEV LeafRay to_leaf(const DevLeaf &L, d3 ow, d3 dw)
{
    LeafRay r;
    if (L.identity) {
        r.o = ow;
        r.d = dw;
    } else {
        // Ray::transform (system.rs:145-152)
        r.o = xf_point(L.W, ow);
        r.d = normalize3(xf_dir(L.W, dw));
    }
    return r;
}

// Intersection::to_world's t (system.rs:190-195)
RT_DEV double leaf_t_to_world(const DevLeaf &L, d3 ow, const LeafRay &r, double t)
{
    if (L.identity) return t;
    d3 pw = xf_point(L.O, r.o + r.d * t);
    return signum(t) * len3(pw - ow);
}

// plane.rs:9-21
RT_DEV void dev_plane_uv(d3 n, d3 &u, d3 &v)
{
    u = cross3(n, mk3(1.0, 0.0, 0.0));
    if (len2(u) < 1e-6) u = cross3(n, mk3(0.0, 1.0, 0.0));
    if (len2(u) < 1e-6) u = cross3(n, mk3(0.0, 0.0, 1.0));
    u = normalize3(u);
    v = cross3(n, u);
}

RT_DEV double axis_of(d3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
RT_DEV d3 axis_vec(int a, double s) { return mk3(a == 0 ? s : 0.0, a == 1 ? s : 0.0, a == 2 ? s : 0.0); }

// ---------------------------------------------------------------------------
// Leaf-space end points of one primitive.  Returns the number of end points
// (0, 2 or 4; intervals are consecutive pairs), t[] in leaf space, sub[] =
// part | TAG_NEG.
RT_TORUS_FN int torus_ends(const double *p, d3 o, d3 d, double *t, uint32_t *sub, Guards &gd)
{
    // torus.rs:52-125
    double R = p[0], S = p[1];
    double T = 4.0 * R * R;
    double G = T * (d.x * d.x + d.y * d.y);
    double H = 2.0 * T * (o.x * d.x + o.y * d.y);
    double I = T * (o.x * o.x + o.y * o.y);
    double J = len2(d);
    double K = 2.0 * dot3(o, d);
    double Lc = len2(o) + R * R - S * S;
    double r[4];
    int n = dev_solve_quartic_f64(J * J, 2.0 * J * K, 2.0 * J * Lc + K * K - G, 2.0 * K * Lc - H, Lc * Lc - I, r);
    if (n > 1) {
        bool nan = false;
        for (int i = 0; i < 4; ++i)
            if (i < n && r[i] != r[i]) nan = true;
        if (nan) {
            gd.g[4]++;
            return 0;
        }
    }
    // stable insertion sort
    for (int i = 1; i < 4; ++i) {
        if (i >= n) break;
        double v = r[i];
        int j = i - 1;
        while (j >= 0 && r[j] > v) {
            r[j + 1] = r[j];
            --j;
        }
        r[j + 1] = v;
    }
    sub[0] = sub[1] = sub[2] = sub[3] = 0u;
    if (n == 0) return 0;
    if (n == 1) {
        t[0] = t[1] = r[0];
        return 2;
    }
    if (n == 2) {
        t[0] = r[0];
        t[1] = r[1];
        return 2;
    }
    if (n == 4) {
        t[0] = r[0];
        t[1] = r[1];
        t[2] = r[2];
        t[3] = r[3];
        return 4;
    }
    // three roots: pair them by the facing of the surface normal (torus.rs:92-110)
    int facing[3];
    for (int k = 0; k < 3; ++k) {
        d3 q = o + d * r[k];
        double a = 1.0 - (R / sqrt(q.x * q.x + q.y * q.y));
        d3 nn = normalize3(mk3(a * q.x, a * q.y, q.z));
        double s = signum(dot3(nn, d));
        facing[k] = (s != s) ? 0 : (int)s;
    }
    if (facing[0] == -1 && facing[1] == 1 && facing[2] == -1) {
        t[0] = r[0];
        t[1] = r[1];
        t[2] = r[2];
        t[3] = r[2];
        return 4;
    }
    if (facing[0] == 1 && facing[1] == -1 && facing[2] == 1) {
        t[0] = r[0];
        t[1] = r[0];
        t[2] = r[1];
        t[3] = r[2];
        return 4;
    }
    gd.g[1]++;
    return 0;
}

// TORUS = false leaves the quartic solver (and its registers) out of the kernel.
template <bool TORUS>
RT_DEV int leaf_ends(const DevLeaf &L, const LeafRay &ry, double *t, uint32_t *sub, Guards &gd)
{
    const d3 o = ry.o, d = ry.d;
    const double *p = L.p;
    switch (L.kind) {
    case DL_SPHERE: {
        // sphere.rs:41-56,69-90
        d3 l = o - mk3(p[0], p[1], p[2]);
        double a = dot3(d, d);
        double b = 2.0 * dot3(d, l);
        double c = dot3(l, l) - p[3];
        double discr = b * b - 4.0 * a * c;
        if (discr < 0.0) return 0;
        double t0, t1;
        if (discr == 0.0) {
            t0 = t1 = -0.5 * b / a;
        } else {
            double sq = sqrt(discr);
            double q = (b > 0.0) ? -0.5 * (b + sq) : -0.5 * (b - sq);
            t0 = q / a;
            t1 = c / q;
        }
        if (t0 > t1) {
            double tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        t[0] = t0;
        t[1] = t1;
        sub[0] = sub[1] = 0u;
        return 2;
    }
    case DL_PLANE: {
        // plane.rs:23-32,58-83
        d3 n = mk3(p[3], p[4], p[5]);
        double denom = dot3(d, n);
        if (!(fabs(denom) > 1e-6)) return 0;
        double tt = dot3(mk3(p[0], p[1], p[2]) - o, n) / denom;
        t[0] = t[1] = tt;
        sub[0] = sub[1] = 0u;
        return 2;
    }
    case DL_RECT_XY:
    case DL_RECT_XZ:
    case DL_RECT_ZY: {
        // plane.rs:111-244
        int axis = L.kind == DL_RECT_XY ? 2 : (L.kind == DL_RECT_XZ ? 1 : 0);
        double denom = axis_of(d, axis) * p[3];
        if (!(fabs(denom) > 1e-6)) return 0;
        double tt = ((axis_of(mk3(p[0], p[1], p[2]), axis) - axis_of(o, axis)) * p[3]) / denom;
        d3 q = o + d * tt;
        double c0 = L.kind == DL_RECT_ZY ? q.z : q.x;
        double c1 = L.kind == DL_RECT_XZ ? q.z : q.y;
        if (c0 < p[4] || c0 > p[5] || c1 < p[6] || c1 > p[7]) return 0;
        t[0] = t[1] = tt;
        sub[0] = sub[1] = 0u;
        return 2;
    }
    case DL_CUBE: {
        // cube.rs:78-106 over the six rectangles of cube.rs:21-37
        int nh = 0;
        double ht[2];
        uint32_t hf[2];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const int axis = f >> 1;
            const double *fq = p + f * 5;
            double da = axis_of(d, axis);
            if (!(fabs(da) > 1e-6)) continue;
            double tt = (fq[0] - axis_of(o, axis)) / da;
            d3 q = o + d * tt;
            double c0 = axis == 0 ? q.z : q.x;
            double c1 = axis == 1 ? q.z : q.y;
            if (c0 < fq[1] || c0 > fq[2] || c1 < fq[3] || c1 > fq[4]) continue;
            if (nh < 2) {
                ht[nh] = tt;
                hf[nh] = (uint32_t)f;
            }
            ++nh;
        }
        if (nh > 2) {
            gd.g[0]++;
            return 0;
        }
        if (nh == 0) return 0;
        if (nh == 1) {
            t[0] = t[1] = ht[0];
            sub[0] = sub[1] = hf[0] << TAG_PART_SHIFT;
            return 2;
        }
        int first = (ht[0] > ht[1]) ? 1 : 0;
        t[0] = ht[first];
        t[1] = ht[1 - first];
        sub[0] = hf[first] << TAG_PART_SHIFT;
        sub[1] = (hf[1 - first] << TAG_PART_SHIFT) | TAG_NEG;
        return 2;
    }
    case DL_CYLINDER: {
        // cylinder.rs:53-144
        double radius = p[0], height = p[1];
        double a = d.x * d.x + d.z * d.z;
        double b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
        double c = o.x * o.x + o.z * o.z - radius * radius;
        double discr = b * b - 4.0 * a * c;
        if (discr < 0.0) return 0;
        double sq = sqrt(discr);
        double t0 = (-b + sq) / (2.0 * a);
        double t1 = (-b - sq) / (2.0 * a);
        if (t0 > t1) {
            double tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        double y0 = o.y + t0 * d.y;
        double y1 = o.y + t1 * d.y;
        double max_y = height / 2.0, min_y = -max_y;
        bool cap_ok = fabs(d.y) > 1e-6; // the caps are Planes (plane.rs:23-32)
        double t_top = (max_y - o.y) / d.y;
        double t_bot = (min_y - o.y) / d.y;
        const uint32_t SIDE = PART_CYL_SIDE << TAG_PART_SHIFT, TOP = PART_CYL_TOP << TAG_PART_SHIFT,
                       BOT = PART_CYL_BOTTOM << TAG_PART_SHIFT;
        if ((y0 < min_y && y1 < min_y) || (y0 > max_y && y1 > max_y)) return 0;
        bool back_side = (y1 > min_y && y1 < max_y);
        if (y0 > max_y) {
            if (!cap_ok) {
                gd.g[3]++;
                return 0;
            }
            t[0] = t_top;
            sub[0] = TOP;
            if (back_side) {
                t[1] = t1;
                sub[1] = SIDE;
            } else {
                if (!(y1 < min_y)) {
                    gd.g[3]++;
                    return 0;
                }
                t[1] = t_bot;
                sub[1] = BOT;
            }
            return 2;
        } else if (y0 < min_y) {
            if (!cap_ok) {
                gd.g[3]++;
                return 0;
            }
            t[0] = t_bot;
            sub[0] = BOT;
            if (back_side) {
                t[1] = t1;
                sub[1] = SIDE;
            } else {
                if (!(y1 > max_y)) {
                    gd.g[3]++;
                    return 0;
                }
                t[1] = t_top;
                sub[1] = TOP;
            }
            return 2;
        } else {
            t[0] = t0;
            sub[0] = SIDE;
            if (back_side) {
                t[1] = t1;
                sub[1] = SIDE;
            } else if (y1 < min_y) {
                if (!cap_ok) {
                    gd.g[3]++;
                    return 0;
                }
                t[1] = t_bot;
                sub[1] = BOT;
            } else {
                if (!(y1 > max_y) || !cap_ok) {
                    gd.g[3]++;
                    return 0;
                }
                t[1] = t_top;
                sub[1] = TOP;
            }
            return 2;
        }
    }
    case DL_TORUS:
        if (TORUS) return torus_ends(p, o, d, t, sub, gd);
        return 0;
    default: return 0;
    }
}

// ---------------------------------------------------------------------------
// Normal (leaf space, before the tag's negation) and uv at a leaf-space point.
RT_DEV void leaf_surface(const DevLeaf &L, const LeafRay &ry, d3 q, uint32_t tag, bool need_uv, d3 &n, double &u, double &v)
{
    const double *p = L.p;
    const uint32_t part = (tag >> TAG_PART_SHIFT) & 7u;
    u = 0.0;
    v = 0.0;
    switch (L.kind) {
    case DL_SPHERE: {
        // sphere.rs:27-38
        n = normalize3(q - mk3(p[0], p[1], p[2]));
        if (need_uv) {
            u = (1.0 - atan2(n.z, n.x) / RT_PI) * 0.5;
            v = acos(n.y) / RT_PI;
        }
        break;
    }
    case DL_PLANE: {
        // plane.rs:62-79; the uv basis of either side is a constant of the plane, worked out by the flattener
        // (plane_constants: dev_plane_uv's own operations on N and on N * -1.0)
        const d3 N = mk3(p[3], p[4], p[5]);
        const bool back = dot3(ry.d, N) > 0.0;
        n = back ? N * -1.0 : N;
        if (need_uv) {
            const double *b = p + (back ? 15 : 9);
            d3 op = q - mk3(p[0], p[1], p[2]);
            u = dot3(mk3(b[0], b[1], b[2]), op);
            v = dot3(mk3(b[3], b[4], b[5]), op);
        }
        break;
    }
    case DL_RECT_XY:
    case DL_RECT_XZ:
    case DL_RECT_ZY: {
        // plane.rs:62-79
        d3 N = axis_vec(L.kind == DL_RECT_XY ? 2 : (L.kind == DL_RECT_XZ ? 1 : 0), p[3]);
        n = (dot3(ry.d, N) > 0.0) ? N * -1.0 : N;
        if (need_uv) {
            d3 bu, bv;
            dev_plane_uv(n, bu, bv);
            d3 op = q - mk3(p[0], p[1], p[2]);
            u = dot3(bu, op);
            v = dot3(bv, op);
        }
        break;
    }
    case DL_CUBE: {
        const int f = (int)part, axis = f >> 1;
        d3 N = axis_vec(axis, (f & 1) ? 1.0 : -1.0); // min faces carry the reversed normal (cube.rs:29-34)
        n = (dot3(ry.d, N) > 0.0) ? N * -1.0 : N;
        if (need_uv) {
            d3 bu, bv;
            dev_plane_uv(n, bu, bv);
            d3 fo = mk3(p[30], p[31], p[32]);
            if (axis == 0) fo.x = p[f * 5];
            if (axis == 1) fo.y = p[f * 5];
            if (axis == 2) fo.z = p[f * 5];
            d3 op = q - fo;
            u = dot3(bu, op);
            v = dot3(bv, op);
        }
        break;
    }
    case DL_CYLINDER: {
        double height = p[1], max_y = height / 2.0;
        if (part == PART_CYL_SIDE) {
            // cylinder.rs:27-39
            n = normalize3(mk3(q.x, 0.0, q.z));
            if (need_uv) {
                u = (1.0 - atan2(n.z, n.x) / RT_PI) * 0.5;
                v = 1.0 - (q.y + max_y) / height;
            }
        } else {
            d3 N = mk3(0.0, part == PART_CYL_TOP ? 1.0 : -1.0, 0.0);
            n = (dot3(ry.d, N) > 0.0) ? N * -1.0 : N;
            if (need_uv) {
                d3 bu, bv;
                dev_plane_uv(n, bu, bv);
                d3 op = q - mk3(0.0, part == PART_CYL_TOP ? max_y : -max_y, 0.0);
                u = dot3(bu, op);
                v = dot3(bv, op);
            }
        }
        break;
    }
    case DL_TORUS: {
        // torus.rs:28-39
        double a = 1.0 - (p[0] / sqrt(q.x * q.x + q.y * q.y));
        n = normalize3(mk3(a * q.x, a * q.y, q.z));
        break;
    }
    default: n = mk3(0.0, 0.0, 0.0); break;
    }
}

// ---------------------------------------------------------------------------
// Interval lists: end points (t, tag), 2 per interval.  In the general kernels the lists of a ray live in scratch
// memory (they are indexed by run-time values), and a launch's waves together hold waves x 64 x (lists x bytes) of it:
// with 12 intervals per list and 9 lists that is 2.9 KB per lane and 600 MB over the chip - far beyond the caches, so
// every list access goes to HBM (room.scene 4K: 6.0 GB of HBM traffic per k_intersect launch against 1.1 GB
// algorithmic before csg_eval stored each list once, 4.0 GB after).  A smaller capacity changes nothing there: only the
// entries a ray stores travel (kernels with 4 intervals x 5 lists measured the same, round 3).
template <int MAXI>
struct IvListT {
    static constexpr int CAP = MAXI;
    double t[2 * MAXI];
    uint32_t tag[2 * MAXI];
    int n; // intervals
    RT_DEV static constexpr int cap() { return MAXI; }
};
typedef IvListT<DEV_MAX_INTERVALS> IvList;

// The same lists in LDS, for scenes whose trees are small enough that every lane of the workgroups a CU holds gets
// room for all of them (the flattener's bounds: `cap` intervals per list, `n_lists` lists at once; rt_kernels.hip
// decides).  Scratch memory has the capacity of HBM and, with thousands of waves in flight, its latency; LDS is a
// hundred cycles away.  Entry e of list l of thread x sits at t[(l * 2 cap + e) * STRIDE + x] (STRIDE = threads per
// workgroup): the lanes of a wave touch neighbouring words whatever entry each of them is at.  The interval counts
// stay in a register, four bits per list.  An LdsList is a view (three pointers and a shift) with the member names of
// IvListT, so the interval algebra below is written once.
template <int STRIDE>
struct LdsList {
    struct TArr {
        double *b;
        RT_DEV double &operator[](int i) const { return b[i * STRIDE]; }
    };
    struct GArr {
        uint32_t *b;
        RT_DEV uint32_t &operator[](int i) const { return b[i * STRIDE]; }
    };
    struct NRef {
        uint32_t *w;
        uint32_t sh;
        RT_DEV operator int() const { return (int)((*w >> sh) & 15u); }
        RT_DEV NRef &operator=(int v)
        {
            *w = (*w & ~(15u << sh)) | ((uint32_t)v << sh);
            return *this;
        }
        RT_DEV NRef &operator=(const NRef &o) { return *this = (int)o; }
        RT_DEV int operator++(int)
        {
            const int v = *this;
            *this = v + 1;
            return v;
        }
    };
    TArr t;
    GArr tag;
    NRef n;
    int cap_;
    RT_DEV int cap() const { return cap_; }
};

// all lists of one thread: stack[l] is list l
template <int STRIDE>
struct LdsLists {
    double *t;       // this thread's column of the end points
    uint32_t *tag;   // ... and of the tags
    int cap_;        // intervals per list (at most 15)
    int n_lists;     // at most 8
    uint32_t counts; // intervals in list l: bits 4l .. 4l + 3
    RT_DEV LdsList<STRIDE> operator[](int l)
    {
        LdsList<STRIDE> v;
        v.t.b = t + l * 2 * cap_ * STRIDE;
        v.tag.b = tag + l * 2 * cap_ * STRIDE;
        v.n.w = &counts;
        v.n.sh = 4u * (uint32_t)l;
        v.cap_ = cap_;
        return v;
    }
};

template <class LST>
RT_DEV void iv_push(LST &l, double t0, uint32_t g0, double t1, uint32_t g1, Guards &gd)
{
    if (l.n >= l.cap()) {
        gd.overflow++;
        return;
    }
    l.t[2 * l.n] = t0;
    l.tag[2 * l.n] = g0;
    l.t[2 * l.n + 1] = t1;
    l.tag[2 * l.n + 1] = g1;
    l.n++;
}

template <class LST>
RT_DEV void iv_copy(LST &dst, const LST &src)
{
    dst.n = src.n;
    for (int i = 0; i < 2 * src.cap(); ++i) {
        if (i >= 2 * src.n) break;
        dst.t[i] = src.t[i];
        dst.tag[i] = src.tag[i];
    }
}

// `intervals.sort_by(partial_cmp)`: stable, lexicographic on (entry.t, exit.t).
// Returns false where the reference would panic on a NaN.
template <class LST>
RT_DEV bool iv_sort(LST &l)
{
    if (l.n > 1) {
        for (int i = 0; i < 2 * l.n; ++i)
            if (l.t[i] != l.t[i]) return false;
    }
    for (int i = 1; i < l.n; ++i) {
        double a0 = l.t[2 * i], a1 = l.t[2 * i + 1];
        uint32_t g0 = l.tag[2 * i], g1 = l.tag[2 * i + 1];
        int j = i - 1;
        while (j >= 0 && (l.t[2 * j] > a0 || (l.t[2 * j] == a0 && l.t[2 * j + 1] > a1))) {
            l.t[2 * j + 2] = l.t[2 * j];
            l.t[2 * j + 3] = l.t[2 * j + 1];
            l.tag[2 * j + 2] = l.tag[2 * j];
            l.tag[2 * j + 3] = l.tag[2 * j + 1];
            --j;
        }
        l.t[2 * j + 2] = a0;
        l.t[2 * j + 3] = a1;
        l.tag[2 * j + 2] = g0;
        l.tag[2 * j + 3] = g1;
    }
    return true;
}

// csg.rs:28-98 / 133-176 / 211-280 on (t, tag) end points.  Returns false on a
// reference panic.
template <class LST>
RT_CSG_FN bool csg_apply(int op, const LST &A, const LST &B, LST &out, Guards &gd)
{
    out.n = 0;
    // The number of intervals written lives in a register (`n_out`) and reaches out.n once, at the end.
    int n_out = 0;
#define PUSH(T0, G0, T1, G1)                                                                                           \
    do {                                                                                                               \
        if (n_out >= out.cap()) {                                                                                      \
            gd.overflow++;                                                                                             \
        } else {                                                                                                       \
            out.t[2 * n_out] = (T0);                                                                                   \
            out.tag[2 * n_out] = (G0);                                                                                 \
            out.t[2 * n_out + 1] = (T1);                                                                               \
            out.tag[2 * n_out + 1] = (G1);                                                                             \
            ++n_out;                                                                                                   \
        }                                                                                                              \
    } while (0)
    if (op == DEV_OP_UNION) {
        if (A.n == 0) {
            iv_copy(out, B);
            return true;
        }
        if (B.n == 0) {
            iv_copy(out, A);
            return true;
        }
    } else if (op == DEV_OP_INTERSECTION) {
        if (A.n == 0 || B.n == 0) return true;
    } else {
        if (A.n == 0) return true;
        if (B.n == 0) {
            iv_copy(out, A);
            return true;
        }
    }
    // ia / ib: index of the interval currently held in (as, ae) / (bs, be); "Some" while the index is inside
    // its list.  (Integer indices rather than boolean flags carried round the loop.)
    int ia = 0, ib = 0;
    double as = A.t[0], ae = A.t[1], bs = B.t[0], be = B.t[1];
    uint32_t asg = A.tag[0], aeg = A.tag[1], bsg = B.tag[0], beg = B.tag[1];
#define NEXT_A()                                                                                                       \
    do {                                                                                                               \
        ++ia;                                                                                                          \
        if (ia < A.n) {                                                                                                \
            as = A.t[2 * ia];                                                                                          \
            ae = A.t[2 * ia + 1];                                                                                      \
            asg = A.tag[2 * ia];                                                                                       \
            aeg = A.tag[2 * ia + 1];                                                                                   \
        }                                                                                                              \
    } while (0)
#define NEXT_B()                                                                                                       \
    do {                                                                                                               \
        ++ib;                                                                                                          \
        if (ib < B.n) {                                                                                                \
            bs = B.t[2 * ib];                                                                                          \
            be = B.t[2 * ib + 1];                                                                                      \
            bsg = B.tag[2 * ib];                                                                                       \
            beg = B.tag[2 * ib + 1];                                                                                   \
        }                                                                                                              \
    } while (0)
    int guard_iter = 0;
    while (ia < A.n && ib < B.n) {
        if (++guard_iter > 8 * out.cap()) break; // cannot happen; keeps the loop provably finite
        if (op == DEV_OP_UNION) {
            if (ae < bs) {
                PUSH(as, asg, ae, aeg);
                NEXT_A();
            } else if (be < as) {
                PUSH(bs, bsg, be, beg);
                NEXT_B();
            } else {
                double ns = as, ne = ae;
                uint32_t nsg = asg, neg = aeg;
                if (bs < ns) {
                    ns = bs;
                    nsg = bsg;
                }
                if (be > ne) {
                    ne = be;
                    neg = beg;
                    NEXT_A();
                    while (ia < A.n) {
                        if (as > ne) break;
                        ne = ae;
                        neg = aeg;
                        NEXT_A();
                    }
                }
                PUSH(ns, nsg, ne, neg);
                NEXT_B();
            }
        } else if (op == DEV_OP_INTERSECTION) {
            if (ae < bs) {
                NEXT_A();
            } else if (be < as) {
                NEXT_B();
            } else {
                bool s_from_a = as > bs, e_from_a = ae < be;
                PUSH(s_from_a ? as : bs, s_from_a ? asg : bsg, e_from_a ? ae : be, e_from_a ? aeg : beg);
                if (be >= ae)
                    NEXT_A();
                else
                    NEXT_B();
            }
        } else {
            if (ae < bs) {
                PUSH(as, asg, ae, aeg);
                NEXT_A();
            } else if (be < as) {
                NEXT_B();
            } else if (bs <= as && be >= ae) {
                NEXT_A();
            } else if (bs <= as && be < ae) {
                as = be;
                asg = beg ^ TAG_NEG;
                NEXT_B();
            } else if (bs < ae && be < ae) {
                PUSH(as, asg, bs, bsg ^ TAG_NEG);
                as = be;
                asg = beg ^ TAG_NEG;
                NEXT_B();
            } else if (bs < ae && be >= ae) {
                PUSH(as, asg, bs, bsg ^ TAG_NEG);
                NEXT_A();
            } else {
                gd.g[2]++;
                return false;
            }
        }
    }
    if (op == DEV_OP_UNION) {
        // the interval in hand first (a's if there is one, else b's), then the untouched rest of both lists
        if (ia < A.n)
            PUSH(as, asg, ae, aeg);
        else if (ib < B.n)
            PUSH(bs, bsg, be, beg);
        for (int k = ia + 1; k < A.n; ++k) PUSH(A.t[2 * k], A.tag[2 * k], A.t[2 * k + 1], A.tag[2 * k + 1]);
        for (int k = ib + 1; k < B.n; ++k) PUSH(B.t[2 * k], B.tag[2 * k], B.t[2 * k + 1], B.tag[2 * k + 1]);
    } else if (op == DEV_OP_DIFFERENCE) {
        if (ia < A.n) PUSH(as, asg, ae, aeg);
        for (int k = ia + 1; k < A.n; ++k) PUSH(A.t[2 * k], A.tag[2 * k], A.t[2 * k + 1], A.tag[2 * k + 1]);
    }
#undef NEXT_A
#undef NEXT_B
#undef PUSH
    out.n = n_out;
    if (!iv_sort(out)) {
        gd.g[4]++;
        return false;
    }
    return true;
}

// Intervals of one leaf as world-space end points
template <bool TORUS, class LST>
RT_DEV void leaf_world_intervals(const DevLeaf &L, uint32_t rel, d3 ow, d3 dw, LST &out, Guards &gd)
{
    LeafRay ry = to_leaf(L, ow, dw);
    double t[4];
    uint32_t sub[4];
    int n = leaf_ends<TORUS>(L, ry, t, sub, gd);
    if (n > 2 * out.cap()) { // cannot happen: the flattener's bound counts two intervals for a torus, one for the others
        gd.overflow++;
        n = 2 * out.cap();
    }
    out.n = n >> 1;
    for (int i = 0; i < 4; ++i) {
        if (i >= n) break;
        out.t[i] = leaf_t_to_world(L, ow, ry, t[i]);
        out.tag[i] = sub[i] | rel;
    }
}

// DEV_PROG_SKIP: length of the ray direction in the node's space over its length in the parent's space
RT_DEV double skip_scale(const DevLeaf &Q, d3 dw)
{
    const double *m = Q.p;
    d3 dp = mk3(dw.x * m[0] + dw.y * m[3] + dw.z * m[6], dw.x * m[1] + dw.y * m[4] + dw.z * m[7],
                dw.x * m[2] + dw.y * m[5] + dw.z * m[8]);
    return len3(xf_dir(Q.W, dw)) / len3(dp);
}

// Evaluate an object's postfix program on the caller's lists `stack` (STK + 1 of them); returns the index of the
// list that holds the result, or -1 (the reference would have panicked, or the program is malformed).
// The lists live in scratch memory, and with thousands of waves in flight that memory is far larger than the caches:
// every list entry stored goes to HBM (room.scene 4K: 4.0 GB written per k_intersect launch, 570 bytes per ray, all
// of it list stores).  So nothing is stored twice: an operation writes its result into the one list that is free
// (never an operand) and the stack entries are renamed - `perm` holds, four bits per stack position, which list a
// position uses - instead of copied; the result is read where it lies.
// `stack`: an array of IvListT (scratch memory) or an LdsLists; `stk` + 1 lists.
template <bool TORUS, class STACK>
RT_CSG_FN int csg_eval(const DevScene &sc, const DevObject &ob, d3 ow, d3 dw, STACK &stack, const int stk, Guards &gd)
{
    uint64_t perm = 0xFEDCBA9876543210ull; // position k uses list (perm >> 4k) & 15 (stk + 1 <= 16)
    int free_list = stk;
    int sp = 0;
#define CSG_LIST(POS) ((int)((perm >> (4 * (POS))) & 15ull))
    for (int pc = 0; pc < ob.prog_len; ++pc) {
        int word = sc.prog[ob.prog0 + pc];
        int op = (word >> 24) & 0x7F;
        if (op == DEV_OP_LEAF) {
            if (sp >= stk) {
                gd.overflow++;
                return -1;
            }
            uint32_t rel = (uint32_t)(word & 0xFFFFFF);
            auto &&dst = stack[CSG_LIST(sp)];
            leaf_world_intervals<TORUS>(sc.leaves[ob.leaf0 + rel], rel, ow, dw, dst, gd);
            ++sp;
        } else {
            if (sp < 2) return -1;
            const int la = CSG_LIST(sp - 2), lb = CSG_LIST(sp - 1);
            auto &&A = stack[la];
            auto &&B = stack[lb];
            const int na = A.n, nb = B.n;
            const bool one_missed = na == 0 || nb == 0;
            // an operand missed: the result is one of the operands as it stands (csg_apply's own first lines, csg.rs:
            // 33-38,137-139,216-221) - taken by renaming, not by copying
            int lr = free_list;
            if (op == DEV_OP_UNION)
                lr = na == 0 ? lb : (nb == 0 ? la : free_list);
            else if (op == DEV_OP_INTERSECTION)
                lr = na == 0 ? la : (nb == 0 ? lb : free_list);
            else
                lr = one_missed ? la : free_list;
            if (lr == free_list) {
                auto &&R = stack[free_list];
                if (!csg_apply(op, A, B, R, gd)) return -1;
            }
            auto &&out = stack[lr];
            if ((word & DEV_PROG_SKIP) && one_missed && out.n > 0) {
                // csg.rs:33-38,216-221: the surviving operand's end points keep this node's space
                const DevLeaf &Q = sc.leaves[ob.leaf0 + (word & 0xFFFF)];
                const double f = skip_scale(Q, dw);
                const uint32_t bit = 1u << (TAG_SKIP_SHIFT + Q.skip_ord - 1);
                for (int i = 0; i < 2 * out.cap(); ++i) {
                    if (i >= 2 * out.n) break;
                    out.t[i] *= f;
                    out.tag[i] |= bit;
                }
            }
            // position sp - 2 now uses the list that holds the result; the positions stay a permutation of the lists
            const int sh = 4 * (sp - 2);
            if (lr == free_list) {
                perm = (perm & ~(15ull << sh)) | ((uint64_t)free_list << sh);
                free_list = la;
            } else if (lr == lb) {
                perm = (perm & ~(255ull << sh)) | ((uint64_t)lb << sh) | ((uint64_t)la << (sh + 4));
            }
            --sp;
        }
    }
#undef CSG_LIST
    if (sp != 1) return -1;
    return (int)(perm & 15ull);
}

// ---------------------------------------------------------------------------
// Meshes.  mesh.rs:66-113 for one triangle record (v0, v0v1, v0v2).
RT_DEV bool tri_hit_v(d3 v0, d3 e1, d3 e2, d3 o, d3 d, double &t, double &u, double &v)
{
    d3 pvec = cross3(d, e2);
    double det = dot3(e1, pvec);
    if (det < 2.220446049250313e-16) return false; // f64::EPSILON: back faces and edge-on triangles
    // Most triangles of a leaf fail one of the tests below by a wide margin.  The numerators
    // decide those cases without the f64 division (det > 0, so signs and the comparison with
    // det carry over); only candidates within 1e-12 of a boundary reach the reference's
    // exact arithmetic, which then decides as the reference does.
    d3 tvec = o - v0;
    double un = dot3(tvec, pvec);
    const double hi = det * (1.0 + 1e-12);
    if (un < -1e-290 || un > hi) return false;
    d3 qvec = cross3(tvec, e1);
    double vn = dot3(d, qvec);
    if (vn < -1e-290 || un + vn > hi) return false;
    double tn = dot3(e2, qvec);
    if (tn < -1e-290) return false;
    double inv_det = 1.0 / det;
    u = un * inv_det;
    if (u < 0.0 || u > 1.0) return false;
    v = vn * inv_det;
    if (v < 0.0 || (u + v) > 1.0) return false;
    t = tn * inv_det;
    if (t < 0.0) return false;
    return true;
}

RT_DEV bool tri_hit(const double *__restrict__ tr, d3 o, d3 d, double &t, double &u, double &v)
{
    return tri_hit_v(mk3(tr[0], tr[1], tr[2]), mk3(tr[3], tr[4], tr[5]), mk3(tr[6], tr[7], tr[8]), o, d, t, u, v);
}

// bounding_box.rs:13-44 against the Mesh::new box (which contains the origin)
RT_DEV bool box_hit(const double *qmin, const double *qmax, d3 o, d3 inv);
RT_DEV bool mesh_box_hit(const DevMesh &m, d3 o, d3 inv) { return box_hit(m.qmin, m.qmax, o, inv); }
RT_DEV bool box_hit(const double *qmin, const double *qmax, d3 o, d3 inv)
{
    int sx = inv.x < 0.0, sy = inv.y < 0.0, sz = inv.z < 0.0;
    double tmin = ((sx ? qmax[0] : qmin[0]) - o.x) * inv.x;
    double tmax = ((sx ? qmin[0] : qmax[0]) - o.x) * inv.x;
    double tymin = ((sy ? qmax[1] : qmin[1]) - o.y) * inv.y;
    double tymax = ((sy ? qmin[1] : qmax[1]) - o.y) * inv.y;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    double tzmin = ((sz ? qmax[2] : qmin[2]) - o.z) * inv.z;
    double tzmax = ((sz ? qmin[2] : qmax[2]) - o.z) * inv.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

// Nearest triangle (smallest t, lowest index on ties) = element 0 of the
// reference's stable sort over all triangle hits (mesh.rs:54-64).  The walk
// through the mesh's grid lives in k_mesh (rt_kernels.hip).
#ifndef RT_MESH_LANES
#define RT_MESH_LANES 256 // = threads per k_mesh workgroup (128 measured equal at the same number of waves, 512: k_mesh +20 %)
#endif

// clip the ray against the tight (padded) bounds of the mesh
RT_DEV bool box_clip(const double *bmin, const double *bmax, d3 o, d3 inv, double &t0, double &t1)
{
    t0 = 0.0;
    t1 = 1.0e300;
    double lo, hi;
    lo = (bmin[0] - o.x) * inv.x;
    hi = (bmax[0] - o.x) * inv.x;
    if (lo > hi) { double s = lo; lo = hi; hi = s; }
    if (lo == lo && lo > t0) t0 = lo;
    if (hi == hi && hi < t1) t1 = hi;
    lo = (bmin[1] - o.y) * inv.y;
    hi = (bmax[1] - o.y) * inv.y;
    if (lo > hi) { double s = lo; lo = hi; hi = s; }
    if (lo == lo && lo > t0) t0 = lo;
    if (hi == hi && hi < t1) t1 = hi;
    lo = (bmin[2] - o.z) * inv.z;
    hi = (bmax[2] - o.z) * inv.z;
    if (lo > hi) { double s = lo; lo = hi; hi = s; }
    if (lo == lo && lo > t0) t0 = lo;
    if (hi == hi && hi < t1) t1 = hi;
    return !(t0 > t1);
}

RT_DEV bool mesh_clip(const DevMesh &m, d3 o, d3 inv, double &t0, double &t1) { return box_clip(m.bmin, m.bmax, o, inv, t0, t1); }
