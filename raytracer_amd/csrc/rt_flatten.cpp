// rt_flatten.cpp — rt_scene_desc -> device layout (rt_device.h): composes the
// per-node transforms along every root-to-primitive path, turns CSG trees into
// postfix programs, pre-computes the rectangle bounds the reference computes in
// its constructors, and builds one uniform grid per mesh.
//
// The grid is new: the reference tests every triangle of a mesh and sorts
// the hits (mesh.rs:54-64).  The grid only decides which triangles a ray
// visits; the triangle test and the "nearest t, lowest index on ties" choice
// are the reference's, so the result equals the brute-force one.
#include "rt_flatten.h"
#include "host_math.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rt_host {

namespace {

struct Err {
    int status;
    std::string msg;
};

void store_xform(const Mat4 &a, DevXform &x)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c) x.m[r * 3 + c] = a.m[r][c];
}

bool is_affine(const Mat4 &a) { return a.m[0][3] == 0.0 && a.m[1][3] == 0.0 && a.m[2][3] == 0.0 && a.m[3][3] == 1.0; }

bool is_identity(const Mat4 &a)
{
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            if (a.m[r][c] != ((r == c) ? 1.0 : 0.0)) return false;
    return true;
}

bool linear_part_is_identity(const Mat4 &a)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            if (a.m[r][c] != ((r == c) ? 1.0 : 0.0)) return false;
    return true;
}

// ---------------------------------------------------------------- mesh grids
struct Box {
    double lo[3], hi[3];
};

// Does triangle `v` (9 doubles: three vertices) touch `cell` grown by `pad`?
// Sutherland-Hodgman clipping of the triangle against the six planes; what is
// left is empty exactly when the triangle misses the grown box.
bool tri_touches_box(const double *v, const Box &cell, double pad)
{
    double poly[16][3], tmp[16][3];
    int n = 3;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) poly[i][k] = v[i * 3 + k];
    for (int axis = 0; axis < 3 && n > 0; ++axis) {
        for (int side = 0; side < 2 && n > 0; ++side) {
            const double plane = side == 0 ? cell.lo[axis] - pad : cell.hi[axis] + pad;
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const double *a = poly[i], *b = poly[(i + 1) % n];
                bool ina = side == 0 ? a[axis] >= plane : a[axis] <= plane;
                bool inb = side == 0 ? b[axis] >= plane : b[axis] <= plane;
                if (ina) {
                    for (int k = 0; k < 3; ++k) tmp[m][k] = a[k];
                    ++m;
                }
                if (ina != inb) {
                    double t = (plane - a[axis]) / (b[axis] - a[axis]);
                    for (int k = 0; k < 3; ++k) tmp[m][k] = a[k] + t * (b[k] - a[k]);
                    tmp[m][axis] = plane;
                    ++m;
                }
            }
            n = m;
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 3; ++k) poly[i][k] = tmp[i][k];
        }
    }
    return n > 0;
}

// ---------------------------------------------------------------- kd-trees (the alternative to the grids)
// One reference to a triangle inside a cell: the triangle and the bounds of the
// part of it that lies in the cell ("split clipping": without it the diagonal,
// chunky triangles of LinkedTorus.obj end up in ten leaves each and no empty
// space is ever cut away).
struct KdRef {
    uint32_t tri;
    Box box;
};

struct KdBuilder {
    const std::vector<double> &verts; // 9 per triangle: v0, v1, v2 (absolute)
    std::vector<DevKdNode> &nodes;
    std::vector<uint32_t> &index;
    uint32_t index_base; // start of this mesh's index list
    int max_depth;
    double pad;
    double kIsect = 20.0, kTrav = 15.0, kEmptyBonus = 0.5; // chosen by measurement on MI355X
    size_t leaf_max = 1;
    double min_empty = 0.1;

    // bounds of (triangle clipped to `cell` grown by pad); false if nothing is left
    bool clip(uint32_t tri, const Box &cell, Box &out) const
    {
        double poly[16][3], tmp[16][3];
        int n = 3;
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) poly[v][k] = verts[(size_t)tri * 9 + v * 3 + k];
        for (int axis = 0; axis < 3 && n > 0; ++axis) {
            for (int side = 0; side < 2 && n > 0; ++side) {
                const double plane = side == 0 ? cell.lo[axis] - pad : cell.hi[axis] + pad;
                int m = 0;
                for (int i = 0; i < n; ++i) {
                    const double *a = poly[i], *b = poly[(i + 1) % n];
                    bool ina = side == 0 ? a[axis] >= plane : a[axis] <= plane;
                    bool inb = side == 0 ? b[axis] >= plane : b[axis] <= plane;
                    if (ina) {
                        for (int k = 0; k < 3; ++k) tmp[m][k] = a[k];
                        ++m;
                    }
                    if (ina != inb) {
                        double t = (plane - a[axis]) / (b[axis] - a[axis]);
                        for (int k = 0; k < 3; ++k) tmp[m][k] = a[k] + t * (b[k] - a[k]);
                        tmp[m][axis] = plane;
                        ++m;
                    }
                }
                n = m;
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k < 3; ++k) poly[i][k] = tmp[i][k];
            }
        }
        if (n == 0) return false;
        for (int k = 0; k < 3; ++k) {
            out.lo[k] = poly[0][k];
            out.hi[k] = poly[0][k];
        }
        for (int i = 1; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                out.lo[k] = std::fmin(out.lo[k], poly[i][k]);
                out.hi[k] = std::fmax(out.hi[k], poly[i][k]);
            }
        for (int k = 0; k < 3; ++k) {
            out.lo[k] -= pad;
            out.hi[k] += pad;
        }
        return true;
    }

    void make_leaf(const std::vector<KdRef> &prims)
    {
        DevKdNode n;
        n.split = 0.0;
        n.a = (uint32_t)(((index.size() - index_base) << 2) | 3u);
        n.b = (uint32_t)prims.size();
        std::vector<uint32_t> sorted;
        for (const KdRef &r : prims) sorted.push_back(r.tri);
        std::sort(sorted.begin(), sorted.end());
        index.insert(index.end(), sorted.begin(), sorted.end());
        nodes.push_back(n);
    }

    void build(const std::vector<KdRef> &prims, const Box &box, int depth, int bad)
    {
        if (prims.size() <= leaf_max || depth >= max_depth) {
            make_leaf(prims);
            return;
        }
        double d[3] = {box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]};
        double total_sa = 2.0 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
        double inv_sa = total_sa > 0.0 ? 1.0 / total_sa : 0.0;
        double old_cost = kIsect * (double)prims.size();
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1;
        double best_split = 0.0;

        struct Edge {
            double t;
            bool start;
        };
        std::vector<Edge> edges(prims.size() * 2);
        for (int axis = 0; axis < 3; ++axis) {
            for (size_t i = 0; i < prims.size(); ++i) {
                edges[2 * i] = Edge{std::fmax(prims[i].box.lo[axis], box.lo[axis]), true};
                edges[2 * i + 1] = Edge{std::fmin(prims[i].box.hi[axis], box.hi[axis]), false};
            }
            std::sort(edges.begin(), edges.end(), [](const Edge &a, const Edge &b) {
                if (a.t == b.t) return (int)a.start > (int)b.start;
                return a.t < b.t;
            });
            size_t n_below = 0, n_above = prims.size();
            int o0 = (axis + 1) % 3, o1 = (axis + 2) % 3;
            for (size_t i = 0; i < edges.size(); ++i) {
                if (!edges[i].start) --n_above;
                double t = edges[i].t;
                if (t > box.lo[axis] && t < box.hi[axis]) {
                    double below_sa = 2.0 * (d[o0] * d[o1] + (t - box.lo[axis]) * (d[o0] + d[o1]));
                    double above_sa = 2.0 * (d[o0] * d[o1] + (box.hi[axis] - t) * (d[o0] + d[o1]));
                    // the bonus for cutting away empty space applies only to a worthwhile slab:
                    // sliver cuts would otherwise win every time and use up the depth budget
                    double frac = n_below == 0 ? (t - box.lo[axis]) / d[axis] : (box.hi[axis] - t) / d[axis];
                    double eb = ((n_above == 0 || n_below == 0) && frac >= min_empty) ? kEmptyBonus : 0.0;
                    double cost = kTrav + kIsect * (1.0 - eb) *
                                              (below_sa * inv_sa * (double)n_below + above_sa * inv_sa * (double)n_above);
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_split = t;
                    }
                }
                if (edges[i].start) ++n_below;
            }
        }
        if (best_cost > old_cost) ++bad;
        if (best_axis < 0 || (best_cost > 4.0 * old_cost && prims.size() < 16) || bad >= 3) {
            make_leaf(prims);
            return;
        }
        Box bl = box, bh = box;
        bl.hi[best_axis] = best_split;
        bh.lo[best_axis] = best_split;
        std::vector<KdRef> below, above;
        for (const KdRef &r : prims) {
            // A reference whose (padded) bounds merely touch the plane stays on its own side:
            // candidate planes are exactly such bounds, and listing the touching triangle on
            // both sides would duplicate one reference at every split.
            bool lo_side = r.box.lo[best_axis] < best_split, hi_side = r.box.hi[best_axis] > best_split;
            if (!lo_side && !hi_side) lo_side = true; // flat on the plane
            if (lo_side && hi_side) {
                KdRef c;
                c.tri = r.tri;
                if (clip(r.tri, bl, c.box)) below.push_back(c);
                if (clip(r.tri, bh, c.box)) above.push_back(c);
            } else if (lo_side) {
                below.push_back(r);
            } else {
                above.push_back(r);
            }
        }
        if (below.size() == prims.size() && above.size() == prims.size()) {
            make_leaf(prims);
            return;
        }
        size_t me = nodes.size();
        DevKdNode n;
        n.split = best_split;
        n.a = (uint32_t)best_axis;
        n.b = 0;
        nodes.push_back(n);
        build(below, bl, depth + 1, bad);
        nodes[me].a = (uint32_t)((nodes.size() << 2) | (uint32_t)best_axis);
        build(above, bh, depth + 1, bad);
    }
};

// ---------------------------------------------------------------- flattening
struct Flattener {
    const rt_scene_desc *d;
    FlatScene &out;
    int skip_count = 0; // unions/differences of the current object that need DEV_PROG_SKIP

    const rt_shape_node &node(int idx) const
    {
        if (idx < 0 || (uint32_t)idx >= d->n_nodes) throw Err{RT_ERR_INVALID, "shape node index out of range"};
        return d->nodes[idx];
    }

    void frames(const rt_shape_node &nd, const Mat4 &Wp, const Mat4 &Op, Mat4 &W, Mat4 &O) const
    {
        Mat4 w2o = mat_load(nd.w2o), o2w = mat_load(nd.o2w);
        if (!is_affine(w2o) || !is_affine(o2w))
            throw Err{RT_ERR_UNSUPPORTED, "projective shape transforms are not supported on the device"};
        W = mat_mul(Wp, w2o); // world -> parent space -> this node's object space
        O = mat_mul(o2w, Op);
    }

    int push_leaf(int kind, const Mat4 &W, const Mat4 &O)
    {
        DevLeaf L;
        std::memset(&L, 0, sizeof(L));
        L.kind = kind;
        L.identity = (is_identity(W) && is_identity(O)) ? 1 : 0;
        store_xform(W, L.W);
        store_xform(O, L.O);
        out.leaves.push_back(L);
        return (int)out.leaves.size() - 1;
    }

    static void rect_bounds(int kind, const double *p, double *b)
    {
        // XYRectangle::new / XZRectangle::new / ZYRectangle::new (plane.rs:119-132,164-177,209-222)
        double ox = p[0], oy = p[1], oz = p[2], w = p[3], h = p[4];
        if (kind == RT_SHAPE_XYRECT) {
            b[0] = ox - w / 2.0;
            b[1] = ox + w / 2.0;
            b[2] = oy - h / 2.0;
            b[3] = oy + h / 2.0;
        } else if (kind == RT_SHAPE_XZRECT) {
            b[0] = ox - (w / 2.0);
            b[1] = ox + w / 2.0;
            b[2] = oz - (h / 2.0);
            b[3] = oz + h / 2.0;
        } else {
            b[0] = oz - (w / 2.0);
            b[1] = oz + w / 2.0;
            b[2] = oy - (h / 2.0);
            b[3] = oy + h / 2.0;
        }
    }

    // What every hit of a plane would work out again (plane.rs:62-79, system.rs:196): the unit world normal and the
    // uv basis, one set for the side the normal points to and one for the other (n = N * -1.0).  The same operations in
    // the same order as the kernels' normalize3 / cross3 / xf_normal / dev_plane_uv (rt_dev_math.h, rt_dev_shapes.h),
    // so the constants are the bits a per-hit evaluation would give.  p[6..8] / p[21..23]: world normal front / back;
    // p[9..14] / p[15..20]: bu, bv front / back.
    static void unit3(const double a[3], double r[3])
    {
        const double l2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        if (l2 > 0.0) {
            const double inv = 1.0 / std::sqrt(l2);
            for (int k = 0; k < 3; ++k) r[k] = a[k] * inv;
        } else {
            for (int k = 0; k < 3; ++k) r[k] = a[k];
        }
    }
    static void cross(const double a[3], const double b[3], double r[3])
    {
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
    }
    static void plane_constants(DevLeaf &L)
    {
        for (int side = 0; side < 2; ++side) {
            double n[3];
            for (int k = 0; k < 3; ++k) n[k] = side ? L.p[3 + k] * -1.0 : L.p[3 + k];
            double w[3];
            if (L.identity) {
                for (int k = 0; k < 3; ++k) w[k] = n[k];
            } else {
                const double *m = L.W.m;
                w[0] = n[0] * m[0] + n[1] * m[1] + n[2] * m[2];
                w[1] = n[0] * m[3] + n[1] * m[4] + n[2] * m[5];
                w[2] = n[0] * m[6] + n[1] * m[7] + n[2] * m[8];
            }
            unit3(w, L.p + (side ? 21 : 6));
            const double ex[3] = {1.0, 0.0, 0.0}, ey[3] = {0.0, 1.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
            double u[3];
            cross(n, ex, u);
            if (u[0] * u[0] + u[1] * u[1] + u[2] * u[2] < 1e-6) cross(n, ey, u);
            if (u[0] * u[0] + u[1] * u[1] + u[2] * u[2] < 1e-6) cross(n, ez, u);
            double *b = L.p + (side ? 15 : 9);
            unit3(u, b);
            cross(n, b, b + 3);
        }
    }

    // a primitive; returns the leaf index
    int emit_primitive(const rt_shape_node &nd, const Mat4 &W, const Mat4 &O)
    {
        const double *p = nd.params;
        int li = -1;
        switch (nd.kind) {
        case RT_SHAPE_SPHERE: {
            li = push_leaf(DL_SPHERE, W, O);
            double *q = out.leaves[li].p;
            q[0] = p[0];
            q[1] = p[1];
            q[2] = p[2];
            q[3] = p[3] * p[3]; // radius.powi(2), sphere.rs:22
            break;
        }
        case RT_SHAPE_PLANE: {
            li = push_leaf(DL_PLANE, W, O);
            for (int k = 0; k < 6; ++k) out.leaves[li].p[k] = p[k];
            plane_constants(out.leaves[li]);
            break;
        }
        case RT_SHAPE_XYRECT:
        case RT_SHAPE_XZRECT:
        case RT_SHAPE_ZYRECT: {
            int kind = nd.kind == RT_SHAPE_XYRECT ? DL_RECT_XY : (nd.kind == RT_SHAPE_XZRECT ? DL_RECT_XZ : DL_RECT_ZY);
            li = push_leaf(kind, W, O);
            double *q = out.leaves[li].p;
            q[0] = p[0];
            q[1] = p[1];
            q[2] = p[2];
            q[3] = (nd.flags & 1) ? -1.0 : 1.0;
            rect_bounds(nd.kind, p, q + 4);
            break;
        }
        case RT_SHAPE_CUBE: {
            // Cube::new (cube.rs:21-65): six rectangles built from centre + extent
            li = push_leaf(DL_CUBE, W, O);
            double *q = out.leaves[li].p;
            double mn[3], mx[3];
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::fmin(p[k], p[3 + k]);
                mx[k] = std::fmax(p[k], p[3 + k]);
            }
            double c[3], e[3];
            for (int k = 0; k < 3; ++k) {
                e[k] = mx[k] - mn[k];
                c[k] = (mx[k] - mn[k]) / 2.0 + mn[k];
            }
            for (int f = 0; f < 6; ++f) {
                double rp[5];
                double b[4];
                double *fq = q + f * 5;
                if (f < 2) { // zyrect at x: origin (x, cy, cz), width = ez, height = ey
                    rp[0] = f == 0 ? mn[0] : mx[0];
                    rp[1] = c[1];
                    rp[2] = c[2];
                    rp[3] = e[2];
                    rp[4] = e[1];
                    rect_bounds(RT_SHAPE_ZYRECT, rp, b);
                    fq[0] = rp[0];
                } else if (f < 4) { // xzrect at y: origin (cx, y, cz), width = ex, height = ez
                    rp[0] = c[0];
                    rp[1] = f == 2 ? mn[1] : mx[1];
                    rp[2] = c[2];
                    rp[3] = e[0];
                    rp[4] = e[2];
                    rect_bounds(RT_SHAPE_XZRECT, rp, b);
                    fq[0] = rp[1];
                } else { // xyrect at z: origin (cx, cy, z), width = ex, height = ey
                    rp[0] = c[0];
                    rp[1] = c[1];
                    rp[2] = f == 4 ? mn[2] : mx[2];
                    rp[3] = e[0];
                    rp[4] = e[1];
                    rect_bounds(RT_SHAPE_XYRECT, rp, b);
                    fq[0] = rp[2];
                }
                for (int k = 0; k < 4; ++k) fq[1 + k] = b[k];
            }
            q[30] = c[0];
            q[31] = c[1];
            q[32] = c[2];
            break;
        }
        case RT_SHAPE_CYLINDER: {
            li = push_leaf(DL_CYLINDER, W, O);
            out.leaves[li].p[0] = p[0];
            out.leaves[li].p[1] = p[1];
            break;
        }
        case RT_SHAPE_TORUS: {
            li = push_leaf(DL_TORUS, W, O);
            out.leaves[li].p[0] = p[0];
            out.leaves[li].p[1] = p[1];
            break;
        }
        default: throw Err{RT_ERR_INVALID, "not a primitive"};
        }
        return li;
    }

    // World bounds of what a ray must meet for a subtree to yield any interval at all.  A primitive: its own bounds.
    // A union: both operands' (either may yield).  A difference a - b: a's (csg.rs:216-218: nothing without a; what b
    // does to the result - including the to_world the reference skips when b is missed - never adds a ray that a's
    // primitives did not meet).  An intersection: one operand's, the smaller (both must be met; the boxes' common
    // part would not do: a ray may meet the two boxes in different places).  `bounded` false: a plane somewhere
    // the result depends on.
    struct NeedBox {
        bool bounded = true, any = false;
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        void add(const NeedBox &o)
        {
            bounded = bounded && o.bounded;
            any = any || o.any;
            for (int k = 0; k < 3; ++k) {
                lo[k] = std::fmin(lo[k], o.lo[k]);
                hi[k] = std::fmax(hi[k], o.hi[k]);
            }
        }
        double area() const
        {
            if (!bounded || !any) return 1e300;
            const double e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
            return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
        }
    };

    static NeedBox leaf_world_box(const DevLeaf &L)
    {
        NeedBox bx;
        const double *p = L.p;
        double b[6]; // leaf-space min xyz, max xyz
        switch (L.kind) {
        case DL_NONE: return bx; // a frame, no surface
        case DL_SPHERE: {
            const double r = std::sqrt(p[3]);
            for (int k = 0; k < 3; ++k) {
                b[k] = p[k] - r;
                b[3 + k] = p[k] + r;
            }
            break;
        }
        case DL_RECT_XY: b[0] = p[4], b[3] = p[5], b[1] = p[6], b[4] = p[7], b[2] = b[5] = p[2]; break;
        case DL_RECT_XZ: b[0] = p[4], b[3] = p[5], b[2] = p[6], b[5] = p[7], b[1] = b[4] = p[1]; break;
        case DL_RECT_ZY: b[2] = p[4], b[5] = p[5], b[1] = p[6], b[4] = p[7], b[0] = b[3] = p[0]; break;
        case DL_CUBE: b[0] = p[0], b[3] = p[5], b[1] = p[10], b[4] = p[15], b[2] = p[20], b[5] = p[25]; break;
        case DL_CYLINDER: // whichever way the axis is placed along y
            b[0] = b[2] = -p[0], b[3] = b[5] = p[0], b[1] = -std::fabs(p[1]), b[4] = std::fabs(p[1]);
            break;
        case DL_TORUS: {
            const double r = std::fabs(p[0]) + std::fabs(p[1]);
            for (int k = 0; k < 3; ++k) b[k] = -r, b[3 + k] = r;
            break;
        }
        default: bx.bounded = false; return bx; // a plane: no bounds
        }
        bx.any = true;
        for (int c = 0; c < 8; ++c) {
            const double q[3] = {b[(c & 1) ? 3 : 0], b[(c & 2) ? 4 : 1], b[(c & 4) ? 5 : 2]};
            for (int k = 0; k < 3; ++k) {
                const double w = L.identity ? q[k] : q[0] * L.O.m[0 * 3 + k] + q[1] * L.O.m[1 * 3 + k] + q[2] * L.O.m[2 * 3 + k] + L.O.m[9 + k];
                bx.lo[k] = std::fmin(bx.lo[k], w);
                bx.hi[k] = std::fmax(bx.hi[k], w);
            }
        }
        return bx;
    }

    void check_stack(int idx)
    {
        const int need = stack_need(idx);
        if (need > DEV_CSG_STACK)
            throw Err{RT_ERR_UNSUPPORTED, "CSG tree needs " + std::to_string(need) + " interval lists at once (the device keeps " +
                                              std::to_string(DEV_CSG_STACK) + "): nest the operations on the first operand instead"};
        out.max_stack = std::max(out.max_stack, need);
        out.max_intervals = std::max(out.max_intervals, interval_need(idx));
    }

    // most intervals any list can hold while the subtree is evaluated (its own result and its operands'):
    // a primitive is one interval, a torus up to two (torus.rs:88-125); a union or a difference of lists of a and b
    // intervals has at most a + b (each interval of b can split one of a), an intersection a + b - 1 (csg.rs)
    int interval_need(int idx, int depth = 0, int *result = nullptr) const
    {
        if (depth > 4096) throw Err{RT_ERR_INVALID, "shape tree too deep (or cyclic)"};
        const rt_shape_node &nd = node(idx);
        int mine = 1, most = 1;
        if (nd.kind == RT_SHAPE_MEDIUM) {
            most = interval_need(nd.child_a, depth + 1, &mine);
        } else if (nd.kind == RT_SHAPE_UNION || nd.kind == RT_SHAPE_INTERSECTION || nd.kind == RT_SHAPE_DIFFERENCE) {
            int a = 1, b = 1;
            const int ma = interval_need(nd.child_a, depth + 1, &a), mb = interval_need(nd.child_b, depth + 1, &b);
            mine = nd.kind == RT_SHAPE_INTERSECTION ? a + b - 1 : a + b;
            if (mine > 1 << 20) mine = 1 << 20;
            most = std::max(std::max(ma, mb), mine);
        } else {
            mine = most = nd.kind == RT_SHAPE_TORUS ? 2 : 1;
        }
        if (result) *result = mine;
        return most;
    }

    // interval lists alive at once while the subtree is evaluated (a, then b, then the operation)
    int stack_need(int idx, int depth = 0) const
    {
        if (depth > 4096) throw Err{RT_ERR_INVALID, "shape tree too deep (or cyclic)"};
        const rt_shape_node &nd = node(idx);
        if (nd.kind == RT_SHAPE_MEDIUM) return stack_need(nd.child_a, depth + 1);
        if (nd.kind != RT_SHAPE_UNION && nd.kind != RT_SHAPE_INTERSECTION && nd.kind != RT_SHAPE_DIFFERENCE) return 1;
        const int na = stack_need(nd.child_a, depth + 1), nb = stack_need(nd.child_b, depth + 1);
        return std::max(na, nb + 1);
    }

    // anything that yields intervals: appends to the program (postfix); returns the bounds a ray must meet (NeedBox)
    NeedBox emit_intervals(int idx, const Mat4 &Wp, const Mat4 &Op, int leaf0)
    {
        const rt_shape_node &nd = node(idx);
        if (nd.kind == RT_SHAPE_MEDIUM) {
            // HomogenousMedium::intersection_intervals forwards the ray untouched
            // (homogenous_medium.rs:64-66): its own transformation plays no part here.
            return emit_intervals(nd.child_a, Wp, Op, leaf0);
        }
        NeedBox bx;
        Mat4 W, O;
        frames(nd, Wp, Op, W, O);
        switch (nd.kind) {
        case RT_SHAPE_UNION:
        case RT_SHAPE_INTERSECTION:
        case RT_SHAPE_DIFFERENCE: {
            // csg.rs:33-38,216-221 return one operand's intervals without this node's to_world when
            // the other operand is missed.  With a pure translation that changes nothing.  With any
            // other transform the end points stay in this node's space: t scaled, normal short of one
            // map.  The program word then names a DL_NONE leaf that holds what csg_eval needs to do the
            // same (DEV_PROG_SKIP): W = world -> this node, p[0..8] = linear part of world -> parent,
            // O = (world -> parent) * (this node -> world), the map the normal misses.
            int skip_leaf = -1;
            const NeedBox ba = emit_intervals(nd.child_a, W, O, leaf0);
            const NeedBox bb = emit_intervals(nd.child_b, W, O, leaf0);
            if (nd.kind == RT_SHAPE_UNION) {
                bx = ba;
                bx.add(bb);
            } else if (nd.kind == RT_SHAPE_DIFFERENCE) {
                bx = ba;
            } else {
                bx = ba.area() <= bb.area() ? ba : bb;
            }
            if (nd.kind != RT_SHAPE_INTERSECTION && !linear_part_is_identity(mat_load(nd.o2w))) {
                if (skip_count >= DEV_MAX_SKIPS)
                    throw Err{RT_ERR_UNSUPPORTED, "more than 20 unions/differences with a rotation or scale of their own in one object"};
                skip_leaf = push_leaf(DL_NONE, W, mat_mul(Wp, O));
                DevLeaf &Q = out.leaves[skip_leaf];
                Q.skip_ord = ++skip_count; // post-order: inner nodes get the lower ordinals
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) Q.p[r * 3 + c] = Wp.m[r][c];
                if (skip_leaf - leaf0 >= DEV_MAX_LEAVES_PER_OBJECT) throw Err{RT_ERR_UNSUPPORTED, "too many primitives in one object"};
            }
            int op = nd.kind == RT_SHAPE_UNION ? DEV_OP_UNION
                                               : (nd.kind == RT_SHAPE_INTERSECTION ? DEV_OP_INTERSECTION : DEV_OP_DIFFERENCE);
            out.prog.push_back((op << 24) | (skip_leaf >= 0 ? (DEV_PROG_SKIP | (skip_leaf - leaf0)) : 0));
            out.uses_csg = true;
            break;
        }
        case RT_SHAPE_COMPOSITE:
        case RT_SHAPE_MESH:
            throw Err{RT_ERR_UNSUPPORTED, "meshes inside CSG or fog boundaries are not supported on the device"};
        default: {
            int li = emit_primitive(nd, W, O);
            int rel = li - leaf0;
            if (rel >= DEV_MAX_LEAVES_PER_OBJECT) throw Err{RT_ERR_UNSUPPORTED, "too many primitives in one object"};
            out.prog.push_back((DEV_OP_LEAF << 24) | rel);
            bx = leaf_world_box(out.leaves[li]);
            break;
        }
        }
        return bx;
    }

    void emit_mesh(const rt_shape_node &nd, DevObject &obj)
    {
        if (nd.kind != RT_SHAPE_MESH) throw Err{RT_ERR_UNSUPPORTED, "composite children other than meshes"};
        if (!is_identity(mat_load(nd.o2w)) || !is_identity(mat_load(nd.w2o)))
            throw Err{RT_ERR_UNSUPPORTED, "mesh with a transformation of its own"};
        if (nd.child_a < 0 || (uint32_t)nd.child_a >= d->n_meshes) throw Err{RT_ERR_INVALID, "mesh index out of range"};
        const rt_mesh &m = d->meshes[nd.child_a];
        // the pools the caller's flatten pass filled: every range must lie inside them
        if ((uint64_t)m.first_vertex + m.n_vertices > d->n_vertices || (m.n_vertices && !d->vertices))
            throw Err{RT_ERR_INVALID, "mesh vertex range outside the vertex pool"};
        if ((uint64_t)m.first_normal + m.n_normals > d->n_normals || (m.n_normals && !d->normals))
            throw Err{RT_ERR_INVALID, "mesh normal range outside the normal pool"};
        if ((uint64_t)m.first_triangle + m.n_triangles > d->n_triangles || (m.n_triangles && (!d->tri_v || !d->tri_n)))
            throw Err{RT_ERR_INVALID, "mesh triangle range outside the triangle pool"};
        // a mesh hit travels as (mesh index in its object) << 24 | triangle index
        if (m.n_triangles > (1u << 24)) throw Err{RT_ERR_UNSUPPORTED, "mesh with more than 16 777 216 triangles"};
        if ((int)out.meshes.size() - obj.mesh0 >= 255) throw Err{RT_ERR_UNSUPPORTED, "composite with more than 255 meshes"};
        DevMesh dm;
        std::memset(&dm, 0, sizeof(dm));
        dm.tri0 = (uint32_t)(out.tris.size() / 9);
        dm.n_tris = m.n_triangles;
        dm.normal0 = m.first_normal;
        dm.smooth = (nd.flags & 1) ? 1u : 0u;
        for (int k = 0; k < 3; ++k) {
            dm.qmin[k] = dm.qmax[k] = 0.0; // Point::zero() seeds (mesh.rs:32-33)
            dm.bmin[k] = std::numeric_limits<double>::infinity();
            dm.bmax[k] = -std::numeric_limits<double>::infinity();
        }
        for (uint32_t v = 0; v < m.n_vertices; ++v) {
            const double *q = d->vertices + 3 * (size_t)(m.first_vertex + v);
            for (int k = 0; k < 3; ++k) {
                dm.qmin[k] = std::fmin(dm.qmin[k], q[k]);
                dm.qmax[k] = std::fmax(dm.qmax[k], q[k]);
            }
        }
        std::vector<Box> pb(m.n_triangles);
        std::vector<double> tri_verts;
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            const uint32_t *tv = d->tri_v + 3 * (size_t)(m.first_triangle + t);
            const uint32_t *tn = d->tri_n + 3 * (size_t)(m.first_triangle + t);
            const double *v[3];
            for (int k = 0; k < 3; ++k) {
                if (tv[k] >= m.n_vertices || tn[k] >= m.n_normals) throw Err{RT_ERR_INVALID, "triangle index out of range"};
                v[k] = d->vertices + 3 * (size_t)(m.first_vertex + tv[k]);
            }
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < 3; ++k) tri_verts.push_back(v[c][k]);
            for (int k = 0; k < 3; ++k) out.tris.push_back(v[0][k]);
            for (int k = 0; k < 3; ++k) out.tris.push_back(v[1][k] - v[0][k]); // v0v1 (mesh.rs:74)
            for (int k = 0; k < 3; ++k) out.tris.push_back(v[2][k] - v[0][k]); // v0v2 (mesh.rs:75)
            for (int k = 0; k < 3; ++k) out.tri_n.push_back(tn[k]);
            for (int k = 0; k < 3; ++k) {
                pb[t].lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k]));
                pb[t].hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
                dm.bmin[k] = std::fmin(dm.bmin[k], pb[t].lo[k]);
                dm.bmax[k] = std::fmax(dm.bmax[k], pb[t].hi[k]);
            }
        }
        // pad the boxes so that rounding in the traversal cannot step over a triangle
        double ext = 0.0;
        for (int k = 0; k < 3; ++k)
            if (m.n_triangles) ext = std::fmax(ext, std::fmax(std::fabs(dm.bmin[k]), std::fabs(dm.bmax[k])));
        const double pad = 1e-9 * (ext > 0.0 ? ext : 1.0);
        for (int k = 0; k < 3; ++k) {
            if (!m.n_triangles) dm.bmin[k] = dm.bmax[k] = 0.0;
            dm.bmin[k] -= 2.0 * pad;
            dm.bmax[k] += 2.0 * pad;
        }
        // Part boxes (DevMesh::part): the triangles split into up to DEV_MESH_PARTS groups, each time the group with the
        // largest box cut where the two halves' boxes have the least surface (a sweep over the triangles sorted by
        // centroid along each axis) - the chance that a ray crossing the mesh's bounds meets a box goes with its
        // surface.  Kept when they shrink the total surface to under 85 %.
        dm.n_parts = 0;
        if (m.n_triangles >= 16 && !std::getenv("RT_NO_MESH_PARTS")) {
            auto area_of = [](const Box &b) {
                const double e[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
                return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
            };
            auto box_of = [&](const std::vector<uint32_t> &g) {
                Box b;
                for (int k = 0; k < 3; ++k) b.lo[k] = std::numeric_limits<double>::infinity(), b.hi[k] = -std::numeric_limits<double>::infinity();
                for (uint32_t t : g)
                    for (int k = 0; k < 3; ++k) {
                        b.lo[k] = std::fmin(b.lo[k], pb[t].lo[k]);
                        b.hi[k] = std::fmax(b.hi[k], pb[t].hi[k]);
                    }
                return b;
            };
            std::vector<std::vector<uint32_t>> groups(1);
            groups[0].resize(m.n_triangles);
            for (uint32_t t = 0; t < m.n_triangles; ++t) groups[0][t] = t;
            while (groups.size() < DEV_MESH_PARTS) {
                size_t gi = 0;
                double worst = -1.0;
                for (size_t k = 0; k < groups.size(); ++k) {
                    const double a = groups[k].size() >= 2 ? area_of(box_of(groups[k])) : -1.0;
                    if (a > worst) worst = a, gi = k;
                }
                if (worst <= 0.0) break;
                std::vector<uint32_t> g = groups[gi];
                double best_cost = std::numeric_limits<double>::infinity();
                std::vector<uint32_t> best_order;
                size_t best_at = 0;
                for (int axis = 0; axis < 3; ++axis) {
                    std::vector<uint32_t> order = g;
                    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                        return pb[a].lo[axis] + pb[a].hi[axis] < pb[b].lo[axis] + pb[b].hi[axis];
                    });
                    std::vector<double> right(order.size() + 1, 0.0);
                    Box acc;
                    for (int k = 0; k < 3; ++k) acc.lo[k] = std::numeric_limits<double>::infinity(), acc.hi[k] = -std::numeric_limits<double>::infinity();
                    for (size_t i = order.size(); i-- > 0;) {
                        for (int k = 0; k < 3; ++k) {
                            acc.lo[k] = std::fmin(acc.lo[k], pb[order[i]].lo[k]);
                            acc.hi[k] = std::fmax(acc.hi[k], pb[order[i]].hi[k]);
                        }
                        right[i] = area_of(acc);
                    }
                    for (int k = 0; k < 3; ++k) acc.lo[k] = std::numeric_limits<double>::infinity(), acc.hi[k] = -std::numeric_limits<double>::infinity();
                    for (size_t i = 0; i + 1 < order.size(); ++i) {
                        for (int k = 0; k < 3; ++k) {
                            acc.lo[k] = std::fmin(acc.lo[k], pb[order[i]].lo[k]);
                            acc.hi[k] = std::fmax(acc.hi[k], pb[order[i]].hi[k]);
                        }
                        const double cost = area_of(acc) + right[i + 1];
                        if (cost < best_cost) {
                            best_cost = cost;
                            best_at = i + 1;
                            best_order = order;
                        }
                    }
                }
                if (best_order.empty()) break;
                groups[gi].assign(best_order.begin(), best_order.begin() + (long)best_at);
                groups.emplace_back(best_order.begin() + (long)best_at, best_order.end());
            }
            double total = 0.0;
            for (const auto &g : groups) total += area_of(box_of(g));
            Box whole;
            for (int k = 0; k < 3; ++k) whole.lo[k] = dm.bmin[k], whole.hi[k] = dm.bmax[k];
            if (groups.size() > 1 && total < 0.85 * area_of(whole)) {
                dm.n_parts = (uint32_t)groups.size();
                for (size_t gk = 0; gk < groups.size(); ++gk) {
                    const Box b = box_of(groups[gk]);
                    for (int k = 0; k < 3; ++k) {
                        dm.part[gk][k] = b.lo[k] - 2.0 * pad;
                        dm.part[gk][3 + k] = b.hi[k] + 2.0 * pad;
                    }
                }
            }
        }
        // resolution: about `density` cells per triangle, cells as close to cubes as the bounds
        // allow, and all grids of the scene together within the LDS budget of k_mesh
        double density = 2.0; // measured on MI355X with LinkedTorus.obj, k_mesh per 4K frame: 1.76 / 1.74 / 1.64 / 1.62 / 1.69 / 1.74 / 1.90 ms at 0.5 / 1 / 1.5 / 2 / 3 / 4 / 8 (with the f64 test alone 4 was best: longer lists cost more then)
        if (const char *e = std::getenv("RT_GRID_DENSITY")) density = std::atof(e); // tuning knob, not API
        uint32_t n_mesh_nodes = 0;
        for (uint32_t i = 0; i < d->n_nodes; ++i) n_mesh_nodes += d->nodes[i].kind == RT_SHAPE_MESH ? 1u : 0u;
        const double budget = (double)DEV_GRID_MAX_WORDS * 32.0 / (double)std::max<uint32_t>(1u, n_mesh_nodes) - 32.0;
        if (budget < 1.0) throw Err{RT_ERR_UNSUPPORTED, "too many meshes for the device grid budget"};
        double target = std::fmin(std::fmax(density * (double)m.n_triangles, 1.0), budget);
        double e3[3], vol = 1.0;
        for (int k = 0; k < 3; ++k) {
            e3[k] = dm.bmax[k] - dm.bmin[k];
            vol *= e3[k];
        }
        const double edge = std::cbrt(vol / target);
        for (int pass = 0; pass < 64; ++pass) {
            double cells = 1.0;
            for (int k = 0; k < 3; ++k) {
                double nk = std::floor(e3[k] / (edge * std::pow(1.05, pass)) + 0.5);
                dm.n[k] = (uint32_t)std::fmin(std::fmax(nk, 1.0), (double)DEV_GRID_MAX_AXIS);
                cells *= (double)dm.n[k];
            }
            if (cells <= budget) break; // rounding up on a flat mesh can overshoot the budget: coarsen
        }
        const size_t n_cells = (size_t)dm.n[0] * dm.n[1] * dm.n[2];
        if ((double)n_cells > budget) throw Err{RT_ERR_UNSUPPORTED, "mesh grid does not fit the device budget"};
        for (int k = 0; k < 3; ++k) {
            dm.cell[k] = e3[k] / (double)dm.n[k];
            dm.inv_cell[k] = 1.0 / dm.cell[k];
        }
        // triangles per cell: the cells of the triangle's bounding box that the triangle touches
        std::vector<std::vector<uint32_t>> cells(n_cells);
        for (uint32_t t = 0; t < m.n_triangles; ++t) {
            int lo[3], hi[3];
            for (int k = 0; k < 3; ++k) {
                lo[k] = (int)std::floor((pb[t].lo[k] - pad - dm.bmin[k]) * dm.inv_cell[k]);
                hi[k] = (int)std::floor((pb[t].hi[k] + pad - dm.bmin[k]) * dm.inv_cell[k]);
                lo[k] = std::max(0, std::min(lo[k] - 1, (int)dm.n[k] - 1)); // one cell of margin, then the exact test
                hi[k] = std::max(0, std::min(hi[k] + 1, (int)dm.n[k] - 1));
            }
            for (int z = lo[2]; z <= hi[2]; ++z)
                for (int y = lo[1]; y <= hi[1]; ++y)
                    for (int x = lo[0]; x <= hi[0]; ++x) {
                        Box c;
                        const int idx[3] = {x, y, z};
                        for (int k = 0; k < 3; ++k) {
                            c.lo[k] = dm.bmin[k] + (double)idx[k] * dm.cell[k];
                            c.hi[k] = dm.bmin[k] + (double)(idx[k] + 1) * dm.cell[k];
                        }
                        if (tri_touches_box(&tri_verts[(size_t)t * 9], c, pad))
                            cells[((size_t)z * dm.n[1] + y) * dm.n[0] + x].push_back(t); // ascending t
                    }
        }
        // occupancy words: (32 cell bits, number of occupied cells of this mesh before the word);
        // cell_start is indexed by that rank, so only occupied cells take an entry
        dm.bits0 = (uint32_t)(out.grid_bits.size() / 2);
        dm.cell0 = (uint32_t)out.cell_start.size();
        const size_t n_words = (n_cells + 31) / 32;
        uint32_t rank = 0;
        for (size_t wd = 0; wd < n_words; ++wd) {
            uint32_t bits = 0;
            for (size_t c = wd * 32; c < std::min(n_cells, wd * 32 + 32); ++c)
                if (!cells[c].empty()) bits |= 1u << (c % 32);
            out.grid_bits.push_back(bits);
            out.grid_bits.push_back(rank);
            rank += (uint32_t)__builtin_popcount(bits);
        }
        // Chebyshev distance to the nearest occupied cell, by repeated dilation, capped at 15
        {
            std::vector<uint8_t> dist(n_cells, 15);
            for (size_t c = 0; c < n_cells; ++c)
                if (!cells[c].empty()) dist[c] = 0;
            const int nx = (int)dm.n[0], ny = (int)dm.n[1], nz = (int)dm.n[2];
            for (int r = 1; r < 15; ++r) {
                bool any = false;
                for (int z = 0; z < nz; ++z)
                    for (int y = 0; y < ny; ++y)
                        for (int x = 0; x < nx; ++x) {
                            const size_t c = ((size_t)z * ny + y) * nx + x;
                            if (dist[c] != 15) continue;
                            bool near = false;
                            for (int dz = -1; dz <= 1 && !near; ++dz)
                                for (int dy = -1; dy <= 1 && !near; ++dy)
                                    for (int dx = -1; dx <= 1 && !near; ++dx) {
                                        int X = x + dx, Y = y + dy, Z = z + dz;
                                        if (X < 0 || Y < 0 || Z < 0 || X >= nx || Y >= ny || Z >= nz) continue;
                                        near = dist[((size_t)Z * ny + Y) * nx + X] == r - 1;
                                    }
                            if (near) {
                                dist[c] = (uint8_t)r;
                                any = true;
                            }
                        }
                if (!any) break;
            }
            dm.dist0 = (uint32_t)out.grid_dist.size();
            out.grid_dist.resize(out.grid_dist.size() + (n_cells + 7) / 8, 0u);
            for (size_t c = 0; c < n_cells; ++c) out.grid_dist[dm.dist0 + c / 8] |= (uint32_t)dist[c] << (4 * (c % 8));
        }
        for (size_t c = 0; c < n_cells; ++c) {
            if (cells[c].empty()) continue;
            out.cell_start.push_back((uint32_t)(out.cell_tris.size() / 10));
            // cell-ordered copies of the triangle records: a cell's triangles are one contiguous
            // run, with no index indirection in the walk's dependent-load chain.  Runs are padded to an
            // even length (the f32 pre-test below takes two records at a time).
            const size_t run0 = out.cell_tris.size() / 10;
            for (uint32_t t : cells[c]) {
                for (int k = 0; k < 9; ++k) out.cell_tris.push_back(out.tris[((size_t)dm.tri0 + t) * 9 + k]);
                double tag = 0.0;
                std::memcpy(&tag, &t, sizeof(t));
                out.cell_tris.push_back(tag);
            }
            if (cells[c].size() & 1u) {
                const uint32_t none = 0xFFFFFFFFu;
                double tag = 0.0;
                std::memcpy(&tag, &none, sizeof(none));
                for (int k = 0; k < 9; ++k) out.cell_tris.push_back(0.0);
                out.cell_tris.push_back(tag);
            }
            // The same run as f32 pairs for k_mesh's conservative pre-test: 20 floats per two records, component
            // by component (A, B): v0 relative to the cell's centre (both operands of o - v0 are then small, so
            // is their rounding), e1, e2, and the record's error bound delta.  In exact arithmetic a hit has
            // det > 0, 0 <= un <= det, vn >= 0, un + vn <= det (mesh.rs:76-92 times det).  Evaluated in f32 from
            // inputs rounded to f32, un and vn (three products of a component of o - v0 (|.| <= R), one of an
            // edge (<= E) and one of the unit direction, summed over six terms with fused multiply-adds) are off
            // by less than 60 eps E R and det by less than 48 eps E^2 (eps = 2^-24); delta = 128 eps E (R + E)
            // covers either with a factor two to spare.  R: farthest corner of the padded cell from v0 (the
            // ray's entry point into the cell is the other operand), E: the longer edge.
            const size_t cz = c / ((size_t)dm.n[0] * dm.n[1]), cy = (c / dm.n[0]) % dm.n[1], cx = c % dm.n[0];
            const size_t idx3[3] = {cx, cy, cz};
            double centre[3], half[3];
            for (int k = 0; k < 3; ++k) {
                centre[k] = dm.bmin[k] + ((double)idx3[k] + 0.5) * dm.cell[k]; // k_mesh evaluates the same expression
                half[k] = 0.5 * dm.cell[k] + 2.0 * pad;
            }
            const size_t n_run = out.cell_tris.size() / 10 - run0;
            for (size_t r = 0; r < n_run; r += 2) {
                float rec[20];
                for (int side = 0; side < 2; ++side) {
                    const double *q = &out.cell_tris[(run0 + r + side) * 10];
                    uint32_t ti;
                    std::memcpy(&ti, &q[9], sizeof(ti));
                    if (ti == 0xFFFFFFFFu) { // the padding record: fails every comparison
                        for (int k = 0; k < 9; ++k) rec[2 * k + side] = 0.f;
                        rec[18 + side] = -3.0e38f;
                        continue;
                    }
                    double E = 0.0, R = 0.0;
                    for (int e = 0; e < 2; ++e) {
                        const double *ev = q + 3 + 3 * e;
                        E = std::fmax(E, std::sqrt(ev[0] * ev[0] + ev[1] * ev[1] + ev[2] * ev[2]));
                    }
                    for (int k = 0; k < 3; ++k) {
                        const double far = std::fabs(q[k] - centre[k]) + half[k];
                        R += far * far;
                    }
                    R = std::sqrt(R) * (1.0 + 1e-6);
                    for (int k = 0; k < 3; ++k) rec[2 * k + side] = (float)(q[k] - centre[k]);
                    for (int k = 3; k < 9; ++k) rec[2 * k + side] = (float)q[k];
                    const double delta = 128.0 * 5.9604644775390625e-08 * E * (R + E) * 1.01 + 1e-37;
                    rec[18 + side] = (float)delta >= delta ? (float)delta : std::nextafterf((float)delta, 3.0e38f);
                }
                out.cell_pairs.insert(out.cell_pairs.end(), rec, rec + 20);
            }
        }
        out.cell_start.push_back((uint32_t)(out.cell_tris.size() / 10));
        {
            double n1 = 0.0, n2 = 0.0;
            for (size_t c = 0; c < n_cells; ++c) {
                n1 += (double)cells[c].size();
                n2 += (double)cells[c].size() * (double)cells[c].size();
            }
            if (n1 > 0.0) out.grid_crowd = std::fmax(out.grid_crowd, n2 / n1);
            // Structure per mesh.  The grid is the default (no stack, LDS-resident tables); it degrades when a
            // mesh crowds many triangles into few cells, which the tree's SAH splits handle.
            // LinkedTorus.obj: 7; Monkey.obj: 5; a "teapot in a stadium": 250.
            dm.crowd = n1 > 0.0 ? (float)(n2 / n1) : 0.f;
            dm.use_kd = dm.crowd > 24.f ? 1u : 0u;
        }
        if (out.grid_bits.size() / 2 > DEV_GRID_MAX_WORDS) throw Err{RT_ERR_UNSUPPORTED, "mesh grids exceed the device budget"};
        // Groups of 64 triangles that lie together (Morton order of the centroids) with their padded
        // bounds: a wave that tests one ray against a whole mesh (k_mesh_brute, k_tail) first tests
        // the group boxes, one per lane, and then only the groups whose box the ray meets.
        {
            std::vector<std::pair<uint64_t, uint32_t>> order(m.n_triangles);
            for (uint32_t t = 0; t < m.n_triangles; ++t) {
                uint64_t code = 0;
                uint32_t qc[3];
                for (int k = 0; k < 3; ++k) {
                    const double c = (pb[t].lo[k] + pb[t].hi[k]) * 0.5;
                    double f = (c - dm.bmin[k]) / (dm.bmax[k] - dm.bmin[k]);
                    f = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
                    qc[k] = (uint32_t)(f * 1023.0);
                }
                for (int b = 9; b >= 0; --b)
                    for (int k = 0; k < 3; ++k) code = (code << 1) | ((qc[k] >> b) & 1u);
                order[t] = {code, t};
            }
            std::sort(order.begin(), order.end());
            dm.grp0 = (uint32_t)(out.grp_idx.size() / 64);
            dm.n_groups = (m.n_triangles + 63u) / 64u;
            for (uint32_t g = 0; g < dm.n_groups; ++g) {
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for (uint32_t sl = 0; sl < 64u; ++sl) {
                    const uint32_t at = g * 64u + sl;
                    if (at < m.n_triangles) {
                        const uint32_t t = order[at].second;
                        out.grp_idx.push_back(t);
                        for (int k = 0; k < 9; ++k) out.grp_tris.push_back(out.tris[((size_t)dm.tri0 + t) * 9 + k]);
                        for (int k = 0; k < 3; ++k) {
                            lo[k] = std::fmin(lo[k], pb[t].lo[k]);
                            hi[k] = std::fmax(hi[k], pb[t].hi[k]);
                        }
                    } else {
                        out.grp_idx.push_back(0xFFFFFFFFu);
                        for (int k = 0; k < 9; ++k) out.grp_tris.push_back(0.0);
                    }
                }
                for (int k = 0; k < 3; ++k) out.grp_box.push_back(lo[k] - 2.0 * pad);
                for (int k = 0; k < 3; ++k) out.grp_box.push_back(hi[k] + 2.0 * pad);
            }
        }
        // The alternative structure, built for every mesh as well (k_mesh<KD>, chosen per scene in
        // rt_scene_create): a kd-tree with SAH + split clipping, 16-byte nodes, leaf-ordered records.
        {
            dm.kd0 = (uint32_t)out.kd.size();
            dm.kd_index0 = (uint32_t)out.kd_index.size();
            std::vector<DevKdNode> nodes; // mesh-relative node indices
            int max_depth = (int)std::lround(8.0 + 1.3 * std::log2((double)std::max<uint32_t>(1u, m.n_triangles)));
            if (max_depth > DEV_KD_STACK - 1) max_depth = DEV_KD_STACK - 1;
            KdBuilder kb{tri_verts, nodes, out.kd_index, dm.kd_index0, max_depth, pad};
            std::vector<KdRef> all(m.n_triangles);
            for (uint32_t t = 0; t < m.n_triangles; ++t) {
                all[t].tri = t;
                for (int k = 0; k < 3; ++k) {
                    all[t].box.lo[k] = pb[t].lo[k] - pad;
                    all[t].box.hi[k] = pb[t].hi[k] + pad;
                }
            }
            Box root;
            for (int k = 0; k < 3; ++k) {
                root.lo[k] = dm.bmin[k];
                root.hi[k] = dm.bmax[k];
            }
            kb.build(all, root, 0, 0);
            out.kd.insert(out.kd.end(), nodes.begin(), nodes.end());
            for (size_t e = dm.kd_index0; e < out.kd_index.size(); ++e) {
                const uint32_t t = out.kd_index[e];
                for (int k = 0; k < 9; ++k) out.leaf_tris.push_back(out.tris[((size_t)dm.tri0 + t) * 9 + k]);
                double tag = 0.0;
                std::memcpy(&tag, &t, sizeof(t));
                out.leaf_tris.push_back(tag);
            }
        }
        out.meshes.push_back(dm);
        ++obj.n_meshes;
    }

    void emit_object(const rt_object &ro)
    {
        DevObject obj;
        std::memset(&obj, 0, sizeof(obj));
        obj.material = ro.material;
        obj.leaf0 = (int)out.leaves.size();
        obj.prog0 = (int)out.prog.size();
        obj.frame = -1;
        skip_count = 0;
        if (ro.material < 0 || (uint32_t)ro.material >= d->n_materials) throw Err{RT_ERR_INVALID, "material index out of range"};
        const rt_material &m = d->materials[ro.material];
        obj.needs_uv = (m.texture >= 0 && d->textures[m.texture].kind != RT_TEX_SOLID) ? 1 : 0;

        const rt_shape_node &root = node(ro.shape);
        Mat4 I = mat_identity();
        NeedBox need; // what a ray must meet for the object to yield an interval (CSG programs, fog boundaries, primitives)
        switch (root.kind) {
        case RT_SHAPE_COMPOSITE: {
            obj.kind = DO_MESH;
            Mat4 W, O;
            frames(root, I, I, W, O);
            obj.frame = push_leaf(DL_NONE, W, O);
            obj.mesh0 = (int)out.meshes.size();
            for (int k = 0; k < root.child_b; ++k) {
                int64_t ci = (int64_t)root.child_a + k;
                if (ci < 0 || (uint64_t)ci >= d->n_child_list || !d->child_list) throw Err{RT_ERR_INVALID, "composite child out of range"};
                emit_mesh(node(d->child_list[ci]), obj);
            }
            break;
        }
        case RT_SHAPE_MESH: {
            obj.kind = DO_MESH;
            obj.frame = push_leaf(DL_NONE, I, I);
            obj.mesh0 = (int)out.meshes.size();
            emit_mesh(root, obj);
            break;
        }
        case RT_SHAPE_MEDIUM: {
            obj.kind = DO_MEDIUM;
            obj.density = root.params[0];
            Mat4 W, O;
            frames(root, I, I, W, O);
            obj.frame = push_leaf(DL_NONE, W, O);
            obj.leaf0 = (int)out.leaves.size();
            check_stack(root.child_a);
            need = emit_intervals(root.child_a, W, O, obj.leaf0);
            break;
        }
        default: {
            check_stack(ro.shape);
            need = emit_intervals(ro.shape, I, I, obj.leaf0);
            bool single = (out.prog.size() - (size_t)obj.prog0) == 1;
            obj.kind = single ? DO_LEAF : DO_CSG;
            bool entry_mode = root.kind == RT_SHAPE_CUBE || root.kind == RT_SHAPE_PLANE || root.kind == RT_SHAPE_XYRECT ||
                              root.kind == RT_SHAPE_XZRECT || root.kind == RT_SHAPE_ZYRECT;
            obj.mode = entry_mode ? DM_FIRST_ENTRY : DM_FIRST_POSITIVE;
            break;
        }
        }
        if (obj.kind == DO_MESH) {
            // world-space copies of the meshes' bounds and part boxes, for k_intersect's conservative test
            const DevLeaf &F = out.leaves[obj.frame];
            auto to_world_box = [&](const double *lo, const double *hi, double *wb) {
                double wl[3] = {1e300, 1e300, 1e300}, wh[3] = {-1e300, -1e300, -1e300};
                for (int c = 0; c < 8; ++c) {
                    const double q[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
                    for (int k = 0; k < 3; ++k) {
                        const double w = F.identity ? q[k] : q[0] * F.O.m[0 * 3 + k] + q[1] * F.O.m[1 * 3 + k] + q[2] * F.O.m[2 * 3 + k] + F.O.m[9 + k];
                        wl[k] = std::fmin(wl[k], w);
                        wh[k] = std::fmax(wh[k], w);
                    }
                }
                double ext = 0.0;
                for (int k = 0; k < 3; ++k) ext = std::fmax(ext, std::fmax(std::fabs(wl[k]), std::fabs(wh[k])));
                const double padw = 1e-7 * (ext > 0.0 ? ext : 1.0); // (k_intersect's test uses approximate reciprocals)
                for (int k = 0; k < 3; ++k) {
                    wb[k] = wl[k] - padw;
                    wb[3 + k] = wh[k] + padw;
                }
            };
            for (int mi = obj.mesh0; mi < (int)out.meshes.size(); ++mi) {
                DevMesh &dm = out.meshes[mi];
                to_world_box(dm.bmin, dm.bmax, dm.wbox);
                for (uint32_t k = 0; k < dm.n_parts; ++k) to_world_box(dm.part[k], dm.part[k] + 3, dm.wpart[k]);
            }
        }
        obj.n_leaves = (int)out.leaves.size() - obj.leaf0;
        obj.prog_len = (int)out.prog.size() - obj.prog0;
        if (obj.kind == DO_MEDIUM && obj.prog_len == 1) obj.prog_len = 0; // single boundary leaf
        if (obj.kind == DO_LEAF) obj.prog_len = 0;
        // world bounds for the objects whose evaluation is expensive: of the primitives a ray must meet for the
        // object to yield anything (NeedBox, above)
        bool wants_bounds = obj.kind == DO_CSG || obj.kind == DO_MEDIUM;
        for (int li = obj.leaf0; li < obj.leaf0 + obj.n_leaves; ++li)
            wants_bounds = wants_bounds || out.leaves[li].kind == DL_TORUS;
        if (wants_bounds && obj.kind != DO_MESH && need.bounded && need.any && need.lo[0] <= need.hi[0]) {
            // bit 1: no hit of this object lies before its bounds.  Not so with a union or difference whose skipped
            // to_world (DEV_PROG_SKIP) leaves a hit's t in the node's space: the hit may then appear nearer than
            // the primitive it came from
            obj.has_bounds = skip_count == 0 ? 3 : 1;
            for (int k = 0; k < 3; ++k) {
                const double padk = 1e-7 * (std::fmax(std::fabs(need.lo[k]), std::fabs(need.hi[k])) + 1.0);
                obj.wmin[k] = need.lo[k] - padk;
                obj.wmax[k] = need.hi[k] + padk;
            }
        }
        out.objects.push_back(obj);
    }
};

} // namespace

int flatten_scene(const rt_scene_desc *desc, FlatScene &out, std::string &err)
{
    try {
        if (!desc) throw Err{RT_ERR_INVALID, "null scene description"};
        {
            const struct {
                uint64_t n;
                const void *p;
                const char *what;
            } arrays[] = {{desc->n_objects, desc->objects, "objects"},       {desc->n_nodes, desc->nodes, "nodes"},
                          {desc->n_child_list, desc->child_list, "child_list"}, {desc->n_materials, desc->materials, "materials"},
                          {desc->n_textures, desc->textures, "textures"},       {desc->n_images, desc->images, "images"},
                          {desc->n_image_bytes, desc->image_data, "image_data"}, {desc->n_meshes, desc->meshes, "meshes"},
                          {desc->n_vertices, desc->vertices, "vertices"},       {desc->n_normals, desc->normals, "normals"},
                          {desc->n_triangles, desc->tri_v, "tri_v"},            {desc->n_triangles, desc->tri_n, "tri_n"}};
            for (const auto &a : arrays)
                if (a.n && !a.p) throw Err{RT_ERR_INVALID, std::string("null array with a non-zero count: ") + a.what};
        }
        Flattener f{desc, out};
        for (uint32_t k = 0; k < desc->n_materials; ++k) {
            const rt_material &m = desc->materials[k];
            if (m.texture >= (int32_t)desc->n_textures) throw Err{RT_ERR_INVALID, "texture index out of range"};
            if (m.kind != RT_MAT_DIELECTRIC && m.texture < 0) throw Err{RT_ERR_INVALID, "material without texture"};
            out.materials.push_back(DevMaterial{m.kind, m.texture, m.fuzz, m.ior, m.intensity});
        }
        for (uint32_t k = 0; k < desc->n_textures; ++k) {
            const rt_texture &t = desc->textures[k];
            DevTexture o;
            std::memset(&o, 0, sizeof(o));
            o.kind = t.kind;
            for (int c = 0; c < 3; ++c) {
                o.c1[c] = t.c1[c];
                o.c2[c] = t.c2[c];
            }
            o.scale = t.scale;
            if (t.kind == RT_TEX_IMAGE) {
                if (t.image < 0 || (uint32_t)t.image >= desc->n_images) throw Err{RT_ERR_INVALID, "image index out of range"};
                const rt_image &im = desc->images[t.image];
                if (im.width == 0 || im.height == 0 ||
                    im.offset + (uint64_t)im.width * im.height * 3 > desc->n_image_bytes)
                    throw Err{RT_ERR_INVALID, "image data out of range"};
                o.width = (int32_t)im.width;
                o.height = (int32_t)im.height;
                o.offset = im.offset;
            }
            out.textures.push_back(o);
        }
        if (desc->n_image_bytes) out.image_data.assign(desc->image_data, desc->image_data + desc->n_image_bytes);
        if (desc->n_normals) out.normals.assign(desc->normals, desc->normals + 3 * (size_t)desc->n_normals);
        for (uint32_t k = 0; k < desc->n_objects; ++k) f.emit_object(desc->objects[k]);
        // per-triangle vertex normals and the (object, mesh) job records of k_mesh
        for (size_t k = 0; k < out.objects.size(); ++k) {
            const DevObject &ob = out.objects[k];
            if (ob.kind != DO_MESH) continue;
            for (int mi = 0; mi < ob.n_meshes; ++mi) {
                const DevMesh &m = out.meshes[ob.mesh0 + mi];
                const DevLeaf &F = out.leaves[ob.frame];
                DevMeshJob j;
                std::memset(&j, 0, sizeof(j));
                j.object = (int32_t)k;
                j.mesh = mi;
                j.identity = F.identity;
                j.smooth = m.smooth;
                for (int a = 0; a < 3; ++a) {
                    j.n[a] = m.n[a];
                    j.qmin[a] = m.qmin[a];
                    j.qmax[a] = m.qmax[a];
                    j.bmin[a] = m.bmin[a];
                    j.bmax[a] = m.bmax[a];
                    j.inv_cell[a] = m.inv_cell[a];
                    j.cell[a] = m.cell[a];
                }
                j.bits0 = m.bits0;
                j.cell0 = m.cell0;
                j.dist0 = m.dist0;
                j.tri0 = m.tri0;
                j.kd0 = m.kd0;
                j.kd_index0 = m.kd_index0;
                j.use_kd = m.use_kd;
                j.W = F.W;
                j.O = F.O;
                out.jobs.push_back(j);
                if (out.tri_nrm.size() < ((size_t)m.tri0 + m.n_tris) * 9) out.tri_nrm.resize(((size_t)m.tri0 + m.n_tris) * 9, 0.0);
                for (uint32_t t = 0; t < m.n_tris; ++t)
                    for (int c = 0; c < 3; ++c) {
                        const uint32_t ni = out.tri_n[((size_t)m.tri0 + t) * 3 + c];
                        for (int a = 0; a < 3; ++a)
                            out.tri_nrm[((size_t)m.tri0 + t) * 9 + c * 3 + a] = out.normals[((size_t)m.normal0 + ni) * 3 + a];
                    }
            }
        }
        for (int k = 0; k < 3; ++k) out.background[k] = desc->background[k];
        out.cam_fov_factor = desc->camera.fov_factor;
        std::memcpy(out.cam_c2w, desc->camera.camera_to_world, sizeof(out.cam_c2w));
    } catch (const Err &e) {
        err = e.msg;
        return e.status;
    } catch (const std::exception &e) { // bad_alloc, length_error: nothing unwinds across the C ABI
        err = std::string("out of memory or size overflow while flattening the scene (") + e.what() + ")";
        return RT_ERR_INVALID;
    }
    return RT_OK;
}

} // namespace rt_host
