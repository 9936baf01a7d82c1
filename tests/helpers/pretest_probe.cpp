// Test probe for the f32 pre-test of the mesh kernels (rt_k_mesh_grid_round.inc): the product's flattener
// (rt_flatten.cpp) builds the grids and the f32 pair records, and this file runs the pre-test's arithmetic on the
// host, operation for operation (single-rounding fmaf = v_pk_fma_f32, plain float products and sums =
// v_pk_mul_f32 / v_pk_add_f32; built with -ffp-contract=off), against the product's own f64 triangle test
// (tri_hit_v, rt_dev_shapes.h, compiled as host code through tools/host_shim) for random rays through every
// occupied cell.  What must hold: a triangle the f64 test accepts is never rejected by the pre-test.
// Test infrastructure only.
#include "rt_dev_shapes.h"
#include "rt_flatten.h"

#include <cmath>
#include <cstdint>
#include <random>
#include <string>

namespace {
struct V2 {
    float a, b;
};
inline V2 fma2(V2 x, V2 y, V2 z) { return {std::fmaf(x.a, y.a, z.a), std::fmaf(x.b, y.b, z.b)}; }
inline V2 mul2(V2 x, V2 y) { return {x.a * y.a, x.b * y.b}; }
inline V2 add2(V2 x, V2 y) { return {x.a + y.a, x.b + y.b}; }
inline V2 sub2(V2 x, V2 y) { return {x.a - y.a, x.b - y.b}; }
inline V2 neg2(V2 x) { return {-x.a, -x.b}; }
inline V2 min2(V2 x, V2 y) { return {std::fmin(x.a, y.a), std::fmin(x.b, y.b)}; }
} // namespace

extern "C" {
// out: [0] occupied cells, [1] triangle records tested, [2] accepted by the f64 test, [3] passed by the pre-test,
// [4] accepted by the f64 test but rejected by the pre-test (must be 0), [5] grid jobs, [6] rays aimed at an edge or a
// corner of a triangle, [7] unused
int probe_pretest(const rt_scene_desc *desc, uint32_t rays_per_cell, uint64_t seed, uint64_t *out)
{
    rt_host::FlatScene flat;
    std::string err;
    int rc = rt_host::flatten_scene(desc, flat, err);
    if (rc) return rc;
    for (int k = 0; k < 8; ++k) out[k] = 0;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (const DevMeshJob &J : flat.jobs) {
        if (J.use_kd) continue;
        ++out[5];
        const uint32_t nx = J.n[0], ny = J.n[1], nz = J.n[2];
        for (uint32_t cz = 0; cz < nz; ++cz)
            for (uint32_t cy = 0; cy < ny; ++cy)
                for (uint32_t cx = 0; cx < nx; ++cx) {
                    const uint32_t cell = (cz * ny + cy) * nx + cx;
                    const uint32_t bits = flat.grid_bits[2 * (size_t)(J.bits0 + (cell >> 5))];
                    const uint32_t rank = flat.grid_bits[2 * (size_t)(J.bits0 + (cell >> 5)) + 1];
                    if (!((bits >> (cell & 31u)) & 1u)) continue;
                    const uint32_t r = J.cell0 + rank + (uint32_t)__builtin_popcount(bits & ((1u << (cell & 31u)) - 1u));
                    const uint32_t first = flat.cell_start[r], last = flat.cell_start[r + 1];
                    ++out[0];
                    const double c0[3] = {J.bmin[0] + (double)cx * J.cell[0], J.bmin[1] + (double)cy * J.cell[1], J.bmin[2] + (double)cz * J.cell[2]};
                    for (uint32_t k = 0; k < rays_per_cell; ++k) {
                        // a point of the cell (every fourth on a face: where a walk enters), a direction, an origin up
                        // to 1000 units back along it: the device sees the ray and the parameter of the entry point
                        double p[3] = {c0[0] + U(rng) * J.cell[0], c0[1] + U(rng) * J.cell[1], c0[2] + U(rng) * J.cell[2]};
                        if (k % 4 == 0) {
                            const int ax = (int)(U(rng) * 3.0) % 3;
                            p[ax] = c0[ax] + (U(rng) < 0.5 ? 0.0 : (ax == 0 ? J.cell[0] : (ax == 1 ? J.cell[1] : J.cell[2])));
                        }
                        double z = 2.0 * U(rng) - 1.0, ph = 6.283185307179586 * U(rng), s = std::sqrt(1.0 - z * z);
                        d3 d = mk3(s * std::cos(ph), s * std::sin(ph), z);
                        if (k % 16 == 1) d = mk3(k % 32 == 1 ? 1.0 : 0.0, k % 32 == 1 ? 0.0 : -1.0, 0.0); // axis-parallel
                        d = normalize3(d);
                        const double back = k % 3 == 0 ? 0.0 : (k % 3 == 1 ? 10.0 * U(rng) : 1000.0 * U(rng));
                        d3 o = mk3(p[0] - d.x * back, p[1] - d.y * back, p[2] - d.z * back);
                        double w_tc = back;
                        if (k % 4 == 2 && last > first) {
                            // aimed at an edge or a corner of one of the cell's triangles, where the f64 test decides by
                            // the last bit and the pre-test must not decide otherwise: the ray enters the cell a little
                            // before that point (kept only if the point of entry lies in the cell)
                            const uint32_t e = first + (uint32_t)(U(rng) * (double)(last - first)) % (last - first);
                            const double *rec = flat.cell_tris.data() + (size_t)e * 10;
                            double a = U(rng), b2 = 0.0;
                            const int where = (int)(U(rng) * 6.0) % 6;
                            if (where == 0) b2 = 0.0;                 // edge v = 0
                            else if (where == 1) { b2 = a; a = 0.0; } // edge u = 0
                            else if (where == 2) b2 = 1.0 - a;        // edge u + v = 1
                            else { a = where == 4 ? 1.0 : 0.0; b2 = where == 5 ? 1.0 : 0.0; } // a corner
                            const double tgt[3] = {rec[0] + a * rec[3] + b2 * rec[6], rec[1] + a * rec[4] + b2 * rec[7], rec[2] + a * rec[5] + b2 * rec[8]};
                            const double tau = U(rng) * 0.3 * std::fmin(J.cell[0], std::fmin(J.cell[1], J.cell[2]));
                            const double ent[3] = {tgt[0] - d.x * tau, tgt[1] - d.y * tau, tgt[2] - d.z * tau};
                            bool inside = true;
                            for (int ax = 0; ax < 3; ++ax) inside = inside && ent[ax] >= c0[ax] && ent[ax] <= c0[ax] + J.cell[ax];
                            if (inside) {
                                o = mk3(tgt[0] - d.x * (back + tau), tgt[1] - d.y * (back + tau), tgt[2] - d.z * (back + tau));
                                w_tc = back;
                                ++out[6];
                            }
                        }
                        // rt_k_mesh_grid_round.inc from here on
                        const float fox = (float)((o.x + d.x * w_tc) - (J.bmin[0] + ((double)cx + 0.5) * J.cell[0]));
                        const float foy = (float)((o.y + d.y * w_tc) - (J.bmin[1] + ((double)cy + 0.5) * J.cell[1]));
                        const float foz = (float)((o.z + d.z * w_tc) - (J.bmin[2] + ((double)cz + 0.5) * J.cell[2]));
                        const V2 OX = {fox, fox}, OY = {foy, foy}, OZ = {foz, foz};
                        const V2 DX = {(float)d.x, (float)d.x}, DY = {(float)d.y, (float)d.y}, DZ = {(float)d.z, (float)d.z};
                        for (uint32_t e = first; e < last; e += 2) {
                            const float *q = flat.cell_pairs.data() + (size_t)(e >> 1) * 20;
                            const V2 v0x = {q[0], q[1]}, v0y = {q[2], q[3]}, v0z = {q[4], q[5]};
                            const V2 e1x = {q[6], q[7]}, e1y = {q[8], q[9]}, e1z = {q[10], q[11]};
                            const V2 e2x = {q[12], q[13]}, e2y = {q[14], q[15]}, e2z = {q[16], q[17]}, dl = {q[18], q[19]};
                            const V2 px = fma2(DY, e2z, neg2(mul2(DZ, e2y))), py = fma2(DZ, e2x, neg2(mul2(DX, e2z))),
                                     pz = fma2(DX, e2y, neg2(mul2(DY, e2x)));
                            const V2 det = fma2(e1z, pz, fma2(e1y, py, mul2(e1x, px)));
                            const V2 tx = sub2(OX, v0x), ty = sub2(OY, v0y), tz = sub2(OZ, v0z);
                            const V2 un = fma2(tz, pz, fma2(ty, py, mul2(tx, px)));
                            const V2 qx = fma2(ty, e1z, neg2(mul2(tz, e1y))), qy = fma2(tz, e1x, neg2(mul2(tx, e1z))),
                                     qz = fma2(tx, e1y, neg2(mul2(ty, e1x)));
                            const V2 vn = fma2(DZ, qz, fma2(DY, qy, mul2(DX, qx)));
                            const V2 two = {2.0f, 2.0f};
                            const V2 m1 = add2(det, dl), m2 = add2(un, dl), m3 = add2(vn, dl), m4 = add2(sub2(det, un), mul2(two, dl)),
                                     m5 = add2(sub2(m4, vn), dl);
                            const V2 m = min2(min2(m1, m2), min2(min2(m3, m4), m5));
                            const bool pass[2] = {m.a >= 0.f, m.b >= 0.f};
                            for (uint32_t h = 0; h < 2 && e + h < last; ++h) {
                                const double *rec = flat.cell_tris.data() + (size_t)(e + h) * 10;
                                uint64_t raw;
                                std::memcpy(&raw, rec + 9, 8);
                                if ((uint32_t)raw == 0xFFFFFFFFu) continue; // the padding record of an odd run
                                ++out[1];
                                double t, u, v;
                                const bool hit = tri_hit_v(mk3(rec[0], rec[1], rec[2]), mk3(rec[3], rec[4], rec[5]), mk3(rec[6], rec[7], rec[8]), o, d, t, u, v);
                                out[2] += hit ? 1u : 0u;
                                out[3] += pass[h] ? 1u : 0u;
                                out[4] += (hit && !pass[h]) ? 1u : 0u;
                            }
                        }
                    }
                }
    }
    return 0;
}

// What the flattener decides about bounds (round 3): per object the world bounds of the primitives a ray must meet
// (DevObject::has_bounds, wmin, wmax), per mesh the part boxes that hold its triangles between them, in object and in
// world space, and the scene's bounds on its interval lists.
//   iout: [0] objects, [1] meshes, [2] max_intervals, [3] max_stack, [8 + k] has_bounds of object k (k < 16),
//         [32 + 2m] n_parts of mesh m (m < 4), [33 + 2m] triangles of mesh m with a vertex outside every part box
//   dout: [6k ..] wmin, wmax of object k; [128 + 64m ..] bmin, bmax, part[4][6], wbox, wpart[4][6] of mesh m
int probe_flat_bounds(const rt_scene_desc *desc, uint64_t *iout, double *dout)
{
    rt_host::FlatScene flat;
    std::string err;
    int rc = rt_host::flatten_scene(desc, flat, err);
    if (rc) return rc;
    iout[0] = flat.objects.size();
    iout[1] = flat.meshes.size();
    iout[2] = (uint64_t)flat.max_intervals;
    iout[3] = (uint64_t)flat.max_stack;
    for (size_t k = 0; k < flat.objects.size() && k < 16; ++k) {
        iout[8 + k] = (uint64_t)flat.objects[k].has_bounds;
        for (int c = 0; c < 3; ++c) {
            dout[6 * k + c] = flat.objects[k].wmin[c];
            dout[6 * k + 3 + c] = flat.objects[k].wmax[c];
        }
    }
    for (size_t m = 0; m < flat.meshes.size() && m < 4; ++m) {
        const DevMesh &M = flat.meshes[m];
        iout[32 + 2 * m] = M.n_parts;
        uint64_t outside = 0;
        for (uint32_t t = 0; t < M.n_tris; ++t) {
            const double *tr = flat.tris.data() + ((size_t)M.tri0 + t) * 9; // v0, v0v1, v0v2
            bool covered = false;
            for (uint32_t p = 0; p < M.n_parts && !covered; ++p) {
                bool in = true;
                for (int v = 0; v < 3 && in; ++v)
                    for (int c = 0; c < 3 && in; ++c) {
                        const double x = tr[c] + (v ? tr[3 * v + c] : 0.0);
                        in = x >= M.part[p][c] && x <= M.part[p][3 + c];
                    }
                covered = in;
            }
            if (!covered) ++outside;
        }
        iout[33 + 2 * m] = outside;
        double *d = dout + 128 + 64 * m;
        for (int c = 0; c < 3; ++c) d[c] = M.bmin[c], d[3 + c] = M.bmax[c];
        for (int p = 0; p < 4; ++p)
            for (int c = 0; c < 6; ++c) d[6 + 6 * p + c] = M.part[p][c];
        for (int c = 0; c < 6; ++c) d[30 + c] = M.wbox[c];
        for (int p = 0; p < 4; ++p)
            for (int c = 0; c < 6; ++c) d[36 + 6 * p + c] = M.wpart[p][c];
    }
    return 0;
}
}
