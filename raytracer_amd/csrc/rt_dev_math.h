// rt_dev_math.h — f64 vector helpers and the complex quartic solver for gfx950
// device code.  Everything lives in registers; no heap, no recursion.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device.h"

#define RT_DEV __device__ __forceinline__
// Waves per SIMD the variants of k_intersect are compiled for.  Measured on MI355X (4K csg.scene /
// 1080p room.scene / 4K x 4 samples fog_light.scene, ms per frame): evaluators out of line at 1 wave
// 7.8 / 54 / 17.8; inlined at 2 waves 5.1 / 34.2 / 12.7; inlined at 3 waves 5.2 / 31.9 / 11.7; at 4 waves
// 5.3 / 30.9 / 12.7 (170 spilled registers in the fog variant).  The torus variants (the quartic chain stays
// out of line) run at 2: at 1 the CSG + torus variant took 257 registers and flipped between 3.4 and 5.2 ms per
// 4K csg.scene frame with unrelated edits; held to 256 it is at 3.3 ms (torus.scene unchanged).
#ifndef RT_PLAIN_WAVES
#define RT_PLAIN_WAVES 3
#endif
#ifndef RT_TORUS_WAVES
#define RT_TORUS_WAVES 2
#endif
#ifndef RT_CSG_WAVES
#define RT_CSG_WAVES 3
#endif
#ifndef RT_MEDIUM_WAVES
#define RT_MEDIUM_WAVES 2
#endif
// The CSG evaluators are inlined into the kernel: device functions cannot carry amdgpu_waves_per_eu, so
// out of line they took 166-250 registers whatever the kernel asked for, and every call passed its
// interval lists through generic pointers into scratch memory (-DRT_CSG_INLINE=0 restores that).
#ifndef RT_CSG_INLINE
#define RT_CSG_INLINE 1
#endif
#if RT_CSG_INLINE
#define RT_CSG_FN __device__ __forceinline__
#else
#define RT_CSG_FN __device__ __noinline__
#endif
// the quartic chain of the torus (torus_ends -> quartic -> cubic -> quadratic)
#ifndef RT_TORUS_INLINE
#define RT_TORUS_INLINE 0
#endif
#if RT_TORUS_INLINE
#define RT_TORUS_FN __device__ __forceinline__
#else
#define RT_TORUS_FN __device__ __noinline__
#endif
#define RT_PI 3.14159265358979323846

struct d3 {
    double x, y, z;
};

RT_DEV d3 mk3(double x, double y, double z)
{
    d3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RT_DEV d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV d3 operator-(d3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_DEV d3 operator*(d3 a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV d3 operator*(double s, d3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV d3 cross3(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV double len2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV double len3(d3 a) { return sqrt(len2(a)); }
// direction.rs:57-65: the zero vector stays as it is
RT_DEV d3 normalize3(d3 a)
{
    double l2 = len2(a);
    if (l2 > 0.0) {
        double inv = 1.0 / sqrt(l2);
        return a * inv;
    }
    return a;
}
// f64::signum
RT_DEV double signum(double v) { return (v != v) ? v : copysign(1.0, v); }

// matrix.rs:198-225 for affine maps
RT_DEV d3 xf_point(const DevXform &M, d3 p)
{
    return mk3(p.x * M.m[0] + p.y * M.m[3] + p.z * M.m[6] + M.m[9], p.x * M.m[1] + p.y * M.m[4] + p.z * M.m[7] + M.m[10],
               p.x * M.m[2] + p.y * M.m[5] + p.z * M.m[8] + M.m[11]);
}
RT_DEV d3 xf_dir(const DevXform &M, d3 v)
{
    return mk3(v.x * M.m[0] + v.y * M.m[3] + v.z * M.m[6], v.x * M.m[1] + v.y * M.m[4] + v.z * M.m[7],
               v.x * M.m[2] + v.y * M.m[5] + v.z * M.m[8]);
}
// n * inverse(o2w)^T with inverse(o2w) == W (system.rs:196)
RT_DEV d3 xf_normal(const DevXform &W, d3 n)
{
    return mk3(n.x * W.m[0] + n.y * W.m[1] + n.z * W.m[2], n.x * W.m[3] + n.y * W.m[4] + n.z * W.m[5],
               n.x * W.m[6] + n.y * W.m[7] + n.z * W.m[8]);
}

// ---------------------------------------------------------------------------
// Complex<f64> as used by algebra.rs (num-complex 0.4: naive mul/div, sqrt with
// the real and imaginary axes special-cased, polar form otherwise).
struct cplx {
    double re, im;
};
RT_DEV cplx mkc(double re, double im)
{
    cplx c;
    c.re = re;
    c.im = im;
    return c;
}
RT_DEV cplx operator+(cplx a, cplx b) { return mkc(a.re + b.re, a.im + b.im); }
RT_DEV cplx operator-(cplx a, cplx b) { return mkc(a.re - b.re, a.im - b.im); }
RT_DEV cplx operator-(cplx a) { return mkc(-a.re, -a.im); }
RT_DEV cplx operator*(cplx a, cplx b) { return mkc(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
RT_DEV cplx operator*(double s, cplx a) { return mkc(s * a.re, s * a.im); }
RT_DEV cplx operator/(cplx a, double s) { return mkc(a.re / s, a.im / s); }
RT_DEV cplx operator/(cplx a, cplx b)
{
    double ns = b.re * b.re + b.im * b.im;
    return mkc((a.re * b.re + a.im * b.im) / ns, (a.im * b.re - a.re * b.im) / ns);
}
RT_DEV cplx csqrt_(cplx a)
{
    if (a.im == 0.0) {
        if (!signbit(a.re)) return mkc(sqrt(a.re), a.im);
        double im = sqrt(-a.re);
        return signbit(a.im) ? mkc(0.0, -im) : mkc(0.0, im);
    } else if (a.re == 0.0) {
        double x = sqrt(fabs(a.im) / 2.0);
        return signbit(a.im) ? mkc(x, -x) : mkc(x, x);
    }
    double r = hypot(a.re, a.im), th = atan2(a.im, a.re);
    double sr = sqrt(r), s, c;
    sincos(th / 2.0, &s, &c);
    return mkc(sr * c, sr * s);
}
#define RT_TOL 1.0e-8
RT_DEV bool cis_zero(cplx c) { return fabs(c.re) < RT_TOL && fabs(c.im) < RT_TOL; } // algebra.rs:26-28
// algebra.rs:34-38
RT_DEV cplx ccbrt(cplx c, int n)
{
    double rho = pow(hypot(c.re, c.im), 1.0 / 3.0);
    double theta = ((2.0 * RT_PI * (double)n) + atan2(c.im, c.re)) / 3.0;
    double s, co;
    sincos(theta, &s, &co);
    return mkc(rho * co, rho * s);
}

// algebra.rs:40-58
RT_TORUS_FN int dev_solve_quadratic(cplx a, cplx b, cplx c, cplx *r)
{
    if (cis_zero(a)) {
        if (cis_zero(b)) return 0;
        r[0] = (-c) / b;
        return 1;
    }
    cplx rad = b * b - 4.0 * (a * c);
    if (cis_zero(rad)) {
        r[0] = (-b) / (2.0 * a);
        return 1;
    }
    cplx s = csqrt_(rad), d = 2.0 * a;
    r[0] = (-b + s) / d;
    r[1] = (-b - s) / d;
    return 2;
}

// algebra.rs:60-86
RT_TORUS_FN int dev_solve_cubic(cplx a, cplx b, cplx c, cplx d, cplx *r)
{
    if (cis_zero(a)) return dev_solve_quadratic(b, c, d, r);
    b = b / a;
    c = c / a;
    d = d / a;
    cplx S = b / 3.0;
    cplx D = c / 3.0 - S * S;
    cplx E = S * S * S + (d - S * c) / 2.0;
    cplx Fr = csqrt_(E * E + D * D * D);
    cplx F = -Fr - E;
    if (cis_zero(F)) F = Fr - E;
    for (int i = 0; i < 3; ++i) {
        cplx G = ccbrt(F, i);
        r[i] = G - D / G - S;
    }
    return 3;
}

// algebra.rs:88-156: real coefficients in, real roots (|im| < 1e-8) out, unsorted
RT_TORUS_FN int dev_solve_quartic_f64(double ar, double br, double cr, double dr, double er, double *roots)
{
    cplx r[4];
    int n;
    cplx a = mkc(ar, 0.0), b = mkc(br, 0.0), c = mkc(cr, 0.0), d = mkc(dr, 0.0), e = mkc(er, 0.0);
    if (cis_zero(a)) {
        n = dev_solve_cubic(b, c, d, e, r);
    } else {
        b = b / a;
        c = c / a;
        d = d / a;
        e = e / a;
        cplx b2 = b * b, b3 = b * b2, b4 = b * b3;
        cplx alpha = (-3.0 / 8.0) * b2 + c;
        cplx beta = b3 / 8.0 - (b * c) / 2.0 + d;
        cplx gamma = (-3.0 / 256.0) * b4 + (b2 * c) / 16.0 - (b * d) / 4.0 + e;
        cplx alpha2 = alpha * alpha;
        cplx t = (-b) / 4.0;
        if (cis_zero(beta)) {
            cplx rad = csqrt_(alpha2 - 4.0 * gamma);
            cplx r1 = csqrt_((-alpha + rad) / 2.0);
            cplx r2 = csqrt_((-alpha - rad) / 2.0);
            r[0] = t + r1;
            r[1] = t - r1;
            r[2] = t + r2;
            r[3] = t - r2;
        } else {
            cplx alpha3 = alpha * alpha2;
            cplx P = -(alpha2 / 12.0 + gamma);
            cplx Q = (-alpha3) / 108.0 + (alpha * gamma) / 3.0 - (beta * beta) / 8.0;
            cplx R = (-Q) / 2.0 + csqrt_((Q * Q) / 4.0 + (P * P * P) / 27.0);
            cplx U = ccbrt(R, 0);
            cplx y = (-5.0 / 6.0) * alpha + U;
            if (cis_zero(U))
                y = y - ccbrt(Q, 0);
            else
                y = y - P / (3.0 * U);
            cplx W = csqrt_(alpha + 2.0 * y);
            cplx bw = (2.0 * beta) / W;
            cplx r1 = csqrt_(-(3.0 * alpha + 2.0 * y + bw));
            cplx r2 = csqrt_(-(3.0 * alpha + 2.0 * y - bw));
            r[0] = t + (W - r1) / 2.0;
            r[1] = t + (W + r1) / 2.0;
            r[2] = t + (-W - r2) / 2.0;
            r[3] = t + (-W + r2) / 2.0;
        }
        n = 4;
    }
    int k = 0;
    for (int i = 0; i < 4; ++i)
        if (i < n && fabs(r[i].im) < RT_TOL) roots[k++] = r[i].re;
    return k;
}
