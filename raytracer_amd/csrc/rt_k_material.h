// rt_k_material.h — Material::scatter / emit, Texture::color_at_uv and the path bookkeeping of one segment
// (device code of librt_amd.so; included by rt_k_intersect.h and rt_k_shade.h)
#pragma once

#include "rt_k_common.h"

// Nearest-hit record of one ray as the kernels pass it on
struct HitRec {
    double t; // -1: nothing hit
    d3 n;
    double u, v;
    int obj;
};

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

