// oracle.cpp — CPU restatement of the reference renderer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (raytracer_amd/, librt_amd.so)
// links, loads or calls this file; it is used by tests/, by
// __graft_entry__.smoke() as the checker, and by bench.py's cpu_baseline leg.
//
// It follows the reference (doxxx/raytracer @ 2025-03-14, Rust) function by
// function, including its cost drivers (linear object scan, per-hit 4x4
// inverse, heap vectors of intervals, brute-force mesh + sort) and its quirks
// (SURVEY.md Q1-Q8).  Every function cites the reference lines it restates.
// Where the reference panics this code throws `Panic`, the
// object is treated as missed for that ray and the event is counted.
//
// The only deliberate departure: `rand::rng()` (ThreadRng, OS-seeded) is
// replaced by the counter-based generator of include/rt_rng.h, keyed by
// (seed, pixel, stratum, ray depth, draw site), so that a GPU that evaluates
// the draws in another order sees the same numbers.
//
// Parity pin: the reference's own unit tests (45, all inline) are transcribed
// as data in tests/golden/reference_unit_vectors.json and checked against this
// file by tests/test_oracle_golden.py.  The reference cannot be built here (no
// Rust toolchain), so paths those tests do not touch (camera, trace/cast, CSG,
// cylinder, mesh, medium, materials, textures) are pinned by code reading only.
// Third-party arithmetic not present in the reference tree is restated from
// the crates' published behaviour: num-complex 0.4.6 (Complex sqrt/div/mul/
// norm/arg), rand 0.9 (53-bit f64 mapping), wavefront_obj 5.1 (fan
// triangulation; done in the host loader).
#include "../include/rt_api.h"
#include "../include/rt_rng.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <thread>
#include <vector>

namespace oracle {

struct Panic {
    int which; // 0 cube, 1 torus, 2 csg difference, 3 cylinder, 4 NaN ordering
};

// ---------------------------------------------------------------- direction.rs / point.rs
struct Direction {
    double x, y, z;
};
struct Point {
    double x, y, z;
};
struct Vector2f {
    double u, v;
};

static inline Direction dir(double x, double y, double z) { return Direction{x, y, z}; }
static inline Point pt(double x, double y, double z) { return Point{x, y, z}; }

// direction.rs:41-47
static inline Direction cross(Direction a, Direction b)
{
    return dir(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// direction.rs:49-51 (powi(2) == x*x)
static inline double length_squared(Direction a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// direction.rs:53-55
static inline double length(Direction a) { return std::sqrt(length_squared(a)); }
static inline Direction operator*(Direction a, double s) { return dir(a.x * s, a.y * s, a.z * s); }
static inline Direction operator*(double s, Direction a) { return a * s; } // direction.rs:144-150
static inline Direction operator+(Direction a, Direction b) { return dir(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Direction operator-(Direction a, Direction b) { return dir(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Direction operator-(Direction a) { return dir(-a.x, -a.y, -a.z); }
// direction.rs:57-65
static inline Direction normalize(Direction a)
{
    double l2 = length_squared(a);
    if (l2 > 0.0) {
        double inv = 1.0 / std::sqrt(l2);
        return a * inv;
    }
    return a;
}
// direction.rs:84-94
static inline double dot(Direction a, Direction b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double dot(Direction a, Point b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// direction.rs:75-77: self - normal * 2.0 * self.dot(normal)
static inline Direction reflect(Direction d, Direction n) { return d - (n * 2.0) * dot(d, n); }
// direction.rs:184-190
static inline Direction recip(Direction a) { return dir(1.0 / a.x, 1.0 / a.y, 1.0 / a.z); }

// point.rs
static inline Point operator+(Point a, Direction b) { return pt(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Point operator-(Point a, Direction b) { return pt(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Direction operator-(Point a, Point b) { return dir(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Direction to_dir(Point a) { return dir(a.x, a.y, a.z); }

// f64::signum: +-1 with the sign bit, NaN stays NaN
static inline double signum(double v)
{
    if (v != v) return v;
    return std::signbit(v) ? -1.0 : 1.0;
}

// ---------------------------------------------------------------- matrix.rs
struct Matrix44f {
    double m[4][4];

    static Matrix44f zero()
    {
        Matrix44f r;
        std::memset(r.m, 0, sizeof(r.m));
        return r;
    }
    static Matrix44f identity()
    {
        Matrix44f r = zero();
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0;
        return r;
    }
    // matrix.rs:29-36
    static Matrix44f translation(Direction d)
    {
        Matrix44f r = identity();
        r.m[3][0] = d.x;
        r.m[3][1] = d.y;
        r.m[3][2] = d.z;
        return r;
    }
    // matrix.rs:38-45
    static Matrix44f scaling(Direction d)
    {
        Matrix44f r = identity();
        r.m[0][0] = d.x;
        r.m[1][1] = d.y;
        r.m[2][2] = d.z;
        return r;
    }
    static double to_radians(double deg) { return deg * (M_PI / 180.0); }
    // matrix.rs:47-55
    static Matrix44f rotation_x(double deg)
    {
        double a = to_radians(deg), s = std::sin(a), c = std::cos(a);
        Matrix44f r = identity();
        r.m[1][1] = c;
        r.m[1][2] = s;
        r.m[2][1] = -s;
        r.m[2][2] = c;
        return r;
    }
    // matrix.rs:57-65
    static Matrix44f rotation_y(double deg)
    {
        double a = to_radians(deg), s = std::sin(a), c = std::cos(a);
        Matrix44f r = identity();
        r.m[0][0] = c;
        r.m[0][2] = -s;
        r.m[2][0] = s;
        r.m[2][2] = c;
        return r;
    }
    // matrix.rs:67-75
    static Matrix44f rotation_z(double deg)
    {
        double a = to_radians(deg), s = std::sin(a), c = std::cos(a);
        Matrix44f r = identity();
        r.m[0][0] = c;
        r.m[0][1] = s;
        r.m[1][0] = -s;
        r.m[1][1] = c;
        return r;
    }

    // matrix.rs:81-156
    Matrix44f inverse() const
    {
        Matrix44f s = identity();
        Matrix44f t = *this;
        for (int i = 0; i < 3; ++i) {
            int pivot = i;
            double pivot_size = t.m[i][i];
            if (pivot_size < 0.0) pivot_size = -pivot_size;
            for (int j = i + 1; j < 4; ++j) {
                double tmp = t.m[j][i];
                if (tmp < 0.0) tmp = -tmp;
                if (tmp > pivot_size) {
                    pivot = j;
                    pivot_size = tmp;
                }
            }
            if (pivot_size == 0.0) return identity();
            if (pivot != i) {
                for (int j = 0; j < 4; ++j) {
                    double tmp = t.m[i][j];
                    t.m[i][j] = t.m[pivot][j];
                    t.m[pivot][j] = tmp;
                    tmp = s.m[i][j];
                    s.m[i][j] = s.m[pivot][j];
                    s.m[pivot][j] = tmp;
                }
            }
            for (int j = i + 1; j < 4; ++j) {
                double f = t.m[j][i] / t.m[i][i];
                for (int k = 0; k < 4; ++k) {
                    t.m[j][k] -= f * t.m[i][k];
                    s.m[j][k] -= f * s.m[i][k];
                }
            }
        }
        for (int i = 3; i >= 0; --i) {
            double f = t.m[i][i];
            if (f == 0.0) return identity();
            for (int j = 0; j < 4; ++j) {
                t.m[i][j] /= f;
                s.m[i][j] /= f;
            }
            for (int j = 0; j < i; ++j) {
                f = t.m[j][i];
                for (int k = 0; k < 4; ++k) {
                    t.m[j][k] -= f * t.m[i][k];
                    s.m[j][k] -= f * s.m[i][k];
                }
            }
        }
        return s;
    }

    // matrix.rs:158-166
    Matrix44f transpose() const
    {
        Matrix44f t = zero();
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) t.m[i][j] = m[j][i];
        return t;
    }
};

// matrix.rs:183-196
static inline Matrix44f operator*(const Matrix44f &a, const Matrix44f &b)
{
    Matrix44f r = Matrix44f::zero();
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}

// matrix.rs:198-213
static inline Point operator*(Point p, const Matrix44f &r)
{
    Point v = pt(p.x * r.m[0][0] + p.y * r.m[1][0] + p.z * r.m[2][0] + r.m[3][0],
                 p.x * r.m[0][1] + p.y * r.m[1][1] + p.z * r.m[2][1] + r.m[3][1],
                 p.x * r.m[0][2] + p.y * r.m[1][2] + p.z * r.m[2][2] + r.m[3][2]);
    double w = p.x * r.m[0][3] + p.y * r.m[1][3] + p.z * r.m[2][3] + r.m[3][3];
    if (w != 1.0 && w != 0.0) {
        v.x /= w;
        v.y /= w;
        v.z /= w;
    }
    return v;
}

// matrix.rs:215-225
static inline Direction operator*(Direction d, const Matrix44f &r)
{
    return dir(d.x * r.m[0][0] + d.y * r.m[1][0] + d.z * r.m[2][0], d.x * r.m[0][1] + d.y * r.m[1][1] + d.z * r.m[2][1],
               d.x * r.m[0][2] + d.y * r.m[1][2] + d.z * r.m[2][2]);
}

// ---------------------------------------------------------------- color.rs
struct Color {
    double r, g, b;
};
static inline Color operator+(Color a, Color b) { return Color{a.r + b.r, a.g + b.g, a.b + b.b}; }
static inline Color operator*(Color a, Color b) { return Color{a.r * b.r, a.g * b.g, a.b * b.b}; }
static inline Color operator*(Color a, double s) { return Color{a.r * s, a.g * s, a.b * s}; }
static inline Color operator*(double s, Color a) { return Color{s * a.r, s * a.g, s * a.b}; }
static inline Color operator/(Color a, double s) { return Color{a.r / s, a.g / s, a.b / s}; }

// ---------------------------------------------------------------- object.rs:7-26
struct Transformation {
    Matrix44f object_to_world = Matrix44f::identity();
    Matrix44f world_to_object = Matrix44f::identity();
    void transform(const Matrix44f &m)
    {
        object_to_world = object_to_world * m;
        world_to_object = object_to_world.inverse();
    }
};

// ---------------------------------------------------------------- random numbers
// Replaces rand::rng(); see the header comment and include/rt_rng.h.
struct RngContext {
    uint64_t seed = 0;
    uint32_t pixel = 0;
    uint32_t stratum = 0;
    uint32_t object = 0; // index of the object being intersected (fog draws)
};
static thread_local RngContext g_rng;
static thread_local uint64_t g_trace_calls = 0;
static thread_local uint32_t g_guards[5] = {0, 0, 0, 0, 0};

// ---------------------------------------------------------------- system.rs:75-152
struct Ray {
    Point origin;
    Direction direction;
    uint16_t depth;
    Direction inverse_direction;
    int sign[3];

    // system.rs:94-104
    static Ray make(Point origin, Direction direction, uint16_t depth)
    {
        Ray r;
        r.origin = origin;
        r.direction = direction;
        r.depth = depth;
        r.inverse_direction = recip(direction);
        r.set_sign();
        return r;
    }
    // direction.rs:67-73
    void set_sign()
    {
        sign[0] = inverse_direction.x < 0.0 ? 1 : 0;
        sign[1] = inverse_direction.y < 0.0 ? 1 : 0;
        sign[2] = inverse_direction.z < 0.0 ? 1 : 0;
    }
    // system.rs:145-152
    void transform(const Matrix44f &m)
    {
        origin = origin * m;
        direction = normalize(direction * m);
        inverse_direction = recip(direction);
        set_sign();
    }
    // system.rs:106-110
    Ray to_object(const Transformation &tx) const
    {
        Ray o = *this;
        o.transform(tx.world_to_object);
        return o;
    }
};

// system.rs:178-206
struct Intersection {
    double t;
    Direction n;
    Vector2f uv;

    Point point(const Ray &ray) const { return ray.origin + ray.direction * t; }

    // system.rs:190-199 — recomputes the inverse on every call, as there
    Intersection to_world(const Ray &world_ray, const Ray &object_ray, const Transformation &tx) const
    {
        Point object_hit_point = point(object_ray);
        Point world_hit_point = object_hit_point * tx.object_to_world;
        double tsign = signum(t);
        Intersection r;
        r.t = tsign * length(world_hit_point - world_ray.origin);
        r.n = normalize(n * tx.object_to_world.inverse().transpose());
        r.uv = uv;
        return r;
    }
};

// shapes/mod.rs:27-37
struct Interval {
    Intersection a, b;
    Interval to_world(const Ray &world_ray, const Ray &object_ray, const Transformation &tx) const
    {
        return Interval{a.to_world(world_ray, object_ray, tx), b.to_world(world_ray, object_ray, tx)};
    }
};

// `sort_by(|a, b| a.partial_cmp(b).unwrap())` on Vec<Interval>: derived
// lexicographic PartialOrd over (entry.t, exit.t); stable; NaN panics.
static void sort_intervals(std::vector<Interval> &v)
{
    for (const Interval &i : v)
        if (i.a.t != i.a.t || i.b.t != i.b.t) {
            if (v.size() > 1) throw Panic{4};
        }
    std::stable_sort(v.begin(), v.end(), [](const Interval &p, const Interval &q) {
        if (p.a.t < q.a.t) return true;
        if (p.a.t > q.a.t) return false;
        return p.b.t < q.b.t;
    });
}

// shapes/mod.rs:39-43
static std::vector<Interval> skip_negative_intervals(const std::vector<Interval> &in)
{
    size_t k = 0;
    while (k < in.size() && in[k].a.t < 0.0 && in[k].b.t < 0.0) ++k;
    return std::vector<Interval>(in.begin() + (long)k, in.end());
}

// shapes/mod.rs:45-51
static std::optional<Intersection> first_positive_intersection(const std::vector<Interval> &in)
{
    std::vector<Intersection> flat;
    for (const Interval &i : in) {
        flat.push_back(i.a);
        flat.push_back(i.b);
    }
    size_t k = 0;
    while (k < flat.size() && flat[k].t < 0.0) ++k;
    if (k < flat.size()) return flat[k];
    return std::nullopt;
}

// shapes/mod.rs:53-55
static std::optional<Intersection> first_intersection(const std::vector<Interval> &in)
{
    if (in.empty()) return std::nullopt;
    return in[0].a;
}

// shapes/mod.rs:57-60
struct Shape {
    virtual ~Shape() {}
    virtual std::optional<Intersection> intersect(const Ray &ray) const = 0;
    virtual std::vector<Interval> intersection_intervals(const Ray &ray) const = 0;
    virtual Transformation &transformation() = 0;
};

// `min_by(|a, b| a.partial_cmp(b).unwrap())`: first minimum wins, NaN panics
static std::optional<Intersection> min_by_t(const std::vector<Intersection> &v)
{
    if (v.empty()) return std::nullopt;
    size_t best = 0;
    for (size_t i = 1; i < v.size(); ++i) {
        if (v[i].t != v[i].t || v[best].t != v[best].t) throw Panic{4};
        if (v[i].t < v[best].t) best = i;
    }
    return v[best];
}

// ---------------------------------------------------------------- shapes/sphere.rs
// sphere.rs:41-56
static bool sphere_solve_quadratic(double a, double b, double c, double &x0, double &x1)
{
    double discr = b * b - 4.0 * a * c;
    if (discr < 0.0) {
        return false;
    } else if (discr == 0.0) {
        double x = -0.5 * b / a;
        x0 = x1 = x;
        return true;
    } else {
        double q = (b > 0.0) ? -0.5 * (b + std::sqrt(discr)) : -0.5 * (b - std::sqrt(discr));
        x0 = q / a;
        x1 = c / q;
        return true;
    }
}

struct Sphere : Shape {
    Point origin;
    double radius_squared;
    Transformation tx;
    Sphere(Point o, double radius) : origin(o), radius_squared(radius * radius) {}
    Transformation &transformation() override { return tx; }

    // sphere.rs:27-38
    Intersection intersection_for_t(const Ray &ray, double t) const
    {
        Point p = ray.origin + ray.direction * t;
        Direction n = normalize(p - origin);
        double u = (1.0 - std::atan2(n.z, n.x) / M_PI) * 0.5;
        double v = std::acos(n.y) / M_PI;
        return Intersection{t, n, Vector2f{u, v}};
    }
    // sphere.rs:58-62
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        return first_positive_intersection(intersection_intervals(ray));
    }
    // sphere.rs:69-90
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        Ray object_ray = ray.to_object(tx);
        Direction l = object_ray.origin - origin;
        double a = dot(object_ray.direction, object_ray.direction);
        double b = 2.0 * dot(object_ray.direction, l);
        double c = dot(l, l) - radius_squared;
        double t0, t1;
        if (sphere_solve_quadratic(a, b, c, t0, t1)) {
            if (t0 > t1) std::swap(t0, t1);
            return {Interval{intersection_for_t(object_ray, t0).to_world(ray, object_ray, tx),
                             intersection_for_t(object_ray, t1).to_world(ray, object_ray, tx)}};
        }
        return {};
    }
};

// ---------------------------------------------------------------- shapes/plane.rs
// plane.rs:9-21
static void plane_uv(Direction n, Direction &u, Direction &v)
{
    u = cross(n, dir(1.0, 0.0, 0.0));
    if (length_squared(u) < 1e-6) u = cross(n, dir(0.0, 1.0, 0.0));
    if (length_squared(u) < 1e-6) u = cross(n, dir(0.0, 0.0, 1.0));
    u = normalize(u);
    v = cross(n, u);
}

// plane.rs:23-32
static bool plane_intersect(Point o, Direction n, const Ray &ray, double &t)
{
    double denom = dot(ray.direction, n);
    if (std::fabs(denom) > 1e-6) {
        Direction w = o - ray.origin;
        t = dot(w, n) / denom;
        return true;
    }
    return false;
}

enum BoundsKind { BOUNDS_NONE, BOUNDS_XY, BOUNDS_XZ, BOUNDS_ZY };

struct Plane : Shape {
    Point origin;
    Direction normal, reverse_normal;
    Direction uv0, uv1, ruv0, ruv1;
    Transformation tx;
    // rectangles (plane.rs:111-244) are a Plane plus an out_of_bounds closure
    BoundsKind bounds = BOUNDS_NONE;
    double lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;

    // plane.rs:43-56
    Plane(Point o, Direction n) : origin(o), normal(n)
    {
        reverse_normal = n * -1.0;
        plane_uv(normal, uv0, uv1);
        plane_uv(reverse_normal, ruv0, ruv1);
    }
    Transformation &transformation() override { return tx; }

    bool out_of_bounds(Point p) const
    {
        switch (bounds) {
        case BOUNDS_XY: return p.x < lo0 || p.x > hi0 || p.y < lo1 || p.y > hi1; // plane.rs:134-136
        case BOUNDS_XZ: return p.x < lo0 || p.x > hi0 || p.z < lo1 || p.z > hi1; // plane.rs:179-181
        case BOUNDS_ZY: return p.z < lo0 || p.z > hi0 || p.y < lo1 || p.y > hi1; // plane.rs:224-226
        default: return false;
        }
    }

    // plane.rs:58-83
    std::optional<Intersection> intersect_with_bounds(const Ray &ray) const
    {
        Ray object_ray = ray.to_object(tx);
        Direction n = normal;
        Direction u0 = uv0, u1 = uv1;
        double t;
        if (plane_intersect(origin, n, object_ray, t)) {
            if (dot(object_ray.direction, normal) > 0.0) {
                n = reverse_normal;
                u0 = ruv0;
                u1 = ruv1;
            }
            Point p = object_ray.origin + object_ray.direction * t;
            if (out_of_bounds(p)) return std::nullopt;
            Direction op = p - origin;
            Intersection i{t, n, Vector2f{dot(u0, op), dot(u1, op)}};
            return i.to_world(ray, object_ray, tx);
        }
        return std::nullopt;
    }
    // plane.rs:95-99 / 139-143 / 184-188 / 229-233
    std::optional<Intersection> intersect(const Ray &ray) const override { return intersect_with_bounds(ray); }
    // plane.rs:85-92
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        std::optional<Intersection> i = intersect_with_bounds(ray);
        if (i) return {Interval{*i, *i}};
        return {};
    }
};

// plane.rs:119-132
static std::unique_ptr<Plane> new_xyrect(Point origin, double width, double height, bool reverse_normal)
{
    Direction normal = dir(0.0, 0.0, 1.0);
    if (reverse_normal) normal = normal * -1.0;
    auto p = std::make_unique<Plane>(origin, normal);
    p->bounds = BOUNDS_XY;
    p->lo0 = origin.x - width / 2.0;
    p->hi0 = origin.x + width / 2.0;
    p->lo1 = origin.y - height / 2.0;
    p->hi1 = origin.y + height / 2.0;
    return p;
}
// plane.rs:164-177
static std::unique_ptr<Plane> new_xzrect(Point origin, double width, double height, bool reverse_normal)
{
    Direction normal = dir(0.0, 1.0, 0.0);
    if (reverse_normal) normal = normal * -1.0;
    auto p = std::make_unique<Plane>(origin, normal);
    p->bounds = BOUNDS_XZ;
    p->lo0 = origin.x - (width / 2.0);
    p->hi0 = origin.x + width / 2.0;
    p->lo1 = origin.z - (height / 2.0);
    p->hi1 = origin.z + height / 2.0;
    return p;
}
// plane.rs:209-222
static std::unique_ptr<Plane> new_zyrect(Point origin, double width, double height, bool reverse_normal)
{
    Direction normal = dir(1.0, 0.0, 0.0);
    if (reverse_normal) normal = normal * -1.0;
    auto p = std::make_unique<Plane>(origin, normal);
    p->bounds = BOUNDS_ZY;
    p->lo0 = origin.z - (width / 2.0);
    p->hi0 = origin.z + width / 2.0;
    p->lo1 = origin.y - (height / 2.0);
    p->hi1 = origin.y + height / 2.0;
    return p;
}

// ---------------------------------------------------------------- shapes/cube.rs
struct Cube : Shape {
    std::unique_ptr<Plane> min_x, max_x, min_y, max_y, min_z, max_z;
    Transformation tx;
    Transformation &transformation() override { return tx; }

    // cube.rs:40-65
    static std::unique_ptr<Plane> xyrect(double x0, double y0, double x1, double y1, double z, bool rev)
    {
        return new_xyrect(pt((x1 - x0) / 2.0 + x0, (y1 - y0) / 2.0 + y0, z), x1 - x0, y1 - y0, rev);
    }
    static std::unique_ptr<Plane> xzrect(double x0, double z0, double x1, double z1, double y, bool rev)
    {
        return new_xzrect(pt((x1 - x0) / 2.0 + x0, y, (z1 - z0) / 2.0 + z0), x1 - x0, z1 - z0, rev);
    }
    static std::unique_ptr<Plane> zyrect(double z0, double y0, double z1, double y1, double x, bool rev)
    {
        return new_zyrect(pt(x, (y1 - y0) / 2.0 + y0, (z1 - z0) / 2.0 + z0), z1 - z0, y1 - y0, rev);
    }
    // cube.rs:21-37
    Cube(Point p1, Point p2)
    {
        double mnx = std::fmin(p1.x, p2.x), mny = std::fmin(p1.y, p2.y), mnz = std::fmin(p1.z, p2.z);
        double mxx = std::fmax(p1.x, p2.x), mxy = std::fmax(p1.y, p2.y), mxz = std::fmax(p1.z, p2.z);
        min_x = zyrect(mnz, mny, mxz, mxy, mnx, true);
        max_x = zyrect(mnz, mny, mxz, mxy, mxx, false);
        min_y = xzrect(mnx, mnz, mxx, mxz, mny, true);
        max_y = xzrect(mnx, mnz, mxx, mxz, mxy, false);
        min_z = xyrect(mnx, mny, mxx, mxy, mnz, true);
        max_z = xyrect(mnx, mny, mxx, mxy, mxz, false);
    }
    // cube.rs:67-71
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        return first_intersection(intersection_intervals(ray));
    }
    // cube.rs:78-106
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        Ray object_ray = ray.to_object(tx);
        std::optional<Intersection> is6[6] = {min_x->intersect(object_ray), max_x->intersect(object_ray),
                                              min_y->intersect(object_ray), max_y->intersect(object_ray),
                                              min_z->intersect(object_ray), max_z->intersect(object_ray)};
        std::vector<Intersection> is;
        for (auto &i : is6)
            if (i) is.push_back(*i);
        if (is.size() > 2) {
            throw Panic{0};
        } else if (is.size() == 2) {
            Intersection a = is[0], b = is[1];
            if (a.t > b.t) std::swap(a, b);
            b.n = b.n * -1.0;
            return {Interval{a, b}.to_world(ray, object_ray, tx)};
        } else if (is.size() == 1) {
            Intersection i = is[0];
            return {Interval{i, i}.to_world(ray, object_ray, tx)};
        }
        return {};
    }
};

// ---------------------------------------------------------------- shapes/cylinder.rs
struct Cylinder : Shape {
    double radius, height;
    Transformation tx;
    Cylinder(double r, double h) : radius(r), height(h) {}
    Transformation &transformation() override { return tx; }

    // cylinder.rs:27-39
    Intersection side_intersection(Point o, Direction d, double t, double y) const
    {
        Point p = o + d * t;
        Direction n = normalize(dir(p.x, 0.0, p.z));
        double u = (1.0 - std::atan2(n.z, n.x) / M_PI) * 0.5;
        double max_y = height / 2.0;
        double v = 1.0 - (y + max_y) / height;
        return Intersection{t, n, Vector2f{u, v}};
    }
    // cylinder.rs:42-46
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        return first_positive_intersection(intersection_intervals(ray));
    }
    static Intersection cap_hit(const Plane &cap, const Ray &object_ray)
    {
        // `.intersection_intervals(&object_ray).pop().expect(..)`
        std::vector<Interval> v = cap.intersection_intervals(object_ray);
        if (v.empty()) throw Panic{3};
        return v.back().a;
    }
    // cylinder.rs:53-144
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        Ray object_ray = ray.to_object(tx);
        Point o = object_ray.origin;
        Direction d = object_ray.direction;
        double a = d.x * d.x + d.z * d.z;
        double b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
        double c = o.x * o.x + o.z * o.z - radius * radius;
        double discr = b * b - 4.0 * a * c;
        if (discr < 0.0) return {};

        double sqrt = std::sqrt(discr);
        double t0 = (-b + sqrt) / (2.0 * a);
        double t1 = (-b - sqrt) / (2.0 * a);
        if (t0 > t1) std::swap(t0, t1);

        double y0 = o.y + t0 * d.y;
        double y1 = o.y + t1 * d.y;

        double max_y = height / 2.0;
        double min_y = -max_y;
        Plane top_cap(pt(0.0, max_y, 0.0), dir(0.0, 1.0, 0.0));
        Plane bottom_cap(pt(0.0, min_y, 0.0), dir(0.0, -1.0, 0.0));

        if ((y0 < min_y && y1 < min_y) || (y0 > max_y && y1 > max_y)) {
            return {};
        } else if (y0 > max_y) {
            Intersection top_i = cap_hit(top_cap, object_ray);
            if (y1 > min_y && y1 < max_y) {
                return {Interval{top_i, side_intersection(o, d, t1, y1)}.to_world(ray, object_ray, tx)};
            } else {
                if (!(y1 < min_y)) throw Panic{3};
                Intersection bottom_i = cap_hit(bottom_cap, object_ray);
                return {Interval{top_i, bottom_i}.to_world(ray, object_ray, tx)};
            }
        } else if (y0 < min_y) {
            Intersection bottom_i = cap_hit(bottom_cap, object_ray);
            if (y1 > min_y && y1 < max_y) {
                return {Interval{bottom_i, side_intersection(o, d, t1, y1)}.to_world(ray, object_ray, tx)};
            } else {
                if (!(y1 > max_y)) throw Panic{3};
                Intersection top_i = cap_hit(top_cap, object_ray);
                return {Interval{bottom_i, top_i}.to_world(ray, object_ray, tx)};
            }
        } else {
            Intersection front_i = side_intersection(o, d, t0, y0);
            if (y1 > min_y && y1 < max_y) {
                return {Interval{front_i, side_intersection(o, d, t1, y1)}.to_world(ray, object_ray, tx)};
            } else if (y1 < min_y) {
                Intersection bottom_i = cap_hit(bottom_cap, object_ray);
                return {Interval{front_i, bottom_i}.to_world(ray, object_ray, tx)};
            } else {
                if (!(y1 > max_y)) throw Panic{3};
                Intersection top_i = cap_hit(top_cap, object_ray);
                return {Interval{front_i, top_i}.to_world(ray, object_ray, tx)};
            }
        }
    }
};

// ---------------------------------------------------------------- algebra.rs
// num_complex::Complex<f64> (0.4.6), restated from the crate's documented
// formulas (crate source not present here).
struct Complex {
    double re, im;
};
static inline Complex cx(double re) { return Complex{re, 0.0}; }
static inline Complex operator+(Complex a, Complex b) { return Complex{a.re + b.re, a.im + b.im}; }
static inline Complex operator-(Complex a, Complex b) { return Complex{a.re - b.re, a.im - b.im}; }
static inline Complex operator-(Complex a) { return Complex{-a.re, -a.im}; }
static inline Complex operator*(Complex a, Complex b)
{
    return Complex{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
static inline Complex operator/(Complex a, Complex b)
{
    double norm_sqr = b.re * b.re + b.im * b.im;
    double re = a.re * b.re + a.im * b.im;
    double im = a.im * b.re - a.re * b.im;
    return Complex{re / norm_sqr, im / norm_sqr};
}
static inline Complex operator*(double s, Complex a) { return Complex{s * a.re, s * a.im}; }
static inline Complex operator*(Complex a, double s) { return Complex{a.re * s, a.im * s}; }
static inline Complex operator/(Complex a, double s) { return Complex{a.re / s, a.im / s}; }
static inline double cnorm(Complex a) { return std::hypot(a.re, a.im); }
static inline double carg(Complex a) { return std::atan2(a.im, a.re); }
// Complex::sqrt: real and imaginary axes special-cased, otherwise polar form
static Complex csqrt(Complex a)
{
    if (a.im == 0.0) {
        if (!std::signbit(a.re)) {
            return Complex{std::sqrt(a.re), a.im};
        } else {
            double im = std::sqrt(-a.re);
            return std::signbit(a.im) ? Complex{0.0, -im} : Complex{0.0, im};
        }
    } else if (a.re == 0.0) {
        double x = std::sqrt(std::fabs(a.im) / 2.0);
        return std::signbit(a.im) ? Complex{x, -x} : Complex{x, x};
    } else {
        double r = cnorm(a), theta = carg(a);
        double sr = std::sqrt(r), ht = theta / 2.0;
        return Complex{sr * std::cos(ht), sr * std::sin(ht)};
    }
}

static const double TOLERANCE = 1.0e-8;
static const double TWO_PI = 2.0 * M_PI;

// algebra.rs:26-28
static inline bool is_zero(Complex c) { return std::fabs(c.re) < TOLERANCE && std::fabs(c.im) < TOLERANCE; }

// algebra.rs:34-38
static Complex cbrt_n(Complex c, int n)
{
    double rho = std::pow(cnorm(c), 1.0 / 3.0);
    double theta = ((TWO_PI * (double)n) + carg(c)) / 3.0;
    return Complex{rho * std::cos(theta), rho * std::sin(theta)};
}

// algebra.rs:40-58
static std::vector<Complex> solve_quadratic(Complex a, Complex b, Complex c)
{
    if (is_zero(a)) {
        if (is_zero(b)) return {};
        return {-c / b};
    }
    Complex radicand = b * b - 4.0 * a * c;
    if (is_zero(radicand)) return {-b / (2.0 * a)};
    Complex r = csqrt(radicand);
    Complex d = 2.0 * a;
    return {(-b + r) / d, (-b - r) / d};
}

// algebra.rs:60-86
static std::vector<Complex> solve_cubic(Complex a, Complex b, Complex c, Complex d)
{
    if (is_zero(a)) return solve_quadratic(b, c, d);
    b = b / a;
    c = c / a;
    d = d / a;
    Complex S = b / 3.0;
    Complex D = c / 3.0 - S * S;
    Complex E = S * S * S + (d - S * c) / 2.0;
    Complex F_root = csqrt(E * E + D * D * D);
    Complex F = -F_root - E;
    if (is_zero(F)) F = F_root - E;
    std::vector<Complex> out;
    for (int i = 0; i < 3; ++i) {
        Complex G = cbrt_n(F, i);
        out.push_back(G - D / G - S);
    }
    return out;
}

// algebra.rs:88-146
static std::vector<Complex> solve_quartic(Complex a, Complex b, Complex c, Complex d, Complex e)
{
    if (is_zero(a)) return solve_cubic(b, c, d, e);
    b = b / a;
    c = c / a;
    d = d / a;
    e = e / a;

    Complex b2 = b * b;
    Complex b3 = b * b2;
    Complex b4 = b * b3;

    Complex alpha = (-3.0 / 8.0) * b2 + c;
    Complex beta = b3 / 8.0 - b * c / 2.0 + d;
    Complex gamma = (-3.0 / 256.0) * b4 + b2 * c / 16.0 - b * d / 4.0 + e;

    Complex alpha2 = alpha * alpha;
    Complex t = -b / 4.0;

    if (is_zero(beta)) {
        Complex rad = csqrt(alpha2 - 4.0 * gamma);
        Complex r1 = csqrt((-alpha + rad) / 2.0);
        Complex r2 = csqrt((-alpha - rad) / 2.0);
        return {t + r1, t - r1, t + r2, t - r2};
    }
    Complex alpha3 = alpha * alpha2;
    Complex P = -(alpha2 / 12.0 + gamma);
    Complex Q = -alpha3 / 108.0 + alpha * gamma / 3.0 - beta * beta / 8.0;
    Complex R = -Q / 2.0 + csqrt(Q * Q / 4.0 + P * P * P / 27.0);
    Complex U = cbrt_n(R, 0);
    Complex y = (-5.0 / 6.0) * alpha + U;
    if (is_zero(U)) {
        y = y - cbrt_n(Q, 0);
    } else {
        y = y - P / (3.0 * U);
    }
    Complex W = csqrt(alpha + 2.0 * y);
    Complex r1 = csqrt(-(3.0 * alpha + 2.0 * y + 2.0 * beta / W));
    Complex r2 = csqrt(-(3.0 * alpha + 2.0 * y - 2.0 * beta / W));
    return {t + (W - r1) / 2.0, t + (W + r1) / 2.0, t + (-W - r2) / 2.0, t + (-W + r2) / 2.0};
}

// algebra.rs:30-32,148-156
static std::vector<double> solve_quartic_f64(double a, double b, double c, double d, double e)
{
    std::vector<double> out;
    for (Complex r : solve_quartic(cx(a), cx(b), cx(c), cx(d), cx(e)))
        if (std::fabs(r.im) < TOLERANCE) out.push_back(r.re);
    return out;
}

// ---------------------------------------------------------------- shapes/torus.rs
struct Torus : Shape {
    double radius1, radius2;
    Transformation tx;
    Torus(double r1, double r2) : radius1(r1), radius2(r2) {}
    Transformation &transformation() override { return tx; }

    // torus.rs:28-39
    Intersection intersection_for(const Ray &ray, double t) const
    {
        Point p = ray.origin + ray.direction * t;
        double a = 1.0 - (radius1 / std::sqrt(p.x * p.x + p.y * p.y));
        Direction n = normalize(dir(a * p.x, a * p.y, p.z));
        return Intersection{t, n, Vector2f{0.0, 0.0}};
    }
    // torus.rs:41-45
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        return first_positive_intersection(intersection_intervals(ray));
    }
    // torus.rs:52-125
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        Ray object_ray = ray.to_object(tx);
        Point o = object_ray.origin;
        Direction d = object_ray.direction;

        double R = radius1, S = radius2;
        double T = 4.0 * R * R;
        double G = T * (d.x * d.x + d.y * d.y);
        double H = 2.0 * T * (o.x * d.x + o.y * d.y);
        double I = T * (o.x * o.x + o.y * o.y);
        double J = length_squared(d);
        double K = 2.0 * dot(to_dir(o), d);
        double L = length_squared(to_dir(o)) + R * R - S * S;

        std::vector<double> roots =
            solve_quartic_f64(J * J, 2.0 * J * K, 2.0 * J * L + K * K - G, 2.0 * K * L - H, L * L - I);
        for (double r : roots)
            if (r != r && roots.size() > 1) throw Panic{4};
        std::stable_sort(roots.begin(), roots.end());

        std::vector<Interval> is;
        if (roots.size() == 0) {
        } else if (roots.size() == 1) {
            Intersection i = intersection_for(object_ray, roots[0]);
            is.push_back(Interval{i, i});
        } else if (roots.size() == 2) {
            is.push_back(Interval{intersection_for(object_ray, roots[0]), intersection_for(object_ray, roots[1])});
        } else if (roots.size() == 3) {
            Intersection iv[3];
            long facing[3];
            for (int k = 0; k < 3; ++k) {
                iv[k] = intersection_for(object_ray, roots[k]);
                double s = signum(dot(iv[k].n, d));
                facing[k] = (s != s) ? 0 : (long)s; // `as isize`: NaN -> 0
            }
            if (facing[0] == -1 && facing[1] == 1 && facing[2] == -1) {
                is.push_back(Interval{iv[0], iv[1]});
                is.push_back(Interval{iv[2], iv[2]});
            } else if (facing[0] == 1 && facing[1] == -1 && facing[2] == 1) {
                is.push_back(Interval{iv[0], iv[0]});
                is.push_back(Interval{iv[1], iv[2]});
            } else {
                throw Panic{1};
            }
        } else if (roots.size() == 4) {
            is.push_back(Interval{intersection_for(object_ray, roots[0]), intersection_for(object_ray, roots[1])});
            is.push_back(Interval{intersection_for(object_ray, roots[2]), intersection_for(object_ray, roots[3])});
        } else {
            throw Panic{1};
        }
        std::vector<Interval> out;
        for (const Interval &i : is) out.push_back(i.to_world(ray, object_ray, tx));
        return out;
    }
};

// ---------------------------------------------------------------- shapes/csg.rs
struct CSG : Shape {
    int op; // RT_SHAPE_UNION / INTERSECTION / DIFFERENCE
    std::unique_ptr<Shape> a, b;
    Transformation tx;
    Transformation &transformation() override { return tx; }

    // csg.rs:101-109,179-187,283-290
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        return first_positive_intersection(intersection_intervals(ray));
    }

    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        switch (op) {
        case RT_SHAPE_UNION: return union_intervals(ray);
        case RT_SHAPE_INTERSECTION: return intersection_op_intervals(ray);
        default: return difference_intervals(ray);
        }
    }

    std::vector<Interval> finish(std::vector<Interval> &intervals, const Ray &ray, const Ray &object_ray) const
    {
        sort_intervals(intervals);
        std::vector<Interval> out;
        for (const Interval &i : intervals) out.push_back(i.to_world(ray, object_ray, tx));
        return out;
    }

    // csg.rs:28-98
    std::vector<Interval> union_intervals(const Ray &ray) const
    {
        Ray object_ray = ray.to_object(tx);
        std::vector<Interval> ia = a->intersection_intervals(object_ray);
        std::vector<Interval> ib = b->intersection_intervals(object_ray);
        if (ia.size() == 0) return ib;
        if (ib.size() == 0) return ia;

        std::vector<Interval> intervals;
        size_t pa = 0, pb = 0;
        auto next = [](const std::vector<Interval> &v, size_t &p) -> std::optional<Interval> {
            if (p < v.size()) return v[p++];
            return std::nullopt;
        };
        std::optional<Interval> interval_a = next(ia, pa);
        std::optional<Interval> interval_b = next(ib, pb);

        while (interval_a && interval_b) {
            Intersection a_start = interval_a->a, a_end = interval_a->b;
            Intersection b_start = interval_b->a, b_end = interval_b->b;
            if (a_end.t < b_start.t) {
                intervals.push_back(*interval_a);
                interval_a = next(ia, pa);
            } else if (b_end.t < a_start.t) {
                intervals.push_back(*interval_b);
                interval_b = next(ib, pb);
            } else {
                Intersection new_start = a_start;
                Intersection new_end = a_end;
                if (b_start.t < new_start.t) new_start = b_start;
                if (b_end.t > new_end.t) {
                    new_end = b_end;
                    interval_a = next(ia, pa);
                    while (interval_a) {
                        if (interval_a->a.t > new_end.t) break;
                        new_end = interval_a->b;
                        interval_a = next(ia, pa);
                    }
                }
                intervals.push_back(Interval{new_start, new_end});
                interval_b = next(ib, pb);
            }
        }
        if (interval_a) {
            intervals.push_back(*interval_a);
        } else if (interval_b) {
            intervals.push_back(*interval_b);
        }
        while (pa < ia.size()) intervals.push_back(ia[pa++]);
        while (pb < ib.size()) intervals.push_back(ib[pb++]);
        return finish(intervals, ray, object_ray);
    }

    // csg.rs:133-176
    std::vector<Interval> intersection_op_intervals(const Ray &ray) const
    {
        Ray object_ray = ray.to_object(tx);
        std::vector<Interval> ia = a->intersection_intervals(object_ray);
        std::vector<Interval> ib = b->intersection_intervals(object_ray);
        if (ia.size() == 0 || ib.size() == 0) return {};

        std::vector<Interval> intervals;
        size_t pa = 0, pb = 0;
        auto next = [](const std::vector<Interval> &v, size_t &p) -> std::optional<Interval> {
            if (p < v.size()) return v[p++];
            return std::nullopt;
        };
        std::optional<Interval> interval_a = next(ia, pa);
        std::optional<Interval> interval_b = next(ib, pb);
        while (interval_a && interval_b) {
            Intersection a_start = interval_a->a, a_end = interval_a->b;
            Intersection b_start = interval_b->a, b_end = interval_b->b;
            if (a_end.t < b_start.t) {
                interval_a = next(ia, pa);
            } else if (b_end.t < a_start.t) {
                interval_b = next(ib, pb);
            } else {
                intervals.push_back(Interval{(a_start.t > b_start.t) ? a_start : b_start,
                                             (a_end.t < b_end.t) ? a_end : b_end});
                if (b_end.t >= a_end.t) {
                    interval_a = next(ia, pa);
                } else {
                    interval_b = next(ib, pb);
                }
            }
        }
        return finish(intervals, ray, object_ray);
    }

    // csg.rs:211-280
    std::vector<Interval> difference_intervals(const Ray &ray) const
    {
        Ray object_ray = ray.to_object(tx);
        std::vector<Interval> ia = a->intersection_intervals(object_ray);
        std::vector<Interval> ib = b->intersection_intervals(object_ray);
        if (ia.size() == 0) return {};
        if (ib.size() == 0) return ia;

        std::vector<Interval> intervals;
        size_t pa = 0, pb = 0;
        auto next = [](const std::vector<Interval> &v, size_t &p) -> std::optional<Interval> {
            if (p < v.size()) return v[p++];
            return std::nullopt;
        };
        std::optional<Interval> interval_a = next(ia, pa);
        std::optional<Interval> interval_b = next(ib, pb);
        while (interval_a && interval_b) {
            Intersection a_start = interval_a->a, a_end = interval_a->b;
            Intersection b_start = interval_b->a, b_end = interval_b->b;
            if (a_end.t < b_start.t) {
                intervals.push_back(*interval_a);
                interval_a = next(ia, pa);
            } else if (b_end.t < a_start.t) {
                interval_b = next(ib, pb);
            } else if (b_start.t <= a_start.t && b_end.t >= a_end.t) {
                interval_a = next(ia, pa);
            } else if (b_start.t <= a_start.t && b_end.t < a_end.t) {
                Intersection new_start = b_end;
                new_start.n = new_start.n * -1.0;
                interval_a = Interval{new_start, a_end};
                interval_b = next(ib, pb);
            } else if (b_start.t < a_end.t && b_end.t < a_end.t) {
                Intersection a_1_end = b_start;
                a_1_end.n = a_1_end.n * -1.0;
                intervals.push_back(Interval{a_start, a_1_end});
                Intersection a_2_start = b_end;
                a_2_start.n = a_2_start.n * -1.0;
                interval_a = Interval{a_2_start, a_end};
                interval_b = next(ib, pb);
            } else if (b_start.t < a_end.t && b_end.t >= a_end.t) {
                Intersection new_end = b_start;
                new_end.n = new_end.n * -1.0;
                intervals.push_back(Interval{a_start, new_end});
                interval_a = next(ia, pa);
            } else {
                throw Panic{2};
            }
        }
        if (interval_a) intervals.push_back(*interval_a);
        while (pa < ia.size()) intervals.push_back(ia[pa++]);
        return finish(intervals, ray, object_ray);
    }
};

// ---------------------------------------------------------------- shapes/bounding_box.rs
struct BoundingBox {
    Point bounds[2];
    // bounding_box.rs:13-44 (the z slab only rejects)
    bool intersect(const Ray &ray) const
    {
        double tmin = (bounds[ray.sign[0]].x - ray.origin.x) * ray.inverse_direction.x;
        double tmax = (bounds[1 - ray.sign[0]].x - ray.origin.x) * ray.inverse_direction.x;
        double tymin = (bounds[ray.sign[1]].y - ray.origin.y) * ray.inverse_direction.y;
        double tymax = (bounds[1 - ray.sign[1]].y - ray.origin.y) * ray.inverse_direction.y;
        if ((tmin > tymax) || (tymin > tmax)) return false;
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        double tzmin = (bounds[ray.sign[2]].z - ray.origin.z) * ray.inverse_direction.z;
        double tzmax = (bounds[1 - ray.sign[2]].z - ray.origin.z) * ray.inverse_direction.z;
        if ((tmin > tzmax) || (tzmin > tmax)) return false;
        return true;
    }
};

// ---------------------------------------------------------------- shapes/mesh.rs
struct MeshTriangle {
    size_t vertex_indices[3];
    size_t normal_indices[3];
};

struct Mesh : Shape {
    std::vector<Point> vertices;
    std::vector<Direction> normals;
    std::vector<MeshTriangle> triangles;
    BoundingBox bounding_box;
    bool smooth_shading;
    Transformation tx;
    Transformation &transformation() override { return tx; }

    // mesh.rs:26-52 — min/max start at the origin
    Mesh(std::vector<Point> v, std::vector<Direction> n, std::vector<MeshTriangle> t, bool smooth)
        : vertices(std::move(v)), normals(std::move(n)), triangles(std::move(t)), smooth_shading(smooth)
    {
        Point mn = pt(0, 0, 0), mx = pt(0, 0, 0);
        for (const Point &p : vertices) {
            mn.x = std::fmin(mn.x, p.x);
            mn.y = std::fmin(mn.y, p.y);
            mn.z = std::fmin(mn.z, p.z);
            mx.x = std::fmax(mx.x, p.x);
            mx.y = std::fmax(mx.y, p.y);
            mx.z = std::fmax(mx.z, p.z);
        }
        bounding_box.bounds[0] = mn;
        bounding_box.bounds[1] = mx;
    }

    // mesh.rs:66-113
    std::optional<Intersection> intersect_triangle(const Ray &ray, const MeshTriangle &tri) const
    {
        Point v0 = vertices[tri.vertex_indices[0]];
        Point v1 = vertices[tri.vertex_indices[1]];
        Point v2 = vertices[tri.vertex_indices[2]];
        Direction n0 = normals[tri.normal_indices[0]];
        Direction n1 = normals[tri.normal_indices[1]];
        Direction n2 = normals[tri.normal_indices[2]];

        Direction v0v1 = v1 - v0;
        Direction v0v2 = v2 - v0;
        Direction pvec = cross(ray.direction, v0v2);
        double det = dot(v0v1, pvec);
        if (det < std::numeric_limits<double>::epsilon()) return std::nullopt;
        double inv_det = 1.0 / det;
        Direction tvec = ray.origin - v0;
        double u = dot(tvec, pvec) * inv_det;
        if (u < 0.0 || u > 1.0) return std::nullopt;
        Direction qvec = cross(tvec, v0v1);
        double v = dot(ray.direction, qvec) * inv_det;
        if (v < 0.0 || (u + v) > 1.0) return std::nullopt;
        double t = dot(v0v2, qvec) * inv_det;
        if (t < 0.0) return std::nullopt;
        Direction n = smooth_shading ? normalize((1.0 - u - v) * n0 + u * n1 + v * n2) : normalize(n0 + n1 + n2);
        return Intersection{t, n, Vector2f{u, v}};
    }

    // mesh.rs:54-64
    std::vector<Intersection> intersect_triangles(const Ray &ray) const
    {
        std::vector<Intersection> is;
        for (const MeshTriangle &tri : triangles) {
            std::optional<Intersection> i = intersect_triangle(ray, tri);
            if (i) is.push_back(*i);
        }
        for (const Intersection &i : is)
            if (i.t != i.t && is.size() > 1) throw Panic{4};
        std::stable_sort(is.begin(), is.end(), [](const Intersection &p, const Intersection &q) { return p.t < q.t; });
        return is;
    }

    // mesh.rs:116-128
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        if (!bounding_box.intersect(ray)) return std::nullopt;
        Ray object_ray = ray.to_object(tx);
        std::vector<Intersection> is = intersect_triangles(object_ray);
        if (is.empty()) return std::nullopt;
        return is[0].to_world(ray, object_ray, tx);
    }
    // mesh.rs:135-149
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        std::optional<Intersection> i = intersect(ray);
        if (i) return {Interval{*i, *i}};
        return {};
    }
};

// ---------------------------------------------------------------- shapes/composite.rs
struct Composite : Shape {
    std::vector<std::unique_ptr<Shape>> shapes;
    Transformation tx;
    Transformation &transformation() override { return tx; }

    // composite.rs:20-34
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        if (shapes.size() == 0) return std::nullopt;
        Ray object_ray = ray.to_object(tx);
        std::vector<Intersection> hits;
        for (const auto &s : shapes) {
            std::optional<Intersection> i = s->intersect(object_ray);
            if (i) hits.push_back(*i);
        }
        std::optional<Intersection> best = min_by_t(hits);
        if (!best) return std::nullopt;
        return best->to_world(ray, object_ray, tx);
    }
    // composite.rs:41-50
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        Ray object_ray = ray.to_object(tx);
        std::vector<Interval> is;
        for (const auto &s : shapes)
            for (const Interval &i : s->intersection_intervals(object_ray)) is.push_back(i);
        sort_intervals(is);
        std::vector<Interval> out;
        for (const Interval &i : is) out.push_back(i.to_world(ray, object_ray, tx));
        return out;
    }
};

// ---------------------------------------------------------------- shapes/homogenous_medium.rs
struct HomogenousMedium : Shape {
    std::unique_ptr<Shape> boundary;
    double density;
    Transformation tx;
    Transformation &transformation() override { return tx; }

    // homogenous_medium.rs:29-57; one draw per interval examined, lazily
    std::optional<Intersection> intersect(const Ray &ray) const override
    {
        Ray object_ray = ray.to_object(tx);
        std::vector<Interval> is = intersection_intervals(object_ray);
        if (is.size() == 0) return std::nullopt;

        uint32_t k = 0;
        for (const Interval &iv : skip_negative_intervals(is)) {
            double at = std::fmax(iv.a.t, 0.0), bt = iv.b.t;
            double distance = length((bt - at) * object_ray.direction);
            rt_rng_block blk = rt_rng_draw(g_rng.seed, g_rng.pixel, g_rng.stratum, ray.depth, RT_RNG_SITE_MEDIUM + k,
                                           g_rng.object);
            ++k;
            double hit_distance = -(1.0 / density) * std::log(rt_rng_u0(&blk));
            if (hit_distance < distance) {
                Intersection i{at + hit_distance / length(object_ray.direction), dir(1.0, 0.0, 0.0),
                               Vector2f{0.0, 0.0}};
                return i.to_world(ray, object_ray, tx);
            }
        }
        return std::nullopt;
    }
    // homogenous_medium.rs:64-66
    std::vector<Interval> intersection_intervals(const Ray &ray) const override
    {
        return boundary->intersection_intervals(ray);
    }
};

// ---------------------------------------------------------------- texture.rs
struct Texture {
    int kind;
    Color c1, c2;
    double scale;
    uint32_t width = 0, height = 0;
    const uint8_t *pixels = nullptr;

    static uint32_t as_u32(double v)
    {
        if (!(v > 0.0)) return 0; // NaN and negatives saturate to 0
        if (v >= 4294967295.0) return 4294967295u;
        return (uint32_t)v;
    }
    // texture.rs:54-94
    Color color_at_uv(Vector2f uv) const
    {
        if (kind == RT_TEX_SOLID) return c1;
        if (kind == RT_TEX_CHECKERBOARD) {
            double sx = uv.u * scale, sy = uv.v * scale;
            double scaled_x_frac = sx - std::trunc(sx);
            double scaled_y_frac = sy - std::trunc(sy);
            bool x_pattern = (std::fabs(scaled_x_frac) > 0.5) ^ (scaled_x_frac < 0.0);
            bool y_pattern = (std::fabs(scaled_y_frac) > 0.5) ^ (scaled_y_frac < 0.0);
            double pattern = (x_pattern ^ y_pattern) ? 1.0 : 0.0;
            return c1 * (1.0 - pattern) + c2 * pattern; // mix, texture.rs:92-94
        }
        double max_x = (double)(width - 1);
        double max_y = (double)(height - 1);
        uint32_t x = as_u32(uv.u * scale * max_x) % width;
        uint32_t y = as_u32(uv.v * scale * max_y) % height;
        const uint8_t *c = pixels + ((size_t)y * width + x) * 3;
        return Color{(double)c[0] / 255.0, (double)c[1] / 255.0, (double)c[2] / 255.0};
    }
};

// ---------------------------------------------------------------- materials/*.rs
struct RayHit {
    const Ray *incident;
    int object;
    double t;
    Direction n;
    Vector2f uv;
    // system.rs:173-175
    Point point() const { return incident->origin + incident->direction * t; }
};

struct ScatteredRay {
    Color attenuation;
    Point origin;
    Direction direction;
};

// direction.rs:29-39 with the two draws taken from the counter-based block
static Direction uniform_sphere_distribution(uint32_t depth)
{
    rt_rng_block blk = rt_rng_draw(g_rng.seed, g_rng.pixel, g_rng.stratum, depth, RT_RNG_SITE_SPHERE, 0);
    double theta = 2.0 * M_PI * rt_rng_u0(&blk);
    double phi = std::acos(1.0 - 2.0 * rt_rng_u1(&blk));
    double x = std::sin(phi) * std::cos(theta);
    double y = std::sin(phi) * std::sin(theta);
    double z = std::cos(phi);
    return dir(x, y, z);
}

// dielectric.rs:64-66
static inline double clamp(double lo, double hi, double val) { return std::fmax(lo, std::fmin(hi, val)); }

// dielectric.rs:68-86
static Direction refract(Direction incident, Direction normal, double ior)
{
    double cos_i = clamp(-1.0, 1.0, dot(incident, normal));
    double eta_i = 1.0, eta_t = ior;
    Direction n = normal;
    if (cos_i < 0.0) {
        cos_i = -cos_i;
    } else {
        std::swap(eta_i, eta_t);
        n = -normal;
    }
    double eta = eta_i / eta_t;
    double k = 1.0 - eta * eta * (1.0 - cos_i * cos_i);
    if (k < 0.0) return dir(0.0, 0.0, 0.0);
    return incident * eta + n * (eta * cos_i - std::sqrt(k));
}

// dielectric.rs:89-108
static double fresnel(Direction incident, Direction normal, double ior)
{
    double cos_i = clamp(-1.0, 1.0, dot(incident, normal));
    double eta_i = 1.0, eta_t = ior;
    if (cos_i > 0.0) std::swap(eta_i, eta_t);
    double sin_t = eta_i / eta_t * std::sqrt(std::fmax(1.0 - cos_i * cos_i, 0.0));
    if (sin_t >= 1.0) return 1.0;
    double cos_t = std::sqrt(std::fmax(1.0 - sin_t * sin_t, 0.0));
    cos_i = std::fabs(cos_i);
    double r_s = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
    double r_p = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
    return (r_s * r_s + r_p * r_p) / 2.0;
}

struct Material {
    int kind;
    double fuzz, ior, intensity;
    const Texture *texture;

    std::optional<ScatteredRay> scatter(double bias, const RayHit &hit) const
    {
        uint32_t depth = hit.incident->depth;
        switch (kind) {
        case RT_MAT_LAMBERTIAN: {
            // lambertian.rs:20-31
            Point p = hit.point();
            Point scattered_origin = p + hit.n * bias;
            Point target = p + hit.n + uniform_sphere_distribution(depth);
            Direction scattered_dir = normalize(target - p);
            return ScatteredRay{texture->color_at_uv(hit.uv), scattered_origin, scattered_dir};
        }
        case RT_MAT_METAL: {
            // metal.rs:21-32
            Direction reflected = normalize(reflect(hit.incident->direction, hit.n));
            Direction fz = fuzz * uniform_sphere_distribution(depth);
            Point scattered_origin = hit.point() + hit.n * bias;
            Direction scattered_dir = normalize(reflected + fz);
            return ScatteredRay{texture->color_at_uv(hit.uv), scattered_origin, scattered_dir};
        }
        case RT_MAT_DIELECTRIC: {
            // dielectric.rs:25-53
            Point p = hit.point();
            bool outside = dot(hit.incident->direction, hit.n) < 0.0;
            Direction b = hit.n * bias;
            double kr = fresnel(hit.incident->direction, hit.n, ior);
            rt_rng_block blk = rt_rng_draw(g_rng.seed, g_rng.pixel, g_rng.stratum, depth, RT_RNG_SITE_DIELECTRIC, 0);
            if (rt_rng_u0(&blk) < kr) {
                Direction reflected = reflect(hit.incident->direction, hit.n);
                Direction fz = fuzz * uniform_sphere_distribution(depth);
                Direction scattered = normalize(reflected + fz);
                return ScatteredRay{Color{1.0, 1.0, 1.0}, outside ? p + b : p - b, scattered};
            } else {
                Direction refracted = refract(hit.incident->direction, hit.n, ior);
                Direction fz = fuzz * uniform_sphere_distribution(depth);
                Direction scattered = normalize(refracted + fz);
                return ScatteredRay{Color{1.0, 1.0, 1.0}, outside ? p - b : p + b, scattered};
            }
        }
        case RT_MAT_DIFFUSE_LIGHT:
            return std::nullopt; // diffuse_light.rs:20-22
        default: {
            // isotropic.rs:20-26
            return ScatteredRay{texture->color_at_uv(hit.uv), hit.point(), uniform_sphere_distribution(depth)};
        }
        }
    }

    Color emit(const RayHit &hit) const
    {
        if (kind == RT_MAT_DIFFUSE_LIGHT) return intensity * texture->color_at_uv(hit.uv); // diffuse_light.rs:24-26
        return Color{0.0, 0.0, 0.0};
    }
};

// ---------------------------------------------------------------- object.rs:28-54, sdl.rs:18-34
struct Object {
    std::unique_ptr<Shape> shape;
    const Material *material;
};

struct Camera {
    double width, height, fov_factor;
    Matrix44f camera_to_world;
};

struct Scene {
    Color background_color;
    Camera camera;
    std::vector<Object> objects;
    std::vector<Material> materials;
    std::vector<Texture> textures;
};

struct Options {
    uint32_t width, height;
    double bias;
    uint16_t max_depth;
    uint32_t samples;
};

struct RenderContext {
    Options options;
    const Scene *scene;
    uint32_t sqrt_spp;
    double recip_sqrt_spp;
};

// system.rs:37-55
static Camera camera_new(double width, double height, double fov, Point origin, Point look_at)
{
    Direction up = dir(0.0, 1.0, 0.0);
    Direction zaxis = normalize(origin - look_at);
    Direction xaxis = cross(normalize(up), zaxis);
    Direction yaxis = cross(zaxis, xaxis);
    Camera c;
    c.width = width;
    c.height = height;
    c.fov_factor = std::tan(Matrix44f::to_radians(fov * 0.5));
    double rows[4][4] = {{xaxis.x, xaxis.y, xaxis.z, 0.0},
                         {yaxis.x, yaxis.y, yaxis.z, 0.0},
                         {zaxis.x, zaxis.y, zaxis.z, 0.0},
                         {origin.x, origin.y, origin.z, 1.0}};
    std::memcpy(c.camera_to_world.m, rows, sizeof(rows));
    return c;
}

// system.rs:57-66
static Ray pixel_ray(const Camera &cam, double x, double y)
{
    double aspect_ratio = cam.width / cam.height;
    double ndcx = x / cam.width;
    double ndcy = y / cam.height;
    double cx = (2.0 * ndcx - 1.0) * cam.fov_factor * aspect_ratio;
    double cy = (1.0 - 2.0 * ndcy) * cam.fov_factor;
    Point origin = pt(0, 0, 0) * cam.camera_to_world;
    Point dir_point = pt(cx, cy, -1.0) * cam.camera_to_world;
    return Ray::make(origin, normalize(dir_point - origin), 0);
}

// object.rs:50-54 with the panic guard
static std::optional<Intersection> object_intersect(const Object &o, const Ray &ray)
{
    try {
        std::optional<Intersection> i = o.shape->intersect(ray);
        if (i && i->t >= 0.0) return i;
        return std::nullopt;
    } catch (const Panic &p) {
        ++g_guards[p.which];
        return std::nullopt;
    }
}

// system.rs:122-133
static bool trace(const Ray &ray, const std::vector<Object> &objects, double max_distance, RayHit &hit)
{
    ++g_trace_calls;
    bool found = false;
    for (size_t k = 0; k < objects.size(); ++k) {
        g_rng.object = (uint32_t)k;
        std::optional<Intersection> i = object_intersect(objects[k], ray);
        if (!i) continue;
        if (!(i->t < max_distance)) continue;
        if (!found || i->t < hit.t) { // min_by keeps the first minimum
            hit.incident = &ray;
            hit.object = (int)k;
            hit.t = i->t;
            hit.n = i->n;
            hit.uv = i->uv;
            found = true;
        }
    }
    return found;
}

static Color cast(const Ray &ray, const RenderContext &ctx, int32_t *primary_id);

// system.rs:135-142
static Color hit_color(const Ray &ray, const RenderContext &ctx, const RayHit &hit)
{
    const Material *m = ctx.scene->objects[(size_t)hit.object].material;
    Color e = m->emit(hit);
    std::optional<ScatteredRay> sr = m->scatter(ctx.options.bias, hit);
    Color s;
    if (sr) {
        Ray next = Ray::make(sr->origin, sr->direction, (uint16_t)(ray.depth + 1));
        s = sr->attenuation * cast(next, ctx, nullptr);
    } else {
        s = ctx.scene->background_color;
    }
    return e + s;
}

// system.rs:112-120
static Color cast(const Ray &ray, const RenderContext &ctx, int32_t *primary_id)
{
    if (ray.depth >= ctx.options.max_depth) return ctx.scene->background_color;
    RayHit hit;
    if (trace(ray, ctx.scene->objects, std::numeric_limits<double>::max(), hit)) {
        if (primary_id) *primary_id = hit.object;
        return hit_color(ray, ctx, hit);
    }
    if (primary_id) *primary_id = -1;
    return ctx.scene->background_color;
}

// system.rs:237-242
static Ray get_stratified_ray(const RenderContext &ctx, uint32_t x, uint32_t y, uint32_t s_i, uint32_t s_j)
{
    rt_rng_block blk = rt_rng_draw(g_rng.seed, g_rng.pixel, g_rng.stratum, 0, RT_RNG_SITE_JITTER, 0);
    double s_x = (((double)s_i + rt_rng_u0(&blk)) * ctx.recip_sqrt_spp) - 0.5;
    double s_y = (((double)s_j + rt_rng_u1(&blk)) * ctx.recip_sqrt_spp) - 0.5;
    return pixel_ray(ctx.scene->camera, (double)x + s_x, (double)y + s_y);
}

// ---------------------------------------------------------------- rt_scene_desc -> Scene
static Matrix44f load_matrix(const double m[16])
{
    Matrix44f r;
    std::memcpy(r.m, m, sizeof(double) * 16);
    return r;
}

static std::unique_ptr<Shape> build_shape(const rt_scene_desc *d, int idx)
{
    if (idx < 0 || (uint32_t)idx >= d->n_nodes) throw std::runtime_error("shape node index out of range");
    const rt_shape_node &nd = d->nodes[idx];
    const double *p = nd.params;
    std::unique_ptr<Shape> s;
    switch (nd.kind) {
    case RT_SHAPE_SPHERE: s = std::make_unique<Sphere>(pt(p[0], p[1], p[2]), p[3]); break;
    case RT_SHAPE_PLANE: s = std::make_unique<Plane>(pt(p[0], p[1], p[2]), dir(p[3], p[4], p[5])); break;
    case RT_SHAPE_XYRECT: s = new_xyrect(pt(p[0], p[1], p[2]), p[3], p[4], (nd.flags & 1) != 0); break;
    case RT_SHAPE_XZRECT: s = new_xzrect(pt(p[0], p[1], p[2]), p[3], p[4], (nd.flags & 1) != 0); break;
    case RT_SHAPE_ZYRECT: s = new_zyrect(pt(p[0], p[1], p[2]), p[3], p[4], (nd.flags & 1) != 0); break;
    case RT_SHAPE_CUBE: s = std::make_unique<Cube>(pt(p[0], p[1], p[2]), pt(p[3], p[4], p[5])); break;
    case RT_SHAPE_CYLINDER: s = std::make_unique<Cylinder>(p[0], p[1]); break;
    case RT_SHAPE_TORUS: s = std::make_unique<Torus>(p[0], p[1]); break;
    case RT_SHAPE_UNION:
    case RT_SHAPE_INTERSECTION:
    case RT_SHAPE_DIFFERENCE: {
        auto c = std::make_unique<CSG>();
        c->op = nd.kind;
        c->a = build_shape(d, nd.child_a);
        c->b = build_shape(d, nd.child_b);
        s = std::move(c);
        break;
    }
    case RT_SHAPE_COMPOSITE: {
        auto c = std::make_unique<Composite>();
        for (int k = 0; k < nd.child_b; ++k) c->shapes.push_back(build_shape(d, d->child_list[nd.child_a + k]));
        s = std::move(c);
        break;
    }
    case RT_SHAPE_MESH: {
        const rt_mesh &m = d->meshes[nd.child_a];
        std::vector<Point> v;
        std::vector<Direction> n;
        std::vector<MeshTriangle> t;
        for (uint32_t k = 0; k < m.n_vertices; ++k) {
            const double *q = d->vertices + 3 * (size_t)(m.first_vertex + k);
            v.push_back(pt(q[0], q[1], q[2]));
        }
        for (uint32_t k = 0; k < m.n_normals; ++k) {
            const double *q = d->normals + 3 * (size_t)(m.first_normal + k);
            n.push_back(dir(q[0], q[1], q[2]));
        }
        for (uint32_t k = 0; k < m.n_triangles; ++k) {
            const uint32_t *tv = d->tri_v + 3 * (size_t)(m.first_triangle + k);
            const uint32_t *tn = d->tri_n + 3 * (size_t)(m.first_triangle + k);
            t.push_back(MeshTriangle{{tv[0], tv[1], tv[2]}, {tn[0], tn[1], tn[2]}});
        }
        s = std::make_unique<Mesh>(std::move(v), std::move(n), std::move(t), (nd.flags & 1) != 0);
        break;
    }
    case RT_SHAPE_MEDIUM: {
        auto m = std::make_unique<HomogenousMedium>();
        m->boundary = build_shape(d, nd.child_a);
        m->density = p[0];
        s = std::move(m);
        break;
    }
    default: throw std::runtime_error("unknown shape kind");
    }
    // The description carries the Transformation as left by the host's
    // transform() call (object.rs:21-26).
    s->transformation().object_to_world = load_matrix(nd.o2w);
    s->transformation().world_to_object = load_matrix(nd.w2o);
    return s;
}

static std::unique_ptr<Scene> build_scene(const rt_scene_desc *d, uint32_t width, uint32_t height)
{
    auto sc = std::make_unique<Scene>();
    sc->background_color = Color{d->background[0], d->background[1], d->background[2]};
    sc->camera.width = (double)width;
    sc->camera.height = (double)height;
    sc->camera.fov_factor = d->camera.fov_factor;
    sc->camera.camera_to_world = load_matrix(d->camera.camera_to_world);
    sc->textures.resize(d->n_textures);
    for (uint32_t k = 0; k < d->n_textures; ++k) {
        const rt_texture &t = d->textures[k];
        Texture &o = sc->textures[k];
        o.kind = t.kind;
        o.c1 = Color{t.c1[0], t.c1[1], t.c1[2]};
        o.c2 = Color{t.c2[0], t.c2[1], t.c2[2]};
        o.scale = t.scale;
        if (t.kind == RT_TEX_IMAGE) {
            const rt_image &im = d->images[t.image];
            o.width = im.width;
            o.height = im.height;
            o.pixels = d->image_data + im.offset;
        }
    }
    sc->materials.resize(d->n_materials);
    for (uint32_t k = 0; k < d->n_materials; ++k) {
        const rt_material &m = d->materials[k];
        sc->materials[k] = Material{m.kind, m.fuzz, m.ior, m.intensity, m.texture >= 0 ? &sc->textures[m.texture] : nullptr};
    }
    for (uint32_t k = 0; k < d->n_objects; ++k) {
        Object o;
        o.shape = build_shape(d, d->objects[k].shape);
        o.material = &sc->materials[d->objects[k].material];
        sc->objects.push_back(std::move(o));
    }
    return sc;
}

// ---------------------------------------------------------------- system.rs:244-307
struct Totals {
    std::atomic<uint64_t> rays{0};
    std::atomic<uint32_t> guards[5];
    Totals()
    {
        for (auto &g : guards) g = 0;
    }
    void flush()
    {
        rays += g_trace_calls;
        g_trace_calls = 0;
        for (int k = 0; k < 5; ++k) {
            guards[k] += g_guards[k];
            g_guards[k] = 0;
        }
    }
};

// render_sample (system.rs:244-253) over rows [y0, y1)
static void render_rows(const RenderContext &ctx, uint64_t seed, double *buf, int32_t *ids, uint32_t s_i, uint32_t s_j,
                        uint32_t y0, uint32_t y1)
{
    const uint32_t W = ctx.options.width;
    g_rng.seed = seed;
    g_rng.stratum = s_i * ctx.sqrt_spp + s_j;
    for (uint32_t y = y0; y < y1; ++y) {
        for (uint32_t x = 0; x < W; ++x) {
            g_rng.pixel = y * W + x;
            Ray ray = get_stratified_ray(ctx, x, y, s_i, s_j);
            int32_t id = -1;
            Color c = cast(ray, ctx, &id);
            double *px = buf + ((size_t)y * W + x) * 3;
            px[0] = c.r;
            px[1] = c.g;
            px[2] = c.b;
            if (ids) ids[(size_t)y * W + x] = id;
        }
    }
}

} // namespace oracle

using namespace oracle;

// =====================================================================================
// C entry points (loaded with ctypes by tests/ and bench.py)
// =====================================================================================
extern "C" {

// threads <= 1: one thread.  strata_parallel != 0: the reference's own
// decomposition (one task per stratum, merged under a mutex in completion
// order, system.rs:288-299).  Otherwise rows of each stratum are split over
// `threads` workers and strata are merged in order.
int oracle_render(const rt_scene_desc *desc, const rt_options *opts, int threads, int strata_parallel, double *rgb_sum,
                  int32_t *primary_ids, rt_stats *stats)
{
    if (!desc || !opts || !rgb_sum) return RT_ERR_INVALID;
    std::unique_ptr<Scene> scene;
    try {
        scene = build_scene(desc, opts->width, opts->height);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
    RenderContext ctx;
    ctx.options = Options{opts->width, opts->height, opts->bias, (uint16_t)opts->max_depth, opts->samples};
    ctx.scene = scene.get();
    ctx.sqrt_spp = (uint32_t)std::sqrt((double)opts->samples);
    ctx.recip_sqrt_spp = 1.0 / std::sqrt((double)opts->samples);

    const uint32_t W = opts->width, H = opts->height;
    const size_t n = (size_t)W * H;
    std::fill(rgb_sum, rgb_sum + n * 3, 0.0); // alloc_render_buf (system.rs:229-235)
    Totals totals;
    if (threads < 1) threads = 1;

    std::vector<std::pair<uint32_t, uint32_t>> strat_coords; // system.rs:284-286
    for (uint32_t i = 0; i < ctx.sqrt_spp; ++i)
        for (uint32_t j = 0; j < ctx.sqrt_spp; ++j) strat_coords.push_back({i, j});

    if (strata_parallel && threads > 1 && strat_coords.size() > 1) {
        std::mutex mu;
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                size_t k = next++;
                if (k >= strat_coords.size()) break;
                std::vector<double> sample_buf(n * 3, 0.0);
                bool first = strat_coords[k].first == 0 && strat_coords[k].second == 0;
                render_rows(ctx, opts->seed, sample_buf.data(), first ? primary_ids : nullptr, strat_coords[k].first,
                            strat_coords[k].second, 0, H);
                std::lock_guard<std::mutex> lock(mu);
                for (size_t q = 0; q < n * 3; ++q) rgb_sum[q] += sample_buf[q]; // combine_renderbuf
            }
            totals.flush();
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    } else {
        std::vector<double> sample_buf(n * 3, 0.0);
        for (auto [s_i, s_j] : strat_coords) {
            bool first = s_i == 0 && s_j == 0;
            int32_t *ids = first ? primary_ids : nullptr;
            if (threads == 1) {
                render_rows(ctx, opts->seed, sample_buf.data(), ids, s_i, s_j, 0, H);
                totals.flush();
            } else {
                std::atomic<uint32_t> next_row{0};
                auto worker = [&]() {
                    for (;;) {
                        uint32_t y = next_row.fetch_add(4);
                        if (y >= H) break;
                        render_rows(ctx, opts->seed, sample_buf.data(), ids, s_i, s_j, y, std::min(H, y + 4));
                    }
                    totals.flush();
                };
                std::vector<std::thread> pool;
                for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
                for (auto &t : pool) t.join();
            }
            for (size_t q = 0; q < n * 3; ++q) rgb_sum[q] += sample_buf[q]; // combine_renderbuf (system.rs:255-261)
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays = totals.rays;
        stats->primary_rays = (uint64_t)n * strat_coords.size();
        stats->strata = (uint32_t)strat_coords.size();
        for (int k = 0; k < 5; ++k) stats->guards[k] = totals.guards[k];
    }
    return RT_OK;
}

// Rows [y0, y1) of stratum (0,0) of a frame, at the frame's full geometry: lets a
// test check a band of a 4K or 8K frame without rendering all of it.
// rgb: (y1-y0)*W*3, ids: (y1-y0)*W or NULL.
int oracle_render_rows(const rt_scene_desc *desc, const rt_options *opts, uint32_t y0, uint32_t y1, int threads,
                       double *rgb, int32_t *ids, rt_stats *stats)
{
    if (!desc || !opts || !rgb || y0 >= y1 || y1 > opts->height) return RT_ERR_INVALID;
    std::unique_ptr<Scene> scene;
    try {
        scene = build_scene(desc, opts->width, opts->height);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
    RenderContext ctx;
    ctx.options = Options{opts->width, opts->height, opts->bias, (uint16_t)opts->max_depth, opts->samples};
    ctx.scene = scene.get();
    ctx.sqrt_spp = (uint32_t)std::sqrt((double)opts->samples);
    ctx.recip_sqrt_spp = 1.0 / std::sqrt((double)opts->samples);
    const uint32_t W = opts->width;
    double *base = rgb - (size_t)y0 * W * 3; // render_rows indexes by absolute row
    int32_t *ibase = ids ? ids - (size_t)y0 * W : nullptr;
    Totals totals;
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next_row{y0};
    auto worker = [&]() {
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= y1) break;
            render_rows(ctx, opts->seed, base, ibase, 0, 0, y, y + 1);
        }
        totals.flush();
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays = totals.rays;
        stats->primary_rays = (uint64_t)(y1 - y0) * W;
        stats->strata = 1;
        for (int k = 0; k < 5; ++k) stats->guards[k] = totals.guards[k];
    }
    return RT_OK;
}

// Rows [y0, y1) of every stratum of a frame, at the frame's full geometry, summed in stratum order
// (render_sample + combine_renderbuf, system.rs:244-261,284-299, restricted to a band): lets a test check a
// band of a multi-sample 8K frame.  ids: primary hits of stratum 0.
int oracle_render_rows_strata(const rt_scene_desc *desc, const rt_options *opts, uint32_t y0, uint32_t y1, int threads,
                              double *rgb_sum, int32_t *ids, rt_stats *stats)
{
    if (!desc || !opts || !rgb_sum || y0 >= y1 || y1 > opts->height) return RT_ERR_INVALID;
    std::unique_ptr<Scene> scene;
    try {
        scene = build_scene(desc, opts->width, opts->height);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
    RenderContext ctx;
    ctx.options = Options{opts->width, opts->height, opts->bias, (uint16_t)opts->max_depth, opts->samples};
    ctx.scene = scene.get();
    ctx.sqrt_spp = (uint32_t)std::sqrt((double)opts->samples);
    ctx.recip_sqrt_spp = 1.0 / std::sqrt((double)opts->samples);
    const uint32_t W = opts->width;
    const size_t n_values = (size_t)(y1 - y0) * W * 3;
    std::vector<double> tmp(n_values);
    std::fill(rgb_sum, rgb_sum + n_values, 0.0);
    double *base = tmp.data() - (size_t)y0 * W * 3; // render_rows indexes by absolute row
    int32_t *ibase = ids ? ids - (size_t)y0 * W : nullptr;
    Totals totals;
    if (threads < 1) threads = 1;
    for (uint32_t s_i = 0; s_i < ctx.sqrt_spp; ++s_i) {
        for (uint32_t s_j = 0; s_j < ctx.sqrt_spp; ++s_j) {
            std::atomic<uint32_t> next_row{y0};
            const bool first = s_i == 0 && s_j == 0;
            auto worker = [&]() {
                for (;;) {
                    uint32_t y = next_row.fetch_add(1);
                    if (y >= y1) break;
                    render_rows(ctx, opts->seed, base, first ? ibase : nullptr, s_i, s_j, y, y + 1);
                }
                totals.flush();
            };
            std::vector<std::thread> pool;
            for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
            for (auto &t : pool) t.join();
            for (size_t k = 0; k < n_values; ++k) rgb_sum[k] += tmp[k]; // combine_renderbuf
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays = totals.rays;
        stats->strata = ctx.sqrt_spp * ctx.sqrt_spp;
        stats->primary_rays = (uint64_t)(y1 - y0) * W * stats->strata;
        for (int k = 0; k < 5; ++k) stats->guards[k] = totals.guards[k];
    }
    return RT_OK;
}

// Ray::trace for a batch of rays (ray i: pixel i, stratum 0, depth 0).
int oracle_trace_rays(const rt_scene_desc *desc, uint64_t seed, uint64_t n, const double *origins,
                      const double *directions, rt_hit *hits, rt_stats *stats)
{
    if (!desc || !origins || !directions || !hits) return RT_ERR_INVALID;
    std::unique_ptr<Scene> scene;
    try {
        scene = build_scene(desc, 1, 1);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
    g_rng.seed = seed;
    g_rng.stratum = 0;
    for (uint64_t i = 0; i < n; ++i) {
        g_rng.pixel = (uint32_t)i;
        Ray ray = Ray::make(pt(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                            dir(directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]), 0);
        RayHit hit;
        rt_hit &o = hits[i];
        std::memset(&o, 0, sizeof(o));
        if (trace(ray, scene->objects, std::numeric_limits<double>::max(), hit)) {
            o.t = hit.t;
            o.n[0] = hit.n.x;
            o.n[1] = hit.n.y;
            o.n[2] = hit.n.z;
            o.uv[0] = hit.uv.u;
            o.uv[1] = hit.uv.v;
            o.object = hit.object;
        } else {
            o.t = -1.0;
            o.object = -1;
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays = g_trace_calls;
        for (int k = 0; k < 5; ++k) stats->guards[k] = g_guards[k];
    }
    g_trace_calls = 0;
    for (int k = 0; k < 5; ++k) g_guards[k] = 0;
    return RT_OK;
}

// main.rs:201-222
int oracle_tonemap(const double *rgb_sum, uint64_t n_pixels, double divisor, uint8_t *rgb8)
{
    for (uint64_t i = 0; i < n_pixels; ++i) {
        Color c = Color{rgb_sum[3 * i], rgb_sum[3 * i + 1], rgb_sum[3 * i + 2]} / divisor;
        c = Color{std::sqrt(c.r), std::sqrt(c.g), std::sqrt(c.b)}; // gamma_2, color.rs:36-38
        double ch[3] = {c.r, c.g, c.b};
        for (int k = 0; k < 3; ++k) {
            double v = std::fmin(ch[k] * 255.0, 255.0); // f64::min ignores NaN
            rgb8[3 * i + k] = (v != v || v <= 0.0) ? 0 : (v >= 255.0 ? 255 : (uint8_t)v);
        }
    }
    return RT_OK;
}

// ---- unit-level entry points for the golden vectors -----------------------------
// coefficient arrays are (re, im) pairs; roots likewise; returns the root count
int oracle_solve_quadratic(const double *coef, double *roots)
{
    auto r = solve_quadratic(Complex{coef[0], coef[1]}, Complex{coef[2], coef[3]}, Complex{coef[4], coef[5]});
    for (size_t k = 0; k < r.size(); ++k) {
        roots[2 * k] = r[k].re;
        roots[2 * k + 1] = r[k].im;
    }
    return (int)r.size();
}
int oracle_solve_cubic(const double *coef, double *roots)
{
    auto r = solve_cubic(Complex{coef[0], coef[1]}, Complex{coef[2], coef[3]}, Complex{coef[4], coef[5]},
                         Complex{coef[6], coef[7]});
    for (size_t k = 0; k < r.size(); ++k) {
        roots[2 * k] = r[k].re;
        roots[2 * k + 1] = r[k].im;
    }
    return (int)r.size();
}
int oracle_solve_quartic(const double *coef, double *roots)
{
    auto r = solve_quartic(Complex{coef[0], coef[1]}, Complex{coef[2], coef[3]}, Complex{coef[4], coef[5]},
                           Complex{coef[6], coef[7]}, Complex{coef[8], coef[9]});
    for (size_t k = 0; k < r.size(); ++k) {
        roots[2 * k] = r[k].re;
        roots[2 * k + 1] = r[k].im;
    }
    return (int)r.size();
}
int oracle_solve_quartic_f64(double a, double b, double c, double d, double e, double *roots)
{
    auto r = solve_quartic_f64(a, b, c, d, e);
    for (size_t k = 0; k < r.size(); ++k) roots[k] = r[k];
    return (int)r.size();
}

// kind: 0 identity, 1 translation, 2 scaling, 3 rotation_x, 4 rotation_y, 5 rotation_z
void oracle_matrix_make(int kind, const double *args, double *out)
{
    Matrix44f m = Matrix44f::identity();
    switch (kind) {
    case 1: m = Matrix44f::translation(dir(args[0], args[1], args[2])); break;
    case 2: m = Matrix44f::scaling(dir(args[0], args[1], args[2])); break;
    case 3: m = Matrix44f::rotation_x(args[0]); break;
    case 4: m = Matrix44f::rotation_y(args[0]); break;
    case 5: m = Matrix44f::rotation_z(args[0]); break;
    default: break;
    }
    std::memcpy(out, m.m, sizeof(m.m));
}
void oracle_matrix_mul(const double *a, const double *b, double *out)
{
    Matrix44f r = load_matrix(a) * load_matrix(b);
    std::memcpy(out, r.m, sizeof(r.m));
}
void oracle_matrix_inverse(const double *a, double *out)
{
    Matrix44f r = load_matrix(a).inverse();
    std::memcpy(out, r.m, sizeof(r.m));
}
void oracle_matrix_transpose(const double *a, double *out)
{
    Matrix44f r = load_matrix(a).transpose();
    std::memcpy(out, r.m, sizeof(r.m));
}
void oracle_point_mul_matrix(const double *p, const double *m, double *out)
{
    Point r = pt(p[0], p[1], p[2]) * load_matrix(m);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
void oracle_direction_mul_matrix(const double *p, const double *m, double *out)
{
    Direction r = dir(p[0], p[1], p[2]) * load_matrix(m);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

// color.rs ops: 0 add, 1 mul color, 2 mul f64 (b[0]), 3 div f64 (b[0])
void oracle_color_op(int op, const double *a, const double *b, double *out)
{
    Color x{a[0], a[1], a[2]}, r{0, 0, 0};
    switch (op) {
    case 0: r = x + Color{b[0], b[1], b[2]}; break;
    case 1: r = x * Color{b[0], b[1], b[2]}; break;
    case 2: r = x * b[0]; break;
    default: r = x / b[0]; break;
    }
    out[0] = r.r;
    out[1] = r.g;
    out[2] = r.b;
}

// Shape::intersection_intervals of one node for one ray (Ray::primary(o, d, 0));
// out: 12 doubles per interval (t, n xyz, uv) x 2.  Returns the interval
// count, or -(10 + which) when the reference would panic.
int oracle_shape_intervals(const rt_scene_desc *desc, int node, const double *o, const double *d, int max_intervals,
                           double *out)
{
    try {
        std::unique_ptr<Shape> s = build_shape(desc, node);
        Ray ray = Ray::make(pt(o[0], o[1], o[2]), dir(d[0], d[1], d[2]), 0);
        std::vector<Interval> v = s->intersection_intervals(ray);
        int n = 0;
        for (const Interval &iv : v) {
            if (n >= max_intervals) break;
            const Intersection *e[2] = {&iv.a, &iv.b};
            for (int k = 0; k < 2; ++k) {
                double *q = out + (size_t)n * 12 + k * 6;
                q[0] = e[k]->t;
                q[1] = e[k]->n.x;
                q[2] = e[k]->n.y;
                q[3] = e[k]->n.z;
                q[4] = e[k]->uv.u;
                q[5] = e[k]->uv.v;
            }
            ++n;
        }
        return n;
    } catch (const Panic &p) {
        return -(10 + p.which);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
}

// Intersectable::intersect of one node (no `t >= 0` filter); out: t, n xyz, uv.
// Returns 1 for Some, 0 for None.
int oracle_shape_intersect(const rt_scene_desc *desc, int node, const double *o, const double *d, double *out)
{
    try {
        std::unique_ptr<Shape> s = build_shape(desc, node);
        Ray ray = Ray::make(pt(o[0], o[1], o[2]), dir(d[0], d[1], d[2]), 0);
        g_rng = RngContext();
        std::optional<Intersection> i = s->intersect(ray);
        if (!i) return 0;
        out[0] = i->t;
        out[1] = i->n.x;
        out[2] = i->n.y;
        out[3] = i->n.z;
        out[4] = i->uv.u;
        out[5] = i->uv.v;
        return 1;
    } catch (const Panic &p) {
        return -(10 + p.which);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
}

// Camera::new + Camera::pixel_ray (system.rs:37-66): out = origin xyz, direction xyz
void oracle_camera_ray(double width, double height, double fov, const double *origin, const double *look_at, double x,
                       double y, double *out_c2w, double *out_fov_factor, double *out_ray)
{
    Camera c = camera_new(width, height, fov, pt(origin[0], origin[1], origin[2]), pt(look_at[0], look_at[1], look_at[2]));
    if (out_c2w) std::memcpy(out_c2w, c.camera_to_world.m, sizeof(c.camera_to_world.m));
    if (out_fov_factor) *out_fov_factor = c.fov_factor;
    Ray r = pixel_ray(c, x, y);
    out_ray[0] = r.origin.x;
    out_ray[1] = r.origin.y;
    out_ray[2] = r.origin.z;
    out_ray[3] = r.direction.x;
    out_ray[4] = r.direction.y;
    out_ray[5] = r.direction.z;
}

// dielectric.rs:68-108 and texture.rs:54-94 on their own, for the analytic checks in tests/test_analytic.py
double oracle_fresnel(const double *incident, const double *normal, double ior)
{
    return fresnel(dir(incident[0], incident[1], incident[2]), dir(normal[0], normal[1], normal[2]), ior);
}

void oracle_refract(const double *incident, const double *normal, double ior, double *out)
{
    Direction r = refract(dir(incident[0], incident[1], incident[2]), dir(normal[0], normal[1], normal[2]), ior);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

void oracle_reflect(const double *incident, const double *normal, double *out)
{
    Direction r = reflect(dir(incident[0], incident[1], incident[2]), dir(normal[0], normal[1], normal[2]));
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}

// kind: RT_TEX_*; pixels: RGB8 rows top to bottom (RT_TEX_IMAGE only)
void oracle_texture_color(int kind, const double *c1, const double *c2, double scale, uint32_t width, uint32_t height,
                          const uint8_t *pixels, double u, double v, double *out)
{
    Texture t;
    t.kind = kind;
    t.c1 = Color{c1[0], c1[1], c1[2]};
    t.c2 = Color{c2[0], c2[1], c2[2]};
    t.scale = scale;
    t.width = width;
    t.height = height;
    t.pixels = pixels;
    Color c = t.color_at_uv(Vector2f{u, v});
    out[0] = c.r;
    out[1] = c.g;
    out[2] = c.b;
}

// Philox block for the known-answer vectors
void oracle_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out)
{
    rt_rng_block b = rt_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int k = 0; k < 4; ++k) out[k] = b.w[k];
}

} // extern "C"
