// host_math.h — host-side 4x4 matrix algebra used while a scene is being
// built (transform composition, inverse).  Runs once per shape at parse time,
// exactly where the reference runs it (sdl.rs:44-47, object.rs:21-26); the
// device never inverts a matrix.
//
// Conventions of reference src/matrix.rs: row-major, row-vector * matrix,
// translation in row 3.
#pragma once
#include <cmath>
#include <cstring>

namespace rt_host {

struct Mat4 {
    double m[4][4];
};

inline Mat4 mat_identity()
{
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.0 : 0.0;
    return r;
}

// matrix.rs:29-36
inline Mat4 mat_translation(double x, double y, double z)
{
    Mat4 r = mat_identity();
    r.m[3][0] = x;
    r.m[3][1] = y;
    r.m[3][2] = z;
    return r;
}

// matrix.rs:38-45
inline Mat4 mat_scaling(double x, double y, double z)
{
    Mat4 r = mat_identity();
    r.m[0][0] = x;
    r.m[1][1] = y;
    r.m[2][2] = z;
    return r;
}

// f64::to_radians: value * (PI / 180)
inline double deg_to_rad(double deg) { return deg * (M_PI / 180.0); }

// matrix.rs:47-55
inline Mat4 mat_rotation_x(double deg)
{
    double a = deg_to_rad(deg), s = std::sin(a), c = std::cos(a);
    Mat4 r = mat_identity();
    r.m[1][1] = c;
    r.m[1][2] = s;
    r.m[2][1] = -s;
    r.m[2][2] = c;
    return r;
}

// matrix.rs:57-65
inline Mat4 mat_rotation_y(double deg)
{
    double a = deg_to_rad(deg), s = std::sin(a), c = std::cos(a);
    Mat4 r = mat_identity();
    r.m[0][0] = c;
    r.m[0][2] = -s;
    r.m[2][0] = s;
    r.m[2][2] = c;
    return r;
}

// matrix.rs:67-75
inline Mat4 mat_rotation_z(double deg)
{
    double a = deg_to_rad(deg), s = std::sin(a), c = std::cos(a);
    Mat4 r = mat_identity();
    r.m[0][0] = c;
    r.m[0][1] = s;
    r.m[1][0] = -s;
    r.m[1][1] = c;
    return r;
}

// matrix.rs:183-196
inline Mat4 mat_mul(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}

// matrix.rs:81-156 — Gauss-Jordan with partial pivoting; a singular input
// yields the identity, as there.
inline Mat4 mat_inverse(const Mat4 &in)
{
    Mat4 s = mat_identity();
    Mat4 t = in;
    for (int i = 0; i < 3; ++i) {
        int pivot = i;
        double pivot_size = std::fabs(t.m[i][i]);
        for (int j = i + 1; j < 4; ++j) {
            double tmp = std::fabs(t.m[j][i]);
            if (tmp > pivot_size) {
                pivot = j;
                pivot_size = tmp;
            }
        }
        if (pivot_size == 0.0) return mat_identity();
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
        if (f == 0.0) return mat_identity();
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

inline void mat_store(const Mat4 &a, double out[16]) { std::memcpy(out, a.m, sizeof(double) * 16); }
inline Mat4 mat_load(const double in[16])
{
    Mat4 a;
    std::memcpy(a.m, in, sizeof(double) * 16);
    return a;
}

} // namespace rt_host
