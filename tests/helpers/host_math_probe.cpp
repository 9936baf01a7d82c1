// Test probe around the product's host-side matrix code (raytracer_amd/csrc/host_math.h): lets
// tests/test_analytic.py call its inverse, product and constructors directly.  Test infrastructure only.
#include "../../raytracer_amd/csrc/host_math.h"

extern "C" {
void probe_inverse(const double *in, double *out) { rt_host::mat_store(rt_host::mat_inverse(rt_host::mat_load(in)), out); }
void probe_mul(const double *a, const double *b, double *out)
{
    rt_host::mat_store(rt_host::mat_mul(rt_host::mat_load(a), rt_host::mat_load(b)), out);
}
// kind: 0 translation, 1 scaling, 2..4 rotation about x, y, z (degrees in args[0])
void probe_make(int kind, const double *args, double *out)
{
    rt_host::Mat4 m = rt_host::mat_identity();
    switch (kind) {
    case 0: m = rt_host::mat_translation(args[0], args[1], args[2]); break;
    case 1: m = rt_host::mat_scaling(args[0], args[1], args[2]); break;
    case 2: m = rt_host::mat_rotation_x(args[0]); break;
    case 3: m = rt_host::mat_rotation_y(args[0]); break;
    case 4: m = rt_host::mat_rotation_z(args[0]); break;
    }
    rt_host::mat_store(m, out);
}
}
