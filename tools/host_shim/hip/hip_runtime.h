// Host-side stand-in for <hip/hip_runtime.h>, for tools/csg_host_check.cpp only: lets g++/clang++ compile this
// repository's own device headers (rt_dev_math.h, rt_dev_shapes.h are plain C++ behind RT_DEV) as host code so
// that they can be run under -fsanitize=undefined,address.  Never used by the product build.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__
#define __restrict__
struct double2 { double x, y; };
struct double4 { double x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
using std::fabs; using std::sqrt; using std::floor; using std::fmin; using std::fmax; using std::pow; using std::atan2;
using std::signbit; using std::acos; using std::cos; using std::sin; using std::log; using std::hypot; using std::cbrt; using std::trunc;
