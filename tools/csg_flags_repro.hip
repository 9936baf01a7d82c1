// csg_flags_repro.hip — does hipcc still turn the flag-carrying merge loop (tools/csg_apply_flags.inc, the
// formulation before commit 1efb9a9) into gfx950 code that differs from the index-based one?  One thread runs
// both over the round-1 failure and over every pair of lists of 1..3 disjoint intervals on a 9-point grid, for
// the three operations, and counts the cases in which the two disagree.  The host check next to this file
// (csg_host_check.cpp) shows that the two are the same function on the host, free of undefined behaviour.
//
//   hipcc -std=c++17 -O3 -ffp-contract=off --offload-arch=gfx950 -Itools -Iraytracer_amd/csrc tools/csg_flags_repro.hip -o csg_flags_repro
#include <hip/hip_runtime.h>

#include <cstdio>

#include "rt_dev_shapes.h"
#define CSG_FLAGS_ATTR __device__ __noinline__
#include "csg_apply_flags.inc"

__device__ void fill_list(IvList &l, uint32_t mask, uint32_t tag0)
{
    l.n = 0;
    int k = 0;
    for (int i = 0; i < 9; ++i)
        if (mask >> i & 1u) {
            l.t[k] = 1.0 + i;
            l.tag[k] = tag0 + k;
            ++k;
        }
    l.n = k / 2;
}

__device__ bool inside_list(const IvList &l, double x)
{
    for (int i = 0; i < l.n; ++i)
        if (x > l.t[2 * i] && x < l.t[2 * i + 1]) return true;
    return false;
}

// intersection and difference are exact set operations in csg.rs: classify the midpoints of the grid
__device__ bool matches_sets(int op, const IvList &a, const IvList &b, const IvList &o)
{
    for (int i = 0; i < 8; ++i) {
        const double x = 1.5 + i;
        const bool ia = inside_list(a, x), ib = inside_list(b, x);
        const bool want = op == DEV_OP_INTERSECTION ? (ia && ib) : (ia && !ib);
        if (inside_list(o, x) != want) return false;
    }
    return true;
}

__global__ void k_compare(unsigned *result)
{
    unsigned differ = 0, cases = 0, first = 0xFFFFFFFFu, bad_index = 0, bad_flags = 0;
    Guards gd{};
    for (int op = DEV_OP_UNION; op <= DEV_OP_DIFFERENCE; ++op)
        for (uint32_t ma = 1; ma < 512; ++ma)
            for (uint32_t mb = 1; mb < 512; ++mb) {
                const int ka = __popc(ma), kb = __popc(mb);
                if ((ka & 1) || (kb & 1) || ka > 6 || kb > 6) continue;
                IvList a, b, o1, o2;
                fill_list(a, ma, 0x100);
                fill_list(b, mb, 0x200);
                const bool k1 = csg_apply(op, a, b, o1, gd), k2 = csg_apply_flags(op, a, b, o2, gd);
                bool same = k1 == k2 && (!k1 || o1.n == o2.n);
                for (int i = 0; same && k1 && i < 2 * o1.n; ++i) same = o1.t[i] == o2.t[i] && o1.tag[i] == o2.tag[i];
                ++cases;
                if (op != DEV_OP_UNION) {
                    if (k1 && !matches_sets(op, a, b, o1)) ++bad_index; // (!k: touching intervals, where csg.rs:262 panics)
                    if (k2 && !matches_sets(op, a, b, o2)) ++bad_flags;
                }
                if (!same) {
                    if (differ < 3) {
                        printf("op %d, a 0x%x, b 0x%x: index loop ok %d:", op, ma, mb, (int)k1);
                        for (int i = 0; i < 2 * o1.n && i < 12; ++i) printf(" %g", o1.t[i]);
                        printf("; flag-carrying loop ok %d:", (int)k2);
                        for (int i = 0; i < 2 * o2.n && i < 12; ++i) printf(" %g", o2.t[i]);
                        printf("\n");
                    }
                    if (!differ) first = (unsigned)op << 24 | ma << 12 | mb;
                    ++differ;
                }
            }
    result[0] = cases;
    result[1] = differ;
    result[2] = first;
    result[3] = bad_index;
    result[4] = bad_flags;
}

int main()
{
    unsigned *d = nullptr, h[5] = {0, 0, 0, 0, 0};
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_compare, dim3(1), dim3(1), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    std::printf("gfx950: %u cases, flag-carrying loop differs from index loop in %u (first: op %u, a mask 0x%x, b mask 0x%x)\n", h[0], h[1],
                h[2] >> 24, (h[2] >> 12) & 0xFFF, h[2] & 0xFFF);
    std::printf("intersections and differences that are not the set operation: index loop %u, flag-carrying loop %u\n", h[3], h[4]);
    return h[1] ? 1 : 0;
}
