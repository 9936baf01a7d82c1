// Test probe for the two stores of a ray's interval lists (rt_dev_shapes.h): IvListT (an array per list: scratch memory
// on the device) and LdsList (a strided view: on the device the words of all lanes of a workgroup interleaved in LDS, the
// interval counts packed four bits each into one word).  The interval algebra of the CSG evaluator - csg_apply with its
// sort, iv_copy, iv_push - is written once against the member names both have; this file runs it on the host
// (tools/host_shim) over the same inputs through both stores - every ordering of up to 3 x 3 intervals on a grid of
// end points and random lists, all three operations, capacities 12 and 4 (the LDS store's capacity is a run-time
// value: overflow must be counted the same way) - and compares the results entry by entry.  Test infrastructure only.
#include "rt_dev_shapes.h"

#include <cstdint>
#include <random>
#include <vector>

namespace {
constexpr int STRIDE = 3; // "threads" per workgroup of the host-side view; this probe is thread 1
template <int CAP>
struct Case {
    IvListT<CAP> a, b, out;
    std::vector<double> tl;
    std::vector<uint32_t> gl;
    LdsLists<STRIDE> ll;
    Case() : tl(3 * 2 * CAP * STRIDE, -777.0), gl(3 * 2 * CAP * STRIDE, 0xBAD0BAD0u)
    {
        ll.t = tl.data() + 1;
        ll.tag = gl.data() + 1;
        ll.cap_ = CAP;
        ll.n_lists = 3;
        ll.counts = 0u;
    }
    void load(const std::vector<double> &ea, const std::vector<double> &eb)
    {
        auto la = ll[0];
        auto lb = ll[1];
        a.n = (int)ea.size() / 2;
        la.n = (int)ea.size() / 2;
        for (size_t i = 0; i < ea.size(); ++i) {
            a.t[i] = ea[i], a.tag[i] = 0x100u + (uint32_t)i;
            la.t[(int)i] = ea[i], la.tag[(int)i] = 0x100u + (uint32_t)i;
        }
        b.n = (int)eb.size() / 2;
        lb.n = (int)eb.size() / 2;
        for (size_t i = 0; i < eb.size(); ++i) {
            b.t[i] = eb[i], b.tag[i] = 0x200u + (uint32_t)i;
            lb.t[(int)i] = eb[i], lb.tag[(int)i] = 0x200u + (uint32_t)i;
        }
    }
    // 0: same; otherwise what differed
    int run(int op)
    {
        Guards g1 = {}, g2 = {};
        auto la = ll[0];
        auto lb = ll[1];
        auto lo = ll[2];
        const bool ok1 = csg_apply(op, a, b, out, g1);
        const bool ok2 = csg_apply(op, la, lb, lo, g2);
        if (ok1 != ok2) return 1;
        if (g1.overflow != g2.overflow || g1.g[2] != g2.g[2] || g1.g[4] != g2.g[4]) return 2;
        if (!ok1) return 0;
        if (out.n != (int)lo.n) return 3;
        for (int i = 0; i < 2 * out.n; ++i)
            if (out.t[i] != lo.t[i] || out.tag[i] != lo.tag[i]) return 4;
        // the words of the neighbouring "threads" are untouched
        for (size_t i = 0; i < tl.size(); ++i)
            if (i % STRIDE != 1 && (tl[i] != -777.0 || gl[i] != 0xBAD0BAD0u)) return 5;
        return 0;
    }
};

template <int CAP>
int sweep(uint64_t seed, uint64_t n_random, uint64_t *cases)
{
    Case<CAP> c;
    // every pair of lists with up to 3 intervals whose end points lie on a grid 0 .. 7 (sorted, disjoint or touching)
    std::vector<std::vector<double>> lists;
    lists.push_back({});
    for (int a0 = 0; a0 < 8; ++a0)
        for (int a1 = a0; a1 < 8; ++a1) {
            lists.push_back({(double)a0, (double)a1});
            for (int b0 = a1; b0 < 8; ++b0)
                for (int b1 = b0; b1 < 8; ++b1) {
                    lists.push_back({(double)a0, (double)a1, (double)b0, (double)b1});
                    if (b1 < 7 && a0 % 2 == 0) lists.push_back({(double)a0, (double)a1, (double)b0, (double)b1, (double)b1 + 0.5, 7.5});
                }
        }
    for (size_t i = 0; i < lists.size(); i += 3)
        for (size_t j = 0; j < lists.size(); j += 5)
            for (int op = DEV_OP_UNION; op <= DEV_OP_DIFFERENCE; ++op) {
                c.load(lists[i], lists[j]);
                const int r = c.run(op);
                ++*cases;
                if (r) return r;
            }
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-2.0, 10.0);
    for (uint64_t k = 0; k < n_random; ++k) {
        std::vector<double> e[2];
        for (int s = 0; s < 2; ++s) {
            const int n = (int)(rng() % (uint64_t)(CAP + 1)); // up to a full list: results overflow and must be counted alike
            double x = U(rng);
            for (int i = 0; i < 2 * n; ++i) {
                e[s].push_back(x);
                x += (rng() % 4 == 0) ? 0.0 : std::fabs(U(rng)) * 0.3; // touching end points now and then
            }
        }
        c.load(e[0], e[1]);
        const int r = c.run(DEV_OP_UNION + (int)(k % 3));
        ++*cases;
        if (r) return 10 + r;
    }
    return 0;
}
} // namespace

extern "C" int probe_list_stores(uint64_t seed, uint64_t n_random, uint64_t *cases)
{
    *cases = 0;
    int r = sweep<12>(seed, n_random, cases);
    if (r) return r;
    r = sweep<4>(seed + 1, n_random, cases);
    return r ? 100 + r : 0;
}
