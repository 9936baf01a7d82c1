// csg_host_check.cpp — the interval merge of the CSG evaluator (csg_apply, rt_dev_shapes.h) as HOST code under
// -fsanitize=undefined,address, old formulation against new.
//
// Round 1 found that `difference { union { a b } c }` lost an interval on the device when c swallowed the
// union's first interval and trimmed the second.  The merge loop then carried two bool flags ("list a / b still
// has an interval in hand"); rewritten with integer indices the device gave the right answer, and the round-1
// notes called it a gfx950 miscompile without a repro.  This tool asks the other question: does the old
// formulation (kept below verbatim from commit 1efb9a9^, as `csg_apply_flags`) contain undefined behaviour or a
// read of uninitialised list slots that would make such a "miscompile" legitimate?  It runs both formulations
// over (1) the failing configuration, (2) every ordering of up to 3 x 3 intervals on a small grid of end points,
// (3) random lists - each twice, with the unused slots of the input lists filled with two different garbage
// patterns (a result that depends on the garbage reads a slot it must not) - and compares them with each other
// and with a merge written independently here from the set definitions (point classification of the midpoints).
//
//   g++ -std=c++17 -O1 -g -fsanitize=undefined,address -fno-sanitize-recover=all -ffp-contract=off \
//       -Itools -Itools/host_shim -Iraytracer_amd/csrc tools/csg_host_check.cpp -o /tmp/csg_host_check && /tmp/csg_host_check
#include "rt_dev_shapes.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CSG_FLAGS_ATTR static __attribute__((noinline))
#include "csg_apply_flags.inc"

static void fill(IvList &l, const std::vector<double> &ends, uint32_t tag0, double garbage)
{
    for (int i = 0; i < 2 * DEV_MAX_INTERVALS; ++i) {
        l.t[i] = garbage;
        l.tag[i] = 0xDEAD0000u + (uint32_t)i;
    }
    l.n = (int)ends.size() / 2;
    for (size_t i = 0; i < ends.size(); ++i) {
        l.t[i] = ends[i];
        l.tag[i] = tag0 + (uint32_t)i;
    }
}

static bool same(const IvList &a, const IvList &b)
{
    if (a.n != b.n) return false;
    for (int i = 0; i < 2 * a.n; ++i)
        if (a.t[i] != b.t[i] || a.tag[i] != b.tag[i]) return false;
    return true;
}

static bool inside(const std::vector<double> &l, double x)
{
    for (size_t i = 0; i + 1 < l.size(); i += 2)
        if (x > l[i] && x < l[i + 1]) return true;
    return false;
}

// the set the result must cover, by classifying the midpoints between all end points (difference and
// intersection are exact set operations in csg.rs; its union may repeat or split intervals, so for the union
// only the covered set is compared, not the list)
static bool covers_like_sets(int op, const std::vector<double> &A, const std::vector<double> &B, const IvList &out)
{
    std::vector<double> pts(A);
    pts.insert(pts.end(), B.begin(), B.end());
    std::sort(pts.begin(), pts.end());
    std::vector<double> o(out.t, out.t + 2 * out.n);
    for (size_t i = 0; i + 1 < pts.size(); ++i) {
        if (pts[i] == pts[i + 1]) continue;
        const double x = 0.5 * (pts[i] + pts[i + 1]);
        const bool a = inside(A, x), b = inside(B, x);
        const bool want = op == DEV_OP_UNION ? (a || b) : op == DEV_OP_INTERSECTION ? (a && b) : (a && !b);
        if (inside(o, x) != want) return false;
    }
    return true;
}

static long n_cases = 0, n_garbage = 0, n_differ = 0, n_sets = 0, n_union_quirk = 0;

static void check(int op, const std::vector<double> &A, const std::vector<double> &B, bool disjoint_sorted)
{
    IvList a, b, out_new[2], out_old[2];
    Guards gd{};
    bool ok_new[2], ok_old[2];
    for (int pass = 0; pass < 2; ++pass) {
        const double garbage = pass ? -7.25e200 : std::nan("");
        fill(a, A, 0x100, garbage);
        fill(b, B, 0x200, garbage);
        for (int i = 0; i < 2 * DEV_MAX_INTERVALS; ++i) {
            out_new[pass].t[i] = out_old[pass].t[i] = garbage;
            out_new[pass].tag[i] = out_old[pass].tag[i] = 0xBEEF0000u;
        }
        ok_new[pass] = csg_apply(op, a, b, out_new[pass], gd);
        ok_old[pass] = csg_apply_flags(op, a, b, out_old[pass], gd);
    }
    ++n_cases;
    if (ok_new[0] != ok_new[1] || ok_old[0] != ok_old[1] || (ok_new[0] && !same(out_new[0], out_new[1])) ||
        (ok_old[0] && !same(out_old[0], out_old[1])))
        ++n_garbage;
    if (ok_new[0] != ok_old[0] || (ok_new[0] && !same(out_new[0], out_old[0]))) ++n_differ;
    // csg.rs:60-71: a union that swallows a later interval of a sets new_end = a_end even when that shrinks it
    // (A = [1,2] [3,4], B = [1,5] gives [1,4]); reproduced, so only the other two operations are held to the sets
    if (disjoint_sorted && ok_new[0] && op == DEV_OP_UNION && !covers_like_sets(op, A, B, out_new[0])) ++n_union_quirk;
    if (disjoint_sorted && ok_new[0] && op != DEV_OP_UNION && !covers_like_sets(op, A, B, out_new[0])) {
        if (n_sets < 6) {
            std::printf("set mismatch op %d: A =", op);
            for (double x : A) std::printf(" %g", x);
            std::printf("; B =");
            for (double x : B) std::printf(" %g", x);
            std::printf("; out =");
            for (int i = 0; i < 2 * out_new[0].n; ++i) std::printf(" %g", out_new[0].t[i]);
            std::printf("\n");
        }
        ++n_sets;
    }
}

int main()
{
    // (1) the round-1 failure: A = two intervals, B swallows the first and trims the second
    {
        IvList a, b, o1, o2;
        Guards gd{};
        fill(a, {1.0, 2.0, 3.0, 5.0}, 0x100, std::nan(""));
        fill(b, {0.5, 4.0}, 0x200, std::nan(""));
        bool k1 = csg_apply(DEV_OP_DIFFERENCE, a, b, o1, gd), k2 = csg_apply_flags(DEV_OP_DIFFERENCE, a, b, o2, gd);
        std::printf("round-1 case: new ok=%d n=%d [%g, %g]; flags ok=%d n=%d [%g, %g]\n", k1, o1.n, o1.t[0], o1.t[1], k2, o2.n,
                    o2.t[0], o2.t[1]);
        if (!k1 || !k2 || o1.n != 1 || o2.n != 1 || o1.t[0] != 4.0 || o1.t[1] != 5.0 || !same(o1, o2)) return 1;
    }
    // (2) exhaustive: lists of 1..3 disjoint sorted intervals with end points on a grid of 9 values
    const int G = 9;
    std::vector<std::vector<double>> lists;
    for (int mask = 0; mask < (1 << G); ++mask) {
        const int k = __builtin_popcount(mask);
        if (k == 0 || k > 6 || (k & 1)) continue;
        std::vector<double> l;
        for (int i = 0; i < G; ++i)
            if (mask >> i & 1) l.push_back(1.0 + i);
        lists.push_back(l);
    }
    for (int op = DEV_OP_UNION; op <= DEV_OP_DIFFERENCE; ++op)
        for (const auto &A : lists)
            for (const auto &B : lists) check(op, A, B, true);
    // (3) random: up to 6 intervals per list, overlapping and unsorted ones included (what a union can hand on)
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(-2.0, 10.0);
    for (int it = 0; it < 2000000; ++it) {
        std::vector<double> A, B;
        const int na = 1 + (int)(rng() % 6), nb = 1 + (int)(rng() % 6);
        const bool tidy = rng() & 1;
        for (int side = 0; side < 2; ++side) {
            std::vector<double> &L = side ? B : A;
            const int n = side ? nb : na;
            if (tidy) {
                std::vector<double> e(2 * n);
                for (auto &x : e) x = std::round(U(rng) * 4.0) / 4.0; // ties on purpose
                std::sort(e.begin(), e.end());
                L = e;
            } else {
                for (int k = 0; k < n; ++k) {
                    double s = U(rng), w = std::fabs(U(rng)) * 0.3;
                    L.push_back(s);
                    L.push_back(s + w);
                }
            }
        }
        check(DEV_OP_UNION + (int)(rng() % 3), A, B, false);
    }
    std::printf("%ld cases: results that depend on unused list slots %ld, flags formulation differs from index "
                "formulation %ld, set mismatches of intersection / difference %ld (unions that shrink as csg.rs:60-71 does: %ld)\n",
                n_cases, n_garbage, n_differ, n_sets, n_union_quirk);
    return (n_garbage || n_differ || n_sets) ? 1 : 0;
}
