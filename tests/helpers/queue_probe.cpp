// Test probe for the queues' regions (raytracer_amd/csrc/rt_queue.h): runs the kernels' own code on the host
// (tools/host_shim; the atomic add of an appender is a plain add here, appenders take turns) and checks, over random
// region sizes, fill levels and append sizes, with appenders going round the regions or all starting at region 0
// (RT_Q_SKEW), that
//   - every appended entry gets a position of its own, inside a region's capacity, and runs of entries stay in order;
//   - as long as no more is appended than the regions hold between them, every entry finds a place within
//     RT_Q_REGIONS tries, whatever the counters of full regions have run up to;
//   - the readers' map built from the counters (QMap) sends 0 .. total-1 one to one onto exactly those positions, region
//     by region in rising order, and its per-block form (the scalar path of a wave's 64 entries, the fallback for blocks
//     that straddle regions) agrees with the per-entry form for every entry of every block, the last partial one included;
//   - a queue filled by position (one region as large as the queue) maps every index to itself.
// Test infrastructure only.
#include "rt_dev_math.h"
#define RT_Q_UNIFORM(x) ((uint32_t)(x))
#include "rt_queue.h"

#include <cstdint>
#include <random>
#include <vector>

extern "C" int probe_queue_regions(uint64_t seed, uint64_t rounds, uint64_t *entries_out)
{
    std::mt19937_64 rng(seed);
    uint64_t entries = 0;
    for (uint64_t round = 0; round < rounds; ++round) {
        const uint32_t rc = 1u + (uint32_t)(rng() % 300u);
        const uint32_t stride = 1u + (uint32_t)(rng() % 5u);
        const uint32_t cap = RT_Q_REGIONS * rc;
        std::vector<uint32_t> words(RT_Q_REGIONS * stride, 0u);
        QRegions q;
        q.cnt = words.data();
        q.rc = rc;
        q.stride = stride;
        q.first_mask = (rng() % 3u == 0u) ? 0u : RT_Q_REGIONS - 1u;
        // fill level: sometimes to the last entry
        const uint32_t want_total = (rng() % 4u == 0u) ? cap : (uint32_t)(rng() % (cap + 1u));
        const uint32_t max_n = (rng() % 2u) ? 64u : 256u;
        std::vector<int> owner(cap, -1);
        uint32_t appended = 0u;
        std::vector<uint32_t> next_region(16);
        for (auto &r : next_region) r = (uint32_t)rng();
        int append_id = 0;
        while (appended < want_total) {
            uint32_t n = 1u + (uint32_t)(rng() % max_n);
            if (n > want_total - appended) n = want_total - appended;
            uint32_t &r = next_region[rng() % next_region.size()];
            uint32_t placed = 0u, last_first = 0u;
            bool bad = false;
            region_reserve(
                q, r, n,
                [](uint32_t *word, uint32_t k) {
                    const uint32_t old = *word;
                    *word += k;
                    return old;
                },
                [&](uint32_t first, uint32_t position, uint32_t count) {
                    if (first != placed || (placed && first < last_first)) bad = true; // runs come in order, without gaps
                    last_first = first;
                    for (uint32_t k = 0; k < count; ++k) {
                        const uint32_t p = position + k;
                        if (p >= cap || p / rc != position / rc || owner[p] != -1) {
                            bad = true;
                            continue;
                        }
                        owner[p] = append_id;
                    }
                    placed += count;
                });
            if (bad) return 1;
            if (placed != n) return 2; // an entry found no place although the regions had room
            if (r >= RT_Q_REGIONS) return 3;
            appended += n;
            ++append_id;
        }
        entries += appended;
        // the readers' map
        const QMap m = qmap_load(q);
        if (m.total() != appended) return 4;
        std::vector<char> seen(cap, 0);
        uint32_t prev = 0u;
        for (uint32_t i = 0; i < appended; ++i) {
            const uint32_t p = m.pos(i);
            if (p >= cap || owner[p] == -1 || seen[p]) return 5;
            seen[p] = 1;
            if (i && p / rc == prev / rc && p != prev + 1u) return 6; // inside a region: consecutive
            if (i && p / rc < prev / rc) return 7;                    // regions in rising order
            prev = p;
        }
        for (uint32_t i0 = 0; i0 < appended; i0 += 64u)
            for (uint32_t i = i0; i < i0 + 64u && i < appended; ++i)
                if (m.pos(i0, i) != m.pos(i)) return 8;
        // blocks that do not start at a multiple of 64 (the mesh kernels' cursors in short queues)
        for (int k = 0; k < 8 && appended; ++k) {
            const uint32_t i0 = (uint32_t)(rng() % appended);
            for (uint32_t i = i0; i < i0 + 64u && i < appended; ++i)
                if (m.pos(i0, i) != m.pos(i)) return 9;
        }
    }
    // a queue filled by position: region 0 as large as the queue
    for (uint32_t n : {0u, 1u, 63u, 64u, 1000u, 4000000u}) {
        std::vector<uint32_t> words(RT_Q_REGIONS * 1024u, 0u);
        words[0] = n;
        QRegions q;
        q.cnt = words.data();
        q.rc = 4000064u;
        q.stride = 1024u;
        q.first_mask = RT_Q_REGIONS - 1u;
        const QMap m = qmap_load(q);
        if (m.total() != n) return 10;
        for (uint32_t i = 0; i < n; i += (n > 5000u ? 997u : 1u))
            if (m.pos(i) != i || m.pos(i & ~63u, i) != i) return 11;
    }
    *entries_out = entries;
    return 0;
}
