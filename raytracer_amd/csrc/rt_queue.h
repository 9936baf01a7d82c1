// rt_queue.h — the queues' regions: their counters, the readers' map, where an append's entries go.  Plain C++ behind
// RT_DEV with the two device operations it needs handed in (the atomic add as a functor, the wave-uniform value as
// RT_Q_UNIFORM), so that tests/helpers/queue_probe.cpp can run it on the host; included by rt_k_common.h.
#pragma once

#ifndef RT_Q_UNIFORM // a value that is the same in every lane of the wave, as a scalar
#define RT_Q_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif

// A queue that kernels append to lies in RT_Q_REGIONS parts of `rc` entries, each with a counter word of its own
// (region r: positions r * rc .. r * rc + min(counter, rc) - 1; the counters `stride` words apart, on different memory
// channels).  A single counter word takes about 90 atomics per microsecond on this chip; eight take a wave's append
// every 32 entries.  Appenders go round the regions, so they fill evenly; one that finds its region full moves on to the
// next (a pass appends at most as many entries as it read, RT_Q_REGIONS * rc hold them all, and a full region stays
// full: after RT_Q_REGIONS tries every entry has a place).  Readers go over 0 .. total-1 and map an index to its
// position (QMap).  A queue filled by position 0 .. n-1 (rt_trace_rays, the camera pass's slots) is the case rc >= n
// with n in the first counter.
#define RT_Q_REGIONS 8u
struct QRegions {
    uint32_t *cnt;
    uint32_t rc, stride;
    uint32_t first_mask; // RT_Q_REGIONS - 1; 0 (RT_Q_SKEW=1, a test knob): every append tries region 0 first, so the regions
                         // fill one after the other and most appends meet a full one
};
struct QMap {
    uint32_t start[RT_Q_REGIONS + 1]; // index of region r's first entry; [RT_Q_REGIONS]: entries in the queue
    uint32_t rc;
    RT_DEV uint32_t total() const { return start[RT_Q_REGIONS]; }
    RT_DEV uint32_t pos(uint32_t i) const
    {
        uint32_t first = 0u, at = 0u;
#pragma unroll
        for (uint32_t r = 1; r < RT_Q_REGIONS; ++r)
            if (i >= start[r]) {
                first = start[r];
                at = r * rc;
            }
        return at + (i - first);
    }
    // ... of entry i, one of the 64 entries i0 .. i0 + 63 that a wave reads together (i0: the same in every lane).  All
    // but seven such blocks lie in one region: scalar work and one addition per lane
    RT_DEV uint32_t pos(uint32_t i0, uint32_t i) const
    {
        const uint32_t u0 = RT_Q_UNIFORM(i0);
        uint32_t first = 0u, at = 0u, next = start[1];
#pragma unroll
        for (uint32_t r = 1; r < RT_Q_REGIONS; ++r)
            if (u0 >= start[r]) {
                first = start[r];
                at = r * rc;
                next = start[r + 1];
            }
        if (u0 + 63u < next) return i + (at - first);
        return pos(i);
    }
};
RT_DEV QMap qmap_load(const QRegions &q)
{
    QMap m;
    m.rc = q.rc;
    uint32_t s = 0u;
#pragma unroll
    for (uint32_t r = 0; r < RT_Q_REGIONS; ++r) {
        m.start[r] = s;
        const uint32_t c = q.cnt[r * q.stride];
        s += c < q.rc ? c : q.rc;
    }
    m.start[RT_Q_REGIONS] = s;
    return m;
}

// Where the n entries of one append go: region `r` first (`r` moves on for the appender's next append), the next
// region when one is full.  add(word, k) adds k to a counter word and returns what it held; seg(first, position, count)
// is called for every run of entries first .. first + count - 1 that found room, at position .. position + count - 1.
template <class ADD, class SEG>
RT_DEV void region_reserve(const QRegions &q, uint32_t &r, uint32_t n, ADD &&add, SEG &&seg)
{
    uint32_t done = 0u;
    r &= q.first_mask;
    for (uint32_t tries = 0; done < n && tries < RT_Q_REGIONS; ++tries) {
        const uint32_t old = add(q.cnt + r * q.stride, n - done);
        const uint32_t room = old < q.rc ? q.rc - old : 0u;
        const uint32_t take = room < n - done ? room : n - done;
        if (take) seg(done, r * q.rc + old, take);
        done += take;
        r = (r + 1u) & (RT_Q_REGIONS - 1u);
    }
}
