// rt_k_common.h — wavefront buffers, queue appends, LDS staging of the scene tables, camera rays
// (device code of librt_amd.so; included by rt_kernels.hip, which holds the host side of the C ABI)
#pragma once

// ------------------------------------------------------------------ device buffers
struct RayQueue {
    double *o[3];
    double *d[3];
    double *thr[3];
    uint32_t *slot;  // index into the shard's frame buffer
    uint32_t *pixel; // y * width + x, the random-number key
    // primary != 0: nothing above is read.  Entry i is slot i of the shard and its ray is that
    // pixel's camera ray, regenerated wherever it is needed (a jitter draw and a normalise) instead
    // of being written by one kernel and read back by three: 1.8 GB less traffic per 4K frame.
    int primary;
    int row_major; // outputs are indexed by pixel instead of by shard slot (RT_FLAG_ROW_MAJOR)
    uint32_t rank, s_i, s_j;
    rt_shard_geom g;
    double recip_sqrt_spp;
};

struct HitBuf {
    double *t;
    double *n[3];
    double *uv[2];
    int32_t *obj;
};

#ifndef RT_MESH_BATCH
#define RT_MESH_BATCH 2
#endif
#ifndef RT_MESH_PAIRS
#define RT_MESH_PAIRS 2 // f32 record pairs k_mesh loads and pre-tests per step
#endif
// k_mesh's waves take their further 64-record blocks from RT_MESH_CURSORS counters (wave w from counter w mod
// RT_MESH_CURSORS; block m of counter p is block m * RT_MESH_CURSORS + p of the queue's dynamic part): one counter
// word sustains about 90 atomics per microsecond, which 3072 waves taking a block every few microseconds exceed
// cells a lane crosses per round before the wave turns to the triangle lists (a lane still in empty cells
// then goes on in the next round).  The same limit on the records tested per round (a lane with a long run going
// on in the next round) gained nothing: 1.29 / 1.30 / 1.42 / 1.71 ms at 32 / 16 / 8 / 4 records.
#ifndef RT_WALK_STEPS
#define RT_WALK_STEPS 4 // 4K mesh.scene, k_mesh_one per frame: 1.69 / 1.44 / 1.27 / 1.26 / 1.28 / 1.36 ms at 1 / 2 / 3 / 4 / 6 / 12, 1.39 without a limit
#endif
// ... and since round 3 the count follows the wave (RT_WALK_DYN, rt_k_mesh_grid_round.inc): after RT_WALK_STEPS steps the
// lanes go on while more than one in RT_WALK_DYN_DEN of the active ones is still in empty cells, RT_WALK_DYN_MAX steps at
// most.  4K mesh.scene, k_mesh_one per frame, same run: 0.777 ms with the fixed 4; 0.737 / 0.738 / 0.738 / 0.745 / 0.762 at
// (DEN, MAX) = (2, 12) / (3, 8) / (4, 8 or 12) / (6, 12) / (8, 16); first steps 2 or 3 instead of 4: the same
#ifndef RT_WALK_DYN
#define RT_WALK_DYN 1
#endif
#ifndef RT_WALK_DYN_MAX
#define RT_WALK_DYN_MAX 8
#endif
#ifndef RT_WALK_DYN_DEN
#define RT_WALK_DYN_DEN 4u
#endif
#ifndef RT_MESH_CURSORS
#define RT_MESH_CURSORS 16
#endif
#ifndef RT_MESH_WAVES
#define RT_MESH_WAVES 3
#endif
#ifndef RT_MESH_MIN_BLOCK
#define RT_MESH_MIN_BLOCK 4u // fewest lanes a wave of k_mesh uses when the queue is short (480x270 frame: k_mesh 0.39 -> 0.34 ms against 16)
#endif
#define RT_MESH_OCC __attribute__((amdgpu_waves_per_eu(RT_MESH_WAVES, RT_MESH_WAVES)))

// k_intersect, waves per SIMD per variant (rt_dev_math.h): with the shading in the same kernel the plain
// variant is best at 3 (4K mesh.scene intersect stage 1.03 ms against 1.08 at 4: 71 spilled registers there),
// the fog variants at 2 (fog_light 8K x 16: 123 ms per frame against 138 at 3), CSG alone stays at 3
// (room 1080p 28.6 ms against 30.3 at 2).
#ifndef RT_TAIL_WAVES
#define RT_TAIL_WAVES 3 // 4K mesh.scene tail 0.36 -> 0.34 ms against 2 (41 spilled registers); cornell_box 0.26 -> 0.27
#endif
// k_tail without CSG / fog / torus code (the bench scene's variant): waves per SIMD it is compiled for
// ... with CSG / fog / torus code: RT_TAIL_HEAVY_WAVES
#ifndef RT_TAIL_HEAVY_WAVES
#define RT_TAIL_HEAVY_WAVES 2
#endif
#define RT_TAIL_OCC __attribute__((amdgpu_waves_per_eu((CSG || MEDIUM || TORUS) ? RT_TAIL_HEAVY_WAVES : RT_TAIL_WAVES)))
// ... and the variants with their interval lists in LDS at RT_LL_WAVES: the lists of two workgroups fill a CU's LDS
#ifndef RT_LL_WAVES
#define RT_LL_WAVES 2
#endif
// the plain variant on a ray queue (PQ == 2): RT_QUEUE_WAVES
#ifndef RT_QUEUE_WAVES
#define RT_QUEUE_WAVES RT_PLAIN_WAVES
#endif
#define RT_ISECT_WAVES(C, M, T, LL, PQ) ((LL) ? RT_LL_WAVES : ((T) ? RT_TORUS_WAVES : ((M) ? RT_MEDIUM_WAVES : ((C) ? RT_CSG_WAVES : ((PQ) == 2 ? RT_QUEUE_WAVES : RT_PLAIN_WAVES)))))
#define RT_ISECT_OCC __attribute__((amdgpu_waves_per_eu(RT_ISECT_WAVES(CSG, MEDIUM, TORUS, LL, PQ))))

struct PassParams {
    uint64_t seed;
    uint32_t stratum;
    uint32_t depth;
    uint32_t max_depth;
    uint32_t force_uv;
    double bias;
};

#define RT_BLOCK 256
#ifndef RT_ISECT_BLOCK
#define RT_ISECT_BLOCK 256 // threads per k_intersect workgroup (512: +3 %, 1024: +20 % on the 4K bench frame's intersect stage)
#endif

RT_DEV uint32_t lane_id() { return threadIdx.x & 63u; }

// A finished path's colour goes to its pixel.  Every slot takes at most one path per launch, so the sum is the same
// whichever unit adds: the L2's f64 add does (global_atomic_add_f64, nothing returned), so the wave has no load to
// wait for; and an exact zero - a path that ends on a black background - is not added at all (the sums start at
// +0.0 and never hold -0.0, so x + 0 is x bit for bit).
RT_DEV void fb_add(double *px, double r, double g, double b)
{
    if (r != 0.0 || g != 0.0 || b != 0.0) {
        unsafeAtomicAdd(px + 0, r);
        unsafeAtomicAdd(px + 1, g);
        unsafeAtomicAdd(px + 2, b);
    }
}


#include "rt_queue.h" // QRegions, QMap, region_reserve

// Append with one atomic per workgroup and list: list a and list b of the same block (a thread may be in at most
// one), region r0 first.  Called by every thread of the block, in block-uniform control flow.
// `lds` holds RT_APPEND_LDS(blockDim.x) words.
#define RT_APPEND_LDS(BLOCK) (2 * ((BLOCK) / 64 + 1 + 2 * (int)RT_Q_REGIONS))
RT_DEV void block_append2(bool a, bool b, const QRegions &qa, const QRegions &qb, uint32_t r0, uint32_t *lds, uint32_t &at_a, uint32_t &at_b)
{
    const unsigned long long ma = __ballot(a), mb = __ballot(b);
    const uint32_t wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    // per list: [w] rank of wave w's first entry; [n_waves] segments; then per segment its first rank and its
    // position - rank (one segment, unless a region filled up under this block)
    const uint32_t per_list = n_waves + 1u + 2u * RT_Q_REGIONS;
    if (lane_id() == 0) {
        lds[wave] = (uint32_t)__popcll(ma);
        lds[per_list + wave] = (uint32_t)__popcll(mb);
    }
    __syncthreads();
    if (threadIdx.x < 2) { // thread 0 places list a, thread 1 list b
        uint32_t *l = lds + threadIdx.x * per_list;
        const QRegions &q = threadIdx.x ? qb : qa;
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; ++w) {
            const uint32_t c = l[w];
            l[w] = total;
            total += c;
        }
        uint32_t n_seg = 0u, r = r0;
        region_reserve(
            q, r, total, [](uint32_t *word, uint32_t k) { return atomicAdd(word, k); },
            [&](uint32_t first, uint32_t position, uint32_t) {
                l[n_waves + 1u + n_seg] = first;
                l[n_waves + 1u + RT_Q_REGIONS + n_seg] = position - first;
                ++n_seg;
            });
        l[n_waves] = n_seg;
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane_id()) - 1ull;
    const uint32_t rank[2] = {lds[wave] + (uint32_t)__popcll(ma & lt), lds[per_list + wave] + (uint32_t)__popcll(mb & lt)};
    uint32_t at[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t *l = lds + k * per_list;
        const uint32_t n_seg = l[n_waves];
        uint32_t off = l[n_waves + 1u + RT_Q_REGIONS];
        for (uint32_t sgm = 1; sgm < n_seg; ++sgm)
            if (rank[k] >= l[n_waves + 1u + sgm]) off = l[n_waves + 1u + RT_Q_REGIONS + sgm];
        at[k] = off + rank[k];
    }
    at_a = at[0];
    at_b = at[1];
    __syncthreads();
}
// ... one list
RT_DEV uint32_t block_append(bool alive, const QRegions &q, uint32_t r0, uint32_t *lds)
{
    uint32_t at, none;
    block_append2(alive, false, q, q, r0, lds, at, none);
    return at;
}

// Append per wave, without barriers (the bounce passes: there the waves of a workgroup differ most in what a round
// costs them, and with block_append2 the other three wait for the slowest at every barrier).
// Position of entry e of the n a wave appends at once (e: an entry's rank among them), region `r` first; `r` moves on
// for the wave's next append.
RT_DEV uint32_t wave_reserve(const QRegions &q, uint32_t &r, uint32_t n, uint32_t e)
{
    uint32_t at = 0u;
    region_reserve(
        q, r, n,
        [](uint32_t *word, uint32_t k) {
            uint32_t old = 0u;
            if (lane_id() == 0u) old = atomicAdd(word, k);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        },
        [&](uint32_t first, uint32_t position, uint32_t count) {
            if (e >= first && e < first + count) at = position + (e - first);
        });
    return at;
}
// One atomic per append is one returned value to wait for per round and list, and 64 times the atomics of a
// workgroup's append when few lanes have an entry.  So a wave with fewer than RT_STAGE_DIRECT_MIN entries collects them
// in LDS (ND doubles and NU words per entry, component k of entry e at d[k * C + e]) and moves them to the queue C at
// a time, with coalesced stores; with more it appends them from its registers.  C == 0: no room in LDS, every append
// from the registers.  Wave-uniform control flow; `flush(n)` stores the staged entries 0..n-1.
#ifndef RT_STAGE_CAP
#define RT_STAGE_CAP 32
#endif
#ifndef RT_STAGE_DIRECT_MIN
#define RT_STAGE_DIRECT_MIN 24u
#endif
template <int C, int ND, int NU>
struct WaveStage {
    double *d;
    uint32_t *u;
    uint32_t fill;
};
RT_DEV void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// returns true in the lanes that have to store their entry themselves, at `at`
template <int C, int ND, int NU, class FLUSH>
RT_DEV bool wave_append(WaveStage<C, ND, NU> &st, bool has, const double (&vd)[ND], const uint32_t (&vu)[NU], const QRegions &q,
                        uint32_t &r, uint32_t &at, FLUSH &&flush)
{
    const unsigned long long m = __ballot(has);
    if (m == 0ull) return false;
    const uint32_t n = (uint32_t)__popcll(m);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
    if (C == 0 || n >= RT_STAGE_DIRECT_MIN) {
        at = wave_reserve(q, r, n, has ? rank : 0xFFFFFFFFu);
        return has;
    }
    uint32_t done = 0u;
    while (done < n) {
        const uint32_t room = (uint32_t)C - st.fill;
        const uint32_t take = room < n - done ? room : n - done;
        if (has && rank >= done && rank < done + take) {
            const uint32_t e = st.fill + rank - done;
#pragma unroll
            for (int k = 0; k < ND; ++k) st.d[k * C + e] = vd[k];
#pragma unroll
            for (int k = 0; k < NU; ++k) st.u[k * C + e] = vu[k];
        }
        st.fill += take;
        done += take;
        if (st.fill == (uint32_t)C) {
            wave_lds_fence();
            flush((uint32_t)C);
            wave_lds_fence();
            st.fill = 0u;
        }
    }
    return false;
}

#ifndef RT_BLOCK_WIDE
#define RT_BLOCK_WIDE 256 // k_shade: one atomic per workgroup and append.  1024 (a quarter of the atomics) measures the same
                          // with the GPU to itself; beside other frames' kernels a 256-lane workgroup finds room on a CU
                          // sooner than a 1024-lane one: 4K mesh.scene, 4 frames in flight, 2.29 against 2.34 ms per frame
#endif

// The scene's small tables (objects, leaves, meshes, materials, textures) are read by every ray
// through chains of dependent loads: objects[k] -> leaves[leaf0] -> parameters.  From memory each
// link costs a cache round trip of several hundred cycles - measured, the object loop of
// k_intersect waited 27 000 cycles per 64 rays on them.  Each workgroup therefore copies the
// tables into LDS once and the kernel works on a DevScene whose pointers lead there.  Scenes whose
// tables exceed RT_SCENE_LDS_MAX keep reading memory.  Ends with a barrier.
#define RT_SCENE_LDS_MAX 32768u
static inline __host__ __device__ uint32_t scene_lds_bytes(const DevScene &sc)
{
    const uint32_t b = (((uint32_t)sc.n_objects * (uint32_t)sizeof(DevObject) + 15u) & ~15u) +
                       ((sc.n_leaves * (uint32_t)sizeof(DevLeaf) + 15u) & ~15u) +
                       ((sc.n_meshes * (uint32_t)sizeof(DevMesh) + 15u) & ~15u) +
                       ((sc.n_materials * (uint32_t)sizeof(DevMaterial) + 15u) & ~15u) +
                       ((sc.n_textures * (uint32_t)sizeof(DevTexture) + 15u) & ~15u);
    return b <= RT_SCENE_LDS_MAX ? b : 0u;
}

// IN_LDS is a compile-time choice on purpose: were the pointers of `v` a run-time choice between memory and
// LDS, every table access would compile to a flat (address-space-agnostic) load, which waits on both the
// memory and the LDS counters; unconditionally derived from the LDS array they become ds_read instructions.
// The launch code picks the IN_LDS = false instantiation for scenes whose tables exceed RT_SCENE_LDS_MAX.
template <bool IN_LDS>
RT_DEV DevScene scene_in_lds(const DevScene &sc, unsigned char *lds)
{
    DevScene v = sc;
    if (IN_LDS) {
        unsigned char *p = lds;
#define RT_STAGE(FIELD, TYPE, COUNT)                                                                                   \
    do {                                                                                                               \
        const uint32_t words_ = (uint32_t)(COUNT) * (uint32_t)(sizeof(TYPE) / 4);                                      \
        const uint32_t *src_ = reinterpret_cast<const uint32_t *>(sc.FIELD);                                           \
        uint32_t *dst_ = reinterpret_cast<uint32_t *>(p);                                                              \
        for (uint32_t i_ = threadIdx.x; i_ < words_; i_ += blockDim.x) dst_[i_] = src_[i_];                            \
        v.FIELD = reinterpret_cast<const TYPE *>(p);                                                                   \
        p += (words_ * 4u + 15u) & ~15u;                                                                               \
    } while (0)
        RT_STAGE(objects, DevObject, sc.n_objects);
        RT_STAGE(leaves, DevLeaf, sc.n_leaves);
        RT_STAGE(meshes, DevMesh, sc.n_meshes);
        RT_STAGE(materials, DevMaterial, sc.n_materials);
        RT_STAGE(textures, DevTexture, sc.n_textures);
#undef RT_STAGE
    }
    __syncthreads();
    return v;
}

// ------------------------------------------------------------------ ray generation
// get_stratified_ray + Camera::pixel_ray (system.rs:237-242, 57-66) for slot `slot` of the shard;
// false for the padding slots beyond the image edge
RT_DEV bool primary_ray(const DevScene &sc, const RayQueue &q, const PassParams &pp, uint32_t slot, d3 &origin, d3 &dir,
                        uint32_t &pixel)
{
    uint32_t x = 0, y = 0;
    if (!rt_shard_decode(&q.g, q.rank, slot, &x, &y)) return false;
    pixel = y * q.g.width + x;
    rt_rng_block blk = rt_rng_draw(pp.seed, pixel, pp.stratum, 0u, RT_RNG_SITE_JITTER, 0u);
    double sx = (((double)q.s_i + rt_rng_u0(&blk)) * q.recip_sqrt_spp) - 0.5;
    double sy = (((double)q.s_j + rt_rng_u1(&blk)) * q.recip_sqrt_spp) - 0.5;
    double px = (double)x + sx, py = (double)y + sy;
    double w = (double)q.g.width, h = (double)q.g.height;
    double aspect = w / h;
    double ndcx = px / w, ndcy = py / h;
    double cx = (2.0 * ndcx - 1.0) * sc.cam_fov_factor * aspect;
    double cy = (1.0 - 2.0 * ndcy) * sc.cam_fov_factor;
    const double *m = sc.cam_c2w;
    origin = mk3(m[12], m[13], m[14]);
    d3 dp = mk3(cx * m[0] + cy * m[4] + -1.0 * m[8] + m[12], cx * m[1] + cy * m[5] + -1.0 * m[9] + m[13],
                cx * m[2] + cy * m[6] + -1.0 * m[10] + m[14]);
    dir = normalize3(dp - origin);
    return true;
}

// origin, direction and random-number key of queue entry i; false for a padding slot
RT_DEV bool fetch_ray(const DevScene &sc, const RayQueue &q, const PassParams &pp, uint32_t i, d3 &o, d3 &d, uint32_t &pixel)
{
    if (q.primary) return primary_ray(sc, q, pp, i, o, d, pixel);
    o = mk3(q.o[0][i], q.o[1][i], q.o[2][i]);
    d = mk3(q.d[0][i], q.d[1][i], q.d[2][i]);
    pixel = q.pixel[i];
    return true;
}

// index of the ray's pixel in the output buffers
RT_DEV uint32_t ray_slot(const RayQueue &q, uint32_t i)
{
    if (!q.primary) return q.slot[i];
    if (!q.row_major) return i;
    uint32_t x = 0, y = 0;
    (void)rt_shard_decode(&q.g, q.rank, i, &x, &y);
    return y * q.g.width + x;
}

#define RT_HIT_PADDING (-2) // hit.obj of a padding slot: no ray, nothing to shade
