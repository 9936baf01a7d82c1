/*
 * rt_rng.h — the counter-based random-number contract shared by the GPU core
 * and by any CPU checker of it.
 *
 * The reference draws every random number from `rand::rng()` (ThreadRng,
 * OS-seeded, non-reproducible):
 *   - pixel jitter            reference src/system.rs:238-240      (2 draws)
 *   - uniform sphere sample   reference src/direction.rs:29-39     (2 draws)
 *   - dielectric coin flip    reference src/materials/dielectric.rs:31-32 (1 draw)
 *   - fog free-path sample    reference src/shapes/homogenous_medium.rs:37-43
 *                             (1 draw per interval examined)
 * `random::<f64>()` there is a 53-bit uniform in [0,1).
 *
 * A wavefront tracer evaluates those draws in a different order from a
 * recursive CPU tracer, so the stream is replaced by a pure function of
 * WHERE the draw happens:
 *
 *   block = Philox4x32-10( key  = (seed_lo, seed_hi),
 *                          ctr  = (pixel, stratum | depth << 16, site, object) )
 *   u0 = f64 from block words (1,0),   u1 = f64 from block words (3,2)
 *
 *   pixel   = y * width + x   (global, independent of tile sharding)
 *   stratum = s_i * sqrt_spp + s_j
 *   depth   = Ray::depth of the ray being traced / shaded (0 = primary)
 *   site    = RT_RNG_SITE_*
 *   object  = index in scene.objects for RT_RNG_SITE_MEDIUM, else 0
 *
 * Pure integer arithmetic: bit-identical on host and device.
 */
#ifndef RT_RNG_H
#define RT_RNG_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD static inline
#endif

#define RT_RNG_SITE_JITTER 0u     /* u0 -> s_x, u1 -> s_y                       */
#define RT_RNG_SITE_SPHERE 1u     /* u0 -> theta draw, u1 -> phi draw           */
#define RT_RNG_SITE_DIELECTRIC 2u /* u0 -> reflect/refract coin                 */
#define RT_RNG_SITE_MEDIUM 16u    /* + k for the k-th interval examined, u0     */

typedef struct rt_rng_block {
    uint32_t w[4];
} rt_rng_block;

RT_HD void rt_philox_mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

/* Philox4x32 with 10 rounds (Salmon et al., SC'11); known-answer vectors in
 * tests/golden/reference_unit_vectors.json ("philox4x32_10": Random123's kat_vectors), checked by
 * tests/test_oracle_golden.py. */
RT_HD rt_rng_block rt_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        rt_philox_mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        rt_philox_mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    rt_rng_block b;
    b.w[0] = c0;
    b.w[1] = c1;
    b.w[2] = c2;
    b.w[3] = c3;
    return b;
}

/* 53-bit uniform in [0,1): (u64 >> 11) * 2^-53, the same mapping rand 0.9 uses
 * for `random::<f64>()`. */
RT_HD double rt_rng_to_unit(uint32_t hi, uint32_t lo)
{
    uint64_t u = ((uint64_t)hi << 32) | (uint64_t)lo;
    return (double)(u >> 11) * (1.0 / 9007199254740992.0);
}

RT_HD rt_rng_block rt_rng_draw(uint64_t seed, uint32_t pixel, uint32_t stratum, uint32_t depth, uint32_t site,
                               uint32_t object)
{
    return rt_philox4x32_10(pixel, (stratum & 0xFFFFu) | (depth << 16), site, object, (uint32_t)seed,
                            (uint32_t)(seed >> 32));
}

RT_HD double rt_rng_u0(const rt_rng_block *b) { return rt_rng_to_unit(b->w[1], b->w[0]); }
RT_HD double rt_rng_u1(const rt_rng_block *b) { return rt_rng_to_unit(b->w[3], b->w[2]); }

#endif /* RT_RNG_H */
