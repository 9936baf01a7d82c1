// rt_shard.h — tile sharding geometry, shared by host code and kernels.
//
// The reference parallelises over strata and shares one frame buffer under a
// mutex (system.rs:284-299).  Here pixels are what is distributed: the image
// is cut into T x T tiles (row-major tile order) and tile k belongs to shard
// k % shard_count, so sky and geometry interleave evenly over the GPUs.
// Inside a tile the slots run over 8x8 pixel blocks, so one 64-lane wave owns
// one 8x8 block of primary rays.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_SHARD_HD __host__ __device__ inline
#else
#define RT_SHARD_HD static inline
#endif

typedef struct rt_shard_geom {
    uint32_t width, height;
    uint32_t tile;     // tile edge in pixels, multiple of 8
    uint32_t tiles_x, tiles_y, n_tiles;
    uint32_t shard_count;
    uint32_t tile_px;  // tile * tile
} rt_shard_geom;

RT_SHARD_HD rt_shard_geom rt_shard_geom_make(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t shard_count)
{
    rt_shard_geom g;
    g.width = width;
    g.height = height;
    uint32_t t = tile_size ? tile_size : 64u;
    t = (t + 7u) & ~7u;
    g.tile = t;
    g.tiles_x = (width + t - 1u) / t;
    g.tiles_y = (height + t - 1u) / t;
    g.n_tiles = g.tiles_x * g.tiles_y;
    g.shard_count = shard_count ? shard_count : 1u;
    g.tile_px = t * t;
    return g;
}

// tiles owned by `rank`
RT_SHARD_HD uint32_t rt_shard_tiles(const rt_shard_geom *g, uint32_t rank)
{
    if (rank >= g->shard_count || rank >= g->n_tiles) return 0u;
    return (g->n_tiles - rank + g->shard_count - 1u) / g->shard_count;
}

// slot in the shard's buffers -> pixel; returns 0 for padding beyond the image edge
RT_SHARD_HD int rt_shard_decode(const rt_shard_geom *g, uint32_t rank, uint32_t slot, uint32_t *x, uint32_t *y)
{
    uint32_t local_tile = slot / g->tile_px;
    uint32_t j = slot - local_tile * g->tile_px;
    uint32_t k = local_tile * g->shard_count + rank;
    uint32_t tx = k % g->tiles_x, ty = k / g->tiles_x;
    uint32_t block = j >> 6, lane = j & 63u;
    uint32_t bpr = g->tile >> 3;
    uint32_t bx = block % bpr, by = block / bpr;
    *x = tx * g->tile + bx * 8u + (lane & 7u);
    *y = ty * g->tile + by * 8u + (lane >> 3);
    return (*x < g->width) && (*y < g->height);
}
