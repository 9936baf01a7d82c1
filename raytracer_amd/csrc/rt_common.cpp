// rt_common.cpp — pieces of the C ABI that need no device: last-error slot,
// shard geometry, tone mapping.
#include "../../include/rt_api.h"
#include "rt_shard.h"

#include <cmath>
#include <string>

namespace rt_host {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }

} // namespace rt_host

extern "C" const char *rt_last_error(void) { return rt_host::g_last_error.c_str(); }

extern "C" uint64_t rt_shard_pixels(const rt_options *opts)
{
    if (!opts) return 0;
    rt_shard_geom g = rt_shard_geom_make(opts->width, opts->height, opts->tile_size, opts->shard_count);
    return (uint64_t)rt_shard_tiles(&g, opts->shard_rank) * g.tile_px;
}

extern "C" int rt_shard_map(const rt_options *opts, uint32_t *pixel_index)
{
    if (!opts || !pixel_index || opts->width == 0 || opts->height == 0) {
        rt_host::set_last_error("rt_shard_map: bad argument");
        return RT_ERR_INVALID;
    }
    rt_shard_geom g = rt_shard_geom_make(opts->width, opts->height, opts->tile_size, opts->shard_count);
    if (opts->shard_rank >= g.shard_count) {
        rt_host::set_last_error("rt_shard_map: shard_rank out of range");
        return RT_ERR_INVALID;
    }
    uint64_t n = (uint64_t)rt_shard_tiles(&g, opts->shard_rank) * g.tile_px;
    for (uint64_t s = 0; s < n; ++s) {
        uint32_t x, y;
        pixel_index[s] = rt_shard_decode(&g, opts->shard_rank, (uint32_t)s, &x, &y) ? y * g.width + x : 0xFFFFFFFFu;
    }
    return RT_OK;
}

// main.rs:201-222: (sum / divisor).gamma_2() -> (v * 255).min(255) as u8.
// `as u8` saturates: negative -> 0, NaN -> 0; f64::min drops a NaN operand, so
// a NaN channel becomes min(NaN, 255) = 255.
extern "C" int rt_tonemap(const double *rgb_sum, uint64_t n_pixels, double divisor, uint8_t *rgb8)
{
    if (!rgb_sum || !rgb8) {
        rt_host::set_last_error("rt_tonemap: null argument");
        return RT_ERR_INVALID;
    }
    for (uint64_t i = 0; i < n_pixels * 3; ++i) {
        double c = std::sqrt(rgb_sum[i] / divisor);
        double v = c * 255.0;
        v = (v != v) ? 255.0 : (v < 255.0 ? v : 255.0);
        rgb8[i] = v <= 0.0 ? 0 : (uint8_t)v;
    }
    return RT_OK;
}
