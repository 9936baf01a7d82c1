// rt_common.cpp — pieces of the C ABI that need no device: last-error slot,
// shard geometry, tone mapping, the PNG file.
#include "../../include/rt_api.h"
#include "rt_shard.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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
// PNG: signature, IHDR, one IDAT (zlib stream of the rows, each behind filter byte 0), IEND
namespace {
void png_chunk(std::vector<unsigned char> &out, const char type[4], const unsigned char *data, size_t n)
{
    auto be32 = [&](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) out.push_back((unsigned char)(v >> s));
    };
    be32((uint32_t)n);
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    be32((uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}
} // namespace

extern "C" int rt_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height)
{
    if (!path || !rgb8 || width == 0 || height == 0) {
        rt_host::set_last_error("rt_write_png: null argument or empty image");
        return RT_ERR_INVALID;
    }
    const size_t row = (size_t)width * 3;
    std::vector<unsigned char> raw((row + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw[(row + 1) * y] = 0; // filter type None
        std::memcpy(&raw[(row + 1) * y + 1], rgb8 + row * y, row);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        rt_host::set_last_error("rt_write_png: zlib compress2 failed");
        return RT_ERR_IO;
    }
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    unsigned char ihdr[13] = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8),
                              (unsigned char)width, (unsigned char)(height >> 24), (unsigned char)(height >> 16),
                              (unsigned char)(height >> 8), (unsigned char)height,
                              8 /* bit depth */, 2 /* truecolour */, 0, 0, 0};
    png_chunk(out, "IHDR", ihdr, sizeof(ihdr));
    png_chunk(out, "IDAT", z.data(), zn);
    png_chunk(out, "IEND", nullptr, 0);
    FILE *f = std::fopen(path, "wb");
    if (!f) {
        rt_host::set_last_error(std::string("rt_write_png: cannot open ") + path);
        return RT_ERR_IO;
    }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    if (std::fclose(f) != 0 || !ok) {
        rt_host::set_last_error(std::string("rt_write_png: short write to ") + path);
        return RT_ERR_IO;
    }
    return RT_OK;
}

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
