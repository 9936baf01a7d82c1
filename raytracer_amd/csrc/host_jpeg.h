// host_jpeg.h — baseline JPEG decoder for `image "…"` textures (host side)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rt_host {

// 8-bit baseline (sequential DCT, Huffman) greyscale or YCbCr JPEG -> interleaved RGB8, rows top to bottom.
// Returns false with a message for files it does not cover (progressive, arithmetic, 12-bit, CMYK) or cannot parse.
bool decode_jpeg(const uint8_t *bytes, size_t n_bytes, int &width, int &height, std::vector<uint8_t> &rgb, std::string &err);

} // namespace rt_host
