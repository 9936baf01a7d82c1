// host_jpeg.cpp — baseline JPEG (ITU-T T.81 sequential DCT, Huffman, 8-bit) to RGB8.
//
// The reference decodes `image "…"` textures with image 0.14 / jpeg-decoder 0.1.22 (sdl.rs:52,
// texture.rs:60-66), crates that are not under /root/reference.  This is the published algorithm, not
// their code: marker parsing (B.2), Huffman decoding (F.2.2), dequantisation, an integer
// Loeffler-Ligtenberg-Moschytz inverse DCT, triangle-filter chroma upsampling and the JFIF YCbCr
// matrix.  Decoders differ in the rounding of the last three steps, so texel values may differ from
// the reference's by a unit or two: **parity unpinned** for texel bytes; both sides of
// the parity tests read the bytes this decoder produced.
// Not supported (reported, never guessed at): progressive and arithmetic-coded files, 12-bit samples,
// CMYK.
#include "host_jpeg.h"

#include <cstring>

namespace rt_host {

namespace {

struct HuffTable {
    bool present = false;
    uint8_t symbols[256];
    int mincode[17], maxcode[18], valptr[17]; // F.2.2.3, per code length 1..16
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0; // in 8x8 blocks, padded to whole MCUs
    int pred = 0;
    std::vector<uint8_t> plane; // blocks_w*8 x blocks_h*8
};

const int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Decoder {
    const uint8_t *d;
    size_t n, pos = 0;
    std::string err;

    int width = 0, height = 0, ncomp = 0;
    Component comp[3];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    bool adobe_rgb = false; // APP14 transform 0 with three components: samples are RGB already

    // entropy-coded segment reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;

    bool fail(const std::string &m)
    {
        if (err.empty()) err = m;
        return false;
    }
    int u8()
    {
        if (pos >= n) return -1;
        return d[pos++];
    }
    int u16()
    {
        if (pos + 2 > n) return -1;
        int v = (d[pos] << 8) | d[pos + 1];
        pos += 2;
        return v;
    }

    // ---- tables (B.2.4)
    bool read_dqt(size_t end)
    {
        while (pos < end) {
            int pt = u8();
            int prec = pt >> 4, id = pt & 15;
            if (id > 3 || prec > 1) return fail("bad quantisation table header");
            for (int k = 0; k < 64; ++k) {
                int v = prec ? u16() : u8();
                if (v < 0) return fail("truncated quantisation table");
                qt[id][kZigzag[k]] = (uint16_t)v;
            }
            qt_present[id] = true;
        }
        return true;
    }
    bool read_dht(size_t end)
    {
        while (pos < end) {
            int tc = u8();
            int cls = tc >> 4, id = tc & 15;
            if (cls > 1 || id > 3) return fail("bad Huffman table header");
            int counts[17];
            int total = 0;
            for (int l = 1; l <= 16; ++l) {
                counts[l] = u8();
                if (counts[l] < 0) return fail("truncated Huffman table");
                total += counts[l];
            }
            if (total > 256 || pos + (size_t)total > n) return fail("bad Huffman table size");
            HuffTable &t = cls ? ac[id] : dc[id];
            std::memcpy(t.symbols, d + pos, (size_t)total);
            pos += (size_t)total;
            int code = 0, k = 0;
            for (int l = 1; l <= 16; ++l) {
                t.valptr[l] = k;
                t.mincode[l] = code;
                code += counts[l];
                k += counts[l];
                t.maxcode[l] = counts[l] ? code - 1 : -1;
                code <<= 1;
            }
            t.maxcode[17] = 0x7FFFFFFF;
            t.present = true;
        }
        return true;
    }
    bool read_sof(int marker)
    {
        if (marker != 0xC0 && marker != 0xC1)
            return fail(marker == 0xC2 ? "progressive JPEG files are not supported (re-save as baseline)"
                                       : "this JPEG coding process is not supported (baseline sequential only)");
        int prec = u8();
        height = u16();
        width = u16();
        ncomp = u8();
        if (prec != 8) return fail("only 8-bit JPEG samples are supported");
        if (width <= 0 || height <= 0) return fail("bad JPEG dimensions");
        if ((uint64_t)width * (uint64_t)height > (1ull << 28)) return fail("JPEG image larger than 2^28 pixels");
        if (ncomp != 1 && ncomp != 3) return fail("only greyscale and three-component JPEG files are supported");
        for (int c = 0; c < ncomp; ++c) {
            comp[c].id = u8();
            int hv = u8();
            comp[c].h = hv >> 4;
            comp[c].v = hv & 15;
            comp[c].tq = u8();
            if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3 || comp[c].tq < 0)
                return fail("unsupported JPEG sampling factors");
        }
        if (ncomp == 3 && (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1))
            return fail("unsupported JPEG chroma sampling (chroma must be 1x1)");
        return true;
    }

    // ---- bits (F.2.2.5): 0xFF00 is a stuffed 0xFF, any other 0xFFxx ends the segment
    void fill()
    {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    int b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) {
                        pos += 2;
                    } else {
                        hit_marker = true; // leave the marker in place; feed zeros from here on
                        b = 0;
                    }
                } else {
                    ++pos;
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int bits(int k)
    {
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k;
        bitcnt -= k;
        return v;
    }
    int decode_symbol(const HuffTable &t)
    {
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | bits(1);
            if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.symbols[t.valptr[l] + code - t.mincode[l]];
        }
        return -1;
    }
    static int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; } // F.2.2.1

    // ---- one 8x8 block: coefficients, dequantised, in natural order
    bool decode_block(Component &c, int *coef)
    {
        std::memset(coef, 0, 64 * sizeof(int));
        int s = decode_symbol(dc[c.td]);
        if (s < 0 || s > 11) return fail("bad DC code in JPEG data");
        c.pred += extend(bits(s), s);
        coef[0] = c.pred * qt[c.tq][0];
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(ac[c.ta]);
            if (rs < 0) return fail("bad AC code in JPEG data");
            int r = rs >> 4;
            s = rs & 15;
            if (s == 0) {
                if (r != 15) break; // end of block
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return fail("AC run past the end of a JPEG block");
            coef[kZigzag[k]] = extend(bits(s), s) * qt[c.tq][kZigzag[k]];
            ++k;
        }
        return true;
    }

    // ---- inverse DCT: two passes of the 8-point LLM butterfly in 12-bit fixed point
    static inline int fx(double x) { return (int)(x * 4096.0 + 0.5); }
    static void idct8(const int *s, int stride, int *even, int *odd)
    {
        // even part -> even[0..3], odd part -> odd[0..3]; output k = even[k] +- odd[3-k]
        int p2 = s[2 * stride], p3 = s[6 * stride];
        int p1 = (p2 + p3) * fx(0.5411961);
        int t2 = p1 + p3 * fx(-1.847759065);
        int t3 = p1 + p2 * fx(0.765366865);
        int t0 = (s[0] + s[4 * stride]) * 4096, t1 = (s[0] - s[4 * stride]) * 4096;
        even[0] = t0 + t3;
        even[3] = t0 - t3;
        even[1] = t1 + t2;
        even[2] = t1 - t2;
        int a0 = s[7 * stride], a1 = s[5 * stride], a2 = s[3 * stride], a3 = s[1 * stride];
        int q3 = a0 + a2, q4 = a1 + a3, q1 = a0 + a3, q2 = a1 + a2;
        int q5 = (q3 + q4) * fx(1.175875602);
        a0 *= fx(0.298631336);
        a1 *= fx(2.053119869);
        a2 *= fx(3.072711026);
        a3 *= fx(1.501321110);
        q1 = q5 + q1 * fx(-0.899976223);
        q2 = q5 + q2 * fx(-2.562915447);
        q3 *= fx(-1.961570560);
        q4 *= fx(-0.390180644);
        odd[3] = a3 + q1 + q4;
        odd[2] = a2 + q2 + q3;
        odd[1] = a1 + q2 + q4;
        odd[0] = a0 + q1 + q3;
    }
    static inline uint8_t clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v); }
    static void idct_block(const int *coef, uint8_t *out, int out_stride)
    {
        int tmp[64], e[4], o[4];
        for (int x = 0; x < 8; ++x) { // columns, keep 2 extra bits
            idct8(coef + x, 8, e, o);
            for (int k = 0; k < 4; ++k) {
                tmp[k * 8 + x] = (e[k] + o[3 - k] + 512) >> 10;
                tmp[(7 - k) * 8 + x] = (e[k] - o[3 - k] + 512) >> 10;
            }
        }
        for (int y = 0; y < 8; ++y) { // rows, level shift by 128
            idct8(tmp + y * 8, 1, e, o);
            const int bias = 65536 + (128 << 17);
            for (int k = 0; k < 4; ++k) {
                out[y * out_stride + k] = clamp8((e[k] + o[3 - k] + bias) >> 17);
                out[y * out_stride + 7 - k] = clamp8((e[k] - o[3 - k] + bias) >> 17);
            }
        }
    }

    // ---- scan (B.2.3, E.2)
    bool read_sos_and_decode()
    {
        int ns = u8();
        if (ns != ncomp) return fail("multi-scan baseline JPEG files are not supported");
        for (int i = 0; i < ns; ++i) {
            int cs = u8(), tt = u8();
            int idx = -1;
            for (int c = 0; c < ncomp; ++c)
                if (comp[c].id == cs) idx = c;
            if (idx != i) return fail("unexpected component order in JPEG scan");
            comp[idx].td = tt >> 4;
            comp[idx].ta = tt & 15;
            if (comp[idx].td > 3 || comp[idx].ta > 3 || !dc[comp[idx].td].present || !ac[comp[idx].ta].present)
                return fail("JPEG scan refers to a missing Huffman table");
            if (!qt_present[comp[idx].tq]) return fail("JPEG scan refers to a missing quantisation table");
        }
        u8(); // Ss
        u8(); // Se
        u8(); // Ah/Al
        // A scan with one component is not interleaved: its MCU is one 8x8 block in raster order, whatever the
        // sampling factors in the frame header say (T.81 A.2.2)
        if (ncomp == 1) comp[0].h = comp[0].v = 1;
        const int hmax = comp[0].h, vmax = comp[0].v;
        const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
        const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
        for (int c = 0; c < ncomp; ++c) {
            comp[c].blocks_w = mcus_x * comp[c].h;
            comp[c].blocks_h = mcus_y * comp[c].v;
            comp[c].plane.assign((size_t)comp[c].blocks_w * 8 * comp[c].blocks_h * 8, 0);
            comp[c].pred = 0;
        }
        int coef[64];
        int until_restart = restart_interval;
        int expect_rst = 0;
        for (int my = 0; my < mcus_y; ++my) {
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (restart_interval && until_restart == 0) {
                    // E.2.4: byte-align, expect RSTm, reset predictors
                    bitbuf = 0;
                    bitcnt = 0;
                    hit_marker = false;
                    while (pos + 1 < n && !(d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7)) ++pos;
                    if (pos + 1 >= n || d[pos + 1] != 0xD0 + expect_rst) return fail("missing restart marker in JPEG data");
                    pos += 2;
                    expect_rst = (expect_rst + 1) & 7;
                    for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                    until_restart = restart_interval;
                }
                for (int c = 0; c < ncomp; ++c) {
                    Component &cc = comp[c];
                    for (int by = 0; by < cc.v; ++by)
                        for (int bx = 0; bx < cc.h; ++bx) {
                            if (!decode_block(cc, coef)) return false;
                            const int stride = cc.blocks_w * 8;
                            uint8_t *dst = cc.plane.data() + (size_t)((my * cc.v + by) * 8) * stride + (size_t)(mx * cc.h + bx) * 8;
                            idct_block(coef, dst, stride);
                        }
                }
                --until_restart;
            }
        }
        return true;
    }

    bool run()
    {
        if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("not a JPEG file");
        pos = 2;
        bool have_sof = false;
        for (;;) {
            int b = u8();
            if (b < 0) return fail("JPEG file ends before its image data");
            if (b != 0xFF) continue;
            int m = u8();
            while (m == 0xFF) m = u8();
            if (m < 0) return fail("JPEG file ends before its image data");
            if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (m == 0xD9) return fail("JPEG file holds no scan");
            int len = u16();
            if (len < 2 || pos + (size_t)len - 2 > n) return fail("bad JPEG segment length");
            size_t end = pos + (size_t)len - 2;
            if (m == 0xDB) {
                if (!read_dqt(end)) return false;
            } else if (m == 0xC4) {
                if (!read_dht(end)) return false;
            } else if (m == 0xDD) {
                restart_interval = u16();
            } else if (m == 0xEE && len >= 14 && std::memcmp(d + pos, "Adobe", 5) == 0) {
                adobe_rgb = d[pos + 11] == 0;
            } else if ((m >= 0xC0 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                if (!read_sof(m)) return false;
                have_sof = true;
            } else if (m == 0xDA) {
                if (!have_sof) return fail("JPEG scan before its frame header");
                return read_sos_and_decode();
            }
            pos = end;
        }
    }

    // ---- chroma to full resolution: the triangle filter (3/4 nearest + 1/4 next nearest sample) along each
    // subsampled axis, then the JFIF matrix
    void to_rgb(std::vector<uint8_t> &rgb) const
    {
        rgb.resize((size_t)width * height * 3);
        const Component &Y = comp[0];
        const int ys = Y.blocks_w * 8;
        if (ncomp == 1) {
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x) {
                    uint8_t v = Y.plane[(size_t)y * ys + x];
                    uint8_t *o = &rgb[((size_t)y * width + x) * 3];
                    o[0] = o[1] = o[2] = v;
                }
            return;
        }
        const int sh = Y.h, sv = Y.v; // chroma is 1x1: luma factors are the subsampling ratios
        const int cw = (width + sh - 1) / sh, ch = (height + sv - 1) / sv;
        std::vector<int> row_cb((size_t)cw), row_cr((size_t)cw); // vertically filtered, scaled by 4 (or 1)
        for (int y = 0; y < height; ++y) {
            const int cy = y / sv;
            int fy = cy;
            if (sv == 2) fy = (y & 1) ? (cy + 1 < ch ? cy + 1 : cy) : (cy > 0 ? cy - 1 : cy);
            for (int c = 1; c <= 2; ++c) {
                const Component &C = comp[c];
                const int cs = C.blocks_w * 8;
                const uint8_t *near = &C.plane[(size_t)cy * cs], *far = &C.plane[(size_t)fy * cs];
                std::vector<int> &row = c == 1 ? row_cb : row_cr;
                for (int x = 0; x < cw; ++x) row[(size_t)x] = sv == 2 ? 3 * near[x] + far[x] : 4 * near[x];
            }
            for (int x = 0; x < width; ++x) {
                int cb, cr;
                if (sh == 2) {
                    const int cx = x / 2;
                    const int fx2 = (x & 1) ? (cx + 1 < cw ? cx + 1 : cx) : (cx > 0 ? cx - 1 : cx);
                    cb = (3 * row_cb[(size_t)cx] + row_cb[(size_t)fx2] + 8) >> 4;
                    cr = (3 * row_cr[(size_t)cx] + row_cr[(size_t)fx2] + 8) >> 4;
                } else {
                    cb = (row_cb[(size_t)x] + 2) >> 2;
                    cr = (row_cr[(size_t)x] + 2) >> 2;
                }
                const int yy = Y.plane[(size_t)y * ys + x];
                uint8_t *o = &rgb[((size_t)y * width + x) * 3];
                if (adobe_rgb) {
                    o[0] = (uint8_t)yy;
                    o[1] = (uint8_t)cb;
                    o[2] = (uint8_t)cr;
                } else {
                    cb -= 128;
                    cr -= 128;
                    // R = Y + 1.402 Cr, G = Y - 0.344136 Cb - 0.714136 Cr, B = Y + 1.772 Cb, in 16.16
                    o[0] = clamp8(yy + ((91881 * cr + 32768) >> 16));
                    o[1] = clamp8(yy - ((22554 * cb + 46802 * cr + 32768) >> 16));
                    o[2] = clamp8(yy + ((116130 * cb + 32768) >> 16));
                }
            }
        }
    }
};

} // namespace

bool decode_jpeg(const uint8_t *bytes, size_t n_bytes, int &width, int &height, std::vector<uint8_t> &rgb, std::string &err)
{
    Decoder dec;
    dec.d = bytes;
    dec.n = n_bytes;
    std::memset(dec.qt, 0, sizeof(dec.qt));
    if (!dec.run()) {
        err = dec.err.empty() ? "could not decode JPEG data" : dec.err;
        return false;
    }
    width = dec.width;
    height = dec.height;
    dec.to_rgb(rgb);
    return true;
}

} // namespace rt_host
