// rt_jpeg.cpp — baseline JPEG decoder for texture maps.
//
// The reference loads the "map" of a TextureMap with image::open (src/materials.rs:37-39), i.e. the
// image crate's jpeg-decoder, and reads texels as 8-bit RGB (src/materials.rs:160-170).  This file
// decodes the same class of files — baseline / extended-sequential Huffman DCT, 8-bit, one or three
// components — with the arithmetic of the IJG library's default path: the "islow" integer inverse DCT
// (Loeffler-Ligtenberg-Moschytz, 13-bit constants, 2 extra bits between the passes) and the 16-bit
// fixed-point YCbCr -> RGB tables.  For files without chroma subsampling (the bundled earth map) the
// output is therefore byte-identical to libjpeg / libjpeg-turbo.  Subsampled chroma is upsampled by
// replication (IJG's default is a triangle filter, so such files differ from it at chroma edges).
// Progressive, arithmetic-coded, lossless and 12-bit files are refused.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "rt_internal.h"

namespace rt {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t counts[17] = {0};
    uint8_t symbols[256] = {0};
    int32_t mincode[17], maxcode[18], valptr[17];
    void prepare() {
        int32_t code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len];
            k += counts[len];
            maxcode[len] = counts[len] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;  // allocated block grid (whole MCUs)
    int stride = 0;
    int64_t pred = 0;
    std::vector<uint8_t> plane;
};

struct Decoder {
    const uint8_t* data;
    size_t len, pos = 0;
    uint16_t quant[4][64];
    bool quant_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1;
    Component comp[3];
    int restart_interval = 0;
    bool adobe_rgb = false;
    bool frame_seen = false;
    // bit reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    const char* error = nullptr;

    bool fail(const char* msg) {
        if (!error) error = msg;
        return false;
    }
    int u8() { return pos < len ? data[pos++] : (fail("unexpected end of file"), 0); }
    int u16() {
        int a = u8();
        return (a << 8) | u8();
    }

    void fill_bits() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < len) {
                b = data[pos];
                if (b == 0xff) {
                    int b2 = pos + 1 < len ? data[pos + 1] : 0xd9;
                    if (b2 == 0) {
                        pos += 2;
                    } else {
                        hit_marker = true;  // leave the marker in place; feed zero bits from here on
                        b = 0;
                    }
                } else {
                    ++pos;
                }
            } else {
                hit_marker = true;
            }
            bitbuf |= uint32_t(b) << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bitcnt < n) fill_bits();
        int v = int(bitbuf >> (32 - n));
        bitbuf <<= n;
        bitcnt -= n;
        return v;
    }
    int decode_symbol(const Huff& h) {
        int32_t code = 0;
        for (int lenb = 1; lenb <= 16; ++lenb) {
            code = (code << 1) | get_bits(1);
            if (h.maxcode[lenb] >= 0 && code <= h.maxcode[lenb] && code >= h.mincode[lenb])
                return h.symbols[h.valptr[lenb] + (code - h.mincode[lenb])];
        }
        fail("bad Huffman code");
        return 0;
    }
    static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

    bool parse_dqt() {
        int l = u16() - 2;
        while (l > 0 && !error) {
            int pt = u8();
            int prec = pt >> 4, t = pt & 15;
            if (t > 3 || prec > 1) return fail("bad quantisation table header");
            for (int i = 0; i < 64; ++i) quant[t][i] = uint16_t(prec ? u16() : u8());  // zigzag order
            quant_present[t] = true;
            l -= 1 + 64 * (prec + 1);
        }
        return !error;
    }
    bool parse_dht() {
        int l = u16() - 2;
        while (l > 0 && !error) {
            int tt = u8();
            int cls = tt >> 4, t = tt & 15;
            if (cls > 1 || t > 3) return fail("bad Huffman table header");
            Huff& h = cls ? ac[t] : dc[t];
            int total = 0;
            for (int i = 1; i <= 16; ++i) total += (h.counts[i] = uint8_t(u8()));
            if (total > 256) return fail("bad Huffman table");
            for (int i = 0; i < total; ++i) h.symbols[i] = uint8_t(u8());
            h.present = true;
            h.prepare();
            l -= 17 + total;
        }
        return !error;
    }
    bool parse_sof() {
        u16();
        if (u8() != 8) return fail("only 8-bit samples are supported");
        height = u16();
        width = u16();
        ncomp = u8();
        if (width <= 0 || height <= 0) return fail("empty image");
        if (int64_t(width) * height > (int64_t(1) << 28)) return fail("image larger than 2^28 pixels");
        if (ncomp != 1 && ncomp != 3) return fail("only 1- or 3-component images are supported");
        for (int i = 0; i < ncomp; ++i) {
            comp[i].id = u8();
            int hv = u8();
            comp[i].h = hv >> 4;
            comp[i].v = hv & 15;
            comp[i].tq = u8();
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3)
                return fail("bad component description");
            if (comp[i].h > hmax) hmax = comp[i].h;
            if (comp[i].v > vmax) vmax = comp[i].v;
        }
        const int mcus_x = (width + 8 * hmax - 1) / (8 * hmax), mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; ++i) {
            comp[i].blocks_w = mcus_x * comp[i].h;
            comp[i].blocks_h = mcus_y * comp[i].v;
            comp[i].stride = comp[i].blocks_w * 8;
            comp[i].plane.assign(size_t(comp[i].stride) * comp[i].blocks_h * 8, 0);
        }
        frame_seen = true;
        return !error;
    }

    // jidctint.c "islow": two passes of the LL&M butterfly over dequantised coefficients.
    // 64-bit like IJG's JLONG: valid streams stay far inside 32 bits, damaged ones must not overflow
    static inline int64_t descale(int64_t x, int n) { return (x + (int64_t(1) << (n - 1))) >> n; }
    static void idct_pass(const int64_t in[8], int64_t out[8], int shift) {
        const int64_t F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                      F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
        int64_t z2 = in[2], z3 = in[6];
        int64_t z1 = (z2 + z3) * F0_541;
        int64_t tmp2 = z1 + z3 * (-F1_847);
        int64_t tmp3 = z1 + z2 * F0_765;
        z2 = in[0];
        z3 = in[4];
        int64_t tmp0 = (z2 + z3) * 8192;
        int64_t tmp1 = (z2 - z3) * 8192;
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[7];
        tmp1 = in[5];
        tmp2 = in[3];
        tmp3 = in[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298;
        tmp1 *= F2_053;
        tmp2 *= F3_072;
        tmp3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        out[0] = descale(tmp10 + tmp3, shift);
        out[7] = descale(tmp10 - tmp3, shift);
        out[1] = descale(tmp11 + tmp2, shift);
        out[6] = descale(tmp11 - tmp2, shift);
        out[2] = descale(tmp12 + tmp1, shift);
        out[5] = descale(tmp12 - tmp1, shift);
        out[3] = descale(tmp13 + tmp0, shift);
        out[4] = descale(tmp13 - tmp0, shift);
    }
    static void idct_block(const int64_t coef[64], uint8_t* dst, int stride) {
        int64_t ws[64];
        for (int c = 0; c < 8; ++c) {  // pass 1: columns, results scaled up by 2 bits
            int64_t in[8], out[8];
            for (int r = 0; r < 8; ++r) in[r] = coef[r * 8 + c];
            idct_pass(in, out, 13 - 2);
            for (int r = 0; r < 8; ++r) ws[r * 8 + c] = out[r];
        }
        for (int r = 0; r < 8; ++r) {  // pass 2: rows, remove the scaling and the factor 8 of the 2-D transform
            int64_t out[8];
            idct_pass(ws + r * 8, out, 13 + 2 + 3);
            for (int c = 0; c < 8; ++c) {
                const int64_t v = out[c] + 128;
                dst[r * stride + c] = uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }

    bool decode_block(Component& c, int bx, int by) {
        int64_t coef[64];
        std::memset(coef, 0, sizeof coef);
        const uint16_t* q = quant[c.tq];
        int s = decode_symbol(dc[c.td]);
        if (s > 11) return fail("bad DC difference");
        int diff = s ? extend(get_bits(s), s) : 0;
        c.pred += diff;
        coef[0] = c.pred * int64_t(q[0]);
        const Huff& h = ac[c.ta];
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(h);
            int r = rs >> 4;
            s = rs & 15;
            if (s == 0) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return fail("AC coefficient index out of range");
            coef[kZigzag[k]] = int64_t(extend(get_bits(s), s)) * int64_t(q[k]);
            ++k;
        }
        if (error) return false;
        idct_block(coef, c.plane.data() + size_t(by) * 8 * c.stride + size_t(bx) * 8, c.stride);
        return true;
    }

    bool restart() {
        bitbuf = 0;
        bitcnt = 0;
        hit_marker = false;
        // a restart marker follows, possibly after fill bytes
        while (pos + 1 < len && !(data[pos] == 0xff && data[pos + 1] >= 0xd0 && data[pos + 1] <= 0xd7)) {
            if (data[pos] == 0xff && data[pos + 1] != 0xff && data[pos + 1] != 0) return fail("restart marker missing");
            ++pos;
        }
        if (pos + 1 >= len) return fail("restart marker missing");
        pos += 2;
        for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
        return true;
    }

    bool parse_sos() {
        if (!frame_seen) return fail("scan before frame header");
        u16();
        const int ns = u8();
        if (ns < 1 || ns > ncomp) return fail("bad scan header");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int cid = u8(), t = u8();
            sc[i] = nullptr;
            for (int j = 0; j < ncomp; ++j)
                if (comp[j].id == cid) sc[i] = &comp[j];
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->td = t >> 4;
            sc[i]->ta = t & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3 || !dc[sc[i]->td].present || !ac[sc[i]->ta].present)
                return fail("scan uses an undefined Huffman table");
            if (!quant_present[sc[i]->tq]) return fail("scan uses an undefined quantisation table");
        }
        const int ss = u8(), se = u8(), ahal = u8();
        if (ss != 0 || se != 63 || ahal != 0) return fail("not a sequential scan");
        if (error) return false;
        bitbuf = 0;
        bitcnt = 0;
        hit_marker = false;
        for (int i = 0; i < ncomp; ++i) comp[i].pred = 0;
        int mcus_x, mcus_y;
        if (ns == 1) {  // non-interleaved: the component's own block grid, clipped to the image
            Component& c = *sc[0];
            const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
            mcus_x = (cw + 7) / 8;
            mcus_y = (ch + 7) / 8;
        } else {
            mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
            mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
        }
        int until_restart = restart_interval;
        for (int my = 0; my < mcus_y; ++my) {
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (restart_interval && until_restart == 0) {
                    if (!restart()) return false;
                    until_restart = restart_interval;
                }
                if (ns == 1) {
                    if (!decode_block(*sc[0], mx, my)) return false;
                } else {
                    for (int i = 0; i < ns; ++i)
                        for (int v = 0; v < sc[i]->v; ++v)
                            for (int h = 0; h < sc[i]->h; ++h)
                                if (!decode_block(*sc[i], mx * sc[i]->h + h, my * sc[i]->v + v)) return false;
                }
                --until_restart;
            }
        }
        return true;
    }

    bool run() {
        if (u8() != 0xff || u8() != 0xd8) return fail("not a JPEG file (no SOI marker)");
        bool scans = false;
        while (!error) {
            int b = u8();
            if (error) break;
            if (b != 0xff) continue;  // resynchronise on the next marker (also skips entropy-coded leftovers)
            int m = u8();
            while (m == 0xff && !error) m = u8();
            if (m == 0 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
            if (m == 0xd9) break;
            switch (m) {
                case 0xdb: parse_dqt(); break;
                case 0xc4: parse_dht(); break;
                case 0xc0:
                case 0xc1: parse_sof(); break;
                case 0xc2: return fail("progressive JPEG is not supported");
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    return fail("this JPEG coding process is not supported (only baseline / sequential Huffman)");
                case 0xdd: u16(); restart_interval = u16(); break;
                case 0xda:
                    if (!parse_sos()) return false;
                    scans = true;
                    break;
                case 0xee: {  // Adobe: transform 0 with three components means the data is RGB already
                    int l = u16();
                    size_t end = pos + size_t(l) - 2;
                    if (l >= 14 && pos + 12 <= len && std::memcmp(data + pos, "Adobe", 5) == 0) adobe_rgb = data[pos + 11] == 0;
                    pos = end;
                    break;
                }
                default: {  // APPn, COM and anything else with a length
                    int l = u16();
                    if (l < 2) return fail("bad segment length");
                    pos += size_t(l) - 2;
                }
            }
            if (pos > len) return fail("unexpected end of file");
        }
        if (error) return false;
        if (!scans) return fail("no image data");
        return true;
    }

    void to_rgb(uint8_t* out) const {
        if (ncomp == 1) {
            for (int y = 0; y < height; ++y)
                for (int x = 0; x < width; ++x) {
                    uint8_t g = comp[0].plane[size_t(y) * comp[0].stride + x];
                    uint8_t* o = out + (size_t(y) * width + x) * 3;
                    o[0] = o[1] = o[2] = g;
                }
            return;
        }
        // jdcolor.c: 16-bit fixed point, FIX(x) = x * 65536 + 0.5
        int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
        for (int i = 0; i < 256; ++i) {
            const int32_t x = i - 128;
            cr_r[i] = (91881 * x + 32768) >> 16;
            cb_b[i] = (116130 * x + 32768) >> 16;
            cr_g[i] = -46802 * x;
            cb_g[i] = -22554 * x + 32768;
        }
        auto clamp = [](int32_t v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        for (int y = 0; y < height; ++y) {
            for (int x = 0; x < width; ++x) {
                uint8_t s[3];
                for (int i = 0; i < 3; ++i) {
                    const int cx = x * comp[i].h / hmax, cy = y * comp[i].v / vmax;
                    s[i] = comp[i].plane[size_t(cy) * comp[i].stride + cx];
                }
                uint8_t* o = out + (size_t(y) * width + x) * 3;
                if (adobe_rgb) {
                    o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
                } else {
                    const int32_t Y = s[0];
                    o[0] = clamp(Y + cr_r[s[2]]);
                    o[1] = clamp(Y + ((cb_g[s[1]] + cr_g[s[2]]) >> 16));
                    o[2] = clamp(Y + cb_b[s[1]]);
                }
            }
        }
    }
};

}  // namespace
}  // namespace rt

extern "C" {

int rt_decode_jpeg(const uint8_t* bytes, size_t len, uint8_t** rgb8, uint32_t* width, uint32_t* height) {
    if (!bytes || !rgb8 || !width || !height) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rt_decode_jpeg: null argument");
    *rgb8 = nullptr;
    rt::Decoder* d = new (std::nothrow) rt::Decoder();
    if (!d) return rt::fail(RT_ERR_LIMIT, "rt_decode_jpeg: out of memory");
    d->data = bytes;
    d->len = len;
    std::memset(d->quant, 0, sizeof d->quant);
    bool ok = false;
    try {
        ok = d->run();
    } catch (const std::bad_alloc&) {  // nothing may unwind across the C boundary
        d->fail("out of memory");
    }
    if (!ok) {
        int rc = rt::fail(RT_ERR_UNSUPPORTED, "cannot open file: JPEG decode failed: %s", d->error ? d->error : "?");
        delete d;
        return rc;
    }
    uint8_t* out = static_cast<uint8_t*>(std::malloc(size_t(d->width) * d->height * 3));
    if (!out) {
        delete d;
        return rt::fail(RT_ERR_LIMIT, "rt_decode_jpeg: out of memory");
    }
    d->to_rgb(out);
    *rgb8 = out;
    *width = uint32_t(d->width);
    *height = uint32_t(d->height);
    delete d;
    return RT_OK;
}

int rt_load_jpeg(const char* path, uint8_t** rgb8, uint32_t* width, uint32_t* height) {
    if (!path) return rt::fail(RT_ERR_INVALID_ARGUMENT, "rt_load_jpeg: null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return rt::fail(RT_ERR_IO, "cannot open file %s", path);  // src/materials.rs:38 panics with "cannot open file"
    std::vector<uint8_t> buf;
    uint8_t chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    std::fclose(f);
    return rt_decode_jpeg(buf.data(), buf.size(), rgb8, width, height);
}

void rt_free(void* p) { std::free(p); }

}  // extern "C"
