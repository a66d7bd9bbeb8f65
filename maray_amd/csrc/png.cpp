// png.cpp — PNG writer/reader used by gen() and the CLI (product code).
//
//   maray_png_write <- `img.save(file)`                 src/lib.rs:1207,1212
//   maray_png_read  <- `image::open(file).to_rgb8()`    examples/maray.rs:60-65
//
// The reference delegates to the `image`/`png` crates; PNG container bytes are
// an encoder choice, so parity is defined on the decoded RGB8 raster
// (SURVEY.md §8(c)).  Writer: 8-bit RGB, filter 0, zlib level 6.  Reader:
// gray / RGB / palette / gray+alpha / RGBA at 8 or 16 bits (and 1/2/4-bit gray
// or palette), interlaced or not, converted like `to_rgb8` (alpha dropped, 16
// bits rounded to 8).  The other texture formats: image.cpp.
//
// Both entry points take sizes from outside (the caller; a texture file's IHDR):
// every product of sizes is checked in 64 bits against MARAY_PNG_MAX_BYTES before
// anything is allocated or indexed, and no exception crosses the C boundary.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "backend.hpp"
#include "maray_hip.h"

using namespace maray;

namespace {

void put_be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

void chunk(std::vector<uint8_t> &out, const char *type, const uint8_t *data, size_t n)
{
    put_be32(out, (uint32_t)n);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4));
    put_be32(out, crc);
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// Raster sizes this library will read or write: 2^20 pixels a side (the evaluators' own limit,
// MARAY_DOMAIN_MAX) and 16 GiB of filtered scanlines.  A crafted IHDR beyond that is rejected, not allocated.
const uint64_t MARAY_PNG_MAX_BYTES = 1ull << 34;

template <typename F>
int png_guard(F f)
{
    try { return f(); }
    catch (const Error &e) { set_last_error(e.msg); return e.code; }
    catch (const std::bad_alloc &) { set_last_error("out of memory"); return MARAY_E_INTERNAL; }
    catch (const std::exception &e) { set_last_error(e.what()); return MARAY_E_INTERNAL; }
    catch (...) { set_last_error("unknown error"); return MARAY_E_INTERNAL; }
}

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

}   // namespace

extern "C" int maray_png_write(const char *path, const uint8_t *rgb8, uint32_t w, uint32_t h)
{
    return png_guard([&]() -> int {
    if (!path || (!rgb8 && (uint64_t)w * h)) { set_last_error("null argument"); return MARAY_E_ARG; }
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX || ((uint64_t)w * 3 + 1) * h > MARAY_PNG_MAX_BYTES) {
        set_last_error("image too large for the PNG writer"); return MARAY_E_LIMIT;
    }
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)w * 3 + 1) * h);
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);   // filter type None
        raw.insert(raw.end(), rgb8 + (size_t)y * w * 3, rgb8 + (size_t)(y + 1) * w * 3);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) { set_last_error("zlib compress failed"); return MARAY_E_INTERNAL; }
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit, truecolour
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    for (size_t off = 0; off < zn || off == 0; off += (size_t)1 << 30) {          // a chunk's length field holds 31 bits
        chunk(out, "IDAT", z.data() + off, std::min<size_t>((size_t)1 << 30, zn - off));
        if (zn == 0) break;
    }
    chunk(out, "IEND", nullptr, 0);
    FILE *f = fopen(path, "wb");
    if (!f) { set_last_error(std::string("cannot create ") + path); return MARAY_E_IO; }
    size_t n = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    if (n != out.size()) { set_last_error("short write"); return MARAY_E_IO; }
    return MARAY_OK;
    });
}

namespace maray {

int read_whole_file(const char *path, std::vector<uint8_t> &b)
{
    FILE *f = fopen(path, "rb");
    if (!f) { set_last_error(std::string("cannot open ") + path); return MARAY_E_IO; }
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) b.insert(b.end(), tmp, tmp + n);
    fclose(f);
    return MARAY_OK;
}

// PNG -> RGB8 like `to_rgb8` (alpha dropped, 16 bits rounded to 8, grey replicated); interlaced (Adam7) files included.
// *rgb8_out is malloc'ed.
int png_decode(const std::vector<uint8_t> &b, uint8_t **rgb8_out, uint32_t *w_out, uint32_t *h_out)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (b.size() < 8 || memcmp(b.data(), sig, 8)) { set_last_error("not a PNG file"); return MARAY_E_DECODE; }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t pos = 8;
    bool seen_ihdr = false;
    while (pos + 12 <= b.size()) {
        uint32_t len = be32(&b[pos]);
        if (pos + 12 + (size_t)len > b.size()) { set_last_error("truncated PNG chunk"); return MARAY_E_DECODE; }
        const uint8_t *type = &b[pos + 4], *data = &b[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; seen_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || !w || !h) { set_last_error("PNG without IHDR"); return MARAY_E_DECODE; }
    if (interlace > 1) { set_last_error("bad PNG interlace method"); return MARAY_E_DECODE; }
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: set_last_error("bad PNG colour type"); return MARAY_E_DECODE;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
        set_last_error("bad PNG bit depth"); return MARAY_E_DECODE;
    }
    const size_t bpp_bits = (size_t)channels * depth;                 // <= 64
    // sizes straight from the file: bound them before any product is formed (w, h <= 2^20 keeps every product below 2^47)
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX) { set_last_error("PNG larger than 1048576 pixels a side"); return MARAY_E_LIMIT; }
    const uint64_t stride64 = ((uint64_t)w * bpp_bits + 7) / 8;
    if ((stride64 + 1) * h > MARAY_PNG_MAX_BYTES || (uint64_t)w * h * 3 > MARAY_PNG_MAX_BYTES) {
        set_last_error("PNG raster exceeds the reader's size limit"); return MARAY_E_LIMIT;
    }
    // (a zlib stream expands at most 1032:1: an IHDR that promises more than the IDAT bytes can hold is rejected unallocated, below)
    const size_t bpp = (bpp_bits + 7) / 8;
    // Scanlines of the image, or of its seven Adam7 passes one after the other (pass p holds the pixels x0 + dx i, y0 + dy j;
    // an empty pass has no scanlines); every pass is filtered on its own
    static const uint32_t ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    struct Pass { uint32_t x0, y0, dx, dy, pw, ph; size_t stride; };
    std::vector<Pass> passes;
    uint64_t total = 0;
    for (int p = 0; p < (interlace ? 7 : 1); p++) {
        Pass q = interlace ? Pass{ax0[p], ay0[p], adx[p], ady[p], 0, 0, 0} : Pass{0, 0, 1, 1, 0, 0, 0};
        q.pw = w > q.x0 ? (w - q.x0 + q.dx - 1) / q.dx : 0;
        q.ph = h > q.y0 ? (h - q.y0 + q.dy - 1) / q.dy : 0;
        if (!q.pw || !q.ph) continue;
        q.stride = (size_t)(((uint64_t)q.pw * bpp_bits + 7) / 8);
        total += ((uint64_t)q.stride + 1) * q.ph;
        passes.push_back(q);
    }
    if (total > MARAY_PNG_MAX_BYTES || total > ((uint64_t)idat.size() + 1) * 1040) { set_last_error("PNG inflate failed (IDAT too short for its IHDR)"); return MARAY_E_DECODE; }
    std::vector<uint8_t> raw((size_t)total);
    uLongf rn = (uLongf)raw.size();
    if (uncompress(raw.data(), &rn, idat.data(), (uLong)idat.size()) != Z_OK || rn != raw.size()) {
        set_last_error("PNG inflate failed"); return MARAY_E_DECODE;
    }
    uint8_t *out = (uint8_t *)malloc((size_t)w * h * 3);
    if (!out) { set_last_error("out of memory"); return MARAY_E_INTERNAL; }
    size_t at = 0;
    for (const Pass &q : passes) {
        const size_t stride = q.stride;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (uint32_t j = 0; j < q.ph; j++) {
            const uint8_t *line = &raw[at];
            at += stride + 1;
            const int ft = line[0];
            for (size_t i = 0; i < stride; i++) {
                int a = i >= bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = line[1 + i];
                switch (ft) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += bb; break;
                case 3: x += (a + bb) / 2; break;
                case 4: x += paeth(a, bb, c); break;
                default: free(out); set_last_error("bad PNG filter"); return MARAY_E_DECODE;
                }
                cur[i] = (uint8_t)x;
            }
            uint8_t *orow = out + (size_t)(q.y0 + (size_t)j * q.dy) * w * 3;
            for (uint32_t i = 0; i < q.pw; i++) {
                auto sample = [&](int ch) -> unsigned {   // 8-bit value of channel ch of pixel i of this scanline
                    if (depth == 8) return cur[(size_t)i * channels + ch];
                    if (depth == 16) {   // image crate: 16 -> 8 bit by rounding division (v * 255 + 32767) / 65535
                        unsigned v = ((unsigned)cur[((size_t)i * channels + ch) * 2] << 8) | cur[((size_t)i * channels + ch) * 2 + 1];
                        return (v * 255u + 32767u) / 65535u;
                    }
                    size_t bit = (size_t)i * depth;
                    unsigned v = (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u);
                    return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
                };
                uint8_t *o = orow + (size_t)(q.x0 + (size_t)i * q.dx) * 3;
                if (ctype == 3) {
                    unsigned idx = depth == 8 ? cur[i] : sample(0);
                    for (int k = 0; k < 3; k++) o[k] = (idx * 3 + k < plte.size()) ? plte[idx * 3 + k] : 0;
                } else if (ctype == 0 || ctype == 4) {
                    unsigned g = sample(0);
                    o[0] = o[1] = o[2] = (uint8_t)g;
                } else {
                    for (int k = 0; k < 3; k++) o[k] = (uint8_t)sample(k);
                }
            }
            prev.swap(cur);
        }
    }
    *rgb8_out = out; *w_out = w; *h_out = h;
    return MARAY_OK;
}

}   // namespace maray

extern "C" int maray_png_read(const char *path, uint8_t **rgb8_out, uint32_t *w_out, uint32_t *h_out)
{
    if (!path || !rgb8_out || !w_out || !h_out) { set_last_error("null argument"); return MARAY_E_ARG; }
    *rgb8_out = nullptr;
    return png_guard([&]() -> int {
        std::vector<uint8_t> b;
        const int rc = read_whole_file(path, b);
        return rc ? rc : png_decode(b, rgb8_out, w_out, h_out);
    });
}
