// image.cpp — texture files other than PNG (product code, host side).
//
//   maray_image_read <- `image::open(file).unwrap().to_rgb8()`     examples/maray.rs:58-65
//
// The reference hands a texture file to the `image` crate (0.25.1, Cargo.lock:808-809), which tells the format from the
// file's first bytes and decodes whatever it knows; `to_rgb8` then drops alpha, replicates grey and rounds 16-bit samples
// to 8 (v * 255 + 32767) / 65535.  Here: PNG (png.cpp; every colour type and bit depth, interlaced or not), BMP
// (uncompressed 8 / 24 / 32 bit, bottom-up or top-down), binary and plain PNM (P1 - P6), TGA (true colour, grey and
// colour-mapped, raw or run-length encoded), QOI, farbfeld, GIF (the first frame) and TIFF (grey / RGB strips, 8 or 16 bits,
// uncompressed, LZW, Deflate, PackBits) -- the formats that are a header and pixels, or whose coding is exact.  JPEG, WebP
// and the HDR formats of that crate are codecs of their own and are not restated: a texture in
// one of them is MARAY_E_DECODE with the format's name ("convert it to PNG"), not a wrong picture.
//
// Sizes come from the file: every product is formed in 64 bits and bounded (2^20 pixels a side, the evaluators' own
// limit), and nothing is allocated, reserved or touched before the file has been shown able to hold that much pixel
// data in its coding (raster(): bytes present x the format's best expansion ratio; TIFF strip by strip; a GIF's blank
// logical screen by a fixed 512 MiB budget).
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

const uint64_t IMAGE_MAX_BYTES = 1ull << 34;

struct Fail { int code; std::string msg; };

uint32_t le16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
uint32_t le32(const uint8_t *p) { return le16(p) | (le16(p + 2) << 16); }
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
unsigned to8(unsigned v16) { return (v16 * 255u + 32767u) / 65535u; }        // DynamicImage::to_rgb8 on 16-bit samples

// The raster of a w x h image -- allocated only once the caller has shown that the FILE can hold that many pixels:
// `max_px` is the most pixels its `data_bytes` of pixel data can encode in this format (data_bytes x the format's best
// ratio), formed by the caller from the file's real size.  A header's word alone never sizes an allocation: a 40-byte
// file that declares 65535 x 65535 pixels is MARAY_E_DECODE, not 12.9 GB of zeroed pages (the reference's image::open runs
// under the crate's default Limits).
uint8_t *raster(uint32_t w, uint32_t h, uint64_t max_px, bool zeroed = false)
{
    if (!w || !h) throw Fail{MARAY_E_DECODE, "image without pixels"};
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX || (uint64_t)w * h * 3 > IMAGE_MAX_BYTES) throw Fail{MARAY_E_LIMIT, "image larger than 1048576 pixels a side"};
    if ((uint64_t)w * h > max_px) throw Fail{MARAY_E_DECODE, "pixel data shorter than the header says (the file cannot hold " + std::to_string((uint64_t)w * h) + " pixels)"};
    uint8_t *p = (uint8_t *)(zeroed ? calloc((size_t)w * h, 3) : malloc((size_t)w * h * 3));
    if (!p) throw Fail{MARAY_E_INTERNAL, "out of memory"};
    return p;
}

uint64_t sat_mul(uint64_t a, uint64_t b) { return (b && a > UINT64_MAX / b) ? UINT64_MAX : a * b; }

// zlib stream -> at most *n bytes of dst; *n = what arrived (a strip's last rows may be short of the nominal size)
bool inflate_into(const uint8_t *src, size_t src_n, uint8_t *dst, size_t *n)
{
    uLongf got = (uLongf)*n;
    const int rc = uncompress(dst, &got, src, (uLong)src_n);
    if (rc != Z_OK && rc != Z_BUF_ERROR) return false;       // (Z_BUF_ERROR: more data than the rows need -- the rows are there)
    *n = (size_t)got;
    return true;
}

struct Owned {          // the raster until it is handed to the caller
    uint8_t *p = nullptr;
    ~Owned() { free(p); }
    uint8_t *release() { uint8_t *q = p; p = nullptr; return q; }
};

// ---- BMP: BITMAPFILEHEADER + BITMAPINFOHEADER (or a later version of it), BI_RGB / BI_BITFIELDS with the usual masks ----
void bmp(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    if (b.size() < 54) throw Fail{MARAY_E_DECODE, "truncated BMP header"};
    const uint32_t off = le32(&b[10]), hdr = le32(&b[14]);
    if (hdr < 40) throw Fail{MARAY_E_DECODE, "BMP with an OS/2 header is not supported"};
    const int32_t sw = (int32_t)le32(&b[18]), sh = (int32_t)le32(&b[22]);
    const uint32_t bpp = le16(&b[28]), comp = le32(&b[30]);
    if (sw <= 0 || sh == 0 || sh == INT32_MIN) throw Fail{MARAY_E_DECODE, "bad BMP size"};
    if (!(comp == 0 || (comp == 3 && (bpp == 32 || bpp == 16)))) throw Fail{MARAY_E_DECODE, "compressed BMP is not supported"};
    if (!(bpp == 8 || bpp == 24 || bpp == 32)) throw Fail{MARAY_E_DECODE, "BMP bit depth not supported (8, 24 and 32 are)"};
    w = (uint32_t)sw; h = (uint32_t)(sh < 0 ? -sh : sh);
    const uint64_t stride = (((uint64_t)w * bpp + 31) / 32) * 4;
    if ((uint64_t)off + stride * h > b.size()) throw Fail{MARAY_E_DECODE, "BMP pixel data shorter than its header says"};
    if ((uint64_t)14 + hdr > b.size()) throw Fail{MARAY_E_DECODE, "BMP header longer than the file"};
    out.p = raster(w, h, (uint64_t)b.size());                    // (a pixel is at least a byte)
    const uint8_t *pal = b.data() + 14 + hdr;
    uint32_t ncol = le32(&b[46]);
    if (bpp == 8) {
        if (!ncol || ncol > 256) ncol = 256;
        if ((uint64_t)14 + hdr + (uint64_t)ncol * 4 > b.size()) throw Fail{MARAY_E_DECODE, "truncated BMP palette"};
    }
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *line = &b[off + stride * (sh < 0 ? y : h - 1 - y)];       // bottom-up unless the height is negative
        uint8_t *o = out.p + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            if (bpp == 8) {
                const unsigned idx = line[x];
                for (int k = 0; k < 3; k++) o[3 * x + k] = idx < ncol ? pal[idx * 4 + 2 - k] : 0;
            } else {
                const uint8_t *px = line + (size_t)x * (bpp / 8);
                o[3 * x] = px[2]; o[3 * x + 1] = px[1]; o[3 * x + 2] = px[0];          // stored blue, green, red
            }
        }
    }
}

// ---- PNM: P1 - P6 (bitmap, greymap, pixmap; plain or raw), maxval up to 65535 ----
void pnm(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    const int kind = b[1] - '0';
    size_t pos = 2;
    auto number = [&]() -> uint32_t {
        for (;;) {
            while (pos < b.size() && (b[pos] == ' ' || b[pos] == '\t' || b[pos] == '\n' || b[pos] == '\r')) pos++;
            if (pos < b.size() && b[pos] == '#') { while (pos < b.size() && b[pos] != '\n') pos++; continue; }
            break;
        }
        if (pos >= b.size() || b[pos] < '0' || b[pos] > '9') throw Fail{MARAY_E_DECODE, "bad PNM header"};
        uint64_t v = 0;
        while (pos < b.size() && b[pos] >= '0' && b[pos] <= '9') { v = v * 10 + (uint64_t)(b[pos++] - '0'); if (v > 0xFFFFFFFFull) throw Fail{MARAY_E_DECODE, "bad PNM header"}; }
        return (uint32_t)v;
    };
    w = number(); h = number();
    const uint32_t maxv = (kind == 1 || kind == 4) ? 1u : number();
    if (!maxv || maxv > 65535) throw Fail{MARAY_E_DECODE, "bad PNM maxval"};
    // plain forms: a sample is at least a digit (P1) or a digit and a separator; raw: P4 packs 8 pixels a byte
    out.p = raster(w, h, sat_mul(b.size() - std::min(pos, b.size()) + 1, kind == 4 ? 8 : 1));
    const int ch = (kind == 3 || kind == 6) ? 3 : 1;
    auto scale = [&](uint32_t v) -> uint8_t {        // image's PNM decoder yields 8- or 16-bit samples scaled to the full range
        if (v > maxv) v = maxv;
        if (maxv == 255) return (uint8_t)v;
        if (maxv < 256) return (uint8_t)((v * 255u + maxv / 2) / maxv);
        return (uint8_t)to8(maxv == 65535 ? v : (uint32_t)(((uint64_t)v * 65535u + maxv / 2) / maxv));
    };
    if (kind <= 3) {                                 // plain: whitespace-separated decimals (P1: digits may touch)
        for (size_t i = 0; i < (size_t)w * h; i++)
            for (int k = 0; k < ch; k++) {
                uint32_t v;
                if (kind == 1) {
                    while (pos < b.size() && b[pos] != '0' && b[pos] != '1') { if (b[pos] == '#') while (pos < b.size() && b[pos] != '\n') pos++; else pos++; }
                    if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "truncated PNM"};
                    v = b[pos++] == '1' ? 0u : 1u;                                 // 1 is black
                } else v = number();
                const uint8_t g = scale(v);
                if (ch == 3) out.p[i * 3 + k] = g; else out.p[i * 3] = out.p[i * 3 + 1] = out.p[i * 3 + 2] = g;
            }
        return;
    }
    if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "truncated PNM"};
    pos++;                                            // the single whitespace byte after the header
    const uint64_t bps = maxv > 255 ? 2 : 1;
    const uint64_t row = kind == 4 ? ((uint64_t)w + 7) / 8 : (uint64_t)w * ch * bps;
    if (pos + row * h > b.size()) throw Fail{MARAY_E_DECODE, "PNM pixel data shorter than its header says"};
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *line = &b[pos + row * y];
        uint8_t *o = out.p + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) {
            if (kind == 4) { const uint8_t g = (line[x / 8] >> (7 - x % 8)) & 1 ? 0 : 255; o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = g; continue; }
            for (int k = 0; k < ch; k++) {
                const uint8_t *q = line + ((size_t)x * ch + k) * bps;
                const uint8_t g = scale(bps == 2 ? ((uint32_t)q[0] << 8) | q[1] : q[0]);
                if (ch == 3) o[3 * x + k] = g; else o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = g;
            }
        }
    }
}

// ---- TGA: types 1 / 2 / 3 and their run-length encoded forms 9 / 10 / 11 ----
void tga(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    if (b.size() < 18) throw Fail{MARAY_E_DECODE, "truncated TGA header"};
    const unsigned idlen = b[0], cmtype = b[1], type = b[2], cmlen = le16(&b[5]), cmbits = b[7], bpp = b[16], desc = b[17];
    const unsigned cmfirst = le16(&b[3]);
    w = le16(&b[12]); h = le16(&b[14]);
    const unsigned base = type & 7u;
    if (!(base >= 1 && base <= 3) || (type & ~0xBu)) throw Fail{MARAY_E_DECODE, "TGA image type not supported"};
    if (base == 1 && (cmtype != 1 || !(cmbits == 24 || cmbits == 32) || bpp != 8)) throw Fail{MARAY_E_DECODE, "TGA colour map not supported"};
    if (base == 2 && !(bpp == 24 || bpp == 32)) throw Fail{MARAY_E_DECODE, "TGA pixel depth not supported (24 and 32 are)"};
    if (base == 3 && !(bpp == 8 || bpp == 16)) throw Fail{MARAY_E_DECODE, "TGA grey depth not supported"};
    size_t pos = 18 + idlen;
    const size_t cmap = pos, cmbytes = cmtype ? (size_t)cmlen * ((cmbits + 7) / 8) : 0;
    pos += cmbytes;
    if (pos > b.size()) throw Fail{MARAY_E_DECODE, "truncated TGA colour map"};
    const size_t pb = bpp / 8;
    const uint64_t npx = (uint64_t)w * h;
    // a run-length packet of 1 + pb bytes gives up to 128 pixels; raw data a pixel per pb bytes
    out.p = raster(w, h, (type & 8u) ? sat_mul((b.size() - pos) / (1 + pb) + 1, 128) : (b.size() - pos) / pb);
    std::vector<uint8_t> px;
    if (type & 8u) {                                   // run-length packets: 1 header byte, then one pixel (run) or n (raw)
        if (npx * pb > IMAGE_MAX_BYTES) throw Fail{MARAY_E_LIMIT, "TGA too large"};
        px.reserve((size_t)(npx * pb));                // (<= 128 pb / (1 + pb) x the file, by the bound above)
        while (px.size() < npx * pb) {
            if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "truncated TGA packet"};
            const unsigned n = (b[pos] & 0x7Fu) + 1, run = b[pos] & 0x80u;
            pos++;
            const size_t need = run ? pb : (size_t)n * pb;
            if (pos + need > b.size()) throw Fail{MARAY_E_DECODE, "truncated TGA packet"};
            for (unsigned i = 0; i < n && px.size() < npx * pb; i++) px.insert(px.end(), &b[pos + (run ? 0 : i * pb)], &b[pos + (run ? 0 : i * pb)] + pb);
            pos += need;
        }
    } else {
        if (pos + npx * pb > b.size()) throw Fail{MARAY_E_DECODE, "TGA pixel data shorter than its header says"};
        px.assign(b.begin() + (long)pos, b.begin() + (long)(pos + npx * pb));
    }
    const bool top = desc & 0x20u, right = desc & 0x10u;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t *q = &px[((size_t)(top ? y : h - 1 - y) * w + (right ? w - 1 - x : x)) * pb];
            uint8_t *o = out.p + ((size_t)y * w + x) * 3;
            if (base == 3) { o[0] = o[1] = o[2] = q[0]; continue; }
            const uint8_t *c = q;
            if (base == 1) {
                const unsigned idx = q[0] >= cmfirst ? q[0] - cmfirst : 0u;
                static const uint8_t zero[4] = {0, 0, 0, 0};
                c = idx < cmlen ? &b[cmap + (size_t)idx * (cmbits / 8)] : zero;
            }
            o[0] = c[2]; o[1] = c[1]; o[2] = c[0];                                // stored blue, green, red
        }
}

// ---- QOI (qoiformat.org): 14-byte header, ops, 8-byte end marker ----
void qoi(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    if (b.size() < 14 + 8) throw Fail{MARAY_E_DECODE, "truncated QOI file"};
    w = be32(&b[4]); h = be32(&b[8]);
    if (!(b[12] == 3 || b[12] == 4)) throw Fail{MARAY_E_DECODE, "bad QOI channel count"};
    out.p = raster(w, h, sat_mul(b.size() - 14 - 8 + 1, 62));      // QOI_OP_RUN: one byte, up to 62 pixels
    uint8_t idx[64][4];
    memset(idx, 0, sizeof idx);
    uint8_t px[4] = {0, 0, 0, 255};
    size_t pos = 14;
    const size_t end = b.size() - 8;
    unsigned run = 0;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        if (run) run--;
        else {
            if (pos >= end) throw Fail{MARAY_E_DECODE, "truncated QOI data"};
            const unsigned op = b[pos++];
            if (op == 0xFE || op == 0xFF) {
                const size_t n = op == 0xFE ? 3 : 4;
                if (pos + n > end) throw Fail{MARAY_E_DECODE, "truncated QOI data"};
                memcpy(px, &b[pos], n); pos += n;
            } else if ((op >> 6) == 0) memcpy(px, idx[op], 4);
            else if ((op >> 6) == 1) { px[0] += (uint8_t)(((op >> 4) & 3) - 2); px[1] += (uint8_t)(((op >> 2) & 3) - 2); px[2] += (uint8_t)((op & 3) - 2); }
            else if ((op >> 6) == 2) {
                if (pos >= end) throw Fail{MARAY_E_DECODE, "truncated QOI data"};
                const unsigned b2 = b[pos++];
                const int dg = (int)(op & 0x3F) - 32;
                px[0] += (uint8_t)(dg - 8 + (int)(b2 >> 4)); px[1] += (uint8_t)dg; px[2] += (uint8_t)(dg - 8 + (int)(b2 & 15));
            } else run = op & 0x3F;
            memcpy(idx[(px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) % 64], px, 4);
        }
        memcpy(out.p + i * 3, px, 3);
    }
}

// ---- farbfeld: "farbfeld", width, height (big endian), RGBA 16 bits a channel ----
void farbfeld(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    if (b.size() < 16) throw Fail{MARAY_E_DECODE, "truncated farbfeld header"};
    w = be32(&b[8]); h = be32(&b[12]);
    if ((uint64_t)w * h > (b.size() - 16) / 8) throw Fail{MARAY_E_DECODE, "farbfeld pixel data shorter than its header says"};
    out.p = raster(w, h, (b.size() - 16) / 8);
    for (size_t i = 0; i < (size_t)w * h; i++)
        for (int k = 0; k < 3; k++) out.p[i * 3 + k] = (uint8_t)to8(((unsigned)b[16 + i * 8 + 2 * k] << 8) | b[16 + i * 8 + 2 * k + 1]);
}

// ---- GIF (87a / 89a): the FIRST frame on the logical screen, as `image`'s GifDecoder hands it to `to_rgb8` ----
// (gif crate, ColorOutput::RGBA: a pixel is its palette colour, the transparent index too -- only its alpha is 0, and
// to_rgb8 drops alpha; the screen outside the frame is transparent black; an index past the palette leaves the pixel as
// it was.)  LZW: codes of min_size + 1 .. 12 bits, least significant bit first, across the data sub-blocks; a full table
// keeps decoding with 12-bit codes until the encoder clears it (deferred clear).
void gif(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    if (b.size() < 13) throw Fail{MARAY_E_DECODE, "truncated GIF header"};
    w = le16(&b[6]); h = le16(&b[8]);
    size_t pos = 13;
    const uint8_t *gct = nullptr; size_t gct_n = 0;
    if (b[10] & 0x80) {
        gct_n = (size_t)2 << (b[10] & 7);
        if (pos + gct_n * 3 > b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF colour table"};
        gct = &b[pos]; pos += gct_n * 3;
    }
    auto skip_sub_blocks = [&]() {
        for (;;) {
            if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF block"};
            const size_t n = b[pos++];
            if (!n) return;
            if (pos + n > b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF block"};
            pos += n;
        }
    };
    for (;;) {
        if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "GIF without an image"};
        const uint8_t tag = b[pos++];
        if (tag == 0x3B) throw Fail{MARAY_E_DECODE, "GIF without an image"};
        if (tag == 0x21) {                                  // extension: label, then sub-blocks (the graphic control one only
            if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF block"};      // carries what to_rgb8 drops)
            pos++;
            skip_sub_blocks();
            continue;
        }
        if (tag != 0x2C) throw Fail{MARAY_E_DECODE, "unknown GIF block"};
        break;
    }
    if (pos + 9 > b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF image descriptor"};
    const uint32_t fl = le16(&b[pos]), ft = le16(&b[pos + 2]), fw = le16(&b[pos + 4]), fh = le16(&b[pos + 6]);
    const uint8_t packed = b[pos + 8];
    pos += 9;
    const uint8_t *pal = gct; size_t pal_n = gct_n;
    if (packed & 0x80) {
        pal_n = (size_t)2 << (packed & 7);
        if (pos + pal_n * 3 > b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF colour table"};
        pal = &b[pos]; pos += pal_n * 3;
    }
    if (!pal) throw Fail{MARAY_E_DECODE, "GIF frame without a colour table"};
    if (!fw || !fh || (uint64_t)fl + fw > w || (uint64_t)ft + fh > h) throw Fail{MARAY_E_DECODE, "GIF frame outside its logical screen"};
    if (pos >= b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF image data"};
    const unsigned min_size = b[pos++];
    if (min_size < 1 || min_size > 11) throw Fail{MARAY_E_DECODE, "bad GIF code size"};      // (the gif crate rejects > 11 as well)
    // What the data can paint: a code of at most 12 bits stands for a string of at most 4,095 pixels, so the frame's pixels
    // are bounded by the bytes left (x 2,730); the logical screen around the frame is blank and costs the file nothing, so
    // it is bounded by a fixed budget instead -- 512 MiB of raster, the `image` crate's default Limits::max_alloc -- and
    // comes from calloc (untouched zero pages), not from a memset.
    if ((uint64_t)fw * fh > sat_mul(b.size() - pos + 1, 2730)) throw Fail{MARAY_E_DECODE, "GIF image data shorter than its frame"};
    if ((uint64_t)w * h * 3 > (512ull << 20)) throw Fail{MARAY_E_LIMIT, "GIF logical screen larger than 512 MiB of pixels"};
    out.p = raster(w, h, UINT64_MAX, true);
    const unsigned clear = 1u << min_size, stop = clear + 1;
    std::vector<uint16_t> prefix(4096);
    std::vector<uint8_t> suffix(4096), first(4096), stack(4097);
    for (unsigned i = 0; i < clear; i++) { prefix[i] = 0xFFFF; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; }
    unsigned size = min_size + 1, next = clear + 2, prev = 0xFFFF;
    uint32_t acc = 0; unsigned bits = 0;
    size_t block_left = 0;
    bool data_end = false;
    const uint64_t n_px = (uint64_t)fw * fh;
    uint64_t done = 0;
    const bool interlaced = (packed & 0x40) != 0;
    // frame pixel i -> raster position (interlace: rows 0 8 16 .., 4 12 .., 2 6 10 .., 1 3 5 ..)
    std::vector<uint32_t> row_of;
    if (interlaced) {
        static const unsigned start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
        for (int p4 = 0; p4 < 4; p4++) for (uint32_t y = start[p4]; y < fh; y += step[p4]) row_of.push_back(y);
    }
    auto put = [&](uint8_t idx) {
        if (done >= n_px) return;                           // (data past the frame's pixels is ignored)
        const uint32_t fy = (uint32_t)(done / fw), fx = (uint32_t)(done % fw);
        const uint32_t y = ft + (interlaced ? row_of[fy] : fy), x = fl + fx;
        if (idx < pal_n) memcpy(out.p + ((size_t)y * w + x) * 3, pal + (size_t)idx * 3, 3);
        done++;
    };
    while (done < n_px) {
        while (bits < size && !data_end) {                  // refill from the sub-blocks
            if (!block_left) {
                if (pos >= b.size()) { data_end = true; break; }
                block_left = b[pos++];
                if (!block_left) { data_end = true; break; }
                if (pos + block_left > b.size()) throw Fail{MARAY_E_DECODE, "truncated GIF image data"};
            }
            acc |= (uint32_t)b[pos++] << bits; bits += 8; block_left--;
        }
        if (bits < size) throw Fail{MARAY_E_DECODE, "GIF image data ends before its pixels do"};
        const unsigned code = acc & ((1u << size) - 1);
        acc >>= size; bits -= size;
        if (code == clear) { size = min_size + 1; next = clear + 2; prev = 0xFFFF; continue; }
        if (code == stop) throw Fail{MARAY_E_DECODE, "GIF image data ends before its pixels do"};
        if (code > next || (code == next && prev == 0xFFFF)) throw Fail{MARAY_E_DECODE, "bad GIF code"};
        // the string of `code` (or of prev + its own first symbol when code is the entry being defined), back to front
        unsigned sp = 0, c = code;
        if (code == next) { stack[sp++] = first[prev]; c = prev; }
        while (c >= clear) {
            if (c < clear + 2 || sp >= 4096) throw Fail{MARAY_E_DECODE, "bad GIF code"};
            stack[sp++] = suffix[c]; c = prefix[c];
        }
        stack[sp++] = (uint8_t)c;
        const uint8_t head = (uint8_t)c;
        while (sp) put(stack[--sp]);
        if (prev != 0xFFFF && next < 4096) {
            prefix[next] = (uint16_t)prev; suffix[next] = head; first[next] = first[prev];
            next++;
            if (next == (1u << size) && size < 12) size++;
        }
        prev = code;
    }
}

// ---- TIFF (baseline + LZW / Deflate / PackBits, strips): the first image of the file ----
// What `image`'s TiffDecoder + to_rgb8 give for it: grey (WhiteIsZero inverted) and RGB, 8 or 16 bits a sample, extra
// samples (alpha) dropped, horizontal differencing undone.  Tiles, palettes, CMYK / YCbCr / Lab, planar layouts and
// samples of other widths are named and refused.
struct Tiff {
    const std::vector<uint8_t> &b;
    bool big;
    uint32_t u16(size_t at) const { if (at + 2 > b.size()) throw Fail{MARAY_E_DECODE, "truncated TIFF"}; return big ? ((uint32_t)b[at] << 8) | b[at + 1] : le16(&b[at]); }
    uint32_t u32(size_t at) const { if (at + 4 > b.size()) throw Fail{MARAY_E_DECODE, "truncated TIFF"}; return big ? be32(&b[at]) : le32(&b[at]); }
    // the values of a SHORT / LONG field (type 3 / 4), inline or behind its offset
    std::vector<uint32_t> values(size_t entry) const {
        const uint32_t type = u16(entry + 2), n = u32(entry + 4);
        if (type != 3 && type != 4) throw Fail{MARAY_E_DECODE, "TIFF field of an unexpected type"};
        const size_t sz = type == 3 ? 2 : 4;
        if ((uint64_t)n * sz > b.size()) throw Fail{MARAY_E_DECODE, "truncated TIFF"};
        size_t at = (uint64_t)n * sz <= 4 ? entry + 8 : u32(entry + 8);
        std::vector<uint32_t> v(n);
        for (uint32_t i = 0; i < n; i++, at += sz) v[i] = type == 3 ? u16(at) : u32(at);
        return v;
    }
};

// TIFF's LZW: codes of 9 .. 12 bits, most significant bit first, 256 = clear, 257 = end, the width grows one code early
void tiff_lzw(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t want)
{
    std::vector<uint16_t> prefix(4096);
    std::vector<uint8_t> suffix(4096), first(4096), stack(4097);
    for (unsigned i = 0; i < 256; i++) { prefix[i] = 0xFFFF; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; }
    unsigned size = 9, next = 258, prev = 0xFFFF;
    uint32_t acc = 0; unsigned bits = 0; size_t pos = 0;
    while (out.size() < want) {
        while (bits < size && pos < n) { acc = (acc << 8) | src[pos++]; bits += 8; }
        if (bits < size) break;                               // (a strip may end without the end code)
        const unsigned code = (acc >> (bits - size)) & ((1u << size) - 1);
        bits -= size;
        if (code == 256) { size = 9; next = 258; prev = 0xFFFF; continue; }
        if (code == 257) break;
        if (code > next || (code == next && prev == 0xFFFF) || (code >= 256 && code < 258)) throw Fail{MARAY_E_DECODE, "bad TIFF LZW code"};
        unsigned sp = 0, c = code;
        if (code == next) { stack[sp++] = first[prev]; c = prev; }
        while (c >= 258) { if (sp >= 4096) throw Fail{MARAY_E_DECODE, "bad TIFF LZW code"}; stack[sp++] = suffix[c]; c = prefix[c]; }
        if (c >= 256) throw Fail{MARAY_E_DECODE, "bad TIFF LZW code"};
        stack[sp++] = (uint8_t)c;
        const uint8_t head = (uint8_t)c;
        while (sp && out.size() < want) out.push_back(stack[--sp]);
        if (prev != 0xFFFF && next < 4096) {
            prefix[next] = (uint16_t)prev; suffix[next] = head; first[next] = first[prev];
            next++;
            if (next + 1 == (1u << size) && size < 12) size++;
        }
        prev = code;
    }
}

void tiff(const std::vector<uint8_t> &b, Owned &out, uint32_t &w, uint32_t &h)
{
    const Tiff t{b, b[0] == 'M'};
    size_t ifd = t.u32(4);
    const uint32_t n_entries = t.u16(ifd);
    uint32_t bits = 1, comp = 1, photo = 0xFFFF, spp = 1, rps = 0xFFFFFFFFu, planar = 1, predictor = 1;
    bool bits_same = true, tiled = false;
    std::vector<uint32_t> offsets, counts;
    w = h = 0;
    for (uint32_t i = 0; i < n_entries; i++) {
        const size_t e = ifd + 2 + (size_t)i * 12;
        const uint32_t tag = t.u16(e);
        auto one = [&]() -> uint32_t { const auto v = t.values(e); if (v.empty()) throw Fail{MARAY_E_DECODE, "empty TIFF field"}; return v[0]; };
        switch (tag) {
        case 256: w = one(); break;
        case 257: h = one(); break;
        case 258: { const auto v = t.values(e); if (v.empty()) throw Fail{MARAY_E_DECODE, "bad TIFF BitsPerSample"}; bits = v[0]; for (uint32_t x : v) bits_same &= x == bits; } break;
        case 259: comp = one(); break;
        case 262: photo = one(); break;
        case 273: offsets = t.values(e); break;
        case 277: spp = one(); break;
        case 278: rps = one(); break;
        case 279: counts = t.values(e); break;
        case 284: planar = one(); break;
        case 317: predictor = one(); break;
        case 322: case 323: case 324: case 325: tiled = true; break;
        default: break;
        }
    }
    if (tiled) throw Fail{MARAY_E_DECODE, "tiled TIFF files are not supported: convert the file to PNG"};
    if (photo == 3) throw Fail{MARAY_E_DECODE, "palette TIFF files are not supported: convert the file to PNG"};
    if (photo > 2) throw Fail{MARAY_E_DECODE, "TIFF colour spaces other than grey and RGB are not supported: convert the file to PNG"};
    if (!bits_same || (bits != 8 && bits != 16)) throw Fail{MARAY_E_DECODE, "TIFF samples of 8 or 16 bits are read, not others: convert the file to PNG"};
    const uint32_t colour = photo == 2 ? 3u : 1u;
    if (spp < colour || spp > 8 || (planar != 1 && spp > 1)) throw Fail{MARAY_E_DECODE, "unsupported TIFF sample layout"};
    if (predictor != 1 && predictor != 2) throw Fail{MARAY_E_DECODE, "unsupported TIFF predictor"};
    if (comp != 1 && comp != 5 && comp != 8 && comp != 32946 && comp != 32773) throw Fail{MARAY_E_DECODE, "unsupported TIFF compression: convert the file to PNG"};
    if (!w || !h) throw Fail{MARAY_E_DECODE, "image without pixels"};
    if (w > MARAY_DOMAIN_MAX || h > MARAY_DOMAIN_MAX) throw Fail{MARAY_E_LIMIT, "image larger than 1048576 pixels a side"};
    if (!rps) throw Fail{MARAY_E_DECODE, "bad TIFF RowsPerStrip"};
    rps = std::min(rps, h);
    const uint32_t n_strips = (h + rps - 1) / rps;
    if (offsets.size() < n_strips || counts.size() < n_strips) throw Fail{MARAY_E_DECODE, "TIFF strip table shorter than the image"};
    const size_t bps = bits / 8, row_bytes = (size_t)w * spp * bps;
    // Every strip must be able to hold its rows before anything is sized by them: stored bytes x the coding's best ratio
    // (Deflate 1,032 : 1 -- 1,040 with margin, as png.cpp bounds IDAT; LZW: a 12-bit code for at most 4,095 bytes; PackBits:
    // 2 bytes for 128).  The raster then is at most 3 x the sum of what the strips can decode to.
    const uint64_t ratio = comp == 1 ? 1 : comp == 5 ? 2730 : comp == 32773 ? 64 : 1040;
    for (uint32_t s = 0; s < n_strips; s++) {
        if ((uint64_t)offsets[s] + counts[s] > b.size()) throw Fail{MARAY_E_DECODE, "TIFF strip outside the file"};
        const uint64_t want = (uint64_t)row_bytes * std::min(rps, h - s * rps);
        if (want > sat_mul((uint64_t)counts[s] + 1, ratio)) throw Fail{MARAY_E_DECODE, "TIFF strip too short for its rows"};
    }
    out.p = raster(w, h, UINT64_MAX);          // (w h spp bps = the strips' rows <= ratio x (file + strips), checked above)
    std::vector<uint8_t> strip;
    for (uint32_t s = 0; s < n_strips; s++) {
        const uint32_t rows = std::min(rps, h - s * rps);
        const size_t want = row_bytes * rows;
        if ((uint64_t)offsets[s] + counts[s] > b.size()) throw Fail{MARAY_E_DECODE, "TIFF strip outside the file"};
        const uint8_t *src = b.data() + offsets[s];
        strip.clear();
        if (comp == 1) { if (counts[s] < want) throw Fail{MARAY_E_DECODE, "TIFF strip shorter than its rows"}; strip.assign(src, src + want); }
        else if (comp == 5) { strip.reserve(want); tiff_lzw(src, counts[s], strip, want); }
        else if (comp == 32773) {                              // PackBits
            size_t pos = 0;
            while (strip.size() < want && pos < counts[s]) {
                const int8_t c = (int8_t)src[pos++];
                if (c >= 0) { const size_t n = (size_t)c + 1; if (pos + n > counts[s]) throw Fail{MARAY_E_DECODE, "truncated TIFF PackBits data"}; strip.insert(strip.end(), src + pos, src + pos + n); pos += n; }
                else if (c != -128) { if (pos >= counts[s]) throw Fail{MARAY_E_DECODE, "truncated TIFF PackBits data"}; strip.insert(strip.end(), (size_t)(1 - c), src[pos++]); }
            }
            if (strip.size() > want) strip.resize(want);
        } else {
            strip.resize(want);
            size_t got = want;
            if (!inflate_into(src, counts[s], strip.data(), &got)) throw Fail{MARAY_E_DECODE, "bad TIFF Deflate data"};
            strip.resize(got);
        }
        if (strip.size() < want) throw Fail{MARAY_E_DECODE, "TIFF strip decodes to fewer bytes than its rows"};
        for (uint32_t r = 0; r < rows; r++) {
            uint8_t *line = strip.data() + (size_t)r * row_bytes;
            if (predictor == 2) {                              // horizontal differencing, per sample, in the file's byte order
                if (bps == 1) for (size_t i = spp; i < row_bytes; i++) line[i] = (uint8_t)(line[i] + line[i - spp]);
                else for (size_t i = spp; i < (size_t)w * spp; i++) {
                    const unsigned a = t.big ? ((unsigned)line[2 * i] << 8) | line[2 * i + 1] : le16(&line[2 * i]);
                    const unsigned p = t.big ? ((unsigned)line[2 * (i - spp)] << 8) | line[2 * (i - spp) + 1] : le16(&line[2 * (i - spp)]);
                    const unsigned v = (a + p) & 0xFFFFu;
                    if (t.big) { line[2 * i] = (uint8_t)(v >> 8); line[2 * i + 1] = (uint8_t)v; } else { line[2 * i] = (uint8_t)v; line[2 * i + 1] = (uint8_t)(v >> 8); }
                }
            }
            uint8_t *dst = out.p + ((size_t)(s * rps + r) * w) * 3;
            for (uint32_t x = 0; x < w; x++) {
                unsigned v[3];
                for (uint32_t k = 0; k < colour; k++) {
                    const uint8_t *q = line + ((size_t)x * spp + k) * bps;
                    unsigned u = bps == 1 ? q[0] : (t.big ? ((unsigned)q[0] << 8) | q[1] : le16(q));
                    if (photo == 0) u = (bps == 1 ? 255u : 65535u) - u;          // WhiteIsZero
                    v[k] = bps == 1 ? u : to8(u);
                }
                dst[3 * x] = (uint8_t)v[0]; dst[3 * x + 1] = (uint8_t)v[colour == 3 ? 1 : 0]; dst[3 * x + 2] = (uint8_t)v[colour == 3 ? 2 : 0];
            }
        }
    }
}

bool ends_with(const char *s, const char *suffix)
{
    const size_t n = strlen(s), m = strlen(suffix);
    if (n < m) return false;
    for (size_t i = 0; i < m; i++) if ((s[n - m + i] | 0x20) != suffix[i]) return false;
    return true;
}

}   // namespace

extern "C" int maray_image_read(const char *path, uint8_t **rgb8_out, uint32_t *w_out, uint32_t *h_out)
{
    if (!path || !rgb8_out || !w_out || !h_out) { set_last_error("null argument"); return MARAY_E_ARG; }
    *rgb8_out = nullptr;
    try {
        std::vector<uint8_t> b;
        const int rc = read_whole_file(path, b);
        if (rc) return rc;
        if (b.size() >= 8 && !memcmp(b.data(), "\x89PNG\r\n\x1a\n", 8)) return png_decode(b, rgb8_out, w_out, h_out);
        Owned out;
        uint32_t w = 0, h = 0;
        if (b.size() >= 2 && b[0] == 'B' && b[1] == 'M') bmp(b, out, w, h);
        else if (b.size() >= 3 && b[0] == 'P' && b[1] >= '1' && b[1] <= '6' && (b[2] == ' ' || b[2] == '\n' || b[2] == '\r' || b[2] == '\t' || b[2] == '#')) pnm(b, out, w, h);
        else if (b.size() >= 4 && !memcmp(b.data(), "qoif", 4)) qoi(b, out, w, h);
        else if (b.size() >= 8 && !memcmp(b.data(), "farbfeld", 8)) farbfeld(b, out, w, h);
        else if (b.size() >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF) throw Fail{MARAY_E_DECODE, "JPEG textures are not supported: convert the file to PNG"};
        else if (b.size() >= 6 && (!memcmp(b.data(), "GIF87a", 6) || !memcmp(b.data(), "GIF89a", 6))) gif(b, out, w, h);
        else if (b.size() >= 12 && !memcmp(b.data(), "RIFF", 4) && !memcmp(&b[8], "WEBP", 4)) throw Fail{MARAY_E_DECODE, "WebP textures are not supported: convert the file to PNG"};
        else if (b.size() >= 8 && (!memcmp(b.data(), "II*\0", 4) || !memcmp(b.data(), "MM\0*", 4))) tiff(b, out, w, h);
        else if (ends_with(path, ".tga") || (b.size() >= 26 && !memcmp(&b[b.size() - 18], "TRUEVISION-XFILE", 16))) tga(b, out, w, h);       // TGA has no signature up front
        else throw Fail{MARAY_E_DECODE, "texture file format not recognised (PNG, BMP, PNM, TGA, QOI, farbfeld, GIF and TIFF are read)"};
        *rgb8_out = out.release(); *w_out = w; *h_out = h;
        return MARAY_OK;
    }
    catch (const Fail &f) { set_last_error(std::string(path) + ": " + f.msg); return f.code; }
    catch (const Error &e) { set_last_error(e.msg); return e.code; }
    catch (const std::bad_alloc &) { set_last_error("out of memory"); return MARAY_E_INTERNAL; }
    catch (const std::exception &e) { set_last_error(e.what()); return MARAY_E_INTERNAL; }
    catch (...) { set_last_error("unknown error"); return MARAY_E_INTERNAL; }
}
