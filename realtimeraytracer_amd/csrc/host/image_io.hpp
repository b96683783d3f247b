// image_io.hpp — the image decoding the reference gets from stb_image (reference src/core/file.cppm:272-311:
// stbi_set_flip_vertically_on_load(true); stbi_load(path, &w, &h, &c, isGrayscale ? STBI_grey : STBI_rgb_alpha)).
// Own decoders, written from the format specifications (PNG/zlib RFC 1950/1951, Radiance RGBE, Netpbm), with
// stb_image's documented channel conversions:
//   * to 4 channels: grey -> (g,g,g,255), grey+alpha -> (g,g,g,a), RGB -> (r,g,b,255);
//   * to 1 channel : luma y = (77 r + 150 g + 29 b) >> 8;
//   * .hdr -> 8 bit: channel conversion on the float data first (grey = (r+g+b)/3), then c = clamp(pow(c, 1/2.2) * 255 + 0.5),
//     alpha 255 (the reference loads its 4k HDRI this way, src/app/application.cppm:250);
//   * 16-bit PNG samples keep their high byte; Adam7 interlacing and tRNS (palette alpha, or one transparent colour on grey /
//     RGB images -> an added alpha channel) are decoded.
//   * TGA (types 1/2/3 and their RLE forms 9/10/11; 8-bit grey, 8-bit colour-mapped, 15/16/24/32-bit true colour; either
//     vertical origin) and BMP (BI_RGB 8-bit palettised / 24 / 32 bit, BI_BITFIELDS 32 bit; bottom-up or top-down; a 32-bit
//     file whose alpha bytes are all zero is opaque, as in stb_image).  TGA has no signature: it is tried for ".tga" files.
//   * JPEG (baseline and progressive Huffman, 8 bit, grey or three components): jpeg_decode.hpp; one wanted channel takes
//     the luma plane, as stb_image does.
// The other stb formats (GIF, PSD, PIC) are not decoded: load_image throws "unsupported image format" rather than guess.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "jpeg_decode.hpp"

namespace rtr::img {

struct Image {
    int width = 0, height = 0, channels = 0;   // channels of `pixels` (1 or 4 after load_image)
    std::vector<uint8_t> pixels;
};

namespace detail {

// ---- RFC 1951 inflate (canonical Huffman, bit-serial decode) --------------------------------------------------
struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint64_t bitbuf = 0; int bitcnt = 0;
    int bits(int need) {                       // need <= 16
        uint64_t val = bitbuf;
        while (bitcnt < need) {
            if (pos >= n) throw std::runtime_error("PNG: truncated zlib stream");
            val |= (uint64_t)p[pos++] << bitcnt; bitcnt += 8;
        }
        bitbuf = val >> need; bitcnt -= need;
        return (int)(val & ((1ull << need) - 1ull));
    }
};
struct Huffman { short count[16]; short symbol[288]; };
inline void build(Huffman& h, const short* length, int n) {
    for (int i = 0; i < 16; ++i) h.count[i] = 0;
    for (int i = 0; i < n; ++i) h.count[length[i]]++;
    short offs[16]; offs[1] = 0;
    for (int i = 1; i < 15; ++i) offs[i + 1] = offs[i] + h.count[i];
    for (int i = 0; i < n; ++i) if (length[i]) h.symbol[offs[length[i]]++] = (short)i;
}
inline int decode(BitReader& br, const Huffman& h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= br.bits(1);
        int count = h.count[len];
        if (code - count < first) return h.symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    throw std::runtime_error("PNG: bad Huffman code");
}
inline void inflate_codes(BitReader& br, std::vector<uint8_t>& out, const Huffman& lencode, const Huffman& distcode) {
    static const short lens[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const short dists[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int sym = decode(br, lencode);
        if (sym < 256) out.push_back((uint8_t)sym);
        else if (sym == 256) return;
        else {
            sym -= 257;
            if (sym >= 29) throw std::runtime_error("PNG: bad length symbol");
            int len = lens[sym] + br.bits(lext[sym]);
            int ds = decode(br, distcode);
            if (ds >= 30) throw std::runtime_error("PNG: bad distance symbol");
            size_t dist = (size_t)dists[ds] + (size_t)br.bits(dext[ds]);
            if (dist > out.size()) throw std::runtime_error("PNG: distance too far back");
            size_t from = out.size() - dist;
            for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
        }
    }
}
inline std::vector<uint8_t> zlib_inflate(const uint8_t* data, size_t n, size_t expect) {
    if (n < 2 || (data[0] & 0x0f) != 8 || ((data[0] << 8 | data[1]) % 31) != 0) throw std::runtime_error("PNG: bad zlib header");
    if (data[1] & 0x20) throw std::runtime_error("PNG: preset dictionary not allowed");
    BitReader br{data + 2, n - 2};
    std::vector<uint8_t> out; out.reserve(expect);
    int last;
    do {
        last = br.bits(1);
        int type = br.bits(2);
        if (type == 0) {
            br.bitbuf = 0; br.bitcnt = 0;
            if (br.pos + 4 > br.n) throw std::runtime_error("PNG: truncated stored block");
            unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) throw std::runtime_error("PNG: bad stored block");
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len); br.pos += len;
        } else if (type == 1) {
            short l[288]; Huffman lc, dc;
            for (int i = 0; i < 144; ++i) l[i] = 8;
            for (int i = 144; i < 256; ++i) l[i] = 9;
            for (int i = 256; i < 280; ++i) l[i] = 7;
            for (int i = 280; i < 288; ++i) l[i] = 8;
            build(lc, l, 288);
            for (int i = 0; i < 30; ++i) l[i] = 5;
            build(dc, l, 30);
            inflate_codes(br, out, lc, dc);
        } else if (type == 2) {
            static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
            if (nlen > 286 || ndist > 30) throw std::runtime_error("PNG: bad dynamic block counts");
            short l[320]; Huffman lc;
            for (int i = 0; i < 19; ++i) l[i] = 0;
            for (int i = 0; i < ncode; ++i) l[order[i]] = (short)br.bits(3);
            build(lc, l, 19);
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = decode(br, lc);
                if (sym < 16) l[idx++] = (short)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (idx == 0) throw std::runtime_error("PNG: repeat without previous length"); val = l[idx - 1]; rep = 3 + br.bits(2); }
                    else if (sym == 17) rep = 3 + br.bits(3);
                    else rep = 11 + br.bits(7);
                    if (idx + rep > nlen + ndist) throw std::runtime_error("PNG: too many code lengths");
                    while (rep--) l[idx++] = (short)val;
                }
            }
            Huffman lenc, distc;
            build(lenc, l, nlen);
            build(distc, l + nlen, ndist);
            inflate_codes(br, out, lenc, distc);
        } else throw std::runtime_error("PNG: reserved block type");
    } while (!last);
    return out;
}

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// decodes to `src_channels` (1,2,3 or 4) 8-bit samples, top row first
inline void decode_png(const std::vector<uint8_t>& f, int& w, int& h, int& src_channels, std::vector<uint8_t>& px, int desired_channels) {
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool haveHdr = false;
    while (pos + 12 <= f.size()) {
        uint32_t len = be32(&f[pos]);
        std::string type(reinterpret_cast<const char*>(&f[pos + 4]), 4);
        if (pos + 12 + len > f.size()) throw std::runtime_error("PNG: truncated chunk");
        const uint8_t* d = &f[pos + 8];
        if (type == "IHDR") {
            if (len != 13) throw std::runtime_error("PNG: bad IHDR");
            w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12];
            if (d[10] != 0 || d[11] != 0) throw std::runtime_error("PNG: unknown compression/filter method");
            haveHdr = true;
        } else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "tRNS") trns.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (!haveHdr || w <= 0 || h <= 0 || w > 1 << 16 || h > 1 << 16 || (size_t)w * (size_t)h > ((size_t)1 << 28)) throw std::runtime_error("PNG: missing or bad IHDR (or more than 2^28 pixels)");
    if (interlace > 1) throw std::runtime_error("PNG: unknown interlace method");
    int samples;
    switch (ctype) { case 0: samples = 1; break; case 2: samples = 3; break; case 3: samples = 1; break; case 4: samples = 2; break; case 6: samples = 4; break;
                     default: throw std::runtime_error("PNG: bad colour type"); }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) throw std::runtime_error("PNG: unsupported bit depth");
    if (ctype == 3 && depth == 16) throw std::runtime_error("PNG: bad palette depth");
    const size_t bitsPerPixel = (size_t)samples * depth, bpp = std::max<size_t>(1, bitsPerPixel / 8);
    // the passes of the image: one, or the seven of Adam7 (x origin, y origin, x spacing, y spacing)
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const int whole[1][4] = {{0, 0, 1, 1}};
    const int (*passes)[4] = interlace ? adam7 : whole;
    const int npass = interlace ? 7 : 1;
    size_t expect = 0;
    for (int k = 0; k < npass; ++k) {
        const int pw = (w - passes[k][0] + passes[k][2] - 1) / passes[k][2], ph = (h - passes[k][1] + passes[k][3] - 1) / passes[k][3];
        if (pw > 0 && ph > 0) expect += (((size_t)bitsPerPixel * pw + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw = zlib_inflate(idat.data(), idat.size(), expect);
    if (raw.size() < expect) throw std::runtime_error("PNG: not enough image data");
    std::vector<uint16_t> val((size_t)w * h * samples);            // raw sample values (full 16 bits for depth 16)
    size_t rpos = 0;
    std::vector<uint8_t> img;
    for (int k = 0; k < npass; ++k) {
        const int pw = (w - passes[k][0] + passes[k][2] - 1) / passes[k][2], ph = (h - passes[k][1] + passes[k][3] - 1) / passes[k][3];
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)bitsPerPixel * pw + 7) / 8;
        img.assign(stride * ph, 0);
        for (int y = 0; y < ph; ++y) {                              // un-filter
            const uint8_t* in = &raw[rpos + (stride + 1) * y];
            uint8_t* out = &img[stride * y];
            const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
            int ft = in[0]; ++in;
            for (size_t i = 0; i < stride; ++i) {
                int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, v;
                switch (ft) {
                    case 0: v = in[i]; break;
                    case 1: v = in[i] + a; break;
                    case 2: v = in[i] + b; break;
                    case 3: v = in[i] + ((a + b) >> 1); break;
                    case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                              v = in[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)); break; }
                    default: throw std::runtime_error("PNG: bad filter type");
                }
                out[i] = (uint8_t)v;
            }
        }
        rpos += (stride + 1) * ph;
        for (int y = 0; y < ph; ++y) {
            const uint8_t* row = &img[stride * y];
            for (int x = 0; x < pw; ++x)
                for (int c = 0; c < samples; ++c) {
                    const size_t idx = (size_t)x * samples + c;
                    int v;
                    if (depth == 8) v = row[idx];
                    else if (depth == 16) v = (row[idx * 2] << 8) | row[idx * 2 + 1];
                    else { const size_t bit = idx * depth; v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1); }
                    val[(((size_t)(passes[k][1] + y * passes[k][3])) * w + (passes[k][0] + x * passes[k][2])) * samples + c] = (uint16_t)v;
                }
        }
    }
    // expand to 8-bit samples: palette look-up, bit-depth scaling (16 bit keeps its high byte), tRNS
    auto to8 = [&](int v) -> uint8_t { return (uint8_t)(depth == 16 ? v >> 8 : (ctype == 0 && depth < 8 ? v * (255 / ((1 << depth) - 1)) : v)); };
    if (ctype == 3) {
        src_channels = trns.empty() ? 3 : 4;
        px.resize((size_t)w * h * src_channels);
        for (size_t q = 0; q < (size_t)w * h; ++q) {
            const int i = val[q];
            if ((size_t)i * 3 + 2 >= plte.size()) throw std::runtime_error("PNG: palette index out of range");
            uint8_t* o = &px[q * src_channels];
            o[0] = plte[3 * i]; o[1] = plte[3 * i + 1]; o[2] = plte[3 * i + 2];
            if (src_channels == 4) o[3] = (size_t)i < trns.size() ? trns[i] : 255;
        }
    } else {
        // a tRNS chunk on a grey / RGB image names ONE transparent colour: it adds an alpha channel that is 0 there
        const bool keyed = !trns.empty() && (ctype == 0 || ctype == 2);
        if (keyed && trns.size() < (size_t)samples * 2) throw std::runtime_error("PNG: bad tRNS");
        int key[3] = {0, 0, 0};
        for (int c = 0; keyed && c < samples; ++c) key[c] = ((trns[2 * c] << 8) | trns[2 * c + 1]) & (depth == 16 ? 0xffff : 0xff);
        src_channels = samples + (keyed ? 1 : 0);
        if (depth == 16 && samples >= 3 && desired_channels == 1) {
            // stb_image reduces a 16-bit colour image to grey on the 16-bit values and only then keeps the high byte
            src_channels = 1;
            px.resize((size_t)w * h);
            for (size_t q = 0; q < (size_t)w * h; ++q) {
                const uint32_t y16 = ((uint32_t)val[q * samples] * 77u + (uint32_t)val[q * samples + 1] * 150u + (uint32_t)val[q * samples + 2] * 29u) >> 8;
                px[q] = (uint8_t)((y16 >> 8) & 0xffu);
            }
            return;
        }
        px.resize((size_t)w * h * src_channels);
        for (size_t q = 0; q < (size_t)w * h; ++q) {
            bool match = keyed;
            for (int c = 0; c < samples; ++c) {
                px[q * src_channels + c] = to8(val[q * samples + c]);
                if (keyed && val[q * samples + c] != key[c]) match = false;
            }
            if (keyed) px[q * src_channels + samples] = match ? 0 : 255;
        }
    }
}

inline void decode_pnm(const std::vector<uint8_t>& f, int& w, int& h, int& src_channels, std::vector<uint8_t>& px) {
    size_t pos = 2;
    auto next_int = [&]() {
        for (;;) {
            while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\t' || f[pos] == '\r' || f[pos] == '\n')) ++pos;
            if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') ++pos; continue; }
            break;
        }
        int v = 0; bool any = false;
        while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (f[pos++] - '0'); any = true; }
        if (!any) throw std::runtime_error("PNM: bad header");
        return v;
    };
    src_channels = f[1] == '5' ? 1 : 3;
    w = next_int(); h = next_int(); int maxv = next_int();
    if (maxv != 255) throw std::runtime_error("PNM: only maxval 255 is supported");
    if (w <= 0 || h <= 0 || (size_t)w * (size_t)h > ((size_t)1 << 28)) throw std::runtime_error("PNM: bad size");
    ++pos;                                                          // single whitespace after maxval
    size_t n = (size_t)w * h * src_channels;
    if (w <= 0 || h <= 0 || pos + n > f.size()) throw std::runtime_error("PNM: truncated data");
    px.assign(f.begin() + pos, f.begin() + pos + n);
}

// Radiance .hdr (RGBE) -> float RGB, top row first
inline void decode_hdr(const std::vector<uint8_t>& f, int& w, int& h, std::vector<float>& rgb) {
    size_t pos = 0;
    auto line = [&]() { std::string s; while (pos < f.size() && f[pos] != '\n') s += (char)f[pos++]; ++pos; return s; };
    std::string first = line();
    if (first != "#?RADIANCE" && first != "#?RGBE") throw std::runtime_error("HDR: bad signature");
    bool fmt = false;
    for (;;) { std::string l = line(); if (l.empty()) break; if (l == "FORMAT=32-bit_rle_rgbe") fmt = true; if (pos >= f.size()) break; }
    if (!fmt) throw std::runtime_error("HDR: unsupported format");
    std::string res = line();
    if (std::sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (size_t)w * (size_t)h > ((size_t)1 << 28)) throw std::runtime_error("HDR: unsupported data layout (or more than 2^28 pixels)");
    rgb.resize((size_t)w * h * 3);
    std::vector<uint8_t> scan((size_t)w * 4);
    auto to_float = [](const uint8_t* c, float* o) {
        if (c[3] == 0) { o[0] = o[1] = o[2] = 0.f; return; }
        float s = std::ldexp(1.0f, (int)c[3] - (128 + 8));
        o[0] = c[0] * s; o[1] = c[1] * s; o[2] = c[2] * s;
    };
    for (int y = 0; y < h; ++y) {
        if (pos + 4 > f.size()) throw std::runtime_error("HDR: truncated");
        if (w >= 8 && w < 32768 && f[pos] == 2 && f[pos + 1] == 2 && !(f[pos + 2] & 0x80)) {      // new-style RLE
            if (((int)f[pos + 2] << 8 | f[pos + 3]) != w) throw std::runtime_error("HDR: bad scanline width");
            pos += 4;
            for (int k = 0; k < 4; ++k) {
                int x = 0;
                while (x < w) {
                    if (pos >= f.size()) throw std::runtime_error("HDR: truncated");
                    int count = f[pos++];
                    if (count > 128) { count -= 128; if (x + count > w || pos >= f.size()) throw std::runtime_error("HDR: bad run"); uint8_t v = f[pos++]; while (count--) scan[(size_t)(x++) * 4 + k] = v; }
                    else { if (count == 0 || x + count > w || pos + count > f.size()) throw std::runtime_error("HDR: bad run"); while (count--) scan[(size_t)(x++) * 4 + k] = f[pos++]; }
                }
            }
        } else {                                                                                  // flat RGBE
            if (pos + (size_t)w * 4 > f.size()) throw std::runtime_error("HDR: truncated");
            std::memcpy(scan.data(), &f[pos], (size_t)w * 4); pos += (size_t)w * 4;
        }
        for (int x = 0; x < w; ++x) to_float(&scan[(size_t)x * 4], &rgb[((size_t)y * w + x) * 3]);
    }
}

inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Truevision TGA 2.0 specification.  Output: 1 (grey), 3 or 4 samples per pixel, top row first.
inline void decode_tga(const std::vector<uint8_t>& f, int& w, int& h, int& src_channels, std::vector<uint8_t>& px) {
    auto bad = [](const char* m) { throw std::runtime_error(std::string("Failed to load image: bad TGA (") + m + ")"); };
    if (f.size() < 18) bad("truncated header");
    const int idLen = f[0], cmapType = f[1], type = f[2];
    const int cmapFirst = le16(&f[3]), cmapLen = le16(&f[5]), cmapBits = f[7];
    w = le16(&f[12]); h = le16(&f[14]);
    const int bpp = f[16], desc = f[17];
    const bool rle = type >= 9 && type <= 11;
    const int base = rle ? type - 8 : type;
    if (base < 1 || base > 3 || w <= 0 || h <= 0 || (size_t)w * (size_t)h > ((size_t)1 << 28)) bad("unsupported image type or size");
    if ((base == 1) != (cmapType == 1)) bad("colour map / image type mismatch");
    auto channels_of = [&](int bits, bool grey) -> int {
        if (grey) return bits == 8 ? 1 : 0;
        if (bits == 15 || bits == 16 || bits == 24) return 3;
        if (bits == 32) return 4;
        return 0;
    };
    const int pixBits = base == 1 ? cmapBits : bpp;
    src_channels = channels_of(pixBits, base == 3);
    if (src_channels == 0 || (base == 1 && bpp != 8)) bad("unsupported bit depth");
    auto expand = [&](const uint8_t* s, int bits, uint8_t* d) {      // file order is B,G,R(,A); 15/16 bit is A1 R5 G5 B5
        if (bits == 8) d[0] = s[0];
        else if (bits == 15 || bits == 16) {
            const int v = le16(s);
            d[0] = (uint8_t)((((v >> 10) & 31) * 255) / 31); d[1] = (uint8_t)((((v >> 5) & 31) * 255) / 31); d[2] = (uint8_t)(((v & 31) * 255) / 31);
        } else { d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; if (bits == 32) d[3] = s[3]; }
    };
    size_t pos = 18 + (size_t)idLen;
    std::vector<uint8_t> palette;
    if (cmapType == 1) {
        const int eb = (cmapBits + 7) / 8;
        if (pos + (size_t)cmapLen * eb > f.size()) bad("truncated colour map");
        if (base == 1) {
            palette.resize((size_t)cmapLen * src_channels);
            for (int i = 0; i < cmapLen; ++i) expand(&f[pos + (size_t)i * eb], cmapBits, &palette[(size_t)i * src_channels]);
        }
        pos += (size_t)cmapLen * eb;
    }
    const int fileBytes = (bpp + 7) / 8;
    px.assign((size_t)w * h * src_channels, 0);
    std::vector<uint8_t> raw((size_t)fileBytes);
    size_t count = 0, total = (size_t)w * h;
    auto put = [&](const uint8_t* s) {
        uint8_t* d = &px[count * src_channels];
        if (base == 1) {
            const int idx = (int)s[0] - cmapFirst;
            if (idx < 0 || idx >= cmapLen) bad("palette index out of range");
            std::memcpy(d, &palette[(size_t)idx * src_channels], (size_t)src_channels);
        } else expand(s, bpp, d);
        ++count;
    };
    while (count < total) {
        if (!rle) {
            if (pos + (size_t)fileBytes > f.size()) bad("truncated pixel data");
            put(&f[pos]); pos += (size_t)fileBytes;
        } else {
            if (pos >= f.size()) bad("truncated RLE stream");
            const int head = f[pos++], n = (head & 127) + 1;
            if (count + (size_t)n > total) bad("RLE packet overruns the image");
            if (head & 128) {
                if (pos + (size_t)fileBytes > f.size()) bad("truncated RLE stream");
                for (int i = 0; i < n; ++i) put(&f[pos]);
                pos += (size_t)fileBytes;
            } else {
                if (pos + (size_t)n * fileBytes > f.size()) bad("truncated RLE stream");
                for (int i = 0; i < n; ++i) { put(&f[pos]); pos += (size_t)fileBytes; }
            }
        }
    }
    if (!(desc & 0x20)) {                                            // bit 5 clear: first row in the file is the bottom row
        const size_t row = (size_t)w * src_channels;
        for (int y = 0; y < h / 2; ++y) std::swap_ranges(px.begin() + (size_t)y * row, px.begin() + (size_t)(y + 1) * row, px.begin() + (size_t)(h - 1 - y) * row);
    }
    if (desc & 0x10) {                                               // bit 4: rows stored right to left
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w / 2; ++x)
                std::swap_ranges(&px[((size_t)y * w + x) * src_channels], &px[((size_t)y * w + x + 1) * src_channels], &px[((size_t)y * w + (w - 1 - x)) * src_channels]);
    }
}

// Windows BMP (BITMAPCOREHEADER / BITMAPINFOHEADER and later).  Output: 3 or 4 samples per pixel, top row first.
inline void decode_bmp(const std::vector<uint8_t>& f, int& w, int& h, int& src_channels, std::vector<uint8_t>& px) {
    auto bad = [](const char* m) { throw std::runtime_error(std::string("Failed to load image: bad BMP (") + m + ")"); };
    if (f.size() < 26) bad("truncated header");
    const uint32_t dataOff = le32(&f[10]), hdr = le32(&f[14]);
    int bpp; uint32_t comp = 0; int32_t hs;
    if (hdr == 12) { w = le16(&f[18]); hs = (int16_t)le16(&f[20]); bpp = le16(&f[24]); }
    else if (hdr >= 40 && f.size() >= 14 + (size_t)hdr) { w = (int32_t)le32(&f[18]); hs = (int32_t)le32(&f[22]); bpp = le16(&f[28]); comp = le32(&f[30]); }
    else { bad("unsupported header size"); return; }
    const bool topDown = hs < 0;
    h = topDown ? -hs : hs;
    if (w <= 0 || h <= 0 || (size_t)w * (size_t)h > ((size_t)1 << 28)) bad("bad size (or more than 2^28 pixels)");
    uint32_t mr = 0x00ff0000u, mg = 0x0000ff00u, mb = 0x000000ffu, ma = 0xff000000u;
    if (comp == 3) {
        if (bpp != 32) bad("BI_BITFIELDS is decoded for 32-bit files only");
        const size_t mo = hdr == 40 ? 54 : 54;                       // masks follow the 40-byte header / are fields of the V4+ headers
        if (f.size() < mo + 12) bad("truncated bit masks");
        mr = le32(&f[mo]); mg = le32(&f[mo + 4]); mb = le32(&f[mo + 8]);
        ma = hdr >= 56 && f.size() >= mo + 16 ? le32(&f[mo + 12]) : 0u;
        auto byte_mask = [](uint32_t m) { return m == 0u || m == 0xffu || m == 0xff00u || m == 0xff0000u || m == 0xff000000u; };
        if (!byte_mask(mr) || !byte_mask(mg) || !byte_mask(mb) || !byte_mask(ma) || !mr || !mg || !mb) bad("only byte-aligned 8-bit masks are decoded");
    } else if (comp != 0) bad("compressed BMP (RLE / JPEG / PNG payload) is not decoded");
    if (bpp != 8 && bpp != 24 && bpp != 32) bad("only 8-bit palettised, 24-bit and 32-bit files are decoded");
    std::vector<uint8_t> palette;
    if (bpp == 8) {
        const int eb = hdr == 12 ? 3 : 4;
        uint32_t used = hdr >= 40 ? le32(&f[46]) : 0u;
        if (used == 0 || used > 256) used = 256;
        const size_t po = 14 + (size_t)hdr;
        if (po + (size_t)used * eb > f.size()) bad("truncated palette");
        palette.assign(256 * 3, 0);
        for (uint32_t i = 0; i < used; ++i) { palette[i * 3] = f[po + i * eb + 2]; palette[i * 3 + 1] = f[po + i * eb + 1]; palette[i * 3 + 2] = f[po + i * eb]; }
    }
    src_channels = bpp == 32 ? 4 : 3;
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)dataOff + stride * h > f.size()) bad("truncated pixel data");
    px.assign((size_t)w * h * src_channels, 0);
    auto shift_of = [](uint32_t m) { int s = 0; while (m && !(m & 1u)) { m >>= 1; ++s; } return s; };
    const int sr = shift_of(mr), sg = shift_of(mg), sb = shift_of(mb), sa = shift_of(ma);
    bool anyAlpha = false;
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = &f[dataOff + stride * (size_t)(topDown ? y : h - 1 - y)];
        for (int x = 0; x < w; ++x) {
            uint8_t* d = &px[((size_t)y * w + x) * src_channels];
            if (bpp == 8) { std::memcpy(d, &palette[(size_t)row[x] * 3], 3); }
            else if (bpp == 24) { d[0] = row[x * 3 + 2]; d[1] = row[x * 3 + 1]; d[2] = row[x * 3]; }
            else {
                const uint32_t v = le32(&row[(size_t)x * 4]);
                d[0] = (uint8_t)((v & mr) >> sr); d[1] = (uint8_t)((v & mg) >> sg); d[2] = (uint8_t)((v & mb) >> sb);
                d[3] = ma ? (uint8_t)((v & ma) >> sa) : 255;
                anyAlpha |= d[3] != 0;
            }
        }
    }
    if (bpp == 32 && !anyAlpha)                                      // all-zero alpha channel = an opaque image (stb_image does the same)
        for (size_t i = 3; i < px.size(); i += 4) px[i] = 255;
}

inline std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("Failed to load image: " + path);
    size_t n = (size_t)in.tellg();
    std::vector<uint8_t> b(n);
    in.seekg(0); in.read(reinterpret_cast<char*>(b.data()), (std::streamsize)n);
    return b;
}

}  // namespace detail

// stbi_load(path, ..., desired_channels) + optional vertical flip.  desired_channels: 1 (STBI_grey) or 4 (STBI_rgb_alpha).
inline Image decode_image(const std::vector<uint8_t>& f, const std::string& path, int desired_channels, bool flip_vertically);
inline Image load_image(const std::string& path, int desired_channels, bool flip_vertically) {
    return decode_image(detail::read_file(path), path, desired_channels, flip_vertically);
}
// The same from memory (`path` only names the file in messages and, for TGA, supplies the extension).
inline Image decode_image(const std::vector<uint8_t>& f, const std::string& path, int desired_channels, bool flip_vertically) {
    if (desired_channels != 1 && desired_channels != 4) throw std::runtime_error("load_image: desired_channels must be 1 or 4");
    int w = 0, h = 0, sc = 0;
    std::vector<uint8_t> px;
    static const uint8_t pngsig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() >= 8 && std::memcmp(f.data(), pngsig, 8) == 0) detail::decode_png(f, w, h, sc, px, desired_channels);
    else if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) detail::decode_pnm(f, w, h, sc, px);
    else if (f.size() >= 3 && f[0] == 0xff && f[1] == 0xd8 && f[2] == 0xff) detail::decode_jpeg(f, w, h, sc, px, desired_channels);
    else if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') detail::decode_bmp(f, w, h, sc, px);
    else if (path.size() >= 4 && (path.compare(path.size() - 4, 4, ".tga") == 0 || path.compare(path.size() - 4, 4, ".TGA") == 0)) detail::decode_tga(f, w, h, sc, px);
    else if (f.size() >= 10 && (std::memcmp(f.data(), "#?RADIANCE", 10) == 0 || std::memcmp(f.data(), "#?RGBE", 6) == 0)) {
        std::vector<float> rgb;
        detail::decode_hdr(f, w, h, rgb);
        // stb_image converts the channel count on the FLOAT data (grey = (r+g+b)/3, not the integer luma weights) and
        // only then maps to 8 bits: (float)pow(x, 1/2.2f) * 255 + 0.5f with a double pow(); alpha is 1.0 -> 255
        sc = desired_channels == 1 ? 1 : 3; px.resize((size_t)w * h * sc);
        const double g = (double)(1.0f / 2.2f);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            float v[3] = {rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]};
            if (sc == 1) v[0] = (v[0] + v[1] + v[2]) / 3.0f;
            for (int k = 0; k < sc; ++k) {
                float z = (float)std::pow((double)(v[k] * 1.0f), g) * 255.0f + 0.5f;
                if (z < 0.f) z = 0.f;
                if (z > 255.f) z = 255.f;
                px[i * sc + k] = (uint8_t)(int)z;
            }
        }
    } else throw std::runtime_error("Failed to load image: " + path + " (unsupported image format: PNG, JPEG, TGA, BMP, binary PGM/PPM and Radiance HDR are decoded)");
    Image out; out.width = w; out.height = h; out.channels = desired_channels;
    out.pixels.resize((size_t)w * h * desired_channels);
    for (int y = 0; y < h; ++y) {
        const int sy = flip_vertically ? h - 1 - y : y;
        for (int x = 0; x < w; ++x) {
            const uint8_t* s = &px[((size_t)sy * w + x) * sc];
            uint8_t* d = &out.pixels[((size_t)y * w + x) * desired_channels];
            if (desired_channels == 4) {
                if (sc == 1) { d[0] = d[1] = d[2] = s[0]; d[3] = 255; }
                else if (sc == 2) { d[0] = d[1] = d[2] = s[0]; d[3] = s[1]; }
                else if (sc == 3) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 255; }
                else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; }
            } else {
                d[0] = sc <= 2 ? s[0] : (uint8_t)((s[0] * 77 + s[1] * 150 + s[2] * 29) >> 8);
            }
        }
    }
    return out;
}

}  // namespace rtr::img
