// jpeg_decode.hpp — baseline and progressive JPEG (ITU-T T.81) decoding for the host image layer.
//
// The reference loads textures through stb_image (reference src/core/file.cppm:272-291); JPEG decoding is not
// bit-exactly specified by T.81, so to hand the renderer the SAME texels this decoder makes the same choices as
// stb_image's documented pipeline: the IJG "slow integer" inverse DCT (Loeffler-Ligtenberg-Moschytz, 12-bit constants,
// 10-bit / 17-bit descaling), triangle-filter chroma upsampling (3:1 weights; 9:3:3:1 for 2x2), and the 20-bit
// fixed-point YCbCr -> RGB matrix.  tests/test_images.py holds it byte-equal to the real stb_image of the reference
// tree (oracle/_ref/stb_dump) on baseline, progressive, grey, 4:4:4 / 4:2:2 / 4:2:0 and restart-interval files.
// Not decoded (refused): arithmetic coding, lossless, 12-bit samples, CMYK / YCCK.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtr::img::detail {

inline void jpeg_fail(const char* what) { throw std::runtime_error(std::string("Failed to load image: bad JPEG (") + what + ")"); }

static const uint8_t kJpegNatural[64 + 15] = {      // zig-zag position -> natural (row-major) position; padded so k may overrun
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct JpegHuffman {
    bool defined = false;
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    void build(const uint8_t* counts, const uint8_t* symbols, int total) {
        std::memcpy(vals, symbols, (size_t)total);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k; mincode[len] = code;
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) jpeg_fail("bad Huffman code lengths");
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        defined = true;
    }
};

struct JpegComponent {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0;             // size in samples
    int w2 = 0, h2 = 0;           // padded plane size (whole MCUs)
    int dcpred = 0;
    std::vector<uint8_t> plane;
    std::vector<int16_t> coef;    // progressive only: 64 per block, w2/8 blocks per row
};

struct JpegDecoder {
    const uint8_t* p; size_t n, pos = 0;
    uint32_t bitbuf = 0; int bitcnt = 0; int marker = -1; bool nomore = false;
    uint16_t dequant[4][64];
    JpegHuffman hdc[4], hac[4];
    JpegComponent comp[4];
    int ncomp = 0, imgx = 0, imgy = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, sawAdobe = false, jfif = false; int adobeTransform = -1; bool rgbIds = false;
    int restartInterval = 0, todo = 0;
    int scanN = 0, order[4];
    int specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;

    int get8() { if (pos >= n) jpeg_fail("truncated file"); return p[pos++]; }
    int get16() { int a = get8(); return (a << 8) | get8(); }

    // ---- entropy-coded segment: MSB-first bits, FF00 unstuffing, markers end the data (zeros are fed afterwards) ----
    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!nomore) {
                b = pos < n ? p[pos++] : 0;
                if (b == 0xff) {
                    int c = pos < n ? p[pos++] : 0;
                    while (c == 0xff) c = pos < n ? p[pos++] : 0;
                    if (c != 0) { marker = c; nomore = true; b = 0; }
                }
            }
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int k) {                       // k <= 16
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        const int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k; bitcnt -= k;
        return v;
    }
    int getbit() { return getbits(1); }
    int decode(const JpegHuffman& h) {
        if (!h.defined) jpeg_fail("scan uses an undefined Huffman table");
        if (bitcnt < 16) fill();
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (int)(bitbuf >> (32 - len));
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) {
                bitbuf <<= len; bitcnt -= len;
                return h.vals[h.valptr[len] + code - h.mincode[len]];
            }
        }
        jpeg_fail("bad Huffman code");
        return 0;
    }
    int extend_receive(int s) {                // T.81 F.2.2.1 EXTEND(RECEIVE(s), s)
        const int v = getbits(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    void reset_entropy() {
        bitbuf = 0; bitcnt = 0; nomore = false; marker = -1;
        for (int i = 0; i < 4; ++i) comp[i].dcpred = 0;
        todo = restartInterval ? restartInterval : 0x7fffffff;
        eobRun = 0;
    }

    // ---- one 8x8 block, sequential mode (T.81 F.2.2) ----
    void decode_block(int16_t* data, JpegComponent& c) {
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int t = decode(hdc[c.td]);
        if (t > 15) jpeg_fail("bad DC category");
        const int diff = t ? extend_receive(t) : 0;
        c.dcpred += diff;
        data[0] = (int16_t)(c.dcpred * dequant[c.tq][0]);
        int k = 1;
        do {
            const int rs = decode(hac[c.ta]), s = rs & 15, r = rs >> 4;
            if (s == 0) { if (rs != 0xf0) break; k += 16; }
            else {
                k += r;
                const int z = kJpegNatural[k++];
                data[z] = (int16_t)(extend_receive(s) * dequant[c.tq][z]);
            }
        } while (k < 64);
    }
    // ---- progressive mode (T.81 G.1.2) ----
    void decode_block_prog_dc(int16_t* data, JpegComponent& c) {
        if (specEnd != 0) jpeg_fail("DC scan with a spectral range");
        if (succHigh == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode(hdc[c.td]);
            if (t > 15) jpeg_fail("bad DC category");
            const int diff = t ? extend_receive(t) : 0;
            c.dcpred += diff;
            data[0] = (int16_t)(c.dcpred * (1 << succLow));
        } else if (getbit()) data[0] = (int16_t)(data[0] + (1 << succLow));
    }
    void decode_block_prog_ac(int16_t* data, const JpegHuffman& h) {
        if (specStart == 0) jpeg_fail("AC scan starting at the DC coefficient");
        if (succHigh == 0) {
            if (eobRun) { --eobRun; return; }
            int k = specStart;
            do {
                const int rs = decode(h), s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) { eobRun = 1 << r; if (r) eobRun += getbits(r); --eobRun; break; }
                    k += 16;
                } else {
                    k += r;
                    const int z = kJpegNatural[k++];
                    data[z] = (int16_t)(extend_receive(s) * (1 << succLow));
                }
            } while (k <= specEnd);
        } else {
            const int16_t bit = (int16_t)(1 << succLow);
            auto refine = [&](int16_t* q) {
                if (getbit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
            };
            if (eobRun) {
                --eobRun;
                for (int k = specStart; k <= specEnd; ++k) { int16_t* q = &data[kJpegNatural[k]]; if (*q != 0) refine(q); }
            } else {
                int k = specStart;
                do {
                    const int rs = decode(h);
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) { eobRun = (1 << r) - 1; if (r) eobRun += getbits(r); r = 64; }
                    } else {
                        if (s != 1) jpeg_fail("bad refinement code");
                        s = getbit() ? bit : -bit;
                    }
                    while (k <= specEnd) {
                        int16_t* q = &data[kJpegNatural[k++]];
                        if (*q != 0) refine(q);
                        else { if (r == 0) { *q = (int16_t)s; break; } --r; }
                    }
                } while (k <= specEnd);
            }
        }
    }

    // ---- IJG "slow integer" inverse DCT, as stb_image scales it: 12-bit constants, columns descaled by 10 bits (2 extra
    //      bits of precision kept), rows by 17 bits with the +128 level shift folded in ----
    static int f2f(float x) { return (int)(x * 4096 + 0.5); }     // the sign stays INSIDE: (int) truncates towards zero
    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    struct Odd { int x0, x1, x2, x3, t0, t1, t2, t3; };
    static Odd idct1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
        Odd o;
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * f2f(0.5411961f);
        int t2 = p1 + p3 * f2f(-1.847759065f);
        int t3 = p1 + p2 * f2f(0.765366865f);
        p2 = s0; p3 = s4;
        int t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
        o.x0 = t0 + t3; o.x3 = t0 - t3; o.x1 = t1 + t2; o.x2 = t1 - t2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2; int p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
        const int p5 = (p3 + p4) * f2f(1.175875602f);
        t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f); t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f);
        p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);
        p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);
        o.t3 = t3 + p1 + p4; o.t2 = t2 + p2 + p3; o.t1 = t1 + p2 + p4; o.t0 = t0 + p1 + p3;
        return o;
    }
    static void idct_block(uint8_t* out, int stride, const int16_t* d) {
        int val[64];
        for (int i = 0; i < 8; ++i) {
            const int16_t* c = d + i; int* v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dc = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
            } else {
                Odd o = idct1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
                o.x0 += 512; o.x1 += 512; o.x2 += 512; o.x3 += 512;
                v[0] = (o.x0 + o.t3) >> 10; v[56] = (o.x0 - o.t3) >> 10;
                v[8] = (o.x1 + o.t2) >> 10; v[48] = (o.x1 - o.t2) >> 10;
                v[16] = (o.x2 + o.t1) >> 10; v[40] = (o.x2 - o.t1) >> 10;
                v[24] = (o.x3 + o.t0) >> 10; v[32] = (o.x3 - o.t0) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) {
            const int* v = val + i * 8; uint8_t* o8 = out + (size_t)i * stride;
            Odd o = idct1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const int bias = 65536 + (128 << 17);
            o.x0 += bias; o.x1 += bias; o.x2 += bias; o.x3 += bias;
            o8[0] = clamp8((o.x0 + o.t3) >> 17); o8[7] = clamp8((o.x0 - o.t3) >> 17);
            o8[1] = clamp8((o.x1 + o.t2) >> 17); o8[6] = clamp8((o.x1 - o.t2) >> 17);
            o8[2] = clamp8((o.x2 + o.t1) >> 17); o8[5] = clamp8((o.x2 - o.t1) >> 17);
            o8[3] = clamp8((o.x3 + o.t0) >> 17); o8[4] = clamp8((o.x3 - o.t0) >> 17);
        }
    }

    // ---- markers ----
    void read_dqt(int len) {
        while (len > 0) {
            const int q = get8(), prec = q >> 4, t = q & 15;
            if (prec > 1 || t > 3) jpeg_fail("bad DQT");
            for (int i = 0; i < 64; ++i) dequant[t][kJpegNatural[i]] = (uint16_t)(prec ? get16() : get8());
            len -= prec ? 129 : 65;
        }
        if (len != 0) jpeg_fail("bad DQT length");
    }
    void read_dht(int len) {
        while (len > 0) {
            const int q = get8(), tc = q >> 4, th = q & 15;
            if (tc > 1 || th > 3) jpeg_fail("bad DHT");
            uint8_t counts[16], symbols[256]; int total = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)get8(); total += counts[i]; }
            if (total > 256) jpeg_fail("bad DHT");
            for (int i = 0; i < total; ++i) symbols[i] = (uint8_t)get8();
            (tc ? hac[th] : hdc[th]).build(counts, symbols, total);
            len -= 17 + total;
        }
        if (len != 0) jpeg_fail("bad DHT length");
    }
    void read_sof(int len) {
        if (len < 8) jpeg_fail("bad SOF length");
        if (get8() != 8) jpeg_fail("only 8-bit samples are decoded");
        imgy = get16(); imgx = get16(); ncomp = get8();
        if (imgx <= 0 || imgy <= 0) jpeg_fail("bad size");
        if ((size_t)imgx * (size_t)imgy > ((size_t)1 << 28)) jpeg_fail("image larger than 2^28 pixels");
        if (ncomp != 1 && ncomp != 3) jpeg_fail("only 1- and 3-component files are decoded (no CMYK / YCCK)");
        if (len != 6 + 3 * ncomp) jpeg_fail("bad SOF length");
        static const char rgb[3] = {'R', 'G', 'B'};
        rgbIds = ncomp == 3;
        for (int i = 0; i < ncomp; ++i) {
            JpegComponent& c = comp[i];
            c.id = get8(); const int q = get8(); c.h = q >> 4; c.v = q & 15; c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) jpeg_fail("bad component");
            if (ncomp == 3 && c.id != rgb[i]) rgbIds = false;
            hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
        }
        for (int i = 0; i < ncomp; ++i) if (hmax % comp[i].h || vmax % comp[i].v) jpeg_fail("fractional sampling ratios are not decoded");
        const int mcuw = hmax * 8, mcuh = vmax * 8;
        mcux = (imgx + mcuw - 1) / mcuw; mcuy = (imgy + mcuh - 1) / mcuh;
        for (int i = 0; i < ncomp; ++i) {
            JpegComponent& c = comp[i];
            c.x = (imgx * c.h + hmax - 1) / hmax; c.y = (imgy * c.v + vmax - 1) / vmax;
            c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) c.coef.assign((size_t)c.w2 * c.h2, 0);
        }
    }
    void read_sos(int len) {
        scanN = get8();
        if (scanN < 1 || scanN > ncomp || len != 4 + 2 * scanN) jpeg_fail("bad SOS");
        for (int i = 0; i < scanN; ++i) {
            const int id = get8(), q = get8();
            int which = -1;
            for (int k = 0; k < ncomp; ++k) if (comp[k].id == id) which = k;
            if (which < 0) jpeg_fail("scan names an unknown component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) jpeg_fail("bad table selector");
            order[i] = which;
        }
        specStart = get8(); specEnd = get8();
        const int a = get8(); succHigh = a >> 4; succLow = a & 15;
        if (progressive) { if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) jpeg_fail("bad progressive scan parameters"); }
        else { if (specStart != 0 || succHigh != 0 || succLow != 0) jpeg_fail("bad sequential scan parameters"); specEnd = 63; }
    }
    void restart_if_due() {
        if (--todo > 0) return;
        if (bitcnt < 24) fill();
        if (!(marker >= 0xd0 && marker <= 0xd7)) { todo = 0x7fffffff; return; }   // no RSTn here: the data simply ends
        reset_entropy();
    }
    void decode_scan() {
        reset_entropy();
        int16_t block[64];
        if (scanN == 1) {
            JpegComponent& c = comp[order[0]];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    if (!progressive) { decode_block(block, c); idct_block(&c.plane[(size_t)j * 8 * c.w2 + (size_t)i * 8], c.w2, block); }
                    else {
                        int16_t* d = &c.coef[64 * ((size_t)i + (size_t)j * (c.w2 >> 3))];
                        if (specStart == 0) decode_block_prog_dc(d, c); else decode_block_prog_ac(d, hac[c.ta]);
                    }
                    restart_if_due();
                }
        } else {
            for (int j = 0; j < mcuy; ++j)
                for (int i = 0; i < mcux; ++i) {
                    for (int k = 0; k < scanN; ++k) {
                        JpegComponent& c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x) {
                                const int bx = i * c.h + x, by = j * c.v + y;
                                if (!progressive) { decode_block(block, c); idct_block(&c.plane[(size_t)by * 8 * c.w2 + (size_t)bx * 8], c.w2, block); }
                                else decode_block_prog_dc(&c.coef[64 * ((size_t)bx + (size_t)by * (c.w2 >> 3))], c);
                            }
                    }
                    restart_if_due();
                }
        }
    }
    void finish_progressive() {
        for (int k = 0; k < ncomp; ++k) {
            JpegComponent& c = comp[k];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    int16_t* d = &c.coef[64 * ((size_t)i + (size_t)j * (c.w2 >> 3))];
                    for (int q = 0; q < 64; ++q) d[q] = (int16_t)(d[q] * dequant[c.tq][q]);
                    idct_block(&c.plane[(size_t)j * 8 * c.w2 + (size_t)i * 8], c.w2, d);
                }
        }
    }

    void decode_file() {
        if (get8() != 0xff || get8() != 0xd8) jpeg_fail("no SOI");
        bool haveFrame = false, done = false;
        int m = -1;
        while (!done) {
            if (m < 0) {
                int b = get8();
                while (b != 0xff) b = get8();            // stray bytes between segments are skipped
                do { b = get8(); } while (b == 0xff);
                m = b;
            }
            const int mk = m; m = -1;
            if (mk == 0xd9) { done = true; break; }
            if (mk >= 0xd0 && mk <= 0xd7) continue;
            const int len = get16() - 2;
            if (len < 0) jpeg_fail("bad segment length");
            const size_t end = pos + (size_t)len;
            if (end > n) jpeg_fail("truncated segment");
            switch (mk) {
                case 0xdb: read_dqt(len); break;
                case 0xc4: read_dht(len); break;
                case 0xc0: case 0xc1: case 0xc2:
                    if (haveFrame) jpeg_fail("more than one frame");
                    progressive = mk == 0xc2; read_sof(len); haveFrame = true; break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    jpeg_fail("lossless / hierarchical / arithmetic-coded JPEG is not decoded"); break;
                case 0xdd: if (len != 2) jpeg_fail("bad DRI"); restartInterval = get16(); break;
                case 0xe0:                                 // APP0 "JFIF": the file is YCbCr whatever an Adobe segment says
                    if (len >= 5 && std::memcmp(p + pos, "JFIF", 5) == 0) jfif = true;
                    pos = end; break;
                case 0xee:                                 // APP14 "Adobe": colour transform flag
                    if (len >= 12 && std::memcmp(p + pos, "Adobe", 5) == 0) { sawAdobe = true; adobeTransform = p[pos + 11]; }
                    pos = end; break;
                case 0xda: {
                    if (!haveFrame) jpeg_fail("scan before the frame header");
                    read_sos(len);
                    decode_scan();
                    if (marker >= 0) { m = marker; }        // the marker that ended the entropy-coded data
                    else {                                 // resynchronise on the next marker in the byte stream
                        while (pos < n) { if (p[pos] == 0xff && pos + 1 < n && p[pos + 1] != 0 && p[pos + 1] != 0xff) break; ++pos; }
                        if (pos >= n) done = true;
                    }
                    continue;
                }
                default: pos = end; break;                  // APPn, COM, DNL ...: skipped
            }
            if (pos != end) jpeg_fail("segment length mismatch");
        }
        if (!haveFrame) jpeg_fail("no frame");
        if (progressive) finish_progressive();
    }
};

// chroma (or any sub-sampled component) up-sampling, one output row
inline const uint8_t* jpeg_resample_row(std::vector<uint8_t>& line, const uint8_t* nearRow, const uint8_t* farRow, int w, int hs, int vs) {
    uint8_t* out = line.data();
    if (hs == 1 && vs == 1) return nearRow;
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * nearRow[i] + farRow[i] + 2) >> 2); return out; }
    if (hs == 2 && vs == 1) {
        const uint8_t* in = nearRow;
        if (w == 1) { out[0] = out[1] = in[0]; return out; }
        out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) {
            const int t = 3 * in[i] + 2;
            out[i * 2] = (uint8_t)((t + in[i - 1]) >> 2); out[i * 2 + 1] = (uint8_t)((t + in[i + 1]) >> 2);
        }
        out[i * 2] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2); out[i * 2 + 1] = in[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * nearRow[0] + farRow[0] + 2) >> 2); return out; }
        int t1 = 3 * nearRow[0] + farRow[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1; t1 = 3 * nearRow[i] + farRow[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4); out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];      // other ratios: nearest
    return out;
}

// decodes to `src_channels` = 1 (grey file, or the luma plane when one channel is wanted) or 3 (RGB) samples, top row first
inline void decode_jpeg(const std::vector<uint8_t>& f, int& w, int& h, int& src_channels, std::vector<uint8_t>& px, int desired_channels) {
    JpegDecoder d; d.p = f.data(); d.n = f.size();
    std::memset(d.dequant, 0, sizeof d.dequant);
    d.decode_file();
    w = d.imgx; h = d.imgy;
    const bool isRgb = d.ncomp == 3 && (d.rgbIds || (d.sawAdobe && d.adobeTransform == 0 && !d.jfif));
    const int decodeN = (d.ncomp == 3 && desired_channels < 3 && !isRgb) ? 1 : d.ncomp;
    src_channels = (d.ncomp == 1 || desired_channels < 3) ? 1 : 3;
    px.assign((size_t)w * h * src_channels, 0);
    struct Up { int hs, vs, wl, ystep, ypos; const uint8_t* line0; const uint8_t* line1; std::vector<uint8_t> buf; };
    Up up[3];
    for (int k = 0; k < decodeN; ++k) {
        const JpegComponent& c = d.comp[k];
        up[k].hs = d.hmax / c.h; up[k].vs = d.vmax / c.v; up[k].wl = (w + up[k].hs - 1) / up[k].hs;
        up[k].ystep = up[k].vs >> 1; up[k].ypos = 0; up[k].line0 = up[k].line1 = c.plane.data();
        up[k].buf.assign((size_t)w + 3 * 8 + 8, 0);
    }
    const int cr_r = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, cr_g = -(((int)(0.71414f * 4096.0f + 0.5f)) << 8);
    const int cb_g = -(((int)(0.34414f * 4096.0f + 0.5f)) << 8), cb_b = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
    for (int j = 0; j < h; ++j) {
        const uint8_t* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < decodeN; ++k) {
            Up& u = up[k];
            const bool bot = u.ystep >= (u.vs >> 1);
            row[k] = jpeg_resample_row(u.buf, bot ? u.line1 : u.line0, bot ? u.line0 : u.line1, u.wl, u.hs, u.vs);
            if (++u.ystep >= u.vs) {
                u.ystep = 0; u.line0 = u.line1;
                if (++u.ypos < d.comp[k].y) u.line1 += d.comp[k].w2;
            }
        }
        uint8_t* out = &px[(size_t)j * w * src_channels];
        if (src_channels == 3) {
            if (isRgb) for (int i = 0; i < w; ++i) { out[i * 3] = row[0][i]; out[i * 3 + 1] = row[1][i]; out[i * 3 + 2] = row[2][i]; }
            else for (int i = 0; i < w; ++i) {
                const int yf = (row[0][i] << 20) + (1 << 19), cb = row[1][i] - 128, cr = row[2][i] - 128;
                int r = yf + cr * cr_r, g = yf + cr * cr_g + (int)((uint32_t)(cb * cb_g) & 0xffff0000u), b = yf + cb * cb_b;
                r >>= 20; g >>= 20; b >>= 20;
                out[i * 3] = JpegDecoder::clamp8(r); out[i * 3 + 1] = JpegDecoder::clamp8(g); out[i * 3 + 2] = JpegDecoder::clamp8(b);
            }
        } else if (d.ncomp == 3 && isRgb) {
            for (int i = 0; i < w; ++i) out[i] = (uint8_t)((row[0][i] * 77 + row[1][i] * 150 + row[2][i] * 29) >> 8);
        } else {
            std::memcpy(out, row[0], (size_t)w);
        }
    }
}

}  // namespace rtr::img::detail
