// HOST half of the JPEG decode split for the loader (reference: engine.py:41-54 DataLoader workers running
// PIL `Image.open(...).convert("RGB")` = libjpeg-turbo, then configs/dataset/cub200.yaml:31-47's transform chain).
//
// A 20k images/s encoder outruns any CPU-side full decode (PIL: ~1.5-2 ms per 500x375 image per core), so the decoder is split where
// the hardware splits it:
//   * HOST (this file, plain C++, `ch_jpeg_plan` / `ch_jpeg_entropy_decode`): marker parsing and the Huffman entropy decode -- a serial
//     bit-stream walk, ~25 % of libjpeg's decode time -- on a pool of host threads, straight into a pinned buffer of int16 coefficient
//     blocks (natural order, DC prediction undone);
//   * GPU (`ch_jpeg_reconstruct`): dequantisation + 8x8 inverse DCT + chroma upsampling + YCbCr -> RGB, i.e. everything that is
//     data-parallel, written as decoded RGB bytes in the layout `ch_preprocess` consumes.
// The GPU half restates libjpeg-turbo's DEFAULT decompression arithmetic (what Pillow runs): `jpeg_idct_islow` (jidctint.c: 13-bit
// fixed-point LL&M, two passes, descale 11 / 18, range limit through the post-IDCT table), `h2v1_fancy_upsample` / `h2v2_fancy_upsample`
// (jdsample.c: triangle filter, biases 1/2 and 8/7, edge columns special-cased, context rows clamped at the image's first / last sample
// row as jdmainct.c's funny pointers do) and `ycc_rgb_convert` (jdcolor.c: 16-bit fixed-point tables).  Integer arithmetic throughout:
// the RGB bytes are BIT-EQUAL to Pillow's on every supported file (tests/test_jpeg.py, against PIL itself).
// Supported: 8-bit Huffman-coded files -- baseline / extended-sequential (SOF0 / SOF1) with one interleaved scan, and PROGRESSIVE (SOF2:
// any scan script of spectral selection + successive approximation that ends with every coefficient at bit 0) -- with 1 component (grey)
// or 3 components (YCbCr), luma sampling 1x1 (4:4:4), 2x1 (4:2:2) or 2x2 (4:2:0) and chroma 1x1, restart intervals.  Anything else
// (arithmetic, lossless, CMYK / Adobe RGB, 12 bit, multi-scan sequential, exotic sampling, tiny images, a progressive file whose low
// coefficients are not fully refined -- libjpeg smooths those) gets a non-zero `status` in its descriptor: the host side of the loader
// decodes exactly those files with PIL -- the reference's own path -- and counts them.
// This file is plain C++ (no HIP): marker parsing, the Huffman / progressive entropy decode and the batch file reads.  It is compiled into
// the library by the normal build AND, by tests/test_jpeg.py, into a stand-alone AddressSanitizer + UBSan object that the mutation test of
// the decoder runs against (the decoder parses bytes from disk).  The GPU half (kernels + ch_jpeg_reconstruct) is jpeg.hip.
#include <emmintrin.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/concepthash_hip.h"
#include "ch_host.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------
// host: marker parsing
// ---------------------------------------------------------------------------------------------------------------------------------
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int LOOK = 9;  // look-ahead bits of the Huffman tables

struct Huff {
    bool present = false;
    uint8_t counts[17] = {};
    uint8_t vals[256] = {};
    // derived
    uint16_t look[1 << LOOK];     // (length << 8) | symbol for codes of <= LOOK bits, 0 otherwise
    int32_t maxcode[18];          // largest code of each length (left-justified compare), -1 if none
    int32_t valoff[17];           // vals index of the first code of each length minus that code
    int16_t fast_ac[1 << LOOK];   // AC tables: (value << 8) | (run << 4) | total bits, when code + magnitude fit in LOOK bits; 0 otherwise
};

bool build_huff(Huff &h, bool is_ac) {
    int code = 0, k = 0;
    uint16_t codes[256];
    uint8_t sizes[256];
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < h.counts[len]; ++i) {
            if (k >= 256) return false;
            codes[k] = (uint16_t)code++;
            sizes[k++] = (uint8_t)len;
        }
        if (code > (1 << len)) return false;   // over-subscribed
        code <<= 1;
    }
    std::memset(h.look, 0, sizeof(h.look));
    int p = 0;
    code = 0;
    for (int len = 1; len <= 16; ++len) {
        if (h.counts[len]) {
            h.valoff[len] = p - codes[p];
            p += h.counts[len];
            h.maxcode[len] = codes[p - 1];
        } else {
            h.maxcode[len] = -1;
            h.valoff[len] = 0;
        }
    }
    h.maxcode[17] = 0x7fffffff;
    for (int i = 0; i < k; ++i) {
        if (sizes[i] <= LOOK) {
            const int first = codes[i] << (LOOK - sizes[i]);
            for (int j = 0; j < (1 << (LOOK - sizes[i])); ++j) h.look[first + j] = (uint16_t)((sizes[i] << 8) | h.vals[i]);
        }
    }
    std::memset(h.fast_ac, 0, sizeof(h.fast_ac));
    if (is_ac) {
        for (int i = 0; i < (1 << LOOK); ++i) {
            const uint16_t e = h.look[i];
            if (!e) continue;
            const int len = e >> 8, rs = e & 255, run = rs >> 4, mag = rs & 15;
            if (mag && len + mag <= LOOK) {
                int v = ((i << len) & ((1 << LOOK) - 1)) >> (LOOK - mag);
                if (v < (1 << (mag - 1))) v += (int)((~0u) << mag) + 1;   // EXTEND
                if (v >= -128 && v <= 127) h.fast_ac[i] = (int16_t)((v * 256) + (run * 16) + (len + mag));
            }
        }
    }
    h.present = true;
    return true;
}

struct Parsed {
    int status = 0;
    int width = 0, height = 0, ncomp = 0, hs = 1, vs = 1, restart = 0;
    int tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    uint16_t quant[4][64];
    bool have_q[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int64_t scan_begin = 0;   // first byte of the entropy-coded segment
    bool progressive = false; // SOF2: several scans (spectral selection / successive approximation); decoded by decode_progressive
    int64_t first_sos = 0;    // progressive: position of the first SOS marker's 0xFF (the scan walk starts there)
    int comp_id[3] = {0, 0, 0};
};

// status codes (also documented in include/concepthash_hip.h)
enum {
    JS_OK = 0, JS_NOT_JPEG = 1, JS_TRUNCATED = 2, JS_PROGRESSIVE_OR_OTHER_SOF = 3, JS_PRECISION = 4, JS_COMPONENTS = 5,
    JS_SAMPLING = 6, JS_MULTISCAN = 7, JS_COLORSPACE = 8, JS_TABLES = 9, JS_TINY = 10, JS_CORRUPT = 11, JS_TOO_LARGE = 12
};
// Pillow refuses images of more than 2 x MAX_IMAGE_PIXELS (DecompressionBombError): such a file is left to it, so that the caller sees the
// reference's error instead of a multi-gigabyte coefficient buffer
constexpr int64_t kMaxPixels = 2 * (int64_t)89478485;

inline int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

// header-only parse when `full` is false (no Huffman table construction)
void parse(const uint8_t *d, int64_t n, Parsed &P, bool full) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { P.status = JS_NOT_JPEG; return; }
    int64_t pos = 2;
    bool jfif = false, adobe = false, sof = false;
    int adobe_transform = -1;
    int comp_id[3] = {0, 0, 0}, comp_h[3] = {1, 1, 1}, comp_v[3] = {1, 1, 1};
    while (true) {
        if (pos + 4 > n) { P.status = JS_TRUNCATED; return; }
        if (d[pos] != 0xFF) { P.status = JS_CORRUPT; return; }
        while (pos < n && d[pos] == 0xFF) ++pos;   // fill bytes
        if (pos >= n) { P.status = JS_TRUNCATED; return; }
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;   // standalone markers
        if (m == 0xD9) { P.status = JS_TRUNCATED; return; }                   // EOI before SOS
        if (pos + 2 > n) { P.status = JS_TRUNCATED; return; }
        const int len = rd16(d + pos);
        if (len < 2 || pos + len > n) { P.status = JS_TRUNCATED; return; }
        const uint8_t *s = d + pos + 2;
        const int sl = len - 2;
        switch (m) {
            case 0xE0: if (sl >= 5 && !std::memcmp(s, "JFIF\0", 5)) jfif = true; break;
            case 0xEE: if (sl >= 12 && !std::memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; } break;
            case 0xDB: {
                int o = 0;
                while (o < sl) {
                    const int pq = s[o] >> 4, t = s[o] & 15;
                    ++o;
                    if (t > 3 || o + (pq ? 128 : 64) > sl) { P.status = JS_TABLES; return; }
                    for (int i = 0; i < 64; ++i) {
                        const int v = pq ? rd16(s + o + 2 * i) : s[o + i];
                        P.quant[t][kZigzag[i]] = (uint16_t)v;
                    }
                    P.have_q[t] = true;
                    o += pq ? 128 : 64;
                }
                break;
            }
            case 0xC0: case 0xC1: case 0xC2: {
                if (sof) { P.status = JS_CORRUPT; return; }
                sof = true;
                P.progressive = m == 0xC2;
                if (sl < 6) { P.status = JS_TRUNCATED; return; }
                if (s[0] != 8) { P.status = JS_PRECISION; return; }
                P.height = rd16(s + 1);
                P.width = rd16(s + 3);
                P.ncomp = s[5];
                if (P.ncomp != 1 && P.ncomp != 3) { P.status = JS_COMPONENTS; return; }
                if (sl < 6 + 3 * P.ncomp || P.height == 0 || P.width == 0) { P.status = JS_CORRUPT; return; }
                if ((int64_t)P.width * P.height > kMaxPixels) { P.status = JS_TOO_LARGE; return; }
                for (int c = 0; c < P.ncomp; ++c) {
                    comp_id[c] = P.comp_id[c] = s[6 + 3 * c];
                    comp_h[c] = s[7 + 3 * c] >> 4;
                    comp_v[c] = s[7 + 3 * c] & 15;
                    P.tq[c] = s[8 + 3 * c];
                    if (P.tq[c] > 3) { P.status = JS_TABLES; return; }
                }
                break;
            }
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                if (sl >= 6) {   // every SOFn has the same header: the caller's own decoder is told the size of the slot it fills
                    P.height = rd16(s + 1);
                    P.width = rd16(s + 3);
                }
                P.status = JS_PROGRESSIVE_OR_OTHER_SOF;
                return;
            case 0xC4: {
                int o = 0;
                while (o < sl) {
                    if (o + 17 > sl) { P.status = JS_TABLES; return; }
                    const int cls = s[o] >> 4, id = s[o] & 15;
                    if (cls > 1 || id > 3) { P.status = JS_TABLES; return; }
                    Huff &h = cls ? P.ac[id] : P.dc[id];
                    int total = 0;
                    h.counts[0] = 0;
                    for (int i = 1; i <= 16; ++i) { h.counts[i] = s[o + i]; total += s[o + i]; }
                    o += 17;
                    if (total > 256 || o + total > sl) { P.status = JS_TABLES; return; }
                    std::memcpy(h.vals, s + o, total);
                    o += total;
                    if (full) { if (!build_huff(h, cls == 1)) { P.status = JS_TABLES; return; } }
                    else h.present = true;
                }
                break;
            }
            case 0xDD: if (sl >= 2) P.restart = rd16(s); break;
            case 0xDA: {
                if (!sof) { P.status = JS_CORRUPT; return; }
                if (!P.progressive) {
                    if (sl < 1 || s[0] != P.ncomp || sl < 1 + 2 * P.ncomp + 3) { P.status = JS_MULTISCAN; return; }
                    for (int c = 0; c < P.ncomp; ++c) {
                        if (s[1 + 2 * c] != comp_id[c]) { P.status = JS_MULTISCAN; return; }
                        P.td[c] = s[2 + 2 * c] >> 4;
                        P.ta[c] = s[2 + 2 * c] & 15;
                        if (P.td[c] > 3 || P.ta[c] > 3 || !P.dc[P.td[c]].present || !P.ac[P.ta[c]].present || !P.have_q[P.tq[c]]) {
                            P.status = JS_TABLES;
                            return;
                        }
                    }
                    const uint8_t *t = s + 1 + 2 * P.ncomp;
                    if (t[0] != 0 || t[1] != 63 || t[2] != 0) { P.status = JS_PROGRESSIVE_OR_OTHER_SOF; return; }
                } else {
                    for (int c = 0; c < P.ncomp; ++c)
                        if (!P.have_q[P.tq[c]]) { P.status = JS_TABLES; return; }   // (libjpeg latches the tables at the first scan of a component)
                    P.first_sos = pos - 2;
                    while (P.first_sos > 0 && d[P.first_sos] != 0xFF) --P.first_sos;   // (fill bytes in front of the marker code)
                }
                // colour space, as libjpeg's default_decompress_parms decides it
                if (P.ncomp == 3) {
                    bool ycc;
                    if (jfif) ycc = true;
                    else if (adobe) ycc = adobe_transform == 1;
                    else ycc = comp_id[0] == 1 && comp_id[1] == 2 && comp_id[2] == 3;
                    if (!ycc) { P.status = JS_COLORSPACE; return; }
                    if (comp_h[1] != 1 || comp_v[1] != 1 || comp_h[2] != 1 || comp_v[2] != 1) { P.status = JS_SAMPLING; return; }
                    if (!((comp_h[0] == 1 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 1) || (comp_h[0] == 2 && comp_v[0] == 2))) {
                        P.status = JS_SAMPLING;
                        return;
                    }
                    P.hs = comp_h[0];
                    P.vs = comp_v[0];
                } else {
                    P.hs = P.vs = 1;   // a single-component scan is non-interleaved: one block per MCU whatever the factors say
                }
                if (P.width < 16 || P.height < 16) { P.status = JS_TINY; return; }   // the fancy upsamplers' narrow-image special cases
                P.scan_begin = pos + len;
                {
                    // every block costs at least one bit of entropy-coded data (its DC code): a header that announces more blocks than the
                    // rest of the file has bits is a damaged header, not a reason to allocate them
                    const int64_t mw = (P.width + 8 * P.hs - 1) / (8 * P.hs), mh = (P.height + 8 * P.vs - 1) / (8 * P.vs);
                    const int64_t blocks = mw * mh * (P.ncomp == 3 ? P.hs * P.vs + 2 : 1);
                    if (blocks > 8 * (n - P.scan_begin)) { P.status = JS_TRUNCATED; return; }
                }
                return;
            }
            default: break;
        }
        pos += len;
    }
}

void fill_desc(const Parsed &P, ch_jpeg_desc &d) {
    std::memset(&d, 0, sizeof(d));
    d.status = P.status;
    d.width = P.width;
    d.height = P.height;
    if (P.status) return;
    d.ncomp = P.ncomp;
    d.hs = P.hs;
    d.vs = P.vs;
    d.mcu_w = (P.width + 8 * P.hs - 1) / (8 * P.hs);
    d.mcu_h = (P.height + 8 * P.vs - 1) / (8 * P.vs);
    for (int c = 0; c < P.ncomp; ++c)
        for (int i = 0; i < 64; ++i) d.quant[c][i] = P.quant[P.tq[c]][i];
    const int64_t y = (int64_t)d.mcu_w * d.hs * d.mcu_h * d.vs;
    d.nblocks = (int32_t)(d.ncomp == 3 ? y + 2 * (int64_t)d.mcu_w * d.mcu_h : y);
}

inline int64_t blocks_of(const ch_jpeg_desc &d) { return d.nblocks; }

// ---------------------------------------------------------------------------------------------------------------------------------
// host: Huffman entropy decode of one image into int16 coefficient blocks (natural order)
// ---------------------------------------------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p, *end;
    uint64_t acc = 0;   // valid bits are the TOP `bits` bits
    int bits = 0;
    bool marker = false;   // a marker (FF xx, xx != 0) was reached: only zero bits are fed from here on
    bool eof = false;      // ... or the FILE ended inside the entropy-coded data (a truncated file: libjpeg warns, Pillow raises)

    // callers need at most 31 valid bits at a time (a 16-bit code + a 15-bit magnitude): top up only when fewer than 32 are left
    inline void ensure32() {
        if (bits < 32) refill();
    }
    void refill() {
        while (bits <= 56) {
            if (!marker && p + 8 <= end) {
                uint64_t v;
                std::memcpy(&v, p, 8);
                // any 0xFF byte among the next 8?  (byte-wise test for a zero byte of ~v)
                const uint64_t nv = ~v;
                if (!((nv - 0x0101010101010101ull) & ~nv & 0x8080808080808080ull)) {
                    const int take = (64 - bits) >> 3;   // whole bytes that fit: 1..8
                    v = __builtin_bswap64(v);
                    if (take == 8) {
                        acc = v;
                        bits = 64;
                    } else {
                        acc |= (v >> (64 - 8 * take)) << (64 - bits - 8 * take);
                        bits += 8 * take;
                    }
                    p += take;
                    continue;
                }
            }
            unsigned b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0) p += 2;
                    else { marker = true; b = 0; }
                } else {
                    ++p;
                }
            } else {
                if (!marker) eof = true;
                marker = true;
            }
            acc |= (uint64_t)b << (56 - bits);
            bits += 8;
        }
    }
    inline unsigned peek(int n) const { return (unsigned)(acc >> (64 - n)); }
    inline void skip(int n) { acc <<= n; bits -= n; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v + (int)((~0u) << s) + 1 : v; }

// -> symbol, or -1 on a bad code.  Needs >= 16 valid bits.
inline int decode_sym(BitReader &br, const Huff &h) {
    const unsigned look = br.peek(LOOK);
    const uint16_t e = h.look[look];
    if (e) {
        br.skip(e >> 8);
        return e & 255;
    }
    const unsigned top = br.peek(16);
    for (int len = LOOK + 1; len <= 16; ++len) {
        const int code = (int)(top >> (16 - len));
        if (code <= h.maxcode[len]) {
            br.skip(len);
            return h.vals[(code + h.valoff[len]) & 255];
        }
    }
    return -1;
}

bool decode_block(BitReader &br, const Huff &dc, const Huff &ac, int &pred, int16_t *blk) {
    std::memset(blk, 0, 128);
    br.ensure32();
    int s = decode_sym(br, dc);
    if (s < 0 || s > 11) return false;
    if (s) {
        const int v = (int)br.peek(s);
        br.skip(s);
        pred += extend(v, s);
    }
    blk[0] = (int16_t)pred;
    int k = 1;
    while (k < 64) {
        br.ensure32();
        const int16_t f = ac.fast_ac[br.peek(LOOK)];
        if (f) {
            k += (f >> 4) & 15;
            if (k > 63) return false;
            br.skip(f & 15);
            blk[kZigzag[k++]] = (int16_t)(f >> 8);
            continue;
        }
        const int rs = decode_sym(br, ac);
        if (rs < 0) return false;
        const int r = rs >> 4, sz = rs & 15;
        if (!sz) {
            if (r != 15) break;   // EOB
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) return false;
        const int v = (int)br.peek(sz);
        br.skip(sz);
        blk[kZigzag[k++]] = (int16_t)extend(v, sz);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host: progressive JPEG (SOF2) -- the scans of ITU-T T.81 annex G (spectral selection + successive approximation) accumulated into the
// same coefficient blocks the sequential decoder fills; the GPU half is unchanged (libjpeg runs the same IDCT / upsampling / colour
// conversion on a progressive file's final coefficients).  One thing libjpeg adds for progressive files only: inter-block smoothing
// (jdcoefct.c) when the low AC coefficients are not fully refined.  That cannot happen for a file whose scans bring DC and the first
// nine AC coefficients of every component down to bit 0; a file that does not is reported (status 3) and decoded by the caller's PIL.
// ---------------------------------------------------------------------------------------------------------------------------------
inline int get_bits(BitReader &br, int n) {   // n <= 16
    br.ensure32();
    const int v = (int)br.peek(n);
    br.skip(n);
    return v;
}

struct CompGrid {
    int16_t *base;   // first block of the component
    int stride;      // blocks per row of the (MCU-padded) component array
    int h, v;        // sampling factors inside an interleaved MCU
    int bw, bh;      // blocks per row / column of a NON-interleaved scan: ceil(component size / 8)
};

// byte-align and consume the expected RSTn; false when it is not there
inline bool take_restart(BitReader &br, int &next_rst) {
    br.acc = 0;
    br.bits = 0;
    br.marker = false;
    const uint8_t *q = br.p;
    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
        if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) return false;
        ++q;
    }
    if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) return false;
    br.p = q + 2;
    next_rst = (next_rst + 1) & 7;
    return true;
}

// One AC refinement pass over a block (T.81 figure G.7): returns false on a bad code.  `nz` is the block's map of non-zero coefficients by
// zig-zag position (kept by the AC passes): the pass has to (a) hand one correction bit to every non-zero coefficient it passes, in order,
// and (b) count ZERO-history positions for the run lengths -- with the map both are bit operations on the positions that matter, instead
// of a walk over all 63 positions of every block of every refinement scan.  (Measured: the four refinement scans of a typical file are
// three quarters of its decode time with or without the map -- ~180 k correction / sign bits per 500 x 375 image at ~8 ns each; the map
// is worth 9 %: 671 -> 731 images/s per thread against 2,100 for baseline files.  Reading the bits 16 at a time and applying them without
// branches measured slower, 640.)
inline bool refine_block(BitReader &br, const Huff &ac, int16_t *blk, uint64_t &nz, int Ss, int Se, int Al, int &eobrun) {
    const int p1 = 1 << Al, m1 = -(1 << Al);
    const uint64_t band = (Se == 63 ? ~0ull : ((1ull << (Se + 1)) - 1)) & ~((1ull << Ss) - 1);
    // correction bits for the non-zero coefficients at the positions of `set`, ascending
    auto correct = [&](uint64_t set) {
        while (set) {
            const int k = __builtin_ctzll(set);
            set &= set - 1;
            int16_t &c = blk[kZigzag[k]];
            if (get_bits(br, 1) && !(c & p1)) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
        }
    };
    int k = Ss;
    if (eobrun == 0) {
        while (k <= Se) {
            br.ensure32();
            const int rs = decode_sym(br, ac);
            if (rs < 0) return false;
            int r = rs >> 4;
            const int sz = rs & 15;
            int value = 0;
            if (sz) {
                if (sz != 1) return false;
                value = get_bits(br, 1) ? p1 : m1;      // a newly non-zero coefficient: its sign now, its position after the run
            } else if (r != 15) {
                eobrun = 1 << r;
                if (r) eobrun += get_bits(br, r);
                break;                                   // the rest of the band belongs to the end-of-band run (below)
            }
            // position of the (r + 1)-th zero-history coefficient at or after k; the non-zero ones in front of it take their bits
            const uint64_t from = ~((1ull << k) - 1) & band;
            uint64_t zeros = ~nz & from;
            for (; r > 0 && zeros; --r) zeros &= zeros - 1;
            if (!zeros) {                                // the band ends first: legal for a run of zeros (ZRL), not for a coefficient
                correct(nz & from);
                if (value) return false;
                k = Se + 1;
                break;
            }
            const int pos = __builtin_ctzll(zeros);
            correct(nz & from & ((1ull << pos) - 1));
            if (value) {
                blk[kZigzag[pos]] = (int16_t)value;
                nz |= 1ull << pos;
            }
            k = pos + 1;
        }
    }
    if (eobrun > 0) {
        if (k <= Se) correct(nz & band & ~((1ull << k) - 1));
        --eobrun;
    }
    return true;
}

int decode_progressive(const uint8_t *d, int64_t n, const ch_jpeg_desc &desc, Parsed &P, int16_t *coef) {
    const int ybw = desc.mcu_w * desc.hs, ybh = desc.mcu_h * desc.vs;
    CompGrid grid[3];
    grid[0] = {coef, ybw, desc.hs, desc.vs, (desc.width + 7) / 8, (desc.height + 7) / 8};
    if (desc.ncomp == 3) {
        const int cw = (desc.width + desc.hs - 1) / desc.hs, chh = (desc.height + desc.vs - 1) / desc.vs;   // component size, rounded up
        grid[1] = {coef + (int64_t)64 * ybw * ybh, desc.mcu_w, 1, 1, (cw + 7) / 8, (chh + 7) / 8};
        grid[2] = {grid[1].base + (int64_t)64 * desc.mcu_w * desc.mcu_h, desc.mcu_w, 1, 1, (cw + 7) / 8, (chh + 7) / 8};
    }
    std::memset(coef, 0, sizeof(int16_t) * 64 * (size_t)desc.nblocks);
    std::vector<uint64_t> nzmap((size_t)desc.nblocks, 0);   // per block: non-zero AC coefficients by zig-zag position (refine_block)
    int8_t coef_al[3][10];   // successive-approximation bit of the last scan that carried zig-zag coefficient 0..9; -1 = never seen
    std::memset(coef_al, -1, sizeof(coef_al));
    int restart = P.restart;
    int64_t pos = P.first_sos;
    bool eoi = false;
    while (!eoi) {
        if (pos + 2 > n) return JS_TRUNCATED;
        if (d[pos] != 0xFF) return JS_CORRUPT;
        while (pos < n && d[pos] == 0xFF) ++pos;
        if (pos >= n) return JS_TRUNCATED;
        const int m = d[pos++];
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > n) return JS_TRUNCATED;
        const int len = rd16(d + pos);
        if (len < 2 || pos + len > n) return JS_TRUNCATED;
        const uint8_t *sg = d + pos + 2;
        const int sl = len - 2;
        if (m == 0xC4) {
            int o = 0;
            while (o < sl) {
                if (o + 17 > sl) return JS_TABLES;
                const int cls = sg[o] >> 4, id = sg[o] & 15;
                if (cls > 1 || id > 3) return JS_TABLES;
                Huff &h = cls ? P.ac[id] : P.dc[id];
                int total = 0;
                for (int i = 1; i <= 16; ++i) { h.counts[i] = sg[o + i]; total += sg[o + i]; }
                o += 17;
                if (total > 256 || o + total > sl) return JS_TABLES;
                std::memcpy(h.vals, sg + o, total);
                o += total;
                if (!build_huff(h, false)) return JS_TABLES;   // (the combined run / size / value table is the sequential decoder's)
            }
        } else if (m == 0xDD) {
            if (sl >= 2) restart = rd16(sg);
        } else if (m == 0xDB || (m >= 0xC0 && m <= 0xCF)) {
            return JS_PROGRESSIVE_OR_OTHER_SOF;   // tables or frames redefined between scans: the caller's decoder
        } else if (m == 0xDA) {
            if (sl < 1) return JS_CORRUPT;
            const int ns = sg[0];
            if (ns < 1 || ns > desc.ncomp || sl < 1 + 2 * ns + 3) return JS_CORRUPT;
            int ci[3], td[3], ta[3];
            for (int i = 0; i < ns; ++i) {
                ci[i] = -1;
                for (int c = 0; c < desc.ncomp; ++c)
                    if (P.comp_id[c] == sg[1 + 2 * i]) ci[i] = c;
                if (ci[i] < 0 || (i && ci[i] <= ci[i - 1])) return JS_CORRUPT;
                td[i] = sg[2 + 2 * i] >> 4;
                ta[i] = sg[2 + 2 * i] & 15;
                if (td[i] > 3 || ta[i] > 3) return JS_TABLES;
            }
            const uint8_t *t = sg + 1 + 2 * ns;
            const int Ss = t[0], Se = t[1], Ah = t[2] >> 4, Al = t[2] & 15;
            if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah && Ah != Al + 1)) return JS_CORRUPT;
            for (int i = 0; i < ns; ++i) {
                if (Ss == 0 && !Ah && !P.dc[td[i]].present) return JS_TABLES;
                if (Ss > 0 && !P.ac[ta[i]].present) return JS_TABLES;
                for (int k = Ss; k <= Se && k < 10; ++k) coef_al[ci[i]][k] = (int8_t)Al;
            }
            BitReader br;
            br.p = d + pos + len;
            br.end = d + n;
            int pred[3] = {0, 0, 0}, eobrun = 0, to_restart = restart, next_rst = 0;
            // MCU raster of the scan: the frame's MCU grid when interleaved, the component's own block raster otherwise
            const bool inter = ns > 1;
            const int rows = inter ? desc.mcu_h : grid[ci[0]].bh, cols = inter ? desc.mcu_w : grid[ci[0]].bw;
            for (int my = 0; my < rows; ++my)
                for (int mx = 0; mx < cols; ++mx) {
                    if (restart) {
                        if (to_restart == 0) {
                            if (!take_restart(br, next_rst)) return JS_CORRUPT;
                            pred[0] = pred[1] = pred[2] = 0;
                            eobrun = 0;
                            to_restart = restart;
                        }
                        --to_restart;
                    }
                    for (int i = 0; i < ns; ++i) {
                        const CompGrid &g = grid[ci[i]];
                        const int nh = inter ? g.h : 1, nv = inter ? g.v : 1;
                        for (int v = 0; v < nv; ++v)
                            for (int h = 0; h < nh; ++h) {
                                int16_t *blk = g.base + (int64_t)64 * ((int64_t)(my * nv + v) * g.stride + mx * nh + h);
                                if (Ss == 0) {
                                    if (!Ah) {   // DC, first pass: the sequential decoder's difference coding, scaled by the point transform
                                        br.ensure32();
                                        const int sz = decode_sym(br, P.dc[td[i]]);
                                        if (sz < 0 || sz > 11) return JS_CORRUPT;
                                        if (sz) pred[i] += extend(get_bits(br, sz), sz);
                                        blk[0] = (int16_t)(pred[i] * (1 << Al));
                                    } else if (get_bits(br, 1)) {   // DC refinement: one more bit
                                        blk[0] = (int16_t)(blk[0] | (1 << Al));
                                    }
                                } else if (!Ah) {   // AC band, first pass (figure G.3 with end-of-band runs)
                                    if (eobrun > 0) {
                                        --eobrun;
                                        continue;
                                    }
                                    const Huff &ac = P.ac[ta[i]];
                                    for (int k = Ss; k <= Se; ++k) {
                                        br.ensure32();
                                        const int rs = decode_sym(br, ac);
                                        if (rs < 0) return JS_CORRUPT;
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz) {
                                            k += r;
                                            if (k > Se) return JS_CORRUPT;
                                            blk[kZigzag[k]] = (int16_t)(extend(get_bits(br, sz), sz) * (1 << Al));
                                            nzmap[(size_t)(blk - coef) >> 6] |= 1ull << k;
                                        } else if (r == 15) {
                                            k += 15;
                                        } else {
                                            eobrun = 1 << r;
                                            if (r) eobrun += get_bits(br, r);
                                            --eobrun;
                                            break;
                                        }
                                    }
                                } else if (!refine_block(br, P.ac[ta[i]], blk, nzmap[(size_t)(blk - coef) >> 6], Ss, Se, Al, eobrun)) {
                                    return JS_CORRUPT;
                                }
                            }
                    }
                }
            if (br.eof) return JS_TRUNCATED;
            // the next marker: the reader never steps over one, so it is at or after br.p
            const uint8_t *q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
            if (q + 1 >= br.end) break;   // no EOI: what has been decoded stands, as libjpeg's premature-end handling leaves it
            pos = q - d;
            continue;
        }
        pos += len;
    }
    // libjpeg smooths blocks of a progressive file whose DC / first AC coefficients are not known to bit 0: not reproduced here
    for (int c = 0; c < desc.ncomp; ++c)
        for (int k = 0; k < 10; ++k)
            if (coef_al[c][k] != 0) return JS_PROGRESSIVE_OR_OTHER_SOF;
    return JS_OK;
}

// coef: this image's blocks -- component 0 [rows][cols][64], then components 1, 2
int entropy_decode_one(const uint8_t *d, int64_t n, const ch_jpeg_desc &desc, int16_t *coef) {
    Parsed P;
    parse(d, n, P, true);
    if (P.status) return P.status;
    if (P.progressive) return decode_progressive(d, n, desc, P, coef);
    BitReader br;
    br.p = d + P.scan_begin;
    br.end = d + n;
    int pred[3] = {0, 0, 0};
    const int ybw = desc.mcu_w * desc.hs;   // luma blocks per row
    int16_t *cb = coef + (int64_t)64 * ybw * desc.mcu_h * desc.vs;
    int16_t *cr = cb + (int64_t)64 * desc.mcu_w * desc.mcu_h;
    const Huff &dc0 = P.dc[P.td[0]], &ac0 = P.ac[P.ta[0]];
    int to_restart = P.restart, next_rst = 0;
    for (int my = 0; my < desc.mcu_h; ++my) {
        for (int mx = 0; mx < desc.mcu_w; ++mx) {
            if (P.restart) {
                if (to_restart == 0) {
                    // byte-align, expect RSTn
                    br.acc = 0;
                    br.bits = 0;
                    br.marker = false;
                    const uint8_t *q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                        if (q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF) return JS_CORRUPT;
                        ++q;
                    }
                    if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) return JS_CORRUPT;
                    br.p = q + 2;
                    next_rst = (next_rst + 1) & 7;
                    pred[0] = pred[1] = pred[2] = 0;
                    to_restart = P.restart;
                }
                --to_restart;
            }
            for (int v = 0; v < desc.vs; ++v)
                for (int h = 0; h < desc.hs; ++h) {
                    int16_t *blk = coef + (int64_t)64 * ((int64_t)(my * desc.vs + v) * ybw + mx * desc.hs + h);
                    if (!decode_block(br, dc0, ac0, pred[0], blk)) return JS_CORRUPT;
                }
            if (desc.ncomp == 3) {
                const int64_t ci = (int64_t)64 * ((int64_t)my * desc.mcu_w + mx);
                if (!decode_block(br, P.dc[P.td[1]], P.ac[P.ta[1]], pred[1], cb + ci)) return JS_CORRUPT;
                if (!decode_block(br, P.dc[P.td[2]], P.ac[P.ta[2]], pred[2], cr + ci)) return JS_CORRUPT;
            }
        }
    }
    // the data ran out before a marker: a truncated file.  What was decoded is zero-padded garbage from there on; Pillow raises on
    // such a file, so it is handed to the caller's decoder, which will say so
    return br.eof ? JS_TRUNCATED : JS_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int ch_jpeg_plan(const uint8_t *const *files, const int64_t *lens, int32_t n, ch_jpeg_desc *desc, int64_t *total_coef,
                            int64_t *total_pix, int64_t *total_plane) {
    CH_REQUIRE(n >= 0 && (n == 0 || (files && lens && desc)), "jpeg_plan: null argument");
    int64_t co = 0, po = 0, pl = 0;
    for (int i = 0; i < n; ++i) {
        Parsed P;
        if (!files[i] || lens[i] < 4) P.status = JS_NOT_JPEG;
        else parse(files[i], lens[i], P, false);
        fill_desc(P, desc[i]);
        desc[i].coef_offset = co;
        desc[i].pix_offset = po;
        desc[i].plane_offset = pl;
        if (!P.status) {
            const int64_t nb = blocks_of(desc[i]);
            co += nb * 64;
            pl += (nb * 64 + 15) / 16 * 16;
        }
        po += (int64_t)desc[i].width * desc[i].height * 3;   // fallback images (status != 0 but a readable size) keep their slot
    }
    if (total_coef) *total_coef = co;
    if (total_pix) *total_pix = po;
    if (total_plane) *total_plane = pl;
    return 0;
}

extern "C" int ch_jpeg_entropy_decode(const uint8_t *const *files, const int64_t *lens, int32_t n, ch_jpeg_desc *desc, int16_t *coef_host,
                                      int32_t nthreads) {
    CH_REQUIRE(n >= 0 && (n == 0 || (files && lens && desc && coef_host)), "jpeg_entropy_decode: null argument");
    if (n == 0) return 0;
    std::atomic<int> next{0};
    // Each thread decodes an image into its OWN scratch (a 500 x 375 image is 590 KB of blocks: it stays in the core's L2 while the
    // per-block zero fill and the scattered coefficient writes happen) and then streams the finished blocks to the (pinned) destination
    // with non-temporal stores.  Built while chasing a 17-30 ms in-pipeline decode time that turned out to be CPU-quota throttling
    // (DESIGN.md section 4c): on the MI355X box it measures the same 4.8 ms per 256 images as decoding in place, 5-10 % faster on one
    // thread; kept because the destination is then written exactly once, front to back.
    auto work = [&]() {
        std::vector<int16_t> scratch;
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            if (desc[i].status) continue;
            const size_t ncoef = (size_t)desc[i].nblocks * 64;
            if (scratch.size() < ncoef) scratch.resize(ncoef);
            const int st = entropy_decode_one(files[i], lens[i], desc[i], scratch.data());
            if (st) {
                desc[i].status = st;   // a corrupt stream: the caller falls back to its host decoder for this file
                continue;
            }
            int16_t *dst = coef_host + desc[i].coef_offset;
            if (((uintptr_t)dst & 15) == 0 && ((uintptr_t)scratch.data() & 15) == 0) {
                const __m128i *sv = (const __m128i *)scratch.data();
                __m128i *dv = (__m128i *)dst;
                for (size_t k = 0; k < ncoef / 8; ++k) _mm_stream_si128(dv + k, _mm_load_si128(sv + k));
                _mm_sfence();   // weakly ordered stores: visible before this thread signals completion (the DMA engine reads them next)
            } else {
                std::memcpy(dst, scratch.data(), ncoef * 2);
            }
        }
    };
    const int nt = std::max(1, std::min<int>(nthreads, n));
    if (nt == 1) {
        work();
        return 0;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return 0;
}

// the same two host calls for a batch whose files sit back to back in ONE buffer (what a `gpu_decode` loader worker produces):
// file i = data[offsets[i], offsets[i + 1]); no per-file pointers for the caller to build
extern "C" int ch_jpeg_plan_packed(const uint8_t *data, const int64_t *offsets, int32_t n, ch_jpeg_desc *desc, int64_t *total_coef,
                                   int64_t *total_pix, int64_t *total_plane) {
    CH_REQUIRE(n >= 0 && (n == 0 || (data && offsets && desc)), "jpeg_plan_packed: null argument");
    std::vector<const uint8_t *> files(n);
    std::vector<int64_t> lens(n);
    for (int i = 0; i < n; ++i) {
        CH_REQUIRE(offsets[i + 1] >= offsets[i], "jpeg_plan_packed: offsets must be non-decreasing");
        files[i] = data + offsets[i];
        lens[i] = offsets[i + 1] - offsets[i];
    }
    return ch_jpeg_plan(files.data(), lens.data(), n, desc, total_coef, total_pix, total_plane);
}
extern "C" int ch_jpeg_entropy_decode_packed(const uint8_t *data, const int64_t *offsets, int32_t n, ch_jpeg_desc *desc, int16_t *coef_host,
                                             int32_t nthreads) {
    CH_REQUIRE(n >= 0 && (n == 0 || (data && offsets && desc && coef_host)), "jpeg_entropy_decode_packed: null argument");
    std::vector<const uint8_t *> files(n);
    std::vector<int64_t> lens(n);
    for (int i = 0; i < n; ++i) {
        files[i] = data + offsets[i];
        lens[i] = offsets[i + 1] - offsets[i];
    }
    return ch_jpeg_entropy_decode(files.data(), lens.data(), n, desc, coef_host, nthreads);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host: reading a batch's files (what is left of the reference's loader workers once nothing is decoded on the CPU: engine.py:41-54 +
// the dataset classes' `Image.open(path)`).  Plain POSIX reads on a few threads, no Python in the loop: a `gpu_decode` loader needs no
// worker PROCESSES (0.35-1.1 s to start per loader, a shared-memory hop per batch) -- concepthash_amd/engine.FileBatchLoader.
// ---------------------------------------------------------------------------------------------------------------------------------
extern "C" int ch_io_file_sizes(const char *const *paths, int32_t n, int64_t *sizes) {
    CH_REQUIRE(n >= 0 && (n == 0 || (paths && sizes)), "io_file_sizes: null argument");
    for (int i = 0; i < n; ++i) {
        struct stat st;
        if (!paths[i] || ::stat(paths[i], &st) != 0 || !S_ISREG(st.st_mode)) {
            ch_set_error((std::string("io_file_sizes: cannot stat '") + (paths[i] ? paths[i] : "(null)") + "'").c_str());
            return 3;
        }
        sizes[i] = (int64_t)st.st_size;
    }
    return 0;
}
// file i -> dst[offsets[i], offsets[i] + sizes[i]); fails (status 3) when a file is shorter or cannot be opened
extern "C" int ch_io_read_files(const char *const *paths, int32_t n, const int64_t *offsets, const int64_t *sizes, uint8_t *dst,
                                int32_t nthreads) {
    CH_REQUIRE(n >= 0 && (n == 0 || (paths && offsets && sizes && dst)), "io_read_files: null argument");
    if (n == 0) return 0;
    std::atomic<int> next{0}, bad{-1};
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            const int fd = ::open(paths[i], O_RDONLY | O_CLOEXEC);
            bool ok = fd >= 0;
            int64_t got = 0;
            while (ok && got < sizes[i]) {
                const ssize_t r = ::read(fd, dst + offsets[i] + got, (size_t)(sizes[i] - got));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) ok = false;
                else got += r;
            }
            if (fd >= 0) ::close(fd);
            if (!ok) {
                int expect = -1;
                bad.compare_exchange_strong(expect, i);
            }
        }
    };
    const int nt = std::max(1, std::min<int>(nthreads, n));
    if (nt == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
    }
    if (bad.load() >= 0) {
        ch_set_error((std::string("io_read_files: cannot read '") + paths[bad.load()] + "' (missing, or shorter than its size a moment ago)").c_str());
        return 3;
    }
    return 0;
}

