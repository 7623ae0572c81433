// jpeg.cpp — baseline JPEG in and out for the host layer: what the reference gets from the `image 0.23.14` crate
// at raytracer/src/texture/mod.rs:89-100 (image::open → jpeg-decoder 0.1.22 → RGB8) and at main.rs:213-221
// (JPEGEncoder::new_with_quality(.., 100)). Neither crate is in /root/reference (Cargo.lock:210-226), so this
// restates the published algorithms they are built from — "parity unpinned": texels may differ from the Rust
// decoder's by an LSB; tests/test_assets.py holds this decoder to libjpeg-turbo (through PIL) within a stated bound.
//
//   decode  baseline sequential DCT (SOF0 / SOF1), 8-bit, 1 or 3 components, sampling factors 1 or 2 per axis,
//           restart intervals; integer IDCT of the stb_image family (the one jpeg-decoder's idct.rs carries),
//           triangle-filter ("fancy") chroma upsampling as in libjpeg / jpeg-decoder's upsampler.rs,
//           JFIF YCbCr -> RGB in f32 with round-half-up and clamp
//   encode  baseline, 4:4:4, Annex-K tables scaled by the IJG quality rule (quality 100 = all ones),
//           Annex-K Huffman tables, AAN-free float forward DCT
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "scene_api.hpp"

namespace rt2022 {

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    // canonical code lengths 1..16: first code, first symbol index and count per length
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    bool present = false;
    void build(const uint8_t counts[16], const uint8_t *symbols, int n) {
        std::memcpy(vals, symbols, (size_t)n);
        int32_t code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;            // blocks per line / column, padded to whole MCUs
    std::vector<uint8_t> plane;                 // blocks_w*8 x blocks_h*8 samples
    int dc_pred = 0;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool marker_hit = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() {
        while (nbits <= 24) {
            uint32_t byte = 0;
            if (!marker_hit && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;          // stuffed zero
                    else { marker_hit = true; byte = 0; }             // a marker: feed zeros from here on
                } else {
                    p++;
                }
            }
            acc |= byte << (24 - nbits);
            nbits += 8;
        }
    }
    int bit() {
        if (nbits < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1; nbits--;
        return b;
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n; nbits -= n;
        return v;
    }
    void reset() { acc = 0; nbits = 0; marker_hit = false; }
};

int decode_symbol(BitReader &br, const Huff &h) {
    int32_t code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + (code - h.mincode[len])];
    }
    return -1;
}
inline int extend(int v, int n) { return (n > 0 && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

// Integer IDCT, 8x8, the stb_image formulation (12-bit fixed point constants; first pass keeps 2 extra bits,
// second pass rounds to the sample and adds the 128 level shift).
inline int f2f(double x) { return (int)(x * 4096.0 + 0.5); }
inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }
#define RT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                   \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                          \
    p2 = s2; p3 = s6;                                                               \
    p1 = (p2 + p3) * f2f(0.5411961);                                                 \
    t2 = p1 + p3 * f2f(-1.847759065);                                                \
    t3 = p1 + p2 * f2f(0.765366865);                                                 \
    p2 = s0; p3 = s4;                                                               \
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                    \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                          \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                             \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                          \
    p5 = (p3 + p4) * f2f(1.175875602);                                               \
    t0 = t0 * f2f(0.298631336); t1 = t1 * f2f(2.053119869);                          \
    t2 = t2 * f2f(3.072711026); t3 = t3 * f2f(1.501321110);                          \
    p1 = p5 + p1 * f2f(-0.899976223); p2 = p5 + p2 * f2f(-2.562915447);              \
    p3 = p3 * f2f(-1.961570560); p4 = p4 * f2f(-0.390180644);                        \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(const int coef[64], uint8_t *out, int stride) {
    int tmp[64];
    for (int i = 0; i < 8; i++) {                         // columns
        const int *d = coef + i;
        int *v = tmp + i;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            int dc = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            RT_IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {                         // rows
        const int *v = tmp + i * 8;
        uint8_t *o = out + i * stride;
        RT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

// One output row of a chroma plane upsampled 2x horizontally with the triangle filter: 3/4 nearer + 1/4 farther
// sample, rounding alternately down and up (libjpeg h2v1_fancy_upsample).
void upsample_h2(const int *in, int n_in, int *out) {
    if (n_in == 1) { out[0] = out[1] = in[0]; return; }
    out[0] = in[0] * 4 + 0;                                   // (scaled by 4: callers divide at the end)
    out[1] = in[0] * 3 + in[1];
    for (int i = 1; i < n_in - 1; i++) {
        out[2 * i] = in[i] * 3 + in[i - 1];
        out[2 * i + 1] = in[i] * 3 + in[i + 1];
    }
    out[2 * (n_in - 1)] = in[n_in - 1] * 3 + in[n_in - 2];
    out[2 * (n_in - 1) + 1] = in[n_in - 1] * 4;
}

} // namespace

bool jpeg_decode_rgb8(const uint8_t *data, size_t size, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgb,
                      std::string &err) {
    auto fail = [&](const char *m) { err = m; return false; };
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG (no SOI)");
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    std::vector<Component> comps;
    int restart_interval = 0, hmax = 1, vmax = 1;
    bool have_frame = false;
    size_t pos = 2;
    while (pos + 4 <= size) {
        if (data[pos] != 0xFF) return fail("marker expected");
        uint8_t m = data[pos + 1];
        pos += 2;
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        if (pos + 2 > size) return fail("truncated segment");
        uint32_t len = be16(data + pos);
        if (len < 2 || pos + len > size) return fail("bad segment length");
        const uint8_t *seg = data + pos + 2, *seg_end = data + pos + len;
        if (m == 0xDB) {                                                       // DQT
            while (seg < seg_end) {
                int pq = seg[0] >> 4, tq = seg[0] & 15;
                if (tq > 3 || pq > 1) return fail("bad DQT");
                if (pq == 1) return fail("16-bit quantisation tables are not supported (8-bit baseline files use 8-bit tables)");
                seg++;
                if (seg + (pq ? 128 : 64) > seg_end) return fail("truncated DQT");
                for (int i = 0; i < 64; i++) { qt[tq][kZigzag[i]] = pq ? (uint16_t)be16(seg + 2 * i) : seg[i]; }
                seg += pq ? 128 : 64;
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) {                                                // DHT
            while (seg < seg_end) {
                if (seg + 17 > seg_end) return fail("truncated DHT");
                int tc = seg[0] >> 4, th = seg[0] & 15;
                if (tc > 1 || th > 3) return fail("bad DHT");
                int n = 0;
                for (int i = 0; i < 16; i++) n += seg[1 + i];
                if (n > 256 || seg + 17 + n > seg_end) return fail("bad DHT counts");
                (tc ? hac[th] : hdc[th]).build(seg + 1, seg + 17, n);
                seg += 17 + n;
            }
        } else if (m == 0xC0 || m == 0xC1) {                                   // SOF0 / SOF1: baseline, extended sequential (Huffman)
            if (seg + 6 > seg_end || seg[0] != 8) return fail("only 8-bit samples are supported");
            height = be16(seg + 1); width = be16(seg + 3);
            int nc = seg[5];
            if (width == 0 || height == 0 || (nc != 1 && nc != 3) || seg + 6 + 3 * nc > seg_end) return fail("unsupported frame header");
            // (the planes and the RGB image are allocated from these two header fields alone: 2^28 pixels = 805 MB of RGB at most)
            if ((uint64_t)width * height > (1ull << 28)) return fail("image larger than 2^28 pixels");
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                Component &c = comps[(size_t)i];
                c.id = seg[6 + 3 * i]; c.h = seg[7 + 3 * i] >> 4; c.v = seg[7 + 3 * i] & 15; c.tq = seg[8 + 3 * i];
                if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail("unsupported sampling factors");
                hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return fail("progressive / lossless / arithmetic JPEG is not supported (baseline only)");
        } else if (m == 0xDD) {                                                // DRI
            if (seg + 2 > seg_end) return fail("bad DRI");
            restart_interval = (int)be16(seg);
        } else if (m == 0xDA) {                                                // SOS: the one scan of a baseline file
            if (!have_frame) return fail("SOS before SOF");
            if (seg >= seg_end) return fail("empty SOS header");
            int ns = seg[0];
            if (ns != (int)comps.size() || seg + 1 + 2 * ns + 3 > seg_end) return fail("scan does not cover all components");
            for (int i = 0; i < ns; i++) {
                int cid = seg[1 + 2 * i], tbl = seg[2 + 2 * i];
                bool found = false;
                for (Component &c : comps) if (c.id == cid) { c.td = tbl >> 4; c.ta = tbl & 15; found = true; }
                if (!found) return fail("scan names an unknown component");
            }
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            const int mcus_x = ((int)width + mcu_w - 1) / mcu_w, mcus_y = ((int)height + mcu_h - 1) / mcu_h;
            for (Component &c : comps) {
                if (!qt_present[c.tq] || c.td > 3 || c.ta > 3 || !hdc[c.td].present || !hac[c.ta].present) return fail("scan uses a table that was never defined");
                c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
                c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
                c.dc_pred = 0;
            }
            BitReader br(data + pos + len, data + size);
            int until_restart = restart_interval;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart_interval && until_restart == 0) {              // RSTn: byte-align, skip the marker, reset predictors
                        const uint8_t *q = br.p;
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                        if (q + 1 < br.end) q += 2;
                        br.p = q; br.reset();
                        for (Component &c : comps) c.dc_pred = 0;
                        until_restart = restart_interval;
                    }
                    for (Component &c : comps)
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++) {
                                int coef[64] = {0};
                                int t = decode_symbol(br, hdc[c.td]);
                                if (t < 0 || t > 11) return fail("bad DC code");
                                int diff = t ? extend(br.bits(t), t) : 0;
                                c.dc_pred += diff;
                                // (a conforming stream keeps the predictor inside 16 bits; a hostile one may not: no overflow either way)
                                if (c.dc_pred < -32768 || c.dc_pred > 32767) return fail("DC predictor out of range");
                                coef[0] = c.dc_pred * (int)qt[c.tq][0];
                                for (int k = 1; k < 64;) {
                                    int rs = decode_symbol(br, hac[c.ta]);
                                    if (rs < 0) return fail("bad AC code");
                                    int r = rs >> 4, s = rs & 15;
                                    if (s == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) return fail("AC run past the block");
                                    int z = kZigzag[k];
                                    coef[z] = extend(br.bits(s), s) * (int)qt[c.tq][z];
                                    k++;
                                }
                                int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                                idct_block(coef, &c.plane[(size_t)py * c.blocks_w * 8 + px], c.blocks_w * 8);
                            }
                    if (restart_interval) until_restart--;
                }
            // ---- to RGB8, top-down ------------------------------------------------------------
            rgb.assign((size_t)width * height * 3, 0);
            if (comps.size() == 1) {
                const Component &c = comps[0];
                for (uint32_t y = 0; y < height; y++)
                    for (uint32_t x = 0; x < width; x++) {
                        uint8_t v = c.plane[(size_t)y * c.blocks_w * 8 + x];
                        uint8_t *o = &rgb[((size_t)y * width + x) * 3];
                        o[0] = o[1] = o[2] = v;
                    }
                return true;
            }
            // chroma rows at full resolution, as sums scaled by 4 (h) x 4 (v) = 16 where upsampled
            std::vector<int> line[3], up_near, up_far, tmp_in;
            for (int i = 0; i < 3; i++) line[i].assign((size_t)width + 2 * 8, 0);
            for (uint32_t y = 0; y < height; y++) {
                int scale[3];
                for (int ci = 0; ci < 3; ci++) {
                    const Component &c = comps[(size_t)ci];
                    const int stride = c.blocks_w * 8;
                    const int hs = hmax / c.h, vs = vmax / c.v;                   // 1 or 2
                    const int cw = ((int)width * c.h + hmax - 1) / hmax;          // real samples per line of this component
                    const int ch = ((int)height * c.v + vmax - 1) / vmax;
                    auto row = [&](int r) { return &c.plane[(size_t)(r < 0 ? 0 : r >= ch ? ch - 1 : r) * stride]; };
                    tmp_in.assign((size_t)cw, 0);
                    int vscale = 1;
                    if (vs == 1) {
                        const uint8_t *r0 = row((int)y);
                        for (int x = 0; x < cw; x++) tmp_in[(size_t)x] = r0[x];
                    } else {                                                      // 3/4 nearer row + 1/4 farther row
                        int cy = (int)y / 2;
                        const uint8_t *rn = row(cy), *rf = row((y & 1) ? cy + 1 : cy - 1);
                        for (int x = 0; x < cw; x++) tmp_in[(size_t)x] = 3 * rn[x] + rf[x];
                        vscale = 4;
                    }
                    if (hs == 1) {
                        for (uint32_t x = 0; x < width; x++) line[ci][x] = tmp_in[x];
                        scale[ci] = vscale;
                    } else {
                        up_near.assign((size_t)cw * 2 + 2, 0);
                        upsample_h2(tmp_in.data(), cw, up_near.data());
                        for (uint32_t x = 0; x < width; x++) line[ci][x] = up_near[x];
                        scale[ci] = vscale * 4;
                    }
                }
                for (uint32_t x = 0; x < width; x++) {
                    auto sample = [&](int ci) {                                   // round to nearest, halves up (libjpeg: +8 >> 4, +2 >> 2 alternately +1 / +2)
                        int s = scale[ci], v = line[ci][x];
                        return s == 1 ? v : (v + s / 2) / s;
                    };
                    float Y = (float)sample(0), cb = (float)sample(1) - 128.0f, cr = (float)sample(2) - 128.0f;
                    float r = Y + 1.40200f * cr, g = Y - 0.34414f * cb - 0.71414f * cr, b = Y + 1.77200f * cb;
                    auto to8 = [](float v) { int i = (int)std::floor(v + 0.5f); return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i); };
                    uint8_t *o = &rgb[((size_t)y * width + x) * 3];
                    o[0] = to8(r); o[1] = to8(g); o[2] = to8(b);
                }
            }
            return true;
        }
        pos += len;
    }
    return fail("no scan found");
}

// ------------------------------------------------------------------------------------------------ encoder ----
namespace {
const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                            14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                            49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                              47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumaCounts[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaCounts[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaCounts[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaCounts[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct EncTable { uint16_t code[256]; uint8_t len[256]; };
void build_enc(const uint8_t counts[16], const uint8_t *vals, EncTable &t) {
    std::memset(&t, 0, sizeof t);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < counts[len - 1]; i++) { t.code[vals[k]] = (uint16_t)code; t.len[vals[k]] = (uint8_t)len; code++; k++; }
        code <<= 1;
    }
}
struct BitWriter {
    std::vector<uint8_t> &out;
    uint32_t acc = 0;
    int n = 0;
    void put(uint32_t bits, int len) {
        acc = (acc << len) | (bits & ((1u << len) - 1u));
        n += len;
        while (n >= 8) {
            uint8_t b = (uint8_t)(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0x00);
            n -= 8;
        }
    }
    void flush() { if (n > 0) put(0x7F, 8 - n); }
};
void put16(std::vector<uint8_t> &o, uint32_t v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
int bit_size(int v) { v = v < 0 ? -v : v; int n = 0; while (v) { n++; v >>= 1; } return n; }

void fdct8x8(const float in[64], float out[64]) {            // separable DCT-II, orthonormal scaling of T.81 A.3.3
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int u = 0; u < 8; u++)
            for (int x = 0; x < 8; x++) c[u][x] = (float)((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) { float s = 0; for (int x = 0; x < 8; x++) s += c[u][x] * in[y * 8 + x]; tmp[y * 8 + u] = s; }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) { float s = 0; for (int y = 0; y < 8; y++) s += c[v][y] * tmp[y * 8 + u]; out[v * 8 + u] = s; }
}
} // namespace

bool jpeg_encode_rgb8(const uint8_t *rgb, uint32_t width, uint32_t height, int quality, std::vector<uint8_t> &out) {
    if (!rgb || width == 0 || height == 0 || width > 65535 || height > 65535) return false;
    quality = quality < 1 ? 1 : quality > 100 ? 100 : quality;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;            // IJG quality scaling
    uint8_t q[2][64];
    for (int i = 0; i < 64; i++) {
        int a = ((int)kLumaQ[i] * scale + 50) / 100, b = ((int)kChromaQ[i] * scale + 50) / 100;
        q[0][i] = (uint8_t)(a < 1 ? 1 : a > 255 ? 255 : a);
        q[1][i] = (uint8_t)(b < 1 ? 1 : b > 255 ? 255 : b);
    }
    out.clear();
    out.push_back(0xFF); out.push_back(0xD8);
    const uint8_t jfif[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 2, 0, 0, 1, 0, 1, 0, 0};
    out.insert(out.end(), jfif, jfif + sizeof jfif);
    for (int t = 0; t < 2; t++) {
        out.push_back(0xFF); out.push_back(0xDB); put16(out, 67); out.push_back((uint8_t)t);
        for (int i = 0; i < 64; i++) out.push_back(q[t][kZigzag[i]]);
    }
    out.push_back(0xFF); out.push_back(0xC0); put16(out, 17); out.push_back(8); put16(out, height); put16(out, width); out.push_back(3);
    for (int i = 0; i < 3; i++) { out.push_back((uint8_t)(i + 1)); out.push_back(0x11); out.push_back((uint8_t)(i ? 1 : 0)); }
    auto dht = [&](int cls_id, const uint8_t *counts, const uint8_t *vals, int n) {
        out.push_back(0xFF); out.push_back(0xC4); put16(out, (uint32_t)(19 + n)); out.push_back((uint8_t)cls_id);
        out.insert(out.end(), counts, counts + 16); out.insert(out.end(), vals, vals + n);
    };
    dht(0x00, kDcLumaCounts, kDcVals, 12); dht(0x10, kAcLumaCounts, kAcLumaVals, 162);
    dht(0x01, kDcChromaCounts, kDcVals, 12); dht(0x11, kAcChromaCounts, kAcChromaVals, 162);
    const uint8_t sos[] = {0xFF, 0xDA, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
    out.insert(out.end(), sos, sos + sizeof sos);
    EncTable dc[2], ac[2];
    build_enc(kDcLumaCounts, kDcVals, dc[0]); build_enc(kDcChromaCounts, kDcVals, dc[1]);
    build_enc(kAcLumaCounts, kAcLumaVals, ac[0]); build_enc(kAcChromaCounts, kAcChromaVals, ac[1]);
    BitWriter bw{out};
    int pred[3] = {0, 0, 0};
    for (uint32_t by = 0; by < height; by += 8)
        for (uint32_t bx = 0; bx < width; bx += 8)
            for (int ci = 0; ci < 3; ci++) {
                float blk[64], coef[64];
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) {
                        uint32_t sy = by + y < height ? by + y : height - 1, sx = bx + x < width ? bx + x : width - 1;
                        const uint8_t *p = rgb + ((size_t)sy * width + sx) * 3;
                        float r = p[0], g = p[1], b = p[2], v;
                        if (ci == 0) v = 0.299f * r + 0.587f * g + 0.114f * b;
                        else if (ci == 1) v = -0.168736f * r - 0.331264f * g + 0.5f * b + 128.0f;
                        else v = 0.5f * r - 0.418688f * g - 0.081312f * b + 128.0f;
                        blk[y * 8 + x] = v - 128.0f;
                    }
                fdct8x8(blk, coef);
                int zz[64];
                const int t = ci ? 1 : 0;
                for (int i = 0; i < 64; i++) zz[i] = (int)std::lround(coef[kZigzag[i]] / (float)q[t][kZigzag[i]]);
                int diff = zz[0] - pred[ci];
                pred[ci] = zz[0];
                int s = bit_size(diff);
                bw.put(dc[t].code[s], dc[t].len[s]);
                if (s) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
                int run = 0;
                for (int k = 1; k < 64; k++) {
                    if (zz[k] == 0) { run++; continue; }
                    while (run > 15) { bw.put(ac[t].code[0xF0], ac[t].len[0xF0]); run -= 16; }
                    int sz = bit_size(zz[k]);
                    int sym = (run << 4) | sz;
                    bw.put(ac[t].code[sym], ac[t].len[sym]);
                    bw.put((uint32_t)(zz[k] < 0 ? zz[k] - 1 : zz[k]), sz);
                    run = 0;
                }
                if (run) bw.put(ac[t].code[0x00], ac[t].len[0x00]);
            }
    bw.flush();
    out.push_back(0xFF); out.push_back(0xD9);
    return true;
}

} // namespace rt2022
