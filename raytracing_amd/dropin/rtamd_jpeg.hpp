// rtamd_jpeg.hpp — JPEG decoder (sequential and progressive Huffman frames) for the drop-in's rtw_image (image_texture files such
// as earthmap.jpg, rtw_stb_image.h:50-64).
//
// The reference decodes through stb_image (external/stb_image.h), and the texels feed the
// renderer, so this decoder reproduces stb_image's integer arithmetic bit for bit:
//   * coefficients are dequantised into 16-bit integers (stb_image.h:1741);
//   * the inverse DCT is the 12-bit fixed-point "islow" transform with 2 extra bits kept
//     between the passes and the +128 level shift folded into the rounding of the second
//     pass (stb_image.h:1905-2003);
//   * chroma is upsampled with the 3:1 tent filters (stb_image.h:2848-2910) or replicated;
//   * YCbCr -> RGB in 20-bit fixed point with 12-bit coefficients, the Cb term of green
//     masked to 16 bits (stb_image.h:3071-3095).
//   * progressive frames keep their coefficients in 16-bit integers, refine them as
//     stb_image.h:1748-1892 does and dequantise in 16 bits after the last scan
//     (stb_image.h:2552-2575).
// Supported: 8-bit Huffman frames, sequential (SOF0 / SOF1) or progressive (SOF2: spectral
// selection and successive approximation), 1 or 3 components, sampling factors 1..4, any
// number of scans (interleaved or one component each), restart intervals.  Lossless,
// hierarchical and arithmetic-coded files are rejected.
// Written from the JPEG specification (ITU T.81) and the numbers above; no code is shared
// with stb_image.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rtamd {
namespace jpeg {

struct Image {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb; // width * height * 3, gamma-encoded as stored in the file
};

namespace detail {

// zig-zag position -> natural (row-major) position, T.81 figure A.6
static const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                     12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                     58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable { // T.81 annex C / F.2.2.3
    bool present = false;
    uint8_t values[256];
    int min_code[17], max_code[18], first_index[17];
    bool build(const uint8_t counts[16], const uint8_t *vals, int n_vals) {
        memcpy(values, vals, (size_t)n_vals);
        int code = 0, index = 0;
        for (int len = 1; len <= 16; len++) {
            first_index[len] = index;
            min_code[len] = code;
            index += counts[len - 1];
            code += counts[len - 1];
            max_code[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false; // over-subscribed
            code <<= 1;
        }
        max_code[17] = 0x7fffffff;
        present = true;
        return index == n_vals;
    }
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n_bits = 0;
    bool hit_marker = false;
    void reset() {
        acc = 0;
        n_bits = 0;
        hit_marker = false;
    }
    void fill() {
        while (n_bits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xff) {
                    int next = p + 1 < end ? p[1] : 0xd9;
                    if (next == 0x00) {
                        p += 2; // stuffed zero
                    } else {
                        hit_marker = true; // leave the marker in place; feed zeros from here on
                        byte = 0;
                    }
                } else {
                    p++;
                }
            }
            acc |= (uint32_t)byte << (24 - n_bits);
            n_bits += 8;
        }
    }
    int bit() {
        if (n_bits < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1;
        n_bits--;
        return b;
    }
    int bits(int n) { // n <= 16
        if (n == 0) return 0;
        if (n_bits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n;
        n_bits -= n;
        return v;
    }
};

inline int decode_symbol(BitReader &br, const HuffTable &h) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (h.max_code[len] >= 0 && code <= h.max_code[len] && code >= h.min_code[len])
            return h.values[h.first_index[len] + code - h.min_code[len]];
    }
    return -1;
}

// T.81 F.2.2.1 EXTEND: the value of an n-bit magnitude field
inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

inline uint8_t clamp_u8(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

// 12-bit fixed-point constants of the islow transform
inline int fx(double x) { return (int)(x * 4096 + 0.5); }

// One 8-point pass on in[0..7] (stride 1 here; callers gather).  Produces the even part e[0..3]
// and the odd part o[0..3], both scaled by 2^12; output k is e[k] + o[3-k] ... (see idct8x8).
// The arithmetic is done modulo 2^32: a valid file stays far inside 32 bits (same values as
// with int), and a damaged one, whose coefficients can be anything, wraps instead of running
// into signed overflow.
typedef uint32_t u32;
inline u32 fxu(double x) { return (u32)fx(x); }
inline void idct_1d(const u32 s[8], u32 e[4], u32 o[4]) {
    static const u32 c0541 = fxu(0.5411961f), c1847 = fxu(-1.847759065f), c0765 = fxu(0.765366865f);
    static const u32 c1175 = fxu(1.175875602f), c0298 = fxu(0.298631336f), c2053 = fxu(2.053119869f);
    static const u32 c3072 = fxu(3.072711026f), c1501 = fxu(1.501321110f), c0899 = fxu(-0.899976223f);
    static const u32 c2562 = fxu(-2.562915447f), c1961 = fxu(-1.961570560f), c0390 = fxu(-0.390180644f);
    // even part: rotation of (s2, s6), butterflies with (s0, s4)
    u32 z = (s[2] + s[6]) * c0541;
    u32 r2 = z + s[6] * c1847;
    u32 r3 = z + s[2] * c0765;
    u32 a = (s[0] + s[4]) * 4096u, b = (s[0] - s[4]) * 4096u;
    e[0] = a + r3;
    e[3] = a - r3;
    e[1] = b + r2;
    e[2] = b - r2;
    // odd part
    u32 q0 = s[7], q1 = s[5], q2 = s[3], q3 = s[1];
    u32 w3 = q0 + q2, w4 = q1 + q3, w1 = q0 + q3, w2 = q1 + q2;
    u32 w5 = (w3 + w4) * c1175;
    q0 *= c0298;
    q1 *= c2053;
    q2 *= c3072;
    q3 *= c1501;
    w1 = w5 + w1 * c0899;
    w2 = w5 + w2 * c2562;
    w3 *= c1961;
    w4 *= c0390;
    o[3] = q3 + w1 + w4;
    o[2] = q2 + w2 + w3;
    o[1] = q1 + w2 + w4;
    o[0] = q0 + w1 + w3;
}
// arithmetic shift of a value kept modulo 2^32
inline int sar(u32 v, int n) { return (int32_t)v >> n; }

// coef: 64 dequantised coefficients in natural order; out: 8 rows of 8 samples
inline void idct8x8(const short coef[64], uint8_t *out, int out_stride) {
    u32 mid[64];
    for (int c = 0; c < 8; c++) { // columns; 2 fractional bits are kept
        u32 s[8], e[4], o[4];
        for (int k = 0; k < 8; k++) s[k] = (u32)(int)coef[k * 8 + c];
        idct_1d(s, e, o);
        for (int k = 0; k < 4; k++) {
            u32 ev = e[k] + 512u;
            mid[k * 8 + c] = (u32)sar(ev + o[3 - k], 10);
            mid[(7 - k) * 8 + c] = (u32)sar(ev - o[3 - k], 10);
        }
    }
    for (int r = 0; r < 8; r++) { // rows; 2^17 to remove, level shift folded into the rounding
        u32 e[4], o[4];
        idct_1d(mid + r * 8, e, o);
        uint8_t *dst = out + r * out_stride;
        for (int k = 0; k < 4; k++) {
            u32 ev = e[k] + 65536u + (128u << 17);
            dst[k] = clamp_u8(sar(ev + o[3 - k], 17));
            dst[7 - k] = clamp_u8(sar(ev - o[3 - k], 17));
        }
    }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0;             // samples of this component that carry image data
    int plane_w = 0, plane_h = 0; // padded to whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<short> coef; // progressive frames: 64 per block, zig-zag positions already undone
};

// One scan's parameters (T.81 B.2.3): spectral band [ss, se], successive-approximation bit
// positions ah (0: first pass over the band) and al.
struct Scan {
    int n = 0, order[4] = {0, 0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;
};

// Progressive DC (T.81 G.1.2.1): first pass = the sequential DC code, the value scaled by
// 2^al; refinement = one more bit.  16-bit arithmetic as in stb_image.h:1756-1769.
inline bool prog_dc(BitReader &br, const HuffTable &hd, Component &c, short *blk, const Scan &sc) {
    if (sc.ah == 0) {
        memset(blk, 0, 64 * sizeof(short));
        int t = decode_symbol(br, hd);
        if (t < 0 || t > 15) return false;
        int diff = t ? extend(br.bits(t), t) : 0;
        c.dc_pred = (int)((u32)c.dc_pred + (u32)diff); // (a damaged file may wrap)
        blk[0] = (short)((u32)c.dc_pred * (1u << sc.al));
    } else if (br.bit()) {
        blk[0] = (short)(blk[0] + (short)(1 << sc.al));
    }
    return true;
}

// One more bit for a coefficient that is already non-zero (T.81 G.1.2.3; a coefficient whose
// bit is already set is left alone, stb_image.h:1834 and 1874).
inline void refine_nonzero(BitReader &br, short *p, short bit) {
    if (br.bit() && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
}

// Progressive AC (T.81 G.1.2.2, G.1.2.3) for one block; eob_run = blocks of this band still
// covered by an end-of-band run.
inline bool prog_ac(BitReader &br, const HuffTable &ha, short *blk, const Scan &sc, int &eob_run) {
    if (sc.ah == 0) {
        if (eob_run) {
            eob_run--;
            return true;
        }
        for (int k = sc.ss; k <= sc.se;) {
            int rs = decode_symbol(br, ha);
            if (rs < 0) return false;
            int run = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (run < 15) { // EOBn: this block and 2^run + bits - 1 more
                    eob_run = (1 << run) - 1;
                    if (run) eob_run += br.bits(run);
                    break;
                }
                k += 16;
                continue;
            }
            k += run;
            if (k > 63) return false;
            blk[kNatural[k]] = (short)(extend(br.bits(sz), sz) * (1 << sc.al));
            k++;
        }
        return true;
    }
    const short bit = (short)(1 << sc.al);
    if (eob_run) {
        eob_run--;
        for (int k = sc.ss; k <= sc.se; k++) {
            short *p = &blk[kNatural[k]];
            if (*p != 0) refine_nonzero(br, p, bit);
        }
        return true;
    }
    int k = sc.ss;
    do {
        int rs = decode_symbol(br, ha);
        if (rs < 0) return false;
        int run = rs >> 4, sz = rs & 15;
        int value = 0;
        if (sz == 0) {
            if (run < 15) {
                eob_run = (1 << run) - 1;
                if (run) eob_run += br.bits(run);
                run = 64; // the rest of the band: only the non-zero coefficients get their bit
            }           // (run 15: sixteen zeros, the last of them "written" as 0)
        } else {
            if (sz != 1) return false;
            value = br.bit() ? bit : -bit;
        }
        while (k <= sc.se) {
            short *p = &blk[kNatural[k++]];
            if (*p != 0) {
                refine_nonzero(br, p, bit);
            } else {
                if (run == 0) {
                    *p = (short)value;
                    break;
                }
                run--;
            }
        }
    } while (k <= sc.se);
    return true;
}

// One output row of a component at full width (stb_image.h:3327-3360 row selection, filters
// 2848-2910 and 3029-3037).
struct Upsampler {
    int hs, vs, ystep, ypos, w_lores, rows;
    const uint8_t *line0, *line1;
    int stride;
    std::vector<uint8_t> buf;
    const uint8_t *next_row() {
        const bool y_bot = ystep >= (vs >> 1);
        const uint8_t *nearr = y_bot ? line1 : line0, *farr = y_bot ? line0 : line1;
        const uint8_t *res;
        const int w = w_lores;
        uint8_t *out = buf.data();
        if (hs == 1 && vs == 1) {
            res = nearr;
        } else if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * nearr[i] + farr[i] + 2) >> 2);
            res = out;
        } else if (hs == 2 && vs == 1) {
            if (w == 1) {
                out[0] = out[1] = nearr[0];
            } else {
                out[0] = nearr[0];
                out[1] = (uint8_t)((nearr[0] * 3 + nearr[1] + 2) >> 2);
                for (int i = 1; i < w - 1; i++) {
                    int n = 3 * nearr[i] + 2;
                    out[2 * i] = (uint8_t)((n + nearr[i - 1]) >> 2);
                    out[2 * i + 1] = (uint8_t)((n + nearr[i + 1]) >> 2);
                }
                out[2 * (w - 1)] = (uint8_t)((nearr[w - 2] * 3 + nearr[w - 1] + 2) >> 2);
                out[2 * (w - 1) + 1] = nearr[w - 1];
            }
            res = out;
        } else if (hs == 2 && vs == 2) {
            if (w == 1) {
                out[0] = out[1] = (uint8_t)((3 * nearr[0] + farr[0] + 2) >> 2);
            } else {
                int t1 = 3 * nearr[0] + farr[0];
                out[0] = (uint8_t)((t1 + 2) >> 2);
                for (int i = 1; i < w; i++) {
                    int t0 = t1;
                    t1 = 3 * nearr[i] + farr[i];
                    out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                    out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
                }
                out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
            }
            res = out;
        } else { // any other factor: nearest neighbour
            for (int i = 0; i < w; i++)
                for (int j = 0; j < hs; j++) out[i * hs + j] = nearr[i];
            res = out;
        }
        if (++ystep >= vs) {
            ystep = 0;
            line0 = line1;
            if (++ypos < rows) line1 += stride;
        }
        return res;
    }
};

inline int be16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

// Entropy-coded data of one scan, from `pos`; returns the position at which marker parsing goes on.
// Sequential frames: every block is dequantised and transformed as it is decoded.  Progressive
// frames: the scan adds its band / bit to the stored coefficients (finish_progressive does the
// rest once all scans are in).  A scan of one component covers that component's own
// ceil(x / 8) x ceil(y / 8) blocks; a scan of several covers whole MCUs (T.81 A.2).
inline bool decode_scan(const uint8_t *data, size_t size, size_t &pos, std::vector<Component> &comp, const Scan &sc,
                        bool progressive, const HuffTable *dc_tab, const HuffTable *ac_tab, const uint8_t quant[4][64],
                        int mcus_x, int mcus_y, int restart_interval, const char **why) {
    BitReader br;
    br.p = data + pos;
    br.end = data + size;
    int until_restart = restart_interval ? restart_interval : 0x7fffffff;
    int eob_run = 0;
    for (int k = 0; k < sc.n; k++) comp[(size_t)sc.order[k]].dc_pred = 0;
    short block[64];
    // one block of component c at block coordinates (bx, by)
    auto do_block = [&](Component &c, int bx, int by) -> bool {
        const HuffTable &hd = dc_tab[c.td], &ha = ac_tab[c.ta];
        if (progressive) {
            short *blk = c.coef.data() + 64 * ((size_t)by * (size_t)(c.plane_w / 8) + (size_t)bx);
            if (sc.ss == 0) {
                if (!prog_dc(br, hd, c, blk, sc)) return (*why = "bad DC code", false);
            } else if (!prog_ac(br, ha, blk, sc, eob_run)) {
                return (*why = "bad AC code", false);
            }
            return true;
        }
        const uint8_t *q = quant[c.tq];
        memset(block, 0, sizeof(block));
        int t = decode_symbol(br, hd);
        if (t < 0 || t > 15) return (*why = "bad DC code", false);
        int diff = t ? extend(br.bits(t), t) : 0;
        c.dc_pred = (int)((u32)c.dc_pred + (u32)diff); // (a damaged file may wrap)
        block[0] = (short)((u32)c.dc_pred * (u32)q[0]);
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(br, ha);
            if (rs < 0) return (*why = "bad AC code", false);
            int run = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (run != 15) break; // end of block
                k += 16;
                continue;
            }
            k += run;
            if (k > 63) return (*why = "AC run past the end of the block", false);
            int nat = kNatural[k];
            block[nat] = (short)(extend(br.bits(sz), sz) * q[nat]);
            k++;
        }
        idct8x8(block, c.plane.data() + (size_t)by * 8 * c.plane_w + (size_t)bx * 8, c.plane_w);
        return true;
    };
    // RSTn after every restart_interval MCUs: byte-align, skip the marker, reset the predictors
    auto mcu_done = [&](bool last) -> bool {
        if (--until_restart != 0) return true;
        until_restart = restart_interval;
        if (last) return true;
        const uint8_t *p = br.p;
        while (p + 1 < br.end && !(p[0] == 0xff && p[1] >= 0xd0 && p[1] <= 0xd7)) p++;
        if (p + 1 >= br.end) return (*why = "missing restart marker", false);
        br.p = p + 2;
        br.reset();
        eob_run = 0;
        for (int k = 0; k < sc.n; k++) comp[(size_t)sc.order[k]].dc_pred = 0;
        return true;
    };
    if (sc.n == 1) {
        Component &c = comp[(size_t)sc.order[0]];
        const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                if (!do_block(c, bx, by)) return false;
                if (!mcu_done(by == bh - 1 && bx == bw - 1)) return false;
            }
    } else {
        for (int my = 0; my < mcus_y; my++)
            for (int mx = 0; mx < mcus_x; mx++) {
                for (int k = 0; k < sc.n; k++) {
                    Component &c = comp[(size_t)sc.order[k]];
                    for (int by = 0; by < c.v; by++)
                        for (int bx = 0; bx < c.h; bx++)
                            if (!do_block(c, mx * c.h + bx, my * c.v + by)) return false;
                }
                if (!mcu_done(my == mcus_y - 1 && mx == mcus_x - 1)) return false;
            }
    }
    pos = (size_t)(br.p - data); // at or before the next marker (the reader never steps over one)
    return true;
}

// Progressive frames, after the last scan: dequantise in 16 bits and transform the blocks that
// carry image data (stb_image.h:2552-2575).
inline void finish_progressive(std::vector<Component> &comp, const uint8_t quant[4][64]) {
    for (Component &c : comp) {
        const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                short *blk = c.coef.data() + 64 * ((size_t)by * (size_t)(c.plane_w / 8) + (size_t)bx);
                for (int i = 0; i < 64; i++) blk[i] = (short)(blk[i] * quant[c.tq][i]);
                idct8x8(blk, c.plane.data() + (size_t)by * 8 * c.plane_w + (size_t)bx * 8, c.plane_w);
            }
    }
}

} // namespace detail

inline bool decode(const uint8_t *data, size_t size, Image &img, std::string *err = nullptr) {
    using namespace detail;
    auto fail = [&](const char *why) {
        if (err) *err = why;
        return false;
    };
    if (size < 4 || data[0] != 0xff || data[1] != 0xd8) return fail("not a JPEG file");
    uint8_t quant[4][64];
    bool quant_ok[4] = {false, false, false, false};
    HuffTable dc_tab[4], ac_tab[4];
    std::vector<Component> comp;
    int width = 0, height = 0, restart_interval = 0, h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0;
    size_t pos = 2;
    bool have_frame = false, progressive = false;
    int n_scans = 0;
    for (;;) {
        while (pos < size && data[pos] != 0xff) pos++; // tolerate garbage between segments
        while (pos < size && data[pos] == 0xff) pos++; // fill bytes
        if (pos >= size) {
            if (n_scans > 0) break; // no EOI marker: what has been decoded stands
            return fail("truncated file (no scan)");
        }
        const int marker = data[pos++];
        if (marker == 0xd9) {
            if (n_scans > 0) break;
            return fail("end of image before any scan");
        }
        if (marker == 0x00 || marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7)) continue; // no payload
        if (pos + 2 > size) return fail("truncated segment");
        const int len = be16(data + pos);
        if (len < 2 || pos + (size_t)len > size) return fail("bad segment length");
        const uint8_t *seg = data + pos + 2;
        const int n = len - 2;
        switch (marker) {
        case 0xdb: { // DQT
            int k = 0;
            while (k < n) {
                int pq = seg[k] >> 4, tq = seg[k] & 15;
                if (pq != 0) return fail("16-bit quantisation tables are not supported");
                if (tq > 3 || k + 65 > n) return fail("bad DQT");
                for (int i = 0; i < 64; i++) quant[tq][kNatural[i]] = seg[k + 1 + i];
                quant_ok[tq] = true;
                k += 65;
            }
            break;
        }
        case 0xc4: { // DHT
            int k = 0;
            while (k < n) {
                if (k + 17 > n) return fail("bad DHT");
                int tc = seg[k] >> 4, th = seg[k] & 15;
                if (tc > 1 || th > 3) return fail("bad DHT header");
                int total = 0;
                for (int i = 0; i < 16; i++) total += seg[k + 1 + i];
                if (total > 256 || k + 17 + total > n) return fail("bad DHT size");
                HuffTable &t = tc == 0 ? dc_tab[th] : ac_tab[th];
                if (!t.build(seg + k + 1, seg + k + 17, total)) return fail("bad Huffman code lengths");
                k += 17 + total;
            }
            break;
        }
        case 0xc0:
        case 0xc1:
        case 0xc2: { // SOF0 / SOF1: Huffman, sequential; SOF2: Huffman, progressive
            if (have_frame) return fail("more than one frame");
            if (n < 6) return fail("bad SOF");
            if (seg[0] != 8) return fail("only 8-bit samples are supported");
            height = be16(seg + 1);
            width = be16(seg + 3);
            int nc = seg[5];
            if (width <= 0 || height <= 0) return fail("bad image size");
            if ((nc != 1 && nc != 3) || n < 6 + 3 * nc) return fail("only 1 or 3 components are supported");
            comp.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                comp[i].id = seg[6 + 3 * i];
                comp[i].h = seg[7 + 3 * i] >> 4;
                comp[i].v = seg[7 + 3 * i] & 15;
                comp[i].tq = seg[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3)
                    return fail("bad component");
                if (comp[i].h > h_max) h_max = comp[i].h;
                if (comp[i].v > v_max) v_max = comp[i].v;
            }
            progressive = marker == 0xc2;
            if (nc == 1) comp[0].h = comp[0].v = h_max = v_max = 1; // a lone component is not interleaved
            mcus_x = (width + 8 * h_max - 1) / (8 * h_max);
            mcus_y = (height + 8 * v_max - 1) / (8 * v_max);
            // A few header bytes can claim 65535 x 65535 samples (tens of GB of planes).  Every coded
            // block takes at least one bit, so a large image in a file that cannot hold a bit per
            // block is refused before anything is allocated; small images are decoded whatever the
            // file's length (a cut file gives a partly grey picture, as with stb_image).
            if ((uint64_t)width * (uint64_t)height > (64ull << 20) &&
                (uint64_t)mcus_x * (uint64_t)mcus_y > 8ull * (uint64_t)size)
                return fail("the image size in the header is implausible for the file's length");
            for (Component &c : comp) {
                c.x = (width * c.h + h_max - 1) / h_max;
                c.y = (height * c.v + v_max - 1) / v_max;
                c.plane_w = mcus_x * c.h * 8;
                c.plane_h = mcus_y * c.v * 8;
                c.plane.assign((size_t)c.plane_w * c.plane_h, 0);
                if (progressive) c.coef.assign((size_t)c.plane_w * c.plane_h, 0);
            }
            have_frame = true;
            break;
        }
        case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd:
        case 0xce: case 0xcf:
            return fail("only Huffman-coded sequential and progressive JPEG is supported (this file is lossless, "
                        "hierarchical or arithmetic-coded)");
        case 0xdd: // DRI
            if (n < 2) return fail("bad DRI");
            restart_interval = be16(seg);
            break;
        case 0xda: { // SOS
            if (!have_frame) return fail("scan before frame");
            if (n < 1) return fail("bad SOS");
            Scan sc;
            sc.n = seg[0];
            if (sc.n < 1 || sc.n > (int)comp.size() || n != 1 + 2 * sc.n + 3) return fail("bad SOS");
            for (int i = 0; i < sc.n; i++) {
                int cid = seg[1 + 2 * i], tt = seg[2 + 2 * i], which = -1;
                for (size_t k = 0; k < comp.size(); k++)
                    if (comp[k].id == cid) {
                        which = (int)k;
                        break;
                    }
                if (which < 0) return fail("scan names a component the frame does not have");
                for (int k = 0; k < i; k++)
                    if (sc.order[k] == which) return fail("scan names a component twice");
                sc.order[i] = which;
                comp[(size_t)which].td = tt >> 4;
                comp[(size_t)which].ta = tt & 15;
                if (comp[(size_t)which].td > 3 || comp[(size_t)which].ta > 3) return fail("bad Huffman table index");
                if (!quant_ok[comp[(size_t)which].tq]) return fail("component refers to a missing quantisation table");
            }
            sc.ss = seg[1 + 2 * sc.n];
            sc.se = seg[2 + 2 * sc.n];
            sc.ah = seg[3 + 2 * sc.n] >> 4;
            sc.al = seg[3 + 2 * sc.n] & 15;
            if (progressive) {
                if (sc.ss > 63 || sc.se > 63 || sc.ss > sc.se || sc.ah > 13 || sc.al > 13) return fail("bad SOS");
                if (sc.ss == 0 && sc.se != 0) return fail("a progressive scan mixes DC and AC coefficients");
                if (sc.ss != 0 && sc.n != 1) return fail("a progressive AC scan covers one component");
            } else {
                if (sc.ss != 0 || sc.ah != 0 || sc.al != 0) return fail("bad SOS");
                sc.se = 63;
            }
            for (int i = 0; i < sc.n; i++) { // the tables this scan decodes with
                const Component &c = comp[(size_t)sc.order[i]];
                const bool need_dc = !progressive || (sc.ss == 0 && sc.ah == 0), need_ac = !progressive || sc.ss != 0;
                if ((need_dc && !dc_tab[c.td].present) || (need_ac && !ac_tab[c.ta].present))
                    return fail("scan refers to a missing Huffman table");
            }
            pos += (size_t)len;
            const char *why = "corrupt scan";
            if (!decode_scan(data, size, pos, comp, sc, progressive, dc_tab, ac_tab, quant, mcus_x, mcus_y,
                             restart_interval, &why))
                return fail(why);
            n_scans++;
            continue;
        }
        default: // APPn, COM and anything else with a length
            break;
        }
        pos += (size_t)len;
    }
    if (progressive) finish_progressive(comp, quant);

    // ---- upsample and convert ----------------------------------------------------------------------
    img.width = width;
    img.height = height;
    img.rgb.assign((size_t)width * height * 3, 0);
    std::vector<Upsampler> up(comp.size());
    for (size_t k = 0; k < comp.size(); k++) {
        Upsampler &u = up[k];
        u.hs = h_max / comp[k].h;
        u.vs = v_max / comp[k].v;
        u.ystep = u.vs >> 1;
        u.ypos = 0;
        u.w_lores = (width + u.hs - 1) / u.hs;
        u.rows = (height * comp[k].v + v_max - 1) / v_max; // rows of this component that carry image data
        u.stride = comp[k].plane_w;
        u.line0 = u.line1 = comp[k].plane.data();
        u.buf.assign((size_t)width + 8 + (size_t)u.hs * 2, 0);
    }
    // 12-bit coefficients shifted to 20 bits (stb_image.h:3071)
    auto c20 = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
    const int k_r_cr = c20(1.40200f), k_g_cr = c20(0.71414f), k_g_cb = c20(0.34414f), k_b_cb = c20(1.77200f);
    for (int y = 0; y < height; y++) {
        const uint8_t *row[3] = {nullptr, nullptr, nullptr};
        for (size_t k = 0; k < comp.size(); k++) row[k] = up[k].next_row();
        uint8_t *out = img.rgb.data() + (size_t)y * width * 3;
        if (comp.size() == 3) {
            for (int x = 0; x < width; x++) {
                int yf = (row[0][x] << 20) + (1 << 19);
                int cr = row[2][x] - 128, cb = row[1][x] - 128;
                int r = yf + cr * k_r_cr;
                int g = yf + (cr * -k_g_cr) + (int)((unsigned)(cb * -k_g_cb) & 0xffff0000u);
                int b = yf + cb * k_b_cb;
                out[3 * x + 0] = clamp_u8(r >> 20);
                out[3 * x + 1] = clamp_u8(g >> 20);
                out[3 * x + 2] = clamp_u8(b >> 20);
            }
        } else {
            for (int x = 0; x < width; x++) out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = row[0][x];
        }
    }
    return true;
}

} // namespace jpeg
} // namespace rtamd
