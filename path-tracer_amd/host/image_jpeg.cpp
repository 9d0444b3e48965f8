// JPEG decoder (ITU-T T.81): baseline, extended sequential and progressive DCT (spectral selection and successive
// approximation, T.81 annex G), Huffman coding, 8-bit samples, 1 (grey) or 3 (YCbCr, or RGB with an Adobe
// transform-0 marker) components, any sampling factors up to 4, interleaved and per-component scans, restart
// intervals, written against the standard.  Arithmetic-coded, lossless, hierarchical and 12-bit files are rejected.
// Every scan adds to per-component coefficient arrays; dequantisation and the inverse DCT run once, after the last scan.
#include "image.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

const uint8_t kZigzag[64] = { 0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct huff_table {
    bool present = false;
    uint8_t bits[17] = { 0 };
    uint8_t values[256] = { 0 };
    int32_t min_code[17], max_code[18], val_ptr[17];
    void prepare() {
        int32_t code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            val_ptr[len] = k;
            min_code[len] = code;
            code += bits[len];
            k += bits[len];
            max_code[len] = bits[len] ? code - 1 : -1;
            if (code > (1 << len)) throw image_error("jpeg: over-subscribed Huffman table");
            code <<= 1;
        }
        max_code[17] = 0x7fffffff;
    }
};

struct component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;   // padded to whole MCUs
    int own_blocks_w = 0, own_blocks_h = 0;  // blocks that hold picture samples: what a scan of this component alone codes
    int pred = 0;
    std::vector<int16_t> coef;        // blocks_w * blocks_h blocks of 64 coefficients in natural order, not dequantised
    std::vector<uint8_t> plane;       // blocks_w*8 x blocks_h*8 samples
};

struct entropy_reader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int count = 0;
    bool hit_marker = false;
    int next_bit() {
        if (count == 0) {
            uint8_t byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    while (p < end && *p == 0xFF) p++;      // fill bytes
                    if (p < end && *p == 0x00) p++;          // stuffed zero: a literal 0xFF
                    else { hit_marker = true; p--; byte = 0; }  // a marker: feed zeros until the caller resynchronises
                }
            }
            acc = byte; count = 8;
        }
        count--;
        return (int)((acc >> count) & 1u);
    }
    int receive(int n) { int v = 0; while (n--) v = (v << 1) | next_bit(); return v; }
    void reset() { acc = 0; count = 0; hit_marker = false; }
};

int decode_symbol(entropy_reader& er, const huff_table& t) {
    int32_t code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | er.next_bit();
        if (t.max_code[len] >= 0 && code <= t.max_code[len] && code >= t.min_code[len]) return t.values[t.val_ptr[len] + (code - t.min_code[len])];
    }
    throw image_error("jpeg: invalid Huffman code");
}

int extend(int v, int n) { return n == 0 ? 0 : (v < (1 << (n - 1)) ? v - (1 << n) + 1 : v); }

// Inverse DCT of one dequantised block (T.81 A.3.3), separable, in double precision; output level-shifted by 128.
void idct_block(const int32_t* coef, uint8_t* dst, size_t stride) {
    static double basis[8][8];
    static bool ready = false;
    if (!ready) {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) basis[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0) * 0.5;
        ready = true;
    }
    double tmp[64];
    for (int v = 0; v < 8; v++)       // rows: over u
        for (int x = 0; x < 8; x++) {
            double s = 0.0;
            for (int u = 0; u < 8; u++) if (coef[v * 8 + u]) s += basis[x][u] * coef[v * 8 + u];
            tmp[v * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)       // columns: over v
        for (int y = 0; y < 8; y++) {
            double s = 0.0;
            for (int v = 0; v < 8; v++) s += basis[y][v] * tmp[v * 8 + x];
            const long r = std::lround(s) + 128;
            dst[y * stride + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
}

uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// Sample of plane `c` (at its own resolution cw x ch) at full-resolution pixel (x, y): replication for general
// factors, the (3/4, 1/4) triangle filter when a direction is subsampled by exactly 2.
struct upsampler {
    const component& c;
    int hmax, vmax, cw, ch;
    double at(int cx, int cy) const {
        cx = cx < 0 ? 0 : cx >= cw ? cw - 1 : cx;
        cy = cy < 0 ? 0 : cy >= ch ? ch - 1 : cy;
        return c.plane[(size_t)cy * c.blocks_w * 8 + cx];
    }
    double sample(int x, int y) const {
        const int fx = hmax / c.h, fy = vmax / c.v;
        int x0 = x / fx, x1 = x0, y0 = y / fy, y1 = y0;
        double wx = 0.0, wy = 0.0;
        if (fx == 2) { x1 = (x & 1) ? x0 + 1 : x0 - 1; wx = 0.25; }
        if (fy == 2) { y1 = (y & 1) ? y0 + 1 : y0 - 1; wy = 0.25; }
        const double top = at(x0, y0) * (1.0 - wx) + at(x1, y0) * wx;
        if (wy == 0.0) return top;
        const double bottom = at(x0, y1) * (1.0 - wx) + at(x1, y1) * wx;
        return top * (1.0 - wy) + bottom * wy;
    }
};

uint8_t clamp8(double v) { const long r = std::lround(v); return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); }

}  // namespace

raw_image decode_jpeg(const uint8_t* bytes, size_t size) {
    if (size < 4 || bytes[0] != 0xFF || bytes[1] != 0xD8) throw image_error("jpeg: bad signature");
    uint16_t quant[4][64];
    bool quant_present[4] = { false, false, false, false };
    huff_table dc[4], ac[4];
    std::vector<component> comps;
    int width = 0, height = 0, restart_interval = 0;
    int adobe_transform = -1;
    bool have_frame = false, decoded = false, progressive = false, scan_seen = false;
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;

    size_t pos = 2;
    while (!decoded) {
        // next marker
        if (pos + 2 > size) { if (scan_seen) break; throw image_error("jpeg: truncated before the scan"); }
        if (bytes[pos] != 0xFF) throw image_error("jpeg: marker expected");
        while (pos < size && bytes[pos] == 0xFF) pos++;
        if (pos >= size) throw image_error("jpeg: truncated marker");
        const uint8_t marker = bytes[pos++];
        if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01) continue;  // no payload
        if (marker == 0xD9) { if (scan_seen) break; throw image_error("jpeg: end of image before any scan"); }
        if (pos + 2 > size) throw image_error("jpeg: truncated segment");
        const size_t seg_len = be16(bytes + pos);
        if (seg_len < 2 || pos + seg_len > size) throw image_error("jpeg: bad segment length");
        const uint8_t* seg = bytes + pos + 2;
        const size_t n = seg_len - 2;
        pos += seg_len;

        if (marker == 0xDB) {  // DQT
            size_t i = 0;
            while (i < n) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                i++;
                if (tq > 3 || pq > 1) throw image_error("jpeg: bad quantisation table");
                if (i + (pq ? 128 : 64) > n) throw image_error("jpeg: truncated quantisation table");
                for (int k = 0; k < 64; k++) { quant[tq][kZigzag[k]] = pq ? be16(seg + i) : seg[i]; i += pq ? 2 : 1; }
                quant_present[tq] = true;
            }
        } else if (marker == 0xC4) {  // DHT
            size_t i = 0;
            while (i < n) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                i++;
                if (tc > 1 || th > 3 || i + 16 > n) throw image_error("jpeg: bad Huffman table");
                huff_table& t = tc ? ac[th] : dc[th];
                int total = 0;
                for (int len = 1; len <= 16; len++) { t.bits[len] = seg[i++]; total += t.bits[len]; }
                if (total > 256 || i + total > n) throw image_error("jpeg: truncated Huffman table");
                std::memcpy(t.values, seg + i, (size_t)total);
                i += total;
                t.present = true;
                t.prepare();
            }
        } else if (marker == 0xC0 || marker == 0xC1 || marker == 0xC2) {  // SOF0 / SOF1 / SOF2
            if (have_frame) throw image_error("jpeg: more than one frame");
            if (n < 6) throw image_error("jpeg: truncated frame header");
            if (seg[0] != 8) throw image_error("jpeg: only 8-bit samples are supported");
            progressive = marker == 0xC2;
            height = be16(seg + 1); width = be16(seg + 3);
            const int nc = seg[5];
            if (width == 0 || height == 0) throw image_error("jpeg: bad dimensions");
            if ((uint64_t)width * height > (1ull << 28)) throw image_error("jpeg: more than 2^28 pixels");
            if ((nc != 1 && nc != 3) || n < (size_t)(6 + 3 * nc)) throw image_error("jpeg: only 1 or 3 components are supported");
            comps.resize(nc);
            for (int c = 0; c < nc; c++) {
                comps[c].id = seg[6 + 3 * c];
                comps[c].h = seg[7 + 3 * c] >> 4; comps[c].v = seg[7 + 3 * c] & 15;
                comps[c].tq = seg[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) throw image_error("jpeg: bad component");
                if (comps[c].h > hmax) hmax = comps[c].h;
                if (comps[c].v > vmax) vmax = comps[c].v;
            }
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }  // a single component is never interleaved (T.81 A.2.2)
            for (auto& c : comps) if (hmax % c.h || vmax % c.v) throw image_error("jpeg: fractional sampling ratios are not supported");
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            mcus_x = (width + mcu_w - 1) / mcu_w; mcus_y = (height + mcu_h - 1) / mcu_h;
            for (auto& c : comps) {
                c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
                c.own_blocks_w = ((width * c.h + hmax - 1) / hmax + 7) / 8;
                c.own_blocks_h = ((height * c.v + vmax - 1) / vmax + 7) / 8;
                c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
            }
            have_frame = true;
        } else if (marker == 0xC3 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
            throw image_error("jpeg: lossless, hierarchical and arithmetic-coded files are not supported");
        } else if (marker == 0xDD) {  // DRI
            if (n < 2) throw image_error("jpeg: bad restart interval");
            restart_interval = be16(seg);
        } else if (marker == 0xEE) {  // APP14 "Adobe"
            if (n >= 12 && std::memcmp(seg, "Adobe", 5) == 0) adobe_transform = seg[11];
        } else if (marker == 0xDA) {  // SOS
            if (!have_frame) throw image_error("jpeg: scan before the frame header");
            if (n < 1) throw image_error("jpeg: truncated scan header");
            const size_t ns = seg[0];
            if (ns < 1 || ns > comps.size() || n < 1 + 2 * ns + 3) throw image_error("jpeg: bad scan header");
            std::vector<component*> in_scan;
            for (size_t k = 0; k < ns; k++) {
                const int cid = seg[1 + 2 * k];
                component* c = nullptr;
                for (auto& cc : comps) if (cc.id == cid) c = &cc;
                if (!c) throw image_error("jpeg: scan names an unknown component");
                for (component* seen : in_scan) if (seen == c) throw image_error("jpeg: scan names a component twice");
                c->td = seg[2 + 2 * k] >> 4; c->ta = seg[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3) throw image_error("jpeg: scan names a bad Huffman table");
                in_scan.push_back(c);
            }
            const int ss = seg[1 + 2 * ns], se = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
            if (!progressive) {
                if (ss != 0 || se != 63 || ah != 0 || al != 0) throw image_error("jpeg: a sequential scan codes all 64 coefficients at full precision");
            } else {
                if (ss > se || se > 63 || al > 13 || ah > 13 || (ah != 0 && ah != al + 1)) throw image_error("jpeg: bad progressive scan parameters");
                if (ss == 0 && se != 0) throw image_error("jpeg: a progressive scan codes either the DC or some AC coefficients");
                if (ss > 0 && ns != 1) throw image_error("jpeg: a progressive AC scan holds one component");
            }
            const bool needs_dc = ss == 0 && ah == 0, needs_ac = se > 0;
            for (component* c : in_scan) {
                if (needs_dc && !dc[c->td].present) throw image_error("jpeg: scan names a missing Huffman table");
                if (needs_ac && !ac[c->ta].present) throw image_error("jpeg: scan names a missing Huffman table");
                c->pred = 0;
            }
            scan_seen = true;

            // A scan of one component walks that component's own blocks; a scan of several walks the frame's MCUs.
            const bool interleaved = ns > 1;
            const int units_x = interleaved ? mcus_x : in_scan[0]->own_blocks_w, units_y = interleaved ? mcus_y : in_scan[0]->own_blocks_h;
            entropy_reader er{ bytes + pos, bytes + size };
            int until_restart = restart_interval;
            int expected_rst = 0;
            int eobrun = 0;
            const int p1 = 1 << al, m1 = -(1 << al);
            auto refine_nonzero = [&](int16_t& c) {  // one correction bit for a coefficient that is already non-zero
                if (er.next_bit() && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
            };
            auto decode_block = [&](component& c, int16_t* block) {
                if (!progressive) {
                    const int t = decode_symbol(er, dc[c.td]);
                    if (t > 11) throw image_error("jpeg: bad DC category");
                    c.pred += extend(er.receive(t), t);
                    block[0] = (int16_t)c.pred;
                    for (int k = 1; k < 64;) {
                        const int rs = decode_symbol(er, ac[c.ta]);
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) { k += 16; continue; }
                            break;  // end of block
                        }
                        k += r;
                        if (k > 63) throw image_error("jpeg: AC coefficient index out of range");
                        block[kZigzag[k]] = (int16_t)extend(er.receive(sz), sz);
                        k++;
                    }
                    return;
                }
                if (ss == 0) {
                    if (ah == 0) {  // first DC scan: the difference, scaled by the point transform
                        const int t = decode_symbol(er, dc[c.td]);
                        if (t > 11) throw image_error("jpeg: bad DC category");
                        c.pred += extend(er.receive(t), t);
                        block[0] = (int16_t)(c.pred * (1 << al));
                    } else if (er.next_bit()) {  // DC refinement: one more bit
                        block[0] = (int16_t)(block[0] | p1);
                    }
                    return;
                }
                if (ah == 0) {  // first AC scan of the band [ss, se]
                    if (eobrun > 0) { eobrun--; return; }
                    for (int k = ss; k <= se;) {
                        const int rs = decode_symbol(er, ac[c.ta]);
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) { k += 16; continue; }
                            eobrun = (1 << r) - 1;                       // this block and eobrun more end here
                            if (r) eobrun += er.receive(r);
                            break;
                        }
                        k += r;
                        if (k > se) throw image_error("jpeg: AC coefficient index out of range");
                        block[kZigzag[k]] = (int16_t)(extend(er.receive(sz), sz) * (1 << al));
                        k++;
                    }
                    return;
                }
                // AC refinement (T.81 G.1.2.3): correction bits for the non-zero history, new +-1 coefficients in between
                int k = ss;
                if (eobrun == 0) {
                    for (; k <= se; k++) {
                        const int rs = decode_symbol(er, ac[c.ta]);
                        int r = rs >> 4;
                        const int sz = rs & 15;
                        int value = 0;
                        if (sz == 0) {
                            if (r < 15) {
                                eobrun = 1 << r;                        // counts this block too
                                if (r) eobrun += er.receive(r);
                                break;
                            }
                        } else {
                            if (sz != 1) throw image_error("jpeg: bad coefficient size in a refinement scan");
                            value = er.next_bit() ? p1 : m1;
                        }
                        while (k <= se) {
                            int16_t& cf = block[kZigzag[k]];
                            if (cf != 0) refine_nonzero(cf);
                            else if (--r < 0) break;
                            k++;
                        }
                        if (value) {
                            if (k > se) throw image_error("jpeg: AC coefficient index out of range");
                            block[kZigzag[k]] = (int16_t)value;
                        }
                    }
                }
                if (eobrun > 0) {
                    for (; k <= se; k++) {
                        int16_t& cf = block[kZigzag[k]];
                        if (cf != 0) refine_nonzero(cf);
                    }
                    eobrun--;
                }
            };
            for (int uy = 0; uy < units_y; uy++)
                for (int ux = 0; ux < units_x; ux++) {
                    if (restart_interval && until_restart == 0) {
                        // byte-align, expect RSTn, reset predictors and the end-of-band run
                        er.count = 0;
                        const uint8_t* q = er.p;
                        while (q + 1 < er.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                        if (q + 1 >= er.end) throw image_error("jpeg: restart marker missing");
                        if ((q[1] - 0xD0) != expected_rst) throw image_error("jpeg: restart markers out of sequence");
                        expected_rst = (expected_rst + 1) & 7;
                        er.p = q + 2;
                        er.reset();
                        for (component* c : in_scan) c->pred = 0;
                        eobrun = 0;
                        until_restart = restart_interval;
                    }
                    if (interleaved) {
                        for (component* c : in_scan)
                            for (int by = 0; by < c->v; by++)
                                for (int bx = 0; bx < c->h; bx++)
                                    decode_block(*c, c->coef.data() + ((size_t)(uy * c->v + by) * c->blocks_w + (size_t)(ux * c->h + bx)) * 64);
                    } else {
                        component* c = in_scan[0];
                        decode_block(*c, c->coef.data() + ((size_t)uy * c->blocks_w + (size_t)ux) * 64);
                    }
                    if (restart_interval) until_restart--;
                }
            // on to the marker that ends the scan's data: 0xFF followed by neither a stuffed zero, a fill byte nor RSTn
            const uint8_t* q = er.p;
            while (q + 1 < er.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
            pos = (size_t)(q - bytes);
            if (q + 1 >= er.end) decoded = true;   // no end-of-image marker: take what the scans gave
        }
        // every other segment (APPn, COM, ...) is skipped
    }

    // dequantise + inverse DCT, once, over everything the scans delivered
    for (auto& c : comps) {
        if (!quant_present[c.tq]) throw image_error("jpeg: missing quantisation table");
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
        const size_t stride = (size_t)c.blocks_w * 8;
        int32_t block[64];
        for (int by = 0; by < c.blocks_h; by++)
            for (int bx = 0; bx < c.blocks_w; bx++) {
                const int16_t* cf = c.coef.data() + ((size_t)by * c.blocks_w + bx) * 64;
                for (int k = 0; k < 64; k++) block[k] = (int32_t)cf[k] * quant[c.tq][k];
                idct_block(block, c.plane.data() + (size_t)by * 8 * stride + (size_t)bx * 8, stride);
            }
        std::vector<int16_t>().swap(c.coef);
    }

    raw_image img;
    img.width = width; img.height = height; img.channels = (int32_t)comps.size();
    img.data.resize((size_t)width * height * 4);
    if (comps.size() == 1) {
        const component& c = comps[0];
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                uint8_t* px = img.data.data() + ((size_t)y * width + x) * 4;
                px[0] = px[1] = px[2] = c.plane[(size_t)y * c.blocks_w * 8 + x];
                px[3] = 255;
            }
        return img;
    }
    const bool ycc = adobe_transform != 0;  // JFIF / no marker: YCbCr; Adobe transform 0: stored RGB
    upsampler up[3] = { { comps[0], hmax, vmax, (width * comps[0].h + hmax - 1) / hmax, (height * comps[0].v + vmax - 1) / vmax },
                        { comps[1], hmax, vmax, (width * comps[1].h + hmax - 1) / hmax, (height * comps[1].v + vmax - 1) / vmax },
                        { comps[2], hmax, vmax, (width * comps[2].h + hmax - 1) / hmax, (height * comps[2].v + vmax - 1) / vmax } };
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const double a = up[0].sample(x, y), b = up[1].sample(x, y), c = up[2].sample(x, y);
            uint8_t* px = img.data.data() + ((size_t)y * width + x) * 4;
            if (ycc) {
                px[0] = clamp8(a + 1.402 * (c - 128.0));
                px[1] = clamp8(a - 0.344136286 * (b - 128.0) - 0.714136286 * (c - 128.0));
                px[2] = clamp8(a + 1.772 * (b - 128.0));
            } else {
                px[0] = clamp8(a); px[1] = clamp8(b); px[2] = clamp8(c);
            }
            px[3] = 255;
        }
    return img;
}

raw_image decode_image(const uint8_t* bytes, size_t size) {
    if (size >= 8 && bytes[0] == 137 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G') return decode_png(bytes, size);
    if (size >= 3 && bytes[0] == 0xFF && bytes[1] == 0xD8 && bytes[2] == 0xFF) return decode_jpeg(bytes, size);
    throw image_error("image: neither a PNG nor a JPEG file");
}

raw_image load_image(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw image_error("image: cannot open " + path);
    std::vector<uint8_t> bytes;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    try {
        return decode_image(bytes.data(), bytes.size());
    } catch (const image_error& e) {
        throw image_error(path + ": " + e.what());
    }
}
