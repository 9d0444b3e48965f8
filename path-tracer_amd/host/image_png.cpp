// zlib inflate and the PNG decoder (ISO/IEC 15948), written against the specifications.
#include "image.hpp"

#include <cstdio>
#include <cstring>

namespace {

// ---- RFC 1951 inflate ------------------------------------------------------------------------------------
struct bit_reader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int count = 0;
    void fill() { while (count <= 56 && p != end) { acc |= (uint64_t)(*p++) << count; count += 8; } }
    uint32_t peek(int n) { if (count < n) fill(); return (uint32_t)(acc & ((1ull << n) - 1ull)); }
    void drop(int n) {
        if (count < n) throw image_error("inflate: unexpected end of stream");
        acc >>= n; count -= n;
    }
    uint32_t take(int n) { if (n == 0) return 0; const uint32_t v = peek(n); drop(n); return v; }
    void align_to_byte() { const int r = count & 7; acc >>= r; count -= r; }
};

// canonical Huffman decoding table: counts per length + symbols sorted by (length, value)
struct huffman {
    uint16_t count[16];
    uint16_t symbol[288];
    void build(const uint8_t* lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left = (left << 1) - count[len];
            if (left < 0) throw image_error("inflate: over-subscribed Huffman code");
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(bit_reader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= (int)br.take(1);
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c;
            first <<= 1; code <<= 1;
        }
        throw image_error("inflate: invalid Huffman code");
    }
};

const uint16_t kLengthBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLengthExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

void inflate_codes(bit_reader& br, const huffman& lit, const huffman& dist, std::vector<uint8_t>& out, size_t limit) {
    for (;;) {
        const int sym = lit.decode(br);
        if (sym < 256) {
            if (limit && out.size() >= limit) throw image_error("inflate: output larger than expected");
            out.push_back((uint8_t)sym);
        } else if (sym == 256) {
            return;
        } else {
            if (sym > 285) throw image_error("inflate: invalid length symbol");
            const size_t len = kLengthBase[sym - 257] + br.take(kLengthExtra[sym - 257]);
            const int ds = dist.decode(br);
            if (ds > 29) throw image_error("inflate: invalid distance symbol");
            const size_t d = kDistBase[ds] + br.take(kDistExtra[ds]);
            if (d > out.size()) throw image_error("inflate: distance reaches before the start of the output");
            if (limit && out.size() + len > limit) throw image_error("inflate: output larger than expected");
            size_t from = out.size() - d;
            for (size_t i = 0; i < len; i++) out.push_back(out[from + i]);  // may overlap the bytes being written
        }
    }
}

void inflate_raw(bit_reader& br, std::vector<uint8_t>& out, size_t limit) {
    static const uint8_t kOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    bool last;
    do {
        last = br.take(1) != 0;
        const uint32_t type = br.take(2);
        if (type == 0) {
            br.align_to_byte();
            const uint32_t len = br.take(16), nlen = br.take(16);
            if ((len ^ 0xFFFFu) != nlen) throw image_error("inflate: stored block length check failed");
            if (limit && out.size() + len > limit) throw image_error("inflate: output larger than expected");
            for (uint32_t i = 0; i < len; i++) out.push_back((uint8_t)br.take(8));
        } else if (type == 1) {
            uint8_t lengths[288];
            for (int i = 0; i < 144; i++) lengths[i] = 8;
            for (int i = 144; i < 256; i++) lengths[i] = 9;
            for (int i = 256; i < 280; i++) lengths[i] = 7;
            for (int i = 280; i < 288; i++) lengths[i] = 8;
            huffman lit, dist;
            lit.build(lengths, 288);
            uint8_t dl[30];
            for (int i = 0; i < 30; i++) dl[i] = 5;
            dist.build(dl, 30);
            inflate_codes(br, lit, dist, out, limit);
        } else if (type == 2) {
            const int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1, hclen = (int)br.take(4) + 4;
            if (hlit > 286 || hdist > 30) throw image_error("inflate: too many codes");
            uint8_t cl[19] = { 0 };
            for (int i = 0; i < hclen; i++) cl[kOrder[i]] = (uint8_t)br.take(3);
            huffman clh;
            clh.build(cl, 19);
            uint8_t lengths[286 + 30];
            int n = 0;
            while (n < hlit + hdist) {
                const int sym = clh.decode(br);
                if (sym < 16) { lengths[n++] = (uint8_t)sym; continue; }
                int repeat; uint8_t value = 0;
                if (sym == 16) {
                    if (n == 0) throw image_error("inflate: repeat without a previous length");
                    value = lengths[n - 1]; repeat = 3 + (int)br.take(2);
                } else if (sym == 17) repeat = 3 + (int)br.take(3);
                else repeat = 11 + (int)br.take(7);
                if (n + repeat > hlit + hdist) throw image_error("inflate: code lengths overrun");
                while (repeat--) lengths[n++] = value;
            }
            if (lengths[256] == 0) throw image_error("inflate: no end-of-block code");
            huffman lit, dist;
            lit.build(lengths, hlit);
            dist.build(lengths + hlit, hdist);
            inflate_codes(br, lit, dist, out, limit);
        } else {
            throw image_error("inflate: reserved block type");
        }
    } while (!last);
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

uint32_t crc32_png(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reverses the scanline filters of one (sub-)image in place; `src` holds (1 + stride) bytes per row.
void unfilter(uint8_t* src, size_t rows, size_t stride, size_t bpp) {
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        uint8_t* row = src + y * (stride + 1);
        const int filter = row[0];
        uint8_t* cur = row + 1;
        switch (filter) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < stride; i++)
                cur[i] = (uint8_t)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0));
            break;
        default: throw image_error("png: unknown filter type");
        }
        prev = cur;
    }
}

}  // namespace

std::vector<uint8_t> zlib_inflate(const uint8_t* src, size_t size, size_t expected_size) {
    if (size < 6) throw image_error("inflate: zlib stream too short");
    const uint32_t cmf = src[0], flg = src[1];
    if ((cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0) throw image_error("inflate: bad zlib header");
    if (flg & 0x20) throw image_error("inflate: preset dictionaries are not supported");
    bit_reader br{ src + 2, src + size };
    std::vector<uint8_t> out;
    if (expected_size) out.reserve(expected_size);
    inflate_raw(br, out, expected_size);
    br.align_to_byte();
    uint32_t stored = 0;
    for (int i = 0; i < 4; i++) stored = (stored << 8) | br.take(8);
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < out.size();) {  // Adler-32, reduced every 5552 bytes
        const size_t stop = i + 5552 < out.size() ? i + 5552 : out.size();
        for (; i < stop; i++) { a += out[i]; b += a; }
        a %= 65521u; b %= 65521u;
    }
    if (((b << 16) | a) != stored) throw image_error("inflate: Adler-32 mismatch");
    return out;
}

std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb, uint32_t width, uint32_t height) {
    if (width == 0 || height == 0) throw image_error("png: empty image");
    std::vector<uint8_t> out = { 137, 80, 78, 71, 13, 10, 26, 10 };
    auto be = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        be(out, (uint32_t)data.size());
        const size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        be(out, crc32_png(out.data() + start, out.size() - start));
    };
    std::vector<uint8_t> ihdr;
    be(ihdr, width); be(ihdr, height);
    ihdr.insert(ihdr.end(), { 8, 2, 0, 0, 0 });
    chunk("IHDR", ihdr);
    // scanlines, each led by filter type 0
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * width * 3, rgb + (size_t)(y + 1) * width * 3);
    }
    // zlib stream of stored blocks (at most 65535 bytes each) + Adler-32
    std::vector<uint8_t> z = { 0x78, 0x01 };
    for (size_t at = 0; at < raw.size();) {
        const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + n);
        at += n;
    }
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < raw.size();) {
        const size_t stop = i + 5552 < raw.size() ? i + 5552 : raw.size();
        for (; i < stop; i++) { a += raw[i]; b += a; }
        a %= 65521u; b %= 65521u;
    }
    be(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});
    return out;
}

raw_image decode_png(const uint8_t* bytes, size_t size) {
    static const uint8_t kSignature[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if (size < 8 || std::memcmp(bytes, kSignature, 8) != 0) throw image_error("png: bad signature");
    uint32_t width = 0, height = 0;
    int depth = 0, colour = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool seen_end = false;
    size_t pos = 8;
    while (!seen_end) {
        if (size - pos < 12) throw image_error("png: truncated chunk");
        const uint32_t len = be32(bytes + pos);
        if (len > size - pos - 12) throw image_error("png: chunk length past the end of the file");
        const uint8_t* type = bytes + pos + 4;
        const uint8_t* data = bytes + pos + 8;
        if (crc32_png(type, 4 + (size_t)len) != be32(data + len)) throw image_error("png: chunk CRC mismatch");
        if (colour < 0 && std::memcmp(type, "IHDR", 4) != 0) throw image_error("png: IHDR must come first");
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13 || colour >= 0) throw image_error("png: bad IHDR");
            width = be32(data); height = be32(data + 4);
            depth = data[8]; colour = data[9]; interlace = data[12];
            if (width == 0 || height == 0 || width > (1u << 24) || height > (1u << 24)) throw image_error("png: bad dimensions");
            if ((uint64_t)width * height > (1ull << 28)) throw image_error("png: more than 2^28 pixels");
            if (data[10] != 0 || data[11] != 0 || interlace > 1) throw image_error("png: unknown compression, filter or interlace method");
            const bool ok = (colour == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                            (colour == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                            ((colour == 2 || colour == 4 || colour == 6) && (depth == 8 || depth == 16));
            if (!ok) throw image_error("png: invalid colour type / bit depth combination");
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 || len > 768) throw image_error("png: bad PLTE");
            palette.assign(data, data + len);
        } else if (std::memcmp(type, "tRNS", 4) == 0) {
            trns.assign(data, data + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            seen_end = true;
        } else if (!(type[0] & 0x20)) {
            throw image_error("png: unknown critical chunk");
        }
        pos += 12 + (size_t)len;
    }
    if (idat.empty()) throw image_error("png: no image data");
    if (colour == 3 && palette.empty()) throw image_error("png: indexed image without a palette");

    const int samples = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : 4;
    const size_t bits_per_pixel = (size_t)samples * depth;
    const size_t bpp = bits_per_pixel >= 8 ? bits_per_pixel / 8 : 1;  // filter distance in bytes
    auto row_bytes = [&](size_t w) { return (w * bits_per_pixel + 7) / 8; };

    // the passes: one for a plain image, the seven Adam7 sub-images otherwise
    static const int kX0[7] = { 0, 4, 0, 2, 0, 1, 0 }, kY0[7] = { 0, 0, 4, 0, 2, 0, 1 };
    static const int kDX[7] = { 8, 8, 4, 4, 2, 2, 1 }, kDY[7] = { 8, 8, 8, 4, 4, 2, 2 };
    struct pass { size_t w, h, x0, y0, dx, dy; };
    std::vector<pass> passes;
    if (!interlace) passes.push_back({ width, height, 0, 0, 1, 1 });
    else for (int i = 0; i < 7; i++) {
        const size_t w = (width + kDX[i] - 1 - kX0[i]) / kDX[i], h = (height + kDY[i] - 1 - kY0[i]) / kDY[i];
        if (width > (size_t)kX0[i] && height > (size_t)kY0[i] && w && h) passes.push_back({ w, h, (size_t)kX0[i], (size_t)kY0[i], (size_t)kDX[i], (size_t)kDY[i] });
    }
    size_t total = 0;
    for (const pass& p : passes) total += p.h * (1 + row_bytes(p.w));
    std::vector<uint8_t> raw = zlib_inflate(idat.data(), idat.size(), total);
    if (raw.size() != total) throw image_error("png: image data has the wrong size");

    raw_image img;
    img.width = (int32_t)width; img.height = (int32_t)height;
    img.channels = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 3 : colour == 4 ? 2 : 4;
    if (!trns.empty() && (colour == 0 || colour == 2 || colour == 3)) img.channels += 1;
    img.data.assign((size_t)width * height * 4, 0);

    // colour key of tRNS for grey / RGB images (16-bit big-endian samples, compared at full depth)
    uint32_t key[3] = { 0, 0, 0 };
    bool has_key = false;
    if (!trns.empty() && colour == 0 && trns.size() >= 2) { key[0] = ((uint32_t)trns[0] << 8) | trns[1]; has_key = true; }
    if (!trns.empty() && colour == 2 && trns.size() >= 6) {
        for (int c = 0; c < 3; c++) key[c] = ((uint32_t)trns[2 * c] << 8) | trns[2 * c + 1];
        has_key = true;
    }
    const uint32_t max_value = (1u << depth) - 1u;

    size_t offset = 0;
    for (const pass& p : passes) {
        const size_t stride = row_bytes(p.w);
        unfilter(raw.data() + offset, p.h, stride, bpp);
        for (size_t y = 0; y < p.h; y++) {
            const uint8_t* row = raw.data() + offset + y * (stride + 1) + 1;
            for (size_t x = 0; x < p.w; x++) {
                uint32_t s[4] = { 0, 0, 0, 0 };  // samples at file depth
                if (depth == 8) for (int c = 0; c < samples; c++) s[c] = row[x * samples + c];
                else if (depth == 16) for (int c = 0; c < samples; c++) s[c] = ((uint32_t)row[(x * samples + c) * 2] << 8) | row[(x * samples + c) * 2 + 1];
                else {
                    const size_t bit = x * depth;
                    s[0] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & max_value;
                }
                uint8_t* px = img.data.data() + (((p.y0 + y * p.dy) * width) + (p.x0 + x * p.dx)) * 4;
                // 8-bit value of a sample: 16-bit keeps the high byte, 1/2/4-bit scale to the full range
                auto to8 = [&](uint32_t v) -> uint8_t { return depth == 16 ? (uint8_t)(v >> 8) : depth == 8 ? (uint8_t)v : (uint8_t)(v * 255u / max_value); };
                switch (colour) {
                case 0:
                    px[0] = px[1] = px[2] = to8(s[0]);
                    px[3] = (has_key && s[0] == key[0]) ? 0 : 255;
                    break;
                case 2:
                    px[0] = to8(s[0]); px[1] = to8(s[1]); px[2] = to8(s[2]);
                    px[3] = (has_key && s[0] == key[0] && s[1] == key[1] && s[2] == key[2]) ? 0 : 255;
                    break;
                case 3:
                    if (s[0] * 3 + 2 >= palette.size()) throw image_error("png: palette index out of range");
                    px[0] = palette[s[0] * 3]; px[1] = palette[s[0] * 3 + 1]; px[2] = palette[s[0] * 3 + 2];
                    px[3] = s[0] < trns.size() ? trns[s[0]] : 255;
                    break;
                case 4:
                    px[0] = px[1] = px[2] = to8(s[0]); px[3] = to8(s[1]);
                    break;
                default:
                    px[0] = to8(s[0]); px[1] = to8(s[1]); px[2] = to8(s[2]); px[3] = to8(s[3]);
                    break;
                }
            }
        }
        offset += p.h * (stride + 1);
    }
    return img;
}
