// libsamrs_io.so: PNG read / write for the generation driver's reader and writer threads (include/samrs_io.h).
// Host code only (g++ -O3, links zlib); no interpreter, no GPU.  The PNG layout follows the PNG specification
// (ISO/IEC 15948): signature, IHDR, IDAT (zlib stream of filtered rows), IEND, each chunk CRC-32 protected.
#include "../../include/samrs_io.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace {

const uint8_t kSignature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
const uint32_t kMaxSide = 1u << 16;          // tiles are 1024^2; anything past 65536 a side is not an input of this driver

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline void put_be32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

struct Header {
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0; }
};

// IHDR must be the first chunk: 8 signature + 4 length + 4 type + 13 data + 4 crc
int parse_header(const uint8_t* f, size_t n, Header* hd) {
    if (n < 33 || memcmp(f, kSignature, 8) != 0) return SAMRS_IO_UNSUPPORTED;      // not a PNG: the caller's other decoder
    if (be32(f + 8) != 13 || memcmp(f + 12, "IHDR", 4) != 0) return SAMRS_IO_ECORRUPT;
    if (uint32_t(crc32(0, f + 12, 17)) != be32(f + 29)) return SAMRS_IO_ECORRUPT;
    hd->w = be32(f + 16);
    hd->h = be32(f + 20);
    hd->depth = f[24];
    hd->color = f[25];
    hd->interlace = f[28];
    if (f[26] != 0 || f[27] != 0) return SAMRS_IO_ECORRUPT;                         // compression / filter method
    if (hd->w == 0 || hd->h == 0 || hd->w > kMaxSide || hd->h > kMaxSide) return SAMRS_IO_ESIZE;
    if (hd->channels() == 0) return SAMRS_IO_ECORRUPT;
    if (hd->depth != 8 || hd->interlace != 0) return SAMRS_IO_UNSUPPORTED;
    return SAMRS_IO_OK;
}

// Scratch that lives as long as its reader / writer thread: a tile needs three multi-MiB buffers, and taking them from the
// allocator per call means an mmap + page faults + munmap each time, all of which serialise on the process's address-space
// lock once a few dozen threads do it at once.
uint8_t* scratch(int slot, size_t bytes) {
    thread_local std::vector<uint8_t> buf[3];
    try {
        if (buf[slot].size() < bytes) buf[slot].resize(bytes);
    } catch (const std::bad_alloc&) {
        return nullptr;
    }
    return buf[slot].data();
}

inline uint8_t paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return uint8_t(pa <= pb && pa <= pc ? a : pb <= pc ? b : c);
}

// undo one row's filter in place; `prev` is the reconstructed row above (zeros for the first row)
int unfilter_row(int type, uint8_t* cur, const uint8_t* prev, size_t n, int bpp) {
    switch (type) {
    case 0: break;
    case 1: for (size_t i = bpp; i < n; ++i) cur[i] = uint8_t(cur[i] + cur[i - bpp]); break;
    case 2: for (size_t i = 0; i < n; ++i) cur[i] = uint8_t(cur[i] + prev[i]); break;
    case 3:
        for (size_t i = 0; i < size_t(bpp) && i < n; ++i) cur[i] = uint8_t(cur[i] + (prev[i] >> 1));
        for (size_t i = bpp; i < n; ++i) cur[i] = uint8_t(cur[i] + ((cur[i - bpp] + prev[i]) >> 1));
        break;
    case 4:
        for (size_t i = 0; i < size_t(bpp) && i < n; ++i) cur[i] = uint8_t(cur[i] + prev[i]);                 // paeth(0, b, 0) = b
        for (size_t i = bpp; i < n; ++i) cur[i] = uint8_t(cur[i] + paeth(cur[i - bpp], prev[i], prev[i - bpp]));
        break;
    default: return SAMRS_IO_ECORRUPT;
    }
    return SAMRS_IO_OK;
}

int decode(const uint8_t* f, size_t n, uint8_t* dst, size_t dst_bytes, int* height, int* width) {
    Header hd;
    int rc = parse_header(f, n, &hd);
    if (rc != SAMRS_IO_OK) return rc;
    if (height) *height = int(hd.h);
    if (width) *width = int(hd.w);
    if (!dst) return SAMRS_IO_OK;
    if (dst_bytes < size_t(3) * hd.w * hd.h) return SAMRS_IO_ESIZE;
    const int ch = hd.channels();
    const size_t row = size_t(hd.w) * ch;
    const size_t raw_bytes = (row + 1) * hd.h;
    if (raw_bytes > 0x7fffffffull) return SAMRS_IO_UNSUPPORTED;                      // zlib's 32-bit counters; far beyond any tile
    uint8_t* raw = scratch(0, raw_bytes + row);                                      // + one zero row used as "row above the first"
    if (!raw) return SAMRS_IO_ENOMEM;
    uint8_t palette[256 * 3];
    memset(palette, 0, sizeof palette);
    bool have_palette = false, ended = false;

    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return SAMRS_IO_ENOMEM;
    zs.next_out = raw;
    zs.avail_out = uInt(raw_bytes);
    bool stream_end = false;
    size_t pos = 33;
    while (pos + 12 <= n) {
        uint32_t len = be32(f + pos);
        const uint8_t* type = f + pos + 4;
        if (len > n - pos - 12) { rc = SAMRS_IO_ECORRUPT; break; }
        const uint8_t* data = f + pos + 8;
        if (uint32_t(crc32(0, type, len + 4)) != be32(data + len)) { rc = SAMRS_IO_ECORRUPT; break; }
        if (memcmp(type, "IDAT", 4) == 0) {
            if (!stream_end && len) {
                zs.next_in = const_cast<Bytef*>(data);
                zs.avail_in = len;
                int z = inflate(&zs, Z_NO_FLUSH);
                if (z == Z_STREAM_END) stream_end = true;
                else if (z != Z_OK && z != Z_BUF_ERROR) { rc = SAMRS_IO_ECORRUPT; break; }
                if (z != Z_STREAM_END && zs.avail_in != 0) { rc = SAMRS_IO_ECORRUPT; break; }   // more pixel data than IHDR announces
            }
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) { rc = SAMRS_IO_ECORRUPT; break; }
            memcpy(palette, data, len);
            have_palette = true;
        } else if (memcmp(type, "IEND", 4) == 0) {
            ended = true;
            break;
        }
        pos += size_t(len) + 12;
    }
    const bool complete = zs.avail_out == 0;
    inflateEnd(&zs);
    if (rc != SAMRS_IO_OK) return rc;
    if (!ended || !complete) return SAMRS_IO_ECORRUPT;
    if (hd.color == 3 && !have_palette) return SAMRS_IO_ECORRUPT;

    uint8_t* zero = raw + raw_bytes;
    memset(zero, 0, row);
    const uint8_t* prev = zero;
    for (uint32_t y = 0; y < hd.h; ++y) {
        uint8_t* line = raw + size_t(y) * (row + 1);
        rc = unfilter_row(line[0], line + 1, prev, row, ch);
        if (rc != SAMRS_IO_OK) return rc;
        prev = line + 1;
        uint8_t* out = dst + size_t(y) * hd.w * 3;
        const uint8_t* s = line + 1;
        switch (hd.color) {
        case 2: memcpy(out, s, row); break;
        case 6: for (uint32_t x = 0; x < hd.w; ++x) { out[3 * x] = s[4 * x]; out[3 * x + 1] = s[4 * x + 1]; out[3 * x + 2] = s[4 * x + 2]; } break;
        case 0: for (uint32_t x = 0; x < hd.w; ++x) out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = s[x]; break;
        case 4: for (uint32_t x = 0; x < hd.w; ++x) out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = s[2 * x]; break;
        case 3: for (uint32_t x = 0; x < hd.w; ++x) memcpy(out + 3 * x, palette + 3 * s[x], 3); break;
        }
    }
    return SAMRS_IO_OK;
}

int read_file(const char* path, std::vector<uint8_t>* buf) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return SAMRS_IO_EOPEN;
    int rc = SAMRS_IO_OK;
    if (fseek(fp, 0, SEEK_END) != 0) rc = SAMRS_IO_EOPEN;
    long sz = rc == SAMRS_IO_OK ? ftell(fp) : -1;
    if (sz < 0 || fseek(fp, 0, SEEK_SET) != 0) rc = SAMRS_IO_EOPEN;
    if (rc == SAMRS_IO_OK) {
        try { buf->resize(size_t(sz)); } catch (const std::bad_alloc&) { rc = SAMRS_IO_ENOMEM; }
    }
    if (rc == SAMRS_IO_OK && sz > 0 && fread(buf->data(), 1, size_t(sz), fp) != size_t(sz)) rc = SAMRS_IO_ECORRUPT;
    fclose(fp);
    return rc;
}

// ---- encoder -----------------------------------------------------------------------------------------------------------------

// Filter one row with each of the five PNG filters and keep the one whose output has the smallest sum of |signed byte|
// (the heuristic the PNG specification suggests for truecolour / gray images).  `out` receives type byte + n bytes.
void filter_row(const uint8_t* cur, const uint8_t* prev, size_t n, int bpp, uint8_t* scratch, uint8_t* out) {
    uint8_t* cand[5];
    for (int t = 0; t < 5; ++t) cand[t] = scratch + size_t(t) * n;
    unsigned long best = ~0ul;
    int best_t = 0;
    for (int t = 0; t < 5; ++t) {
        uint8_t* o = cand[t];
        switch (t) {
        case 0: memcpy(o, cur, n); break;
        case 1:
            for (size_t i = 0; i < size_t(bpp) && i < n; ++i) o[i] = cur[i];
            for (size_t i = bpp; i < n; ++i) o[i] = uint8_t(cur[i] - cur[i - bpp]);
            break;
        case 2: for (size_t i = 0; i < n; ++i) o[i] = uint8_t(cur[i] - prev[i]); break;
        case 3:
            for (size_t i = 0; i < size_t(bpp) && i < n; ++i) o[i] = uint8_t(cur[i] - (prev[i] >> 1));
            for (size_t i = bpp; i < n; ++i) o[i] = uint8_t(cur[i] - ((cur[i - bpp] + prev[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < size_t(bpp) && i < n; ++i) o[i] = uint8_t(cur[i] - prev[i]);
            for (size_t i = bpp; i < n; ++i) o[i] = uint8_t(cur[i] - paeth(cur[i - bpp], prev[i], prev[i - bpp]));
            break;
        }
        unsigned long sum = 0;
        for (size_t i = 0; i < n; ++i) sum += o[i] < 128 ? o[i] : 256 - o[i];
        if (sum < best) { best = sum; best_t = t; }
        if (sum == 0) break;                      // a constant run: nothing can beat it
    }
    out[0] = uint8_t(best_t);
    memcpy(out + 1, cand[best_t], n);
}

int write_chunk(FILE* fp, const char* type, const uint8_t* data, uint32_t len) {
    uint8_t head[8];
    put_be32(head, len);
    memcpy(head + 4, type, 4);
    uLong crc = crc32(0, head + 4, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t tail[4];
    put_be32(tail, uint32_t(crc));
    if (fwrite(head, 1, 8, fp) != 8) return SAMRS_IO_EWRITE;
    if (len && fwrite(data, 1, len, fp) != len) return SAMRS_IO_EWRITE;
    if (fwrite(tail, 1, 4, fp) != 4) return SAMRS_IO_EWRITE;
    return SAMRS_IO_OK;
}

// src: [h, w] bytes (stride), expanded through `lut` to RGB when lut != nullptr; or [h, w * ch] when lut == nullptr
int encode(const char* path, const uint8_t* src, int h, int w, size_t stride, int ch, const uint8_t* lut, int level) {
    if (!path || !src || h <= 0 || w <= 0 || uint32_t(h) > kMaxSide || uint32_t(w) > kMaxSide) return SAMRS_IO_ESIZE;
    const int out_ch = lut ? 3 : ch;
    if (stride < size_t(w) * (lut ? 1 : ch)) return SAMRS_IO_ESIZE;
    if (level != SAMRS_IO_LEVEL_RUNS && (level < 1 || level > 9)) level = 6;
    const size_t row = size_t(w) * out_ch;
    const size_t filtered_bytes = (row + 1) * h;
    if (filtered_bytes > 0x7fffffffull) return SAMRS_IO_ESIZE;                  // one IDAT chunk holds < 4 GiB
    uint8_t* filtered = scratch(0, filtered_bytes);
    uint8_t* rows = scratch(1, row * 8);                                              // 5 candidates + current + previous + zeros
    if (!filtered || !rows) return SAMRS_IO_ENOMEM;
    uint8_t* line[2] = {rows + row * 5, rows + row * 6};
    uint8_t* zero = rows + row * 7;
    memset(zero, 0, row);
    const uint8_t* prev = zero;
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src + size_t(y) * stride;
        const uint8_t* cur = s;
        if (lut) {
            uint8_t* d = line[y & 1];
            for (int x = 0; x < w; ++x) { const uint8_t* c = lut + 3 * s[x]; d[3 * x] = c[0]; d[3 * x + 1] = c[1]; d[3 * x + 2] = c[2]; }
            cur = d;
        }
        filter_row(cur, prev, row, out_ch, rows, filtered + size_t(y) * (row + 1));
        prev = cur;
    }
    // one deflate stream over all rows.  SAMRS_IO_LEVEL_RUNS: zlib's run-length strategy (matches at distance 1 only), which is
    // what a filtered class map consists of -- faster than level 6 and no larger on such data
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    const bool runs = level == SAMRS_IO_LEVEL_RUNS;
    if (deflateInit2(&zs, runs ? 6 : level, Z_DEFLATED, 15, 9, runs ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK) return SAMRS_IO_ENOMEM;
    const uLong bound = deflateBound(&zs, uLong(filtered_bytes)) + 64;
    uint8_t* packed = scratch(2, bound);
    if (!packed) { deflateEnd(&zs); return SAMRS_IO_ENOMEM; }
    zs.next_in = filtered;
    zs.avail_in = uInt(filtered_bytes);
    zs.next_out = packed;
    zs.avail_out = uInt(bound);
    const int z = deflate(&zs, Z_FINISH);
    const uLong packed_bytes = zs.total_out;
    deflateEnd(&zs);
    if (z != Z_STREAM_END) return SAMRS_IO_EWRITE;

    std::string tmp;
    // per process: two ranks that (wrongly) took the same image can clobber the final file with identical bytes, never a half-written tmp
    try { tmp = std::string(path) + ".tmp." + std::to_string((long)getpid()); } catch (const std::bad_alloc&) { return SAMRS_IO_ENOMEM; }
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) return SAMRS_IO_EOPEN;
    uint8_t ihdr[13];
    put_be32(ihdr, uint32_t(w));
    put_be32(ihdr + 4, uint32_t(h));
    ihdr[8] = 8;
    ihdr[9] = out_ch == 1 ? 0 : 2;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    int rc = fwrite(kSignature, 1, 8, fp) == 8 ? SAMRS_IO_OK : SAMRS_IO_EWRITE;
    if (rc == SAMRS_IO_OK) rc = write_chunk(fp, "IHDR", ihdr, 13);
    if (rc == SAMRS_IO_OK) rc = write_chunk(fp, "IDAT", packed, uint32_t(packed_bytes));
    if (rc == SAMRS_IO_OK) rc = write_chunk(fp, "IEND", nullptr, 0);
    if (fclose(fp) != 0 && rc == SAMRS_IO_OK) rc = SAMRS_IO_EWRITE;
    if (rc == SAMRS_IO_OK && rename(tmp.c_str(), path) != 0) rc = SAMRS_IO_EWRITE;
    if (rc != SAMRS_IO_OK) remove(tmp.c_str());
    return rc;
}

// ---- label-aware encoder (SAMRS_IO_LEVEL_LABELS) -------------------------------------------------------------------------------
// A class map is piecewise constant, and the colour image of the reference (`seg_color`, main_sam_hbox_semantic.py:163,199) is the
// class map seen through a palette: pixel (y, x) repeats its left neighbour or the pixel above it wherever the LABEL does.  zlib
// has to rediscover that byte by byte (hash chains over 3 MiB of RGB: 40 - 60 ms per 1024^2 tile at level 6 on noise-like maps,
// the dominant host cost of the generation CLI); here the LZ77 parse is done on the 1-byte label map and written out as deflate
// tokens of the RGB stream: a run of labels equal to the row above = one match at distance (row bytes + 1), a run equal to the
// left neighbour = one match at distance (bytes per pixel), anything else = the pixel's literal bytes.  All rows use PNG filter 0
// (none), one dynamic-Huffman deflate block per <= 2^20 tokens; the file is a plain PNG any reader decodes to the same pixels.

struct BitWriter {                                // LSB first; the caller guarantees 8 writable bytes past the current position
    uint8_t* p;
    uint64_t acc = 0;
    int n = 0;                                    // pending bits, < 8 between calls
    explicit BitWriter(uint8_t* at) : p(at) {}
    inline void put(uint32_t v, int bits) {       // bits <= 32; bits == 0 is a no-op (the table-driven emitter relies on it)
        acc |= uint64_t(v) << n;
        n += bits;
        memcpy(p, &acc, 8);                       // little-endian host (x86-64): byte 0 = the oldest bits
        p += n >> 3;
        acc >>= (n & ~7);
        n &= 7;
    }
    inline void put2(uint32_t v1, int bits1, uint32_t v2, int bits2) {      // two strings at once; bits1 + bits2 <= 56
        acc |= (uint64_t(v1) | (uint64_t(v2) << bits1)) << n;
        n += bits1 + bits2;
        memcpy(p, &acc, 8);
        p += n >> 3;
        acc >>= (n & ~7);
        n &= 7;
    }
    uint8_t* flush() { if (n > 0) { *p++ = uint8_t(acc); acc = 0; n = 0; } return p; }
};

// code lengths (<= maxbits) of a Huffman code for freq[0 .. n): plain Huffman; if the tree is too deep the counts are flattened
// (halved, rounding up) and the tree rebuilt -- converges in a few rounds, costs a fraction of a percent of size when it triggers
void huffman_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* len) {
    std::vector<uint32_t> f(freq, freq + n);
    for (;;) {
        struct Node { uint64_t w; int l, r; };
        std::vector<Node> nodes;
        std::vector<int> heap;
        for (int i = 0; i < n; ++i) if (f[i]) { nodes.push_back({f[i], -1 - i, 0}); heap.push_back(int(nodes.size()) - 1); }
        for (int i = 0; i < n; ++i) len[i] = 0;
        if (heap.empty()) return;
        if (heap.size() == 1) { len[-1 - nodes[heap[0]].l] = 1; return; }
        auto cmp = [&](int a, int b) { return nodes[a].w > nodes[b].w || (nodes[a].w == nodes[b].w && a > b); };
        std::make_heap(heap.begin(), heap.end(), cmp);
        while (heap.size() > 1) {
            std::pop_heap(heap.begin(), heap.end(), cmp); int a = heap.back(); heap.pop_back();
            std::pop_heap(heap.begin(), heap.end(), cmp); int b = heap.back(); heap.pop_back();
            nodes.push_back({nodes[a].w + nodes[b].w, a, b});
            heap.push_back(int(nodes.size()) - 1);
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
        int deepest = 0;
        std::vector<std::pair<int, int>> stack{{heap[0], 0}};
        while (!stack.empty()) {
            auto [id, d] = stack.back(); stack.pop_back();
            if (nodes[id].l < 0) { len[-1 - nodes[id].l] = uint8_t(d < 255 ? d : 255); if (d > deepest) deepest = d; }
            else { stack.push_back({nodes[id].l, d + 1}); stack.push_back({nodes[id].r, d + 1}); }
        }
        if (deepest <= maxbits) return;
        for (int i = 0; i < n; ++i) if (f[i]) f[i] = (f[i] + 1) / 2;
    }
}

// canonical codes, bit-reversed for the LSB-first writer
void canonical_codes(const uint8_t* len, int n, uint16_t* code) {
    int count[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; ++i) count[len[i]]++;
    count[0] = 0;
    int c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) { code[i] = 0; continue; }
        unsigned v = unsigned(next[len[i]]++), r = 0;
        for (int b = 0; b < len[i]; ++b) { r = (r << 1) | (v & 1); v >>= 1; }
        code[i] = uint16_t(r);
    }
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct LenCode { uint16_t sym; uint8_t ebits; uint16_t eval; };
struct LenTable {
    uint8_t c[259];
    LenTable() { for (int k = 0; k < 29; ++k) for (int l = kLenBase[k]; l <= (k == 28 ? 258 : kLenBase[k + 1] - 1) && l <= 258; ++l) c[l] = uint8_t(k); c[258] = 28; }
};
const LenTable kLenTable;
inline LenCode len_code(int len) {                 // 3 .. 258
    const int c = kLenTable.c[len];
    return {uint16_t(257 + c), kLenExtra[c], uint16_t(len - kLenBase[c])};
}
inline void dist_code(int dist, int* sym, int* ebits, int* eval) {
    int c = 29;
    while (kDistBase[c] > dist) --c;
    *sym = c; *ebits = kDistExtra[c]; *eval = dist - kDistBase[c];
}

// ---- the parse: one pass over the label map, tokens in PIXEL units (shared by the gray and the colour stream) ----
// token: 0x00000000 | byte   a raw literal byte (the row's filter-type byte)
//        0x40000000 | label  one literal pixel
//        0x80000000 | up << 16 | npx   a run of npx pixels (1 .. 258) equal to the row above (up) or to the left neighbour; the
//                                      emitter cuts it into matches of at most 258 bytes of ITS stream (86 pixels of RGB)
// `unit` = bytes per pixel the parse is tuned for: a match must cover >= 3 BYTES, and an "up" match costs ~10 more bits than a
// "left" one.  A parse made for unit 3 (min 1 pixel) also serves the gray stream: its emitter turns runs shorter than 3 pixels
// into literals (labels read from the map) -- a few bytes larger than the unit-1 parse would give, one parse instead of two.
struct TokenBuf {                                 // per writer thread; grows, never shrinks, never zero-filled
    uint32_t* p = nullptr;
    size_t cap = 0, n = 0;
    ~TokenBuf() { free(p); }
    const uint32_t& operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
};

int parse_labels(const uint8_t* src, int h, int w, size_t stride, int unit, bool up_ok, TokenBuf* tokens) {
    const int max_px = 258;                        // per token; the emitter cuts a run into matches of <= 258 BYTES of its stream
    const int min_px = unit == 1 ? 3 : 1;
    const int up_bias = unit == 3 ? 1 : 2;
    const size_t worst = size_t(h) * (size_t(w) + 1);                     // all literals
    if (tokens->cap < worst) {
        uint32_t* q = static_cast<uint32_t*>(realloc(tokens->p, worst * sizeof(uint32_t)));
        if (!q) return SAMRS_IO_ENOMEM;
        tokens->p = q; tokens->cap = worst;
    }
    uint32_t* t = tokens->p;
    for (int y = 0; y < h; ++y) {
        const uint8_t* g = src + size_t(y) * stride;
        const uint8_t* up = (y > 0 && up_ok) ? g - stride : nullptr;
        *t++ = 0;                                  // the filter-type byte
        int x = 0;
        while (x < w) {
            int lu = 0, ll = 0;
            if (up) {
                while (x + lu + 8 <= w) {          // eight labels per compare
                    uint64_t a, b;
                    memcpy(&a, g + x + lu, 8); memcpy(&b, up + x + lu, 8);
                    if (a != b) { lu += __builtin_ctzll(a ^ b) >> 3; goto up_done; }
                    lu += 8;
                }
                while (x + lu < w && g[x + lu] == up[x + lu]) ++lu;
            }
        up_done:
            if (x > 0) {
                while (x + ll + 8 <= w) {
                    uint64_t a, b;
                    memcpy(&a, g + x + ll, 8); memcpy(&b, g + x + ll - 1, 8);
                    if (a != b) { ll += __builtin_ctzll(a ^ b) >> 3; goto left_done; }
                    ll += 8;
                }
                while (x + ll < w && g[x + ll] == g[x + ll - 1]) ++ll;
            }
        left_done:
            // the colour of a label never changes inside an image, so equal labels <=> equal pixels for the match
            int n = 0;
            uint32_t kind = 0;
            if (lu >= min_px && lu > ll + up_bias) { n = lu; kind = 0x80010000u; }
            else if (ll >= min_px) { n = ll; kind = 0x80000000u; }
            else if (lu >= min_px) { n = lu > max_px ? max_px : lu; kind = 0x80010000u; }
            if (!n) { *t++ = 0x40000000u | g[x]; ++x; continue; }
            while (n > 0) { const int m = n > max_px ? max_px : n; if (m < min_px) break; *t++ = kind | uint32_t(m); x += m; n -= m; }
        }
    }
    tokens->n = size_t(t - tokens->p);
    return SAMRS_IO_OK;
}

// ---- the emitter: the token list as the zlib stream of a bpp-byte-per-pixel image (bpp 1: the labels; bpp 3: lut[label]) ----
// Token types alternate at random on a noise-like map, so neither pass branches on them: a token becomes two table indices
// (literal pixel: its label twice; match: 256 + length in bytes, 512 + up) whose histogram is counted in the first pass and whose
// pre-merged bit strings are written in the second (a string of 0 bits is a no-op of the bit writer).  The rare shapes -- the
// filter byte of a row, runs longer than one match, runs of the gray stream shorter than deflate's 3-byte minimum -- take a side path.
struct ByteBuf {                                  // per writer thread; grows, never zero-filled
    uint8_t* p = nullptr;
    size_t cap = 0;
    ~ByteBuf() { free(p); }
    bool reserve(size_t n) {
        if (cap >= n) return true;
        uint8_t* q = static_cast<uint8_t*>(realloc(p, n));
        if (!q) return false;
        p = q; cap = n;
        return true;
    }
};

int emit_labels(const char* path, const TokenBuf& tok, const uint8_t* src, int h, int w, size_t stride,
                const uint8_t* lut, bool up_ok) {
    const int bpp = lut ? 3 : 1;
    const size_t row = size_t(w) * bpp;
    const int d_left = bpp, d_up = int(row) + 1;
    const uint32_t lit_px = bpp == 1 ? 3 : 1;      // runs below this many pixels are written as literals (deflate: >= 3 bytes)
    const uint32_t max_px = 258 / bpp;             // pixels per match of this stream (a token carries up to 258 pixels)
    thread_local ByteBuf out_tl, rowbuf_tl;
    // worst case per token: three matches of (15 + 5) + (15 + 13) bits; per block: ~600 bytes of code lengths
    const size_t nblocks = (tok.size() >> 20) + 1;
    if (!out_tl.reserve(tok.size() * 18 + nblocks * 1024 + 64) || !rowbuf_tl.reserve(row + 8)) return SAMRS_IO_ENOMEM;
    uint8_t* const out = out_tl.p;
    uint8_t* const rowbuf = rowbuf_tl.p;
    size_t out_size = 0;
    uLong adler = adler32(0L, Z_NULL, 0);
    try {
        uint32_t lut32[256];
        if (lut) for (int v = 0; v < 256; ++v) lut32[v] = uint32_t(lut[3 * v]) | (uint32_t(lut[3 * v + 1]) << 8) | (uint32_t(lut[3 * v + 2]) << 16);
        for (int y = 0; y < h; ++y) {              // the raw (filter 0) rows, only for the zlib checksum
            const uint8_t* g = src + size_t(y) * stride;
            rowbuf[0] = 0;
            if (lut) { uint8_t* d = rowbuf + 1; for (int x = 0; x < w; ++x, d += 3) memcpy(d, &lut32[g[x]], 4); }   // 4-byte stores, 3-byte steps
            else memcpy(rowbuf + 1, g, size_t(w));
            adler = adler32(adler, rowbuf, uInt(row + 1));
        }
        // ---- zlib stream: header, dynamic-Huffman blocks of <= 2^20 tokens, adler32 ----
        out[0] = 0x78; out[1] = 0x9c;
        BitWriter bw(out + 2);
        int ls, le, lv, us = 0, ue = 0, uv = 0;
        dist_code(d_left, &ls, &le, &lv);
        if (up_ok) dist_code(d_up, &us, &ue, &uv);
        const size_t BLOCK = size_t(1) << 20;
        // position of the token walk in the map (short runs of the gray stream read their labels from it): the row the walk is in
        // (a row's first token is its filter byte) and the column, per pass, carried from block to block
        size_t y_count = 0, y_write = 0;
        const uint8_t *cur_count = src, *cur_write = src;
        uint32_t x_count = 0, x_write = 0;
        for (size_t t0 = 0; t0 < tok.size() || t0 == 0; t0 += BLOCK) {
            const size_t t1 = t0 + BLOCK < tok.size() ? t0 + BLOCK : tok.size();
            uint32_t fl[286] = {0}, fd[30] = {0};
            // first index: label | 256 + match bytes; second: label (colour: the pixel's third code) | 256 + label (gray: the second
            // pixel of a two-pixel run written as literals) | 512 + up
            uint32_t h1[256 + 259] = {0}, h2[514] = {0};
            {
                const uint8_t* cur = cur_count;
                uint32_t x = x_count;
                for (size_t t = t0; t < t1; ++t) {
                    const uint32_t k = tok[t];
                    const uint32_t n = k & 0x1ff, m = k >> 31, upf = (k >> 16) & 1;
                    if (__builtin_expect(k >= 0x40000000u && (!m || n <= max_px), 1)) {
                        const uint32_t lit = m & uint32_t(n < lit_px);           // gray only: a run of 1 or 2 pixels
                        const uint32_t v0 = cur[x], v1 = cur[x + 1 < uint32_t(w) ? x + 1 : x];
                        h1[lit ? v0 : (m ? 256 + n * bpp : (k & 0xff))]++;
                        h2[lit ? (n == 2 ? 256 + v1 : v0) : (m ? 512 + upf : (k & 0xff))]++;
                        x += m ? n : 1;
                    } else if (!m) { fl[k & 0xff]++; cur = src + y_count * stride; ++y_count; x = 0; }      // the filter byte: a new row
                    else {
                        for (uint32_t r = n; r > 0; r -= (r > max_px ? max_px : r)) { h1[256 + (r > max_px ? max_px : r) * bpp]++; h2[512 + upf]++; }
                        x += n;
                    }
                }
                cur_count = cur; x_count = x;
            }
            for (int l = 3; l <= 258; ++l) if (h1[256 + l]) fl[len_code(l).sym] += h1[256 + l];
            for (int v = 0; v < 256; ++v) {
                if (lut) { const uint8_t* c = lut + 3 * v; fl[c[0]] += h1[v]; fl[c[1]] += h1[v]; fl[c[2]] += h1[v]; }
                else fl[v] += h1[v] + h2[256 + v];
            }
            fd[ls] += h2[512];
            if (up_ok) fd[us] += h2[513];
            fl[256] = 1;
            if (!fd[ls] && !(up_ok && fd[us])) fd[0] = 1;                    // at least one distance code must be defined
            uint8_t ll_len[286], d_len[30];
            uint16_t ll_code[286], d_code[30];
            huffman_lengths(fl, 286, 15, ll_len);
            huffman_lengths(fd, 30, 15, d_len);
            canonical_codes(ll_len, 286, ll_code);
            canonical_codes(d_len, 30, d_code);
            int hlit = 286, hdist = 30;
            while (hlit > 257 && !ll_len[hlit - 1]) --hlit;
            while (hdist > 1 && !d_len[hdist - 1]) --hdist;
            // code-length sequence with the 16 / 17 / 18 run symbols
            std::vector<uint16_t> cl;              // low 5 bits symbol, then extra value << 5
            std::vector<uint8_t> seq(ll_len, ll_len + hlit);
            seq.insert(seq.end(), d_len, d_len + hdist);
            for (size_t i = 0; i < seq.size();) {
                size_t j = i;
                while (j < seq.size() && seq[j] == seq[i]) ++j;
                size_t run = j - i;
                if (seq[i] == 0) {
                    while (run >= 11) { const size_t r = run > 138 ? 138 : run; cl.push_back(uint16_t(18 | ((r - 11) << 5))); run -= r; }
                    if (run >= 3) { cl.push_back(uint16_t(17 | ((run - 3) << 5))); run = 0; }
                    while (run--) cl.push_back(0);
                } else {
                    cl.push_back(seq[i]); --run;
                    while (run >= 3) { const size_t r = run > 6 ? 6 : run; cl.push_back(uint16_t(16 | ((r - 3) << 5))); run -= r; }
                    while (run--) cl.push_back(seq[i]);
                }
                i = j;
            }
            uint32_t fc[19] = {0};
            for (uint16_t v : cl) fc[v & 31]++;
            uint8_t c_len[19];
            uint16_t c_code[19];
            huffman_lengths(fc, 19, 7, c_len);
            canonical_codes(c_len, 19, c_code);
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            int hclen = 19;
            while (hclen > 4 && !c_len[order[hclen - 1]]) --hclen;
            bw.put(t1 == tok.size() ? 1u : 0u, 1);
            bw.put(2u, 2);                          // dynamic Huffman
            bw.put(uint32_t(hlit - 257), 5); bw.put(uint32_t(hdist - 1), 5); bw.put(uint32_t(hclen - 4), 4);
            for (int i = 0; i < hclen; ++i) bw.put(c_len[order[i]], 3);
            for (uint16_t v : cl) {
                const int sym = v & 31;
                bw.put(c_code[sym], c_len[sym]);
                if (sym == 16) bw.put(uint32_t(v >> 5), 2);
                else if (sym == 17) bw.put(uint32_t(v >> 5), 3);
                else if (sym == 18) bw.put(uint32_t(v >> 5), 7);
            }
            // pre-merged bit strings per table index.  b1 / n1: a literal pixel's first code (colour: codes 0 + 1 merged, <= 30 bits),
            // or a match length with its extra bits (<= 20 bits); b2 / n2: the colour pixel's third code (gray: nothing), or the
            // distance with its extra bits (<= 28 bits)
            uint32_t b1[256 + 259], b2[514];
            uint8_t n1[256 + 259], n2[514];
            for (int v = 0; v < 256; ++v) {
                if (lut) {
                    const uint8_t* c = lut + 3 * v;
                    b1[v] = uint32_t(ll_code[c[0]]) | (uint32_t(ll_code[c[1]]) << ll_len[c[0]]);
                    n1[v] = uint8_t(ll_len[c[0]] + ll_len[c[1]]);
                    b2[v] = ll_code[c[2]]; n2[v] = ll_len[c[2]];
                    b2[256 + v] = 0; n2[256 + v] = 0;
                } else { b1[v] = ll_code[v]; n1[v] = ll_len[v]; b2[v] = 0; n2[v] = 0; b2[256 + v] = ll_code[v]; n2[256 + v] = ll_len[v]; }
            }
            for (int l = 0; l <= 258; ++l) {
                if (l < 3) { b1[256 + l] = 0; n1[256 + l] = 0; continue; }
                const LenCode lc = len_code(l);
                b1[256 + l] = uint32_t(ll_code[lc.sym]) | (uint32_t(lc.eval) << ll_len[lc.sym]);
                n1[256 + l] = uint8_t(ll_len[lc.sym] + lc.ebits);
            }
            b2[512] = uint32_t(d_code[ls]) | (uint32_t(lv) << d_len[ls]); n2[512] = uint8_t(d_len[ls] + le);
            b2[513] = uint32_t(d_code[us]) | (uint32_t(uv) << d_len[us]); n2[513] = uint8_t(d_len[us] + ue);
            const uint8_t* cur = cur_write;
            uint32_t x = x_write;
            for (size_t t = t0; t < t1; ++t) {
                const uint32_t k = tok[t];
                const uint32_t n = k & 0x1ff, m = k >> 31, upf = (k >> 16) & 1;
                if (__builtin_expect(k >= 0x40000000u && (!m || n <= max_px), 1)) {
                    const uint32_t lit = m & uint32_t(n < lit_px);
                    const uint32_t v0 = cur[x], v1 = cur[x + 1 < uint32_t(w) ? x + 1 : x];
                    const uint32_t i1 = lit ? v0 : (m ? 256 + n * bpp : (k & 0xff));
                    const uint32_t i2 = lit ? (n == 2 ? 256 + v1 : v0) : (m ? 512 + upf : (k & 0xff));
                    bw.put2(b1[i1], n1[i1], b2[i2], n2[i2]);           // <= 30 + 15 (a colour pixel) or 20 + 28 (a match) bits
                    x += m ? n : 1;
                } else if (!m) { bw.put(ll_code[k & 0xff], ll_len[k & 0xff]); cur = src + y_write * stride; ++y_write; x = 0; }
                else {
                    for (uint32_t r = n; r > 0; r -= (r > max_px ? max_px : r)) {
                        const uint32_t len = (r > max_px ? max_px : r) * bpp;
                        bw.put(b1[256 + len], n1[256 + len]);
                        bw.put(b2[512 + upf], n2[512 + upf]);
                    }
                    x += n;
                }
            }
            cur_write = cur; x_write = x;
            bw.put(ll_code[256], ll_len[256]);
            if (t1 == tok.size()) break;
        }
        uint8_t* e = bw.flush();
        *e++ = uint8_t(adler >> 24); *e++ = uint8_t(adler >> 16); *e++ = uint8_t(adler >> 8); *e++ = uint8_t(adler);
        out_size = size_t(e - out);
    } catch (const std::bad_alloc&) { return SAMRS_IO_ENOMEM; }
    if (out_size > 0x7fffffffull) return SAMRS_IO_ESIZE;

    std::string tmp;
    try { tmp = std::string(path) + ".tmp." + std::to_string((long)getpid()); } catch (const std::bad_alloc&) { return SAMRS_IO_ENOMEM; }
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) return SAMRS_IO_EOPEN;
    uint8_t ihdr[13];
    put_be32(ihdr, uint32_t(w));
    put_be32(ihdr + 4, uint32_t(h));
    ihdr[8] = 8;
    ihdr[9] = bpp == 1 ? 0 : 2;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    int rc = fwrite(kSignature, 1, 8, fp) == 8 ? SAMRS_IO_OK : SAMRS_IO_EWRITE;
    if (rc == SAMRS_IO_OK) rc = write_chunk(fp, "IHDR", ihdr, 13);
    if (rc == SAMRS_IO_OK) rc = write_chunk(fp, "IDAT", out, uint32_t(out_size));
    if (rc == SAMRS_IO_OK) rc = write_chunk(fp, "IEND", nullptr, 0);
    if (fclose(fp) != 0 && rc == SAMRS_IO_OK) rc = SAMRS_IO_EWRITE;
    if (rc == SAMRS_IO_OK && rename(tmp.c_str(), path) != 0) rc = SAMRS_IO_EWRITE;
    if (rc != SAMRS_IO_OK) remove(tmp.c_str());
    return rc;
}

// gray_path: the class map as an 8-bit gray PNG; color_path + lut: the same map through the palette; either may be null (not both)
int encode_labels(const char* gray_path, const char* color_path, const uint8_t* src, int h, int w, size_t stride, const uint8_t* lut) {
    if ((!gray_path && !color_path) || (color_path && !lut) || !src || h <= 0 || w <= 0 || uint32_t(h) > kMaxSide || uint32_t(w) > kMaxSide)
        return SAMRS_IO_ESIZE;
    if (stride < size_t(w)) return SAMRS_IO_ESIZE;
    thread_local TokenBuf tok_tl;
    const int unit = color_path ? 3 : 1;
    // the "up" distance (row bytes + 1) must fit deflate's 32 KiB window in EVERY stream written from this parse
    const bool up_ok = size_t(w) * unit + 1 <= 32768;
    int rc = parse_labels(src, h, w, stride, unit, up_ok, &tok_tl);
    if (rc == SAMRS_IO_OK && gray_path) rc = emit_labels(gray_path, tok_tl, src, h, w, stride, nullptr, up_ok);
    if (rc == SAMRS_IO_OK && color_path) rc = emit_labels(color_path, tok_tl, src, h, w, stride, lut, up_ok);
    return rc;
}

}  // namespace

extern "C" {

int samrs_io_abi_version(void) { return SAMRS_IO_ABI_VERSION; }

int samrs_io_png_decode_rgb(const uint8_t* file, size_t file_bytes, uint8_t* dst, size_t dst_bytes, int* height, int* width) {
    if (!file) return SAMRS_IO_ESIZE;
    return decode(file, file_bytes, dst, dst_bytes, height, width);
}

int samrs_io_png_info(const char* path, int* height, int* width) {
    if (!path) return SAMRS_IO_EOPEN;
    FILE* fp = fopen(path, "rb");
    if (!fp) return SAMRS_IO_EOPEN;
    uint8_t head[33];
    size_t got = fread(head, 1, sizeof head, fp);
    fclose(fp);
    return decode(head, got, nullptr, 0, height, width);
}

int samrs_io_png_read_rgb(const char* path, uint8_t* dst, size_t dst_bytes, int* height, int* width) {
    if (!path || !dst) return SAMRS_IO_ESIZE;
    thread_local std::vector<uint8_t> buf;
    int rc = read_file(path, &buf);
    if (rc != SAMRS_IO_OK) return rc;
    return decode(buf.data(), buf.size(), dst, dst_bytes, height, width);
}

int samrs_io_png_write_gray(const char* path, const uint8_t* src, int height, int width, size_t stride, int level) {
    if (level == SAMRS_IO_LEVEL_LABELS) return encode_labels(path, nullptr, src, height, width, stride, nullptr);
    return encode(path, src, height, width, stride, 1, nullptr, level);
}

int samrs_io_png_write_lut_rgb(const char* path, const uint8_t* src, int height, int width, size_t stride, const uint8_t* lut, int level) {
    if (!lut) return SAMRS_IO_ESIZE;
    if (level == SAMRS_IO_LEVEL_LABELS) return encode_labels(nullptr, path, src, height, width, stride, lut);
    return encode(path, src, height, width, stride, 1, lut, level);
}

int samrs_io_png_write_label_pair(const char* gray_path, const char* color_path, const uint8_t* src, int height, int width, size_t stride,
                                  const uint8_t* lut) {
    if (!gray_path || !color_path || !lut) return SAMRS_IO_ESIZE;
    return encode_labels(gray_path, color_path, src, height, width, stride, lut);
}

int samrs_io_png_write_rgb(const char* path, const uint8_t* src, int height, int width, size_t stride, int level) {
    return encode(path, src, height, width, stride, 3, nullptr, level);
}

}  // extern "C"
