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

#include <new>
#include <string>
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
    return encode(path, src, height, width, stride, 1, nullptr, level);
}

int samrs_io_png_write_lut_rgb(const char* path, const uint8_t* src, int height, int width, size_t stride, const uint8_t* lut, int level) {
    if (!lut) return SAMRS_IO_ESIZE;
    return encode(path, src, height, width, stride, 1, lut, level);
}

int samrs_io_png_write_rgb(const char* path, const uint8_t* src, int height, int width, size_t stride, int level) {
    return encode(path, src, height, width, stride, 3, nullptr, level);
}

}  // extern "C"
