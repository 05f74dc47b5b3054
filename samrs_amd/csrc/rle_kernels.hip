// rle_kernels.hip -- COCO run-length encoding of full-resolution binary masks on the device (gfx950).
//
// Replaces, per instance, `maskUtils.encode(np.asfortranarray(mask))` + `.decode('ascii')` of
// Generate Dataset/main_sam_hbox_semantic.py:201-202 (the in-tree statement of the counts half is
// Generate Dataset/segment_anything/utils/amg.py:107-135): column-major run lengths starting with the run of
// zeros, each count delta-coded against the count two positions earlier (from the 4th on), cut into 5-bit
// groups with a continuation bit, + 48 -> printable ASCII (cocoapi rleToString).
//
// HBM-bound byte / integer work, three kernels per batch of masks:
//   1. rle_bitpack_kernel   u8 [n][H][W] -> bit matrix [n][W][ceil(H/32)] (bit b of word yw of column x = pixel
//                           (y = 32 yw + b, x)): coalesced 4-byte row reads, a lane owns 4 columns x 128 rows and
//                           writes 16 bytes per column.  After this the column-major pixel sequence IS the word
//                           sequence, and a run boundary is a set bit of  w ^ ((w << 1) | last bit before w).
//   2. rle_encode_kernel    one 1024-thread block per mask: thread = column: popcount of the boundary words ->
//                           block scan -> boundary positions (uint32) -> counts = position differences -> chars
//                           per count -> block scan -> the ASCII string in a per-mask scratch row.
//   3. rle_compact_kernel   packs the strings of the batch behind a device-side cursor into the caller's buffer
//                           (16-byte aligned starts, 16-byte copies) and writes (offset, length, n_counts) per mask.
// Integer arithmetic only: bit-exact with samrs_amd/rle.py (tests/test_rle_gpu.py).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int RLE_THREADS = 1024;

__device__ __forceinline__ uint32_t nz_bytes(uint32_t w) {      // 0x80 in every byte of w that is not 0
    return (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
}

// grid (ceil(H / 128), n, ceil(W / 1024)), 256 threads.  Thread t: columns x0 + 4t .. +3, rows y0 .. y0 + 127.
__global__ __launch_bounds__(256) void rle_bitpack_kernel(const uint8_t* __restrict__ masks, uint32_t* __restrict__ bits,
                                                          int H, int W, int YW) {
    const int m = blockIdx.y;
    const int x = blockIdx.z * 1024 + threadIdx.x * 4;
    const int y0 = blockIdx.x * 128;
    if (x >= W) return;
    const uint8_t* src = masks + (size_t)m * H * W;
    uint32_t w[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) w[c][r] = 0u;
    const bool wide = (W & 3) == 0 && (((uintptr_t)src) & 3) == 0;       // whole 4-byte words of a row
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            const int y = y0 + 32 * r + b;
            uint32_t px = 0u;
            if (y < H) {
                if (wide) {
                    px = *reinterpret_cast<const uint32_t*>(src + (size_t)y * W + x);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (x + c < W) px |= (uint32_t)src[(size_t)y * W + x + c] << (8 * c);
                }
            }
            const uint32_t t = nz_bytes(px);
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c][r] |= ((t >> (8 * c + 7)) & 1u) << b;
        }
    }
    const int yw0 = blockIdx.x * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (x + c >= W) break;
        uint32_t* dst = bits + ((size_t)m * W + x + c) * YW + yw0;
        if (yw0 + 4 <= YW && (YW & 3) == 0) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(w[c][0], w[c][1], w[c][2], w[c][3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (yw0 + r < YW) dst[r] = w[c][r];
        }
    }
}

// exclusive block scan of one value per thread (1024 threads = 16 waves); returns the exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* sm /*[17]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    __syncthreads();                       // sm free (previous use)
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < RLE_THREADS / 64; ++i) { const uint32_t t = sm[i]; sm[i] = run; run += t; }
        sm[16] = run;
    }
    __syncthreads();
    *total = sm[16];
    return sm[wave] + inc - v;
}

// number of chars of one delta-coded count (cocoapi rleToString) and the chars themselves
__device__ __forceinline__ int rle_chars(long long x, unsigned char* __restrict__ out /* or null: count only */) {
    int n = 0;
    bool more = true;
    while (more) {
        int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        if (out) out[n] = (unsigned char)(c + 48);
        ++n;
    }
    return n;
}

// One block per mask.  bits [W][YW]; pos scratch [HW + 2] uint32; str scratch [str_cap] bytes; meta[m] = (length, n_counts).
__global__ __launch_bounds__(RLE_THREADS) void rle_encode_kernel(const uint32_t* __restrict__ bits_all, uint32_t* __restrict__ pos_all,
                                                                 unsigned char* __restrict__ str_all, long long* __restrict__ meta,
                                                                 int H, int W, int YW, size_t pos_stride, size_t str_cap) {
    __shared__ uint32_t sm[17];
    const int m = blockIdx.x;
    const uint32_t* bits = bits_all + (size_t)m * W * YW;
    uint32_t* pos = pos_all + (size_t)m * pos_stride;
    unsigned char* str = str_all + (size_t)m * str_cap;
    const uint32_t HW = (uint32_t)H * (uint32_t)W;
    const int lastw = (H - 1) >> 5, lastb = (H - 1) & 31;
    const uint32_t lastmask = lastb == 31 ? 0xFFFFFFFFu : ((1u << (lastb + 1)) - 1u);

    // ---- phase 1: run boundaries.  Boundary at element i (column-major) <=> v[i] != v[i-1], v[-1] = 0 ----
    // A column's words are read four at a time (16-byte loads when the row of words allows it).
    const bool vec = (YW & 3) == 0;
#define RLE_LOAD4(dst_, x_, c4_)                                                                          \
    if (vec) {                                                                                           \
        const uint4 q_ = *reinterpret_cast<const uint4*>(bits + (size_t)(x_) * YW + 4 * (c4_));          \
        dst_[0] = q_.x; dst_[1] = q_.y; dst_[2] = q_.z; dst_[3] = q_.w;                                  \
    } else {                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                 \
            dst_[i_] = 4 * (c4_) + i_ <= lastw ? bits[(size_t)(x_) * YW + 4 * (c4_) + i_] : 0u;          \
    }
    uint32_t base = 0;
    const int nc4 = lastw / 4 + 1;
    for (int x0 = 0; x0 < W; x0 += RLE_THREADS) {
        const int x = x0 + (int)threadIdx.x;
        uint32_t cnt = 0;
        uint32_t prev = 0;
        if (x < W) {
            if (x > 0) prev = (bits[(size_t)(x - 1) * YW + lastw] >> lastb) & 1u;
            uint32_t p = prev;
            for (int c4 = 0; c4 < nc4; ++c4) {
                uint32_t wv[4];
                RLE_LOAD4(wv, x, c4)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int yw = 4 * c4 + i;
                    const uint32_t vm = yw < lastw ? 0xFFFFFFFFu : (yw == lastw ? lastmask : 0u);
                    const uint32_t w = wv[i] & vm;
                    cnt += __popc((w ^ ((w << 1) | p)) & vm);
                    p = w >> 31;
                }
            }
        }
        uint32_t total;
        uint32_t off = base + block_scan_excl(cnt, sm, &total);
        if (x < W && cnt) {
            uint32_t p = prev;
            for (int c4 = 0; c4 < nc4; ++c4) {
                uint32_t wv[4];
                RLE_LOAD4(wv, x, c4)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int yw = 4 * c4 + i;
                    const uint32_t vm = yw < lastw ? 0xFFFFFFFFu : (yw == lastw ? lastmask : 0u);
                    const uint32_t w = wv[i] & vm;
                    uint32_t t = (w ^ ((w << 1) | p)) & vm;
                    p = w >> 31;
                    while (t) {
                        const int b = __ffs((int)t) - 1;
                        t &= t - 1;
                        pos[off++] = (uint32_t)x * (uint32_t)H + (uint32_t)(yw * 32 + b);
                    }
                }
            }
        }
        base += total;
    }
#undef RLE_LOAD4
    const uint32_t ntrans = base;
    const uint32_t ncounts = ntrans + 1;                      // counts[k] = P(k) - P(k-1), P(-1) = 0, P(ntrans) = HW
    __threadfence_block();
    __syncthreads();

    // ---- phase 2: chars per count, scan, write ----
#define RLE_P(k_) ((k_) < 0 ? 0ll : ((uint32_t)(k_) >= ntrans ? (long long)HW : (long long)pos[(k_)]))
    const uint32_t per = (ncounts + RLE_THREADS - 1) / RLE_THREADS;
    const uint32_t k0 = threadIdx.x * per, k1 = (k0 + per < ncounts) ? k0 + per : ncounts;
    uint32_t nch = 0;
    if (k0 < ncounts) {
        long long pm3 = RLE_P((long long)k0 - 3), pm2 = RLE_P((long long)k0 - 2), pm1 = RLE_P((long long)k0 - 1);
        for (uint32_t k = k0; k < k1; ++k) {
            const long long pk = RLE_P((long long)k);
            long long d = pk - pm1;
            if (k > 2) d -= pm2 - pm3;
            nch += rle_chars(d, nullptr);
            pm3 = pm2; pm2 = pm1; pm1 = pk;
        }
    }
    uint32_t total_chars;
    uint32_t o = block_scan_excl(nch, sm, &total_chars);
    if (k0 < ncounts && (size_t)total_chars <= str_cap) {
        long long pm3 = RLE_P((long long)k0 - 3), pm2 = RLE_P((long long)k0 - 2), pm1 = RLE_P((long long)k0 - 1);
        for (uint32_t k = k0; k < k1; ++k) {
            const long long pk = RLE_P((long long)k);
            long long d = pk - pm1;
            if (k > 2) d -= pm2 - pm3;
            o += rle_chars(d, str + o);
            pm3 = pm2; pm2 = pm1; pm1 = pk;
        }
    }
#undef RLE_P
    if (threadIdx.x == 0) {
        meta[2 * m] = (size_t)total_chars <= str_cap ? (long long)total_chars : -(long long)total_chars;
        meta[2 * m + 1] = (long long)ncounts;
    }
}

// grid (RLE_CP_BLOCKS, n): mask m's string goes to out + align16(cursor) + sum_{j < m} align16(len_j).
// table[m] = (offset, length, n_counts); length < 0: the string did not fit (scratch row or output buffer).
constexpr int RLE_CP_BLOCKS = 8;
__global__ __launch_bounds__(256) void rle_compact_kernel(const unsigned char* __restrict__ str_all, const long long* __restrict__ meta,
                                                          size_t str_cap, unsigned char* __restrict__ out, long long out_cap,
                                                          const long long* __restrict__ cursor, long long* __restrict__ table, int n) {
    const int m = blockIdx.y;
    long long off = (*cursor + 15) & ~15ll;
    for (int j = 0; j < m; ++j) {
        const long long l = meta[2 * j];
        if (l > 0) off += (l + 15) & ~15ll;
    }
    const long long len = meta[2 * m];
    const bool fits = len >= 0 && off + ((len + 15) & ~15ll) <= out_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        table[3 * m] = off;
        table[3 * m + 1] = fits ? len : (len >= 0 ? -len - 1 : len - 1);
        table[3 * m + 2] = meta[2 * m + 1];
    }
    if (!fits || len == 0) return;
    const uint4* src = reinterpret_cast<const uint4*>(str_all + (size_t)m * str_cap);       // str_cap % 16 == 0
    uint4* dst = reinterpret_cast<uint4*>(out + off);
    const long long n16 = (len + 15) >> 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)RLE_CP_BLOCKS * 256) dst[i] = src[i];
}
__global__ void rle_cursor_kernel(const long long* __restrict__ meta, long long* __restrict__ cursor, long long out_cap, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long off = (*cursor + 15) & ~15ll;
    for (int j = 0; j < n; ++j) {
        const long long l = meta[2 * j];
        if (l > 0 && off + ((l + 15) & ~15ll) <= out_cap) off += (l + 15) & ~15ll;
        else if (l > 0) break;
    }
    *cursor = off;
}

}  // namespace

size_t rle_str_capacity(int h, int w) {
    // chars(d) <= 1 + |d| / 16 and sum |d_k| <= 2 HW, n_counts <= HW + 1  ->  total <= 1.125 HW + 1 (+ slack, 16-byte multiple)
    const size_t hw = (size_t)h * w;
    return ((hw + hw / 8 + 64) + 15) & ~(size_t)15;
}
size_t rle_scratch_bytes(int n, int h, int w) {
    const size_t yw = (size_t)(h + 31) / 32;
    const size_t bits = (((size_t)n * w * yw * 4) + 255) & ~(size_t)255;
    const size_t pos = (((size_t)n * ((size_t)h * w + 2) * 4) + 255) & ~(size_t)255;
    const size_t str = (size_t)n * rle_str_capacity(h, w);
    const size_t meta = (size_t)n * 2 * 8 + 256;
    return bits + pos + str + meta;
}

hipError_t launch_rle_encode(const uint8_t* masks, int n, int h, int w, void* scratch, unsigned char* out, long long out_cap,
                             long long* cursor, long long* table, hipStream_t s) {
    if (n < 1 || h < 1 || w < 1 || (size_t)h * w >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const int YW = (h + 31) / 32;
    unsigned char* p = reinterpret_cast<unsigned char*>(scratch);
    uint32_t* bits = reinterpret_cast<uint32_t*>(p);
    p += (((size_t)n * w * YW * 4) + 255) & ~(size_t)255;
    uint32_t* pos = reinterpret_cast<uint32_t*>(p);
    const size_t pos_stride = (size_t)h * w + 2;
    p += (((size_t)n * pos_stride * 4) + 255) & ~(size_t)255;
    unsigned char* str = p;
    const size_t cap = rle_str_capacity(h, w);
    p += (size_t)n * cap;
    long long* meta = reinterpret_cast<long long*>(p);
    rle_bitpack_kernel<<<dim3((h + 127) / 128, n, (w + 1023) / 1024), 256, 0, s>>>(masks, bits, h, w, YW);
    rle_encode_kernel<<<n, RLE_THREADS, 0, s>>>(bits, pos, str, meta, h, w, YW, pos_stride, cap);
    rle_compact_kernel<<<dim3(RLE_CP_BLOCKS, n), 256, 0, s>>>(str, meta, cap, out, out_cap, cursor, table, n);
    rle_cursor_kernel<<<1, 64, 0, s>>>(meta, cursor, out_cap, n);
    return hipGetLastError();
}
