// rle_kernels.hip -- COCO run-length encoding of full-resolution binary masks on the device (gfx950).
//
// Replaces, per instance, `maskUtils.encode(np.asfortranarray(mask))` + `.decode('ascii')` of
// Generate Dataset/main_sam_hbox_semantic.py:201-202 (the in-tree statement of the counts half is
// Generate Dataset/segment_anything/utils/amg.py:107-135): column-major run lengths starting with the run of
// zeros, each count delta-coded against the count two positions earlier (from the 4th on), cut into 5-bit
// groups with a continuation bit, + 48 -> printable ASCII (cocoapi rleToString).
//
// HBM-bound byte / integer work, per batch of masks:
//   1. rle_bitpack_kernel   u8 [n][H][W] -> bit matrix [n][W][ceil(H/32)] (bit b of word yw of column x = pixel
//                           (y = 32 yw + b, x)): coalesced 4-byte row reads, a lane owns 4 columns x 128 rows and
//                           writes 16 bytes per column.  After this the column-major pixel sequence IS the word
//                           sequence, and a run boundary is a set bit of  w ^ ((w << 1) | last bit before w).
//   2. rle_positions_kernel thread = column, 128 columns per block, twice (count, then write): popcount of the boundary words ->
//                           block scan -> boundary positions (uint32, ascending).
//   3. rle_chars_kernel     8 blocks per mask, twice (count, then write): counts = position differences -> chars per count ->
//                           block scan -> the ASCII string, written straight into the caller's buffer behind a device-side
//                           cursor (packed, 16-byte aligned starts) + (offset, length, n_counts) per mask.
// Every cross-block dependence is a kernel boundary (no grid barriers, no atomics): the result does not depend on scheduling.
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

// exclusive block scan of one value per thread (blockDim.x <= 1024 threads, a multiple of 64); returns the exclusive prefix,
// *total = sum
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* sm /*[17]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
        for (int i = 0; i < nw; ++i) { const uint32_t t = sm[i]; sm[i] = run; run += t; }
        sm[16] = run;
    }
    __syncthreads();
    *total = sm[16];
    return sm[wave] + inc - v;
}

// number of chars of one delta-coded count (cocoapi rleToString) and the chars themselves
__device__ __forceinline__ int rle_chars(int x, unsigned char* __restrict__ out /* or null: count only */) {
    int n = 0;
    bool more = true;
    while (more) {
        int c = x & 0x1f;
        x >>= 5;
        more = (c & 0x10) ? x != -1 : x != 0;
        if (more) c |= 0x20;
        if (out) out[n] = (unsigned char)(c + 48);
        ++n;
    }
    return n;
}

// Scratch counters of a batch of n masks (uint32): n_counts[n] | col_chunk[n][RLE_CC_MAX] boundaries per column chunk |
// blk_chars[n][RLE_NB] chars per count block
constexpr int RLE_COLS = 128;                   // columns per block of the position kernels (thread = column)
constexpr int RLE_CC_MAX = 64;                  // column chunks per mask: W <= 8192
constexpr int RLE_NB = 8;                       // count blocks per mask of the char kernels
__host__ __device__ inline size_t rle_ctr_ncounts(int m) { return (size_t)m; }
__host__ __device__ inline size_t rle_ctr_chunk(int n, int m, int c) { return (size_t)n + (size_t)m * RLE_CC_MAX + c; }
__host__ __device__ inline size_t rle_ctr_blk(int n, int m, int b) { return (size_t)n * (1 + RLE_CC_MAX) + (size_t)m * RLE_NB + b; }

// Run boundaries: element i (column-major) is one <=> v[i] != v[i-1], v[-1] = 0.  grid (column chunks, n), thread = column.
// WRITE = false: boundaries per chunk -> col_chunk; WRITE = true: positions (uint32, ascending) behind the chunks before it, and
// the last chunk's block publishes n_counts = boundaries + 1.  A column's words are read four at a time.
template <bool WRITE>
__global__ __launch_bounds__(RLE_COLS) void rle_positions_kernel(const uint32_t* __restrict__ bits_all, uint32_t* __restrict__ pos_all,
                                                                 uint32_t* __restrict__ ctr, int n, int H, int W, int YW,
                                                                 size_t pos_stride) {
    __shared__ uint32_t sm[17];
    const int m = blockIdx.y, chunk = blockIdx.x;
    const uint32_t* bits = bits_all + (size_t)m * W * YW;
    uint32_t* pos = pos_all + (size_t)m * pos_stride;
    const int lastw = (H - 1) >> 5, lastb = (H - 1) & 31;
    const uint32_t lastmask = lastb == 31 ? 0xFFFFFFFFu : ((1u << (lastb + 1)) - 1u);
    const bool vec = (YW & 3) == 0;
#define RLE_LOAD4(dst_, x_, c4_)                                                                          \
    if (vec) {                                                                                           \
        const uint4 q_ = *reinterpret_cast<const uint4*>(bits + (size_t)(x_) * YW + 4 * (c4_));          \
        dst_[0] = q_.x; dst_[1] = q_.y; dst_[2] = q_.z; dst_[3] = q_.w;                                  \
    } else {                                                                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                 \
            dst_[i_] = 4 * (c4_) + i_ <= lastw ? bits[(size_t)(x_) * YW + 4 * (c4_) + i_] : 0u;          \
    }
    const int nc4 = lastw / 4 + 1;
    const int x = chunk * RLE_COLS + (int)threadIdx.x;
    uint32_t cnt = 0, prev = 0;
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
    uint32_t off = block_scan_excl(cnt, sm, &total);
    if (!WRITE) {
        if (threadIdx.x == 0) ctr[rle_ctr_chunk(n, m, chunk)] = total;
        return;
    }
    for (int c = 0; c < chunk; ++c) off += ctr[rle_ctr_chunk(n, m, c)];
    if (chunk == (int)gridDim.x - 1 && threadIdx.x == 0) {
        uint32_t all = total;
        for (int c = 0; c < chunk; ++c) all += ctr[rle_ctr_chunk(n, m, c)];
        ctr[rle_ctr_ncounts(m)] = all + 1;                  // counts[k] = P(k) - P(k-1), P(-1) = 0, P(boundaries) = H W
    }
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
#undef RLE_LOAD4
}

// Counts = position differences, delta-coded against the count two back, 5-bit groups + 48.  RLE_NB blocks per mask (a mask of
// random-init weights has ~130 k counts: one block per mask kept 32 of 256 CUs busy for 0.4 ms); a block owns a contiguous range
// of counts, a thread a contiguous sub-range.  WRITE = false: chars per block -> blk_chars; WRITE = true (same grid): every
// mask's string straight into the caller's buffer at  align16(cursor) + sum of the 16-byte-rounded lengths of the masks before
// it  (so the strings of a batch are packed, 16-byte aligned), each block behind the blocks before it; table[m] = (offset,
// length, n_counts), length < 0: the string did not fit (-length - 1 bytes needed) and nothing of it was written.
// int32 arithmetic (H W < 2^30).
template <bool WRITE>
__global__ __launch_bounds__(RLE_THREADS) void rle_chars_kernel(const uint32_t* __restrict__ pos_all, uint32_t* __restrict__ ctr, int n,
                                                                uint32_t HW, size_t pos_stride, unsigned char* __restrict__ out,
                                                                long long out_cap, const long long* __restrict__ cursor,
                                                                long long* __restrict__ table) {
    __shared__ uint32_t sm[17];
    const int m = blockIdx.y, blk = blockIdx.x;
    const uint32_t* pos = pos_all + (size_t)m * pos_stride;
    const uint32_t ncounts = ctr[rle_ctr_ncounts(m)], ntrans = ncounts - 1;
    unsigned char* str = nullptr;
    uint32_t start = 0;
    if (WRITE) {
        long long off = (*cursor + 15) & ~15ll;
        long long len = 0;
        bool before_fit = true;                 // a mask that does not fit takes no room (the cursor kernel agrees)
        for (int j = 0; j <= m; ++j) {
            long long l = 0;
            for (int b = 0; b < RLE_NB; ++b) {
                const uint32_t c = ctr[rle_ctr_blk(n, j, b)];
                if (j == m && b < blk) start += c;
                l += c;
            }
            const bool fits = off + ((l + 15) & ~15ll) <= out_cap;
            if (j < m) { if (fits) off += (l + 15) & ~15ll; }
            else { len = l; before_fit = fits; }
        }
        if (blk == 0 && threadIdx.x == 0) {
            table[3 * m] = off;
            table[3 * m + 1] = before_fit ? len : -len - 1;
            table[3 * m + 2] = (long long)ncounts;
        }
        if (!before_fit) return;
        str = out + off;
    }
#define RLE_P(k_) ((k_) < 0 ? 0 : ((uint32_t)(k_) >= ntrans ? (int)HW : (int)pos[(k_)]))
    const uint32_t per_blk = (ncounts + RLE_NB - 1) / RLE_NB;
    const uint32_t b0 = blk * per_blk < ncounts ? blk * per_blk : ncounts, b1 = b0 + per_blk < ncounts ? b0 + per_blk : ncounts;
    const uint32_t per = (b1 - b0 + RLE_THREADS - 1) / RLE_THREADS;
    const uint32_t k0 = b0 + threadIdx.x * per < b1 ? b0 + threadIdx.x * per : b1, k1 = k0 + per < b1 ? k0 + per : b1;
    uint32_t nch = 0;
    {
        int pm3 = RLE_P((int)k0 - 3), pm2 = RLE_P((int)k0 - 2), pm1 = RLE_P((int)k0 - 1);
        for (uint32_t k = k0; k < k1; ++k) {
            const int pk = RLE_P((int)k);
            int d = pk - pm1;
            if (k > 2) d -= pm2 - pm3;
            nch += rle_chars(d, nullptr);
            pm3 = pm2; pm2 = pm1; pm1 = pk;
        }
    }
    uint32_t blk_total;
    uint32_t o = start + block_scan_excl(nch, sm, &blk_total);
    if (!WRITE) {
        if (threadIdx.x == 0) ctr[rle_ctr_blk(n, m, blk)] = blk_total;
        return;
    }
    int pm3 = RLE_P((int)k0 - 3), pm2 = RLE_P((int)k0 - 2), pm1 = RLE_P((int)k0 - 1);
    for (uint32_t k = k0; k < k1; ++k) {
        const int pk = RLE_P((int)k);
        int d = pk - pm1;
        if (k > 2) d -= pm2 - pm3;
        o += rle_chars(d, str + o);
        pm3 = pm2; pm2 = pm1; pm1 = pk;
    }
#undef RLE_P
}

// one thread: the cursor moves behind the strings that were written
__global__ void rle_cursor_kernel(const uint32_t* __restrict__ ctr, long long* __restrict__ cursor, long long out_cap, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long long off = (*cursor + 15) & ~15ll;
    for (int j = 0; j < n; ++j) {
        long long l = 0;
        for (int b = 0; b < RLE_NB; ++b) l += ctr[rle_ctr_blk(n, j, b)];
        if (off + ((l + 15) & ~15ll) <= out_cap) off += (l + 15) & ~15ll;
    }
    *cursor = off;
}

}  // namespace

size_t rle_scratch_bytes(int n, int h, int w) {
    const size_t yw = (size_t)(h + 31) / 32;
    const size_t bits = (((size_t)n * w * yw * 4) + 255) & ~(size_t)255;
    const size_t pos = (((size_t)n * ((size_t)h * w + 2) * 4) + 255) & ~(size_t)255;
    const size_t ctr = (size_t)n * (1 + RLE_CC_MAX + RLE_NB) * 4 + 256;
    return bits + pos + ctr;
}

hipError_t launch_rle_encode(const uint8_t* masks, int n, int h, int w, void* scratch, unsigned char* out, long long out_cap,
                             long long* cursor, long long* table, hipStream_t s) {
    if (n < 1 || h < 1 || w < 1 || (size_t)h * w >= (1ull << 30)) return hipErrorInvalidValue;     // int32 delta arithmetic
    const int chunks = (w + RLE_COLS - 1) / RLE_COLS;
    if (chunks > RLE_CC_MAX) return hipErrorInvalidValue;
    const int YW = (h + 31) / 32;
    unsigned char* p = reinterpret_cast<unsigned char*>(scratch);
    uint32_t* bits = reinterpret_cast<uint32_t*>(p);
    p += (((size_t)n * w * YW * 4) + 255) & ~(size_t)255;
    uint32_t* pos = reinterpret_cast<uint32_t*>(p);
    const size_t pos_stride = (size_t)h * w + 2;
    p += (((size_t)n * pos_stride * 4) + 255) & ~(size_t)255;
    uint32_t* ctr = reinterpret_cast<uint32_t*>(p);
    const uint32_t HW = (uint32_t)((size_t)h * w);
    rle_bitpack_kernel<<<dim3((h + 127) / 128, n, (w + 1023) / 1024), 256, 0, s>>>(masks, bits, h, w, YW);
    rle_positions_kernel<false><<<dim3(chunks, n), RLE_COLS, 0, s>>>(bits, pos, ctr, n, h, w, YW, pos_stride);
    rle_positions_kernel<true><<<dim3(chunks, n), RLE_COLS, 0, s>>>(bits, pos, ctr, n, h, w, YW, pos_stride);
    rle_chars_kernel<false><<<dim3(RLE_NB, n), RLE_THREADS, 0, s>>>(pos, ctr, n, HW, pos_stride, out, out_cap, cursor, table);
    rle_chars_kernel<true><<<dim3(RLE_NB, n), RLE_THREADS, 0, s>>>(pos, ctr, n, HW, pos_stride, out, out_cap, cursor, table);
    rle_cursor_kernel<<<1, 64, 0, s>>>(ctr, cursor, out_cap, n);
    return hipGetLastError();
}
