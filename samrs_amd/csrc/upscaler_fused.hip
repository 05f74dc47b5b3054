// upscaler_fused.hip -- the mask decoder's output upscaling + hypernetwork product in ONE kernel (gfx950).
//
// Reference: Generate Dataset/segment_anything/modeling/mask_decoder.py:53-59 (output_upscaling = ConvTranspose2d(256, 64, 2, 2)
// -> LayerNorm2d(64) -> GELU -> ConvTranspose2d(64, 32, 2, 2) -> GELU) and :154-167 (masks = hyper_in @ upscaled).
//
// A 2x2 / stride-2 transposed conv is a per-token GEMM: token (256 channels) -> 4 sub-pixels x 64 channels, then every
// sub-pixel (64 channels) -> 4 sub-sub-pixels x 32 channels, and the mask logit of each of the 16 output pixels of a token is
// a 32-channel dot product with the prompt's hypernetwork vector.  As separate kernels the intermediate [rows][256] tensor
// makes a round trip through HBM (134 MB written + read per 32-prompt predict, twice that in split precision); here it
// never leaves the CU:
//   block = 8 waves, tile = 16 tokens.  Wave (s1, p): s1 = sub-pixel 1 (its 64 output columns of ConvT #1), p = which half
//   of K (GEMM 1) / which two sub-pixels 2 (GEMM 2) it computes.
//   GEMM 1: W1 fragments of the wave stay in registers for the whole launch (persistent blocks), the 16 x 256 key tile comes
//           through LDS (XOR-swizzled chunks, conflict-free fragment reads); the two K halves of a column group meet in LDS.
//   LayerNorm2d + GELU on the accumulators (a row's 64 values sit in the 4 lane quarters: two shuffles).
//   GEMM 2: the GELU output in the ACCUMULATOR layout of GEMM 1 (lane = token, 4 consecutive channels per n-tile) already is
//           a valid B-operand fragment of GEMM 2 up to a permutation of k -- and a sum over k does not care which physical
//           channel sits in which slot as long as both operands agree.  So W2 is stored in LDS in that permuted order once
//           per block and the activations never move.
//   product: GELU, dot with the 32 hypernetwork weights on the accumulators, lane-quarter reduction, the 4 x 64 output
//           pixels of the tile leave through an LDS tile as whole 256-byte rows.
// SPLIT: keys and both weight matrices as hi + lo (common.h split2_pack), three MFMAs per step, the GELU output of ConvT #1
// split in registers: the upscaler's operand rounding was 376 + 395 of the 899 class-map pixels the round-2 engine lost at
// ViT-H (oracle/error_budget.py "only dec.up1" / "only dec.up2").
#include "common.h"
#include "kernels.h"

namespace {

constexpr int UF_THREADS = 512;
constexpr int UF_ROWS = 16;                    // tokens per tile

struct UFArgs {
    const uint16_t *keys, *keys_lo;            // [rows][256] ET; lo = split remainder (SPLIT only)
    const uint16_t *w1, *w1_lo;                // [256 = (s1, c)][256] ET
    const float* b1;                           // [256]
    const float* ln;                           // gamma[64] | beta[64]
    const uint16_t *w2, *w2_lo;                // [128 = (s2, c2)][64] ET
    const float* b2;                           // [128]
    const float* hyper;                        // [n][n_mask_tokens][32]
    float* low;                                // [n][NSEL][4 grid][4 grid]
    int tokens, grid, n_mask_tokens, sel0, n_tiles;
};

template <int PREC, int NSEL, bool SPLIT>
__global__ __launch_bounds__(UF_THREADS) void upscaler_fused_kernel(UFArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char uf_lds[];
    // LDS map
    unsigned char* At = uf_lds;                                   // [2 (hi, lo)][16 rows][32 chunks] x 16 B = 16 KB
    uint4* W2f = reinterpret_cast<uint4*>(uf_lds + 16384);        // [2 (hi, lo)][8 i2][2 ks][64 lanes] x 16 B = 32 KB
    float* Part = reinterpret_cast<float*>(uf_lds + 16384 + 32768);                 // [4 s1][2 p][16][64] floats = 32 KB
    float* Out = Part + 4 * 2 * 16 * 64;                                              // [NSEL][4][64] floats
    float* cst = Out + 3 * 4 * 64;                                                    // b1[256] | gamma[64] | beta[64] | b2[128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s1 = wave >> 1, p = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- once per block: resident W1 fragments, permuted W2 image, constants ----
    uint4 wf1[4][4], wl1[SPLIT ? 4 : 1][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const size_t o = (size_t)(s1 * 64 + i * 16 + fr) * 256 + (4 * p + ks) * 32 + fq * 8;
            wf1[i][ks] = *reinterpret_cast<const uint4*>(a.w1 + o);
            if constexpr (SPLIT) wl1[i][ks] = *reinterpret_cast<const uint4*>(a.w1_lo + o);
        }
    for (int idx = tid; idx < 8 * 2 * 64; idx += UF_THREADS) {
        // slot (fq', j) of k-step ks2 <-> channel 16 (2 ks2 + (j >> 2)) + 4 fq' + (j & 3): the accumulator layout of GEMM 1
        const int i2 = idx >> 7, ks2 = (idx >> 6) & 1, l = idx & 63, r_ = l & 15, q_ = l >> 4;
        const size_t o0 = (size_t)(i2 * 16 + r_) * 64 + 16 * (2 * ks2) + 4 * q_, o1 = o0 + 16;
        const uint2 x0 = *reinterpret_cast<const uint2*>(a.w2 + o0), x1 = *reinterpret_cast<const uint2*>(a.w2 + o1);
        W2f[idx] = make_uint4(x0.x, x0.y, x1.x, x1.y);
        if constexpr (SPLIT) {
            const uint2 y0 = *reinterpret_cast<const uint2*>(a.w2_lo + o0), y1 = *reinterpret_cast<const uint2*>(a.w2_lo + o1);
            W2f[8 * 2 * 64 + idx] = make_uint4(y0.x, y0.y, y1.x, y1.y);
        }
    }
    for (int i = tid; i < 512; i += UF_THREADS)
        cst[i] = i < 256 ? a.b1[i] : (i < 384 ? a.ln[i - 256] : a.b2[i - 384]);

    const int tiles_per_prompt = a.tokens / UF_ROWS;
    const int S = 4 * a.grid;
    // this thread's piece of a key tile: row tid >> 5, 16-byte chunk tid & 31 (hi and lo)
    const int prow = tid >> 5, pch = tid & 31;
    int tile = blockIdx.x;
    uint4 nh = make_uint4(0u, 0u, 0u, 0u), nl = nh;
    if (tile < a.n_tiles) {
        const size_t o = ((size_t)tile * UF_ROWS + prow) * 256 + pch * 8;
        nh = *reinterpret_cast<const uint4*>(a.keys + o);
        if constexpr (SPLIT) nl = *reinterpret_cast<const uint4*>(a.keys_lo + o);
    }
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_prompt, t0 = (tile - b * tiles_per_prompt) * UF_ROWS;
        // ---- key tile -> LDS (chunk c of row r at position c ^ r: fragment reads of 16 rows are conflict-free) ----
        *reinterpret_cast<uint4*>(At + ((prow * 32 + (pch ^ prow)) << 4)) = nh;
        if constexpr (SPLIT) *reinterpret_cast<uint4*>(At + 8192 + ((prow * 32 + (pch ^ prow)) << 4)) = nl;
        __syncthreads();                                                                   // B1
        {
            const int nt = tile + (int)gridDim.x;
            if (nt < a.n_tiles) {                  // the next tile's pieces fly during this tile's math
                const size_t o = ((size_t)nt * UF_ROWS + prow) * 256 + pch * 8;
                nh = *reinterpret_cast<const uint4*>(a.keys + o);
                if constexpr (SPLIT) nl = *reinterpret_cast<const uint4*>(a.keys_lo + o);
            }
        }
        // hypernetwork weights of this prompt, this lane's channels c2 = (i2 & 1) * 16 + 4 fq + r
        float hy[NSEL][8];
#pragma unroll
        for (int c = 0; c < NSEL; ++c) {
            const float* h = a.hyper + ((size_t)b * a.n_mask_tokens + a.sel0 + c) * 32;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const float4 t = *reinterpret_cast<const float4*>(h + half * 16 + 4 * fq);
                hy[c][half * 4 + 0] = t.x; hy[c][half * 4 + 1] = t.y; hy[c][half * 4 + 2] = t.z; hy[c][half * 4 + 3] = t.w;
            }
        }
        // ---- GEMM 1: this wave's 64 columns x 16 tokens over its half of K ----
        f32x4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int pos = (fr * 32 + (((4 * p + ks) * 4 + fq) ^ fr)) << 4;
            const uint4 af = *reinterpret_cast<const uint4*>(At + pos);
            if constexpr (SPLIT) {
                const uint4 al = *reinterpret_cast<const uint4*>(At + 8192 + pos);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = ET<PREC>::mfma16(wf1[i][ks], al, acc[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = ET<PREC>::mfma16(wl1[i][ks], af, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = ET<PREC>::mfma16(wf1[i][ks], af, acc[i]);
        }
        float* mine = Part + ((s1 * 2 + p) * 16) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(i * 4 + r) * 64] = acc[i][r];
        __syncthreads();                                                                   // B2
        // both waves of a column group add the two K halves in the same order -> identical values
        float v[4][4];
        float s = 0.f;
        {
            const float* p0 = Part + ((s1 * 2 + 0) * 16) * 64 + lane;
            const float* p1 = Part + ((s1 * 2 + 1) * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 bb = *reinterpret_cast<const float4*>(cst + s1 * 64 + i * 16 + 4 * fq);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[i][r] = (p0[(i * 4 + r) * 64] + p1[(i * 4 + r) * 64]) + bv[r];
                    s += v[i][r];
                }
            }
        }
        // ---- LayerNorm2d over the 64 channels of (token fr, sub-pixel s1) + GELU ----
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[i][r] -= mean; q += v[i][r] * v[i][r]; }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-6f);
        uint4 bh[2], bl[2];
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = 2 * ks2 + k;
                const float4 gm = *reinterpret_cast<const float4*>(cst + 256 + i * 16 + 4 * fq);
                const float4 bt = *reinterpret_cast<const float4*>(cst + 320 + i * 16 + 4 * fq);
                const float2_t g01 = gelu_erf2(float2_t{v[i][0] * rstd * gm.x + bt.x, v[i][1] * rstd * gm.y + bt.y});
                const float2_t g23 = gelu_erf2(float2_t{v[i][2] * rstd * gm.z + bt.z, v[i][3] * rstd * gm.w + bt.w});
                split2_pack<PREC>(g01.x, g01.y, hh[2 * k], ll[2 * k]);
                split2_pack<PREC>(g23.x, g23.y, hh[2 * k + 1], ll[2 * k + 1]);
            }
            bh[ks2] = make_uint4(hh[0], hh[1], hh[2], hh[3]);
            bl[ks2] = make_uint4(ll[0], ll[1], ll[2], ll[3]);
        }
        // ---- GEMM 2 (this wave: sub-pixels 2 = 2p, 2p + 1), GELU, hypernetwork dot ----
        float part[NSEL][2];
#pragma unroll
        for (int c = 0; c < NSEL; ++c) part[c][0] = part[c][1] = 0.f;
#pragma unroll
        for (int il = 0; il < 4; ++il) {
            const int i2 = 4 * p + il;
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            const uint4 w0 = W2f[(i2 * 2 + 0) * 64 + lane], w1 = W2f[(i2 * 2 + 1) * 64 + lane];
            if constexpr (SPLIT) {
                const uint4 x0 = W2f[8 * 2 * 64 + (i2 * 2 + 0) * 64 + lane], x1 = W2f[8 * 2 * 64 + (i2 * 2 + 1) * 64 + lane];
                d = ET<PREC>::mfma16(w0, bl[0], d);
                d = ET<PREC>::mfma16(w1, bl[1], d);
                d = ET<PREC>::mfma16(x0, bh[0], d);
                d = ET<PREC>::mfma16(x1, bh[1], d);
            }
            d = ET<PREC>::mfma16(w0, bh[0], d);
            d = ET<PREC>::mfma16(w1, bh[1], d);
            const float4 bb = *reinterpret_cast<const float4*>(cst + 384 + i2 * 16 + 4 * fq);
            const float2_t g01 = gelu_erf2(float2_t{d[0] + bb.x, d[1] + bb.y});
            const float2_t g23 = gelu_erf2(float2_t{d[2] + bb.z, d[3] + bb.w});
#pragma unroll
            for (int c = 0; c < NSEL; ++c) {
                const float* h = hy[c] + (il & 1) * 4;
                part[c][il >> 1] += h[0] * g01.x + h[1] * g01.y + h[2] * g23.x + h[3] * g23.y;
            }
        }
        // the 4 lane quarters hold partial sums over disjoint channels; quarter q keeps sub-pixel 2 = 2p + q (q = 0, 1)
#pragma unroll
        for (int c = 0; c < NSEL; ++c)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                float x = part[c][sl];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
                if (fq == sl) {
                    const int s2 = 2 * p + sl;
                    Out[(c * 4 + 2 * (s1 >> 1) + (s2 >> 1)) * 64 + 4 * fr + 2 * (s1 & 1) + (s2 & 1)] = x;
                }
            }
        __syncthreads();                                                                   // B3
        for (int idx = tid; idx < NSEL * 256; idx += UF_THREADS) {
            const int c = idx >> 8, yl = (idx >> 6) & 3, xl = idx & 63;
            const int y = t0 / a.grid, x0 = t0 - y * a.grid;
            a.low[(((size_t)b * NSEL + c) * S + 4 * y + yl) * S + 4 * x0 + xl] = Out[(c * 4 + yl) * 64 + xl];
        }
        // Out is rewritten only after the next tile's B2, At after its reads (before B2), Part after the next B1
    }
}

constexpr int UF_LDS = 16384 + 32768 + 32768 + 3 * 4 * 64 * 4 + 512 * 4;

}  // namespace

hipError_t launch_upscaler_fused(int prec, const void* keys, const void* keys_lo, const void* w1, const void* w1_lo, const float* b1,
                                 const float* ln, const void* w2, const void* w2_lo, const float* b2, const float* hyper, float* low,
                                 int n, int grid, int n_mask_tokens, int sel0, int n_sel, hipStream_t s) {
    const bool split = keys_lo && w1_lo && w2_lo;
    if ((keys_lo || w1_lo || w2_lo) && !split) return hipErrorInvalidValue;
    if (n < 1 || grid % UF_ROWS || (n_sel != 1 && n_sel != 3)) return hipErrorInvalidValue;
    static const int n_cu = [] {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        return c > 0 ? c : 256;
    }();
    UFArgs a;
    a.keys = (const uint16_t*)keys; a.keys_lo = (const uint16_t*)keys_lo;
    a.w1 = (const uint16_t*)w1; a.w1_lo = (const uint16_t*)w1_lo; a.b1 = b1; a.ln = ln;
    a.w2 = (const uint16_t*)w2; a.w2_lo = (const uint16_t*)w2_lo; a.b2 = b2; a.hyper = hyper; a.low = low;
    a.tokens = grid * grid; a.grid = grid; a.n_mask_tokens = n_mask_tokens; a.sel0 = sel0;
    a.n_tiles = n * (a.tokens / UF_ROWS);
    const int blocks = a.n_tiles < n_cu ? a.n_tiles : n_cu;
#define UF_LAUNCH(P, NS, SP)                                                                                                 \
    do {                                                                                                                     \
        auto k = upscaler_fused_kernel<P, NS, SP>;                                                                           \
        HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, UF_LDS)); \
        k<<<blocks, UF_THREADS, UF_LDS, s>>>(a);                                                                             \
    } while (0)
#define UF_SEL(P, SP) do { if (n_sel == 1) UF_LAUNCH(P, 1, SP); else UF_LAUNCH(P, 3, SP); } while (0)
    if (prec == PREC_BF16) { if (split) UF_SEL(PREC_BF16, true); else UF_SEL(PREC_BF16, false); }
    else if (prec == PREC_F16) { if (split) UF_SEL(PREC_F16, true); else UF_SEL(PREC_F16, false); }
    else return hipErrorInvalidValue;
#undef UF_SEL
#undef UF_LAUNCH
    return hipGetLastError();
}
