// encoder_kernels.hip -- non-GEMM kernels of the ViT image encoder (gfx950).
//
// Reference semantics (paths under Generate Dataset/segment_anything/):
//   preprocess + patch im2col : modeling/sam.py:164-174, modeling/image_encoder.py:364-395
//   LayerNorm (+window gather): modeling/image_encoder.py:168-172,243-264 (zero pad AFTER norm1)
//   windowed attention        : modeling/image_encoder.py:224-240,325-361,267-289
//   global attention          : same with window_size == 0
//   neck im2col / transposes  : modeling/image_encoder.py:88-104,114
//
// Attention formulation (both kernels).  Everything is computed TRANSPOSED so that one lane owns
// one query: S^T = K * Q^T with v_mfma_f32_32x32x16 (first operand = 32 keys x 16 d, second =
// 16 d x 32 queries).  In the 32x32 accumulator layout lane l holds column q = l & 31 and rows
// key = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15, so the softmax row-reduction is in-lane plus one
// xor-32 shuffle.  The un-normalised probabilities feed the PV MFMA (O^T = V^T * P^T) straight
// from registers: MFMA sums over k in any order, so the k-slot <-> key assignment is chosen to be
// exactly the accumulator layout (slot (half, jj) of MFMA u of tile t <-> key
// 32t + 16u + 8*(jj>>2) + 4*half + (jj&3)); V^T fragments are gathered with the same assignment
// (two 8-byte LDS reads).  No permutes, no P round-trip through LDS.
//
// Decomposed rel-pos (image_encoder.py:325-361) uses the UNSCALED q: T = Q * [rel_h; rel_w]^T is
// computed with the same MFMA (table rows play the role of keys), bounced through a per-wave LDS
// scratch [q][row] and gathered as bias(q, k) = T_h[q][qh - kh + S-1] + T_w[q][qw - kw + S-1].
#include "common.h"
#include "kernels.h"

namespace {

__constant__ float c_mean[3] = {123.675f, 116.28f, 103.53f};  // modeling/sam.py:27
__constant__ float c_std[3] = {58.395f, 57.12f, 57.375f};     // modeling/sam.py:28

// -----------------------------------------------------------------------------------------
// preprocess + im2col for the 16x16/16 patch conv.  A[row = (img, py, px)][k = c*256 + ky*16 + kx]
// One thread = 8 horizontally adjacent pixels (24 bytes of HWC input) -> three 16-byte chunks.
// -----------------------------------------------------------------------------------------
template <int PREC>
__global__ void patch_im2col_kernel(const uint8_t* __restrict__ img, uint16_t* __restrict__ A, uint16_t* __restrict__ A_lo,
                                    int n_images, int in_h, int in_w, int grid, int patch) {
    const int P2 = patch * patch;            // 256
    const int halves = patch / 8;            // 2
    const long total = (long)n_images * grid * grid * patch * halves;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int half = t % halves;
    long u = t / halves;
    const int ky = u % patch; u /= patch;
    const int px = u % grid; u /= grid;
    const int py = u % grid;
    const int im = u / grid;
    const int y = py * patch + ky, x0 = px * patch + half * 8;
    float v[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int x = x0 + e;
        const bool in = (y < in_h) && (x < in_w);
        const uint8_t* p = img + ((size_t)im * in_h * in_w + (size_t)y * in_w + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][e] = in ? ((float)p[c] - c_mean[c]) / c_std[c] : 0.f;
    }
    const size_t row = ((size_t)im * grid + py) * grid + px;
    const size_t off = row * (3 * P2) + ky * patch + half * 8;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint4 o, l;
        split2_pack<PREC>(v[c][0], v[c][1], o.x, l.x);
        split2_pack<PREC>(v[c][2], v[c][3], o.y, l.y);
        split2_pack<PREC>(v[c][4], v[c][5], o.z, l.z);
        split2_pack<PREC>(v[c][6], v[c][7], o.w, l.w);
        *reinterpret_cast<uint4*>(A + off + c * P2) = o;
        if (A_lo) *reinterpret_cast<uint4*>(A_lo + off + c * P2) = l;
    }
}

template <int PREC>
__global__ void convert_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, uint16_t* __restrict__ out_lo, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    uint2 o, l;
    split2_pack<PREC>(v.x, v.y, o.x, l.x);
    split2_pack<PREC>(v.z, v.w, o.y, l.y);
    reinterpret_cast<uint2*>(out)[i] = o;
    if (out_lo) reinterpret_cast<uint2*>(out_lo)[i] = l;       // the split remainder (weights / neck operand)
}

// -----------------------------------------------------------------------------------------
// LayerNorm over the last dim, one wave per OUTPUT row.  window_mode: output rows are in window
// order [(img, wy, wx), (iy, ix)] and rows that fall in the bottom/right padding are ZERO (the
// reference pads after norm1, image_encoder.py:168-172,256-259).
// -----------------------------------------------------------------------------------------
// Measured alternatives (tools/ln_bench.py, 32768 x 1280, input cold): this kernel 51.6 us = 4.9 TB/s; non-temporal
// loads 45.5 us; two / four rows per wave 50 / 57 us.  In the tile loop the non-temporal variant LOSES 4 % of the whole
// step (59.3 vs 57.0 ms, three alternations on one box): the row it reads was written by the GEMM just before and is
// still partly in L2 / MALL, which a streaming load does not use.  Plain loads stay.
constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 2048

// A/B build switches (tools/ab_env.sh with SAMRS_LIB_PATH): all on in the product build
#ifndef SAMRS_WIN_ONES
#define SAMRS_WIN_ONES 1
#endif
#ifndef SAMRS_GLB_ONES
#define SAMRS_GLB_ONES 1
#endif
// A/B switch: 1 = the two wave groups of global_attention_kernel a third of a tile apart (group 1 takes its block barrier between its
// softmax and its PV product; V^T triple-buffered).  Built in round 4 on the theory that the phase-locked waves of a SIMD want the
// matrix pipe and the VALU at the same times; measured on MI355X (tools/attn_bench.py, alternating libraries): 1217 - 1233 us against
// 1182 - 1208 us for the phase-locked schedule -- no gain, so the default stays 0 (profiles/r04_attention_lds.txt).
#ifndef SAMRS_GLB_SKEW
#define SAMRS_GLB_SKEW 0
#endif

template <int PREC>
__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, uint16_t* __restrict__ out_et, float* __restrict__ out_f32, int rows_out, int D,
    int window_mode, int grid, int window, uint16_t* __restrict__ out_lo /* optional: the split remainder of out_et */,
    MxOut mx /* optional (plain row order only): hi and lo of the output as MXFP4 codes + scale tiles, gemm_et_mx_kernel's A operands */,
    int ld_out /* row stride of out_et in elements (>= D; gemm.hip tl_gemm_ld); out_lo / out_f32 / mx rows stay dense */,
    const int* __restrict__ oc_idx /* optional: n_oc <= 32 OUTLIER columns of the output (engine.hip outlier_columns) */, int n_oc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_out) return;
    long src = row;
    bool valid = true;
    if (window_mode) {
        const int nw = (grid + window - 1) / window;
        const int w2 = window * window;
        int t = row % w2;
        int win = (row / w2) % (nw * nw);
        int im = row / (w2 * nw * nw);
        const int y = (win / nw) * window + t / window;
        const int x = (win % nw) * window + t % window;
        valid = (y < grid) && (x < grid);
        src = ((long)im * grid + y) * grid + x;
    }
    const int nv = D >> 2;
    if (!valid) {
        for (int i = lane; i < nv; i += 64) {
            if (out_et) reinterpret_cast<uint2*>(out_et + (size_t)row * ld_out)[i] = make_uint2(0u, 0u);
            if (out_lo) reinterpret_cast<uint2*>(out_lo + (size_t)row * D)[i] = make_uint2(0u, 0u);
            if (out_f32) reinterpret_cast<float4*>(out_f32 + (size_t)row * D)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float4* xr = reinterpret_cast<const float4*>(X + (size_t)src * D);
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            v[i] = xr[idx];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[idx];
            const float4 b = reinterpret_cast<const float4*>(beta)[idx];
            const float o0 = (v[i].x - mean) * rstd * g.x + b.x;
            const float o1 = (v[i].y - mean) * rstd * g.y + b.y;
            const float o2 = (v[i].z - mean) * rstd * g.z + b.z;
            const float o3 = (v[i].w - mean) * rstd * g.w + b.w;
            if (out_et) {
                uint2 o;
                o.x = pack2<PREC>(o0, o1);
                o.y = pack2<PREC>(o2, o3);
                reinterpret_cast<uint2*>(out_et + (size_t)row * ld_out)[idx] = o;
                if (out_lo) {          // neck only: the remainder of the split (same hi bits as above)
                    uint2 h, l;
                    split2_pack<PREC>(o0, o1, h.x, l.x);
                    split2_pack<PREC>(o2, o3, h.y, l.y);
                    reinterpret_cast<uint2*>(out_lo + (size_t)row * D)[idx] = l;
                }
                if (mx.q_hi) {
                    // a lane holds elements 4 idx .. 4 idx + 3: an MX block (32 elements) = 8 consecutive lanes of one pass.
                    // hi = the ET value just written, lo = the fp32 remainder; one E8M0 scale per block and tensor.
                    float h[4], l[4];
                    const float ov[4] = {o0, o1, o2, o3};
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        h[e2] = ET<PREC>::to_float((uint16_t)(((e2 & 2) ? o.y : o.x) >> (16 * (e2 & 1))));
                        l[e2] = ov[e2] - h[e2];
                    }
                    float ah = fmaxf(fmaxf(fabsf(h[0]), fabsf(h[1])), fmaxf(fabsf(h[2]), fabsf(h[3])));
                    float al = fmaxf(fmaxf(fabsf(l[0]), fabsf(l[1])), fmaxf(fabsf(l[2]), fabsf(l[3])));
#pragma unroll
                    for (int sh = 1; sh < 8; sh <<= 1) { ah = fmaxf(ah, __shfl_xor(ah, sh, 64)); al = fmaxf(al, __shfl_xor(al, sh, 64)); }
                    const int bh = mx_scale_byte(ah), bl = mx_scale_byte(al);
                    const uint32_t ch = fp4_pack4(h[0], h[1], h[2], h[3], bh), cl = fp4_pack4(l[0], l[1], l[2], l[3], bl);
                    const size_t dst = (size_t)row * (D >> 1) + (size_t)idx * 2;
                    *reinterpret_cast<uint16_t*>(mx.q_hi + dst) = (uint16_t)ch;
                    *reinterpret_cast<uint16_t*>(mx.q_lo + dst) = (uint16_t)cl;
                    if ((lane & 7) == 0) {
                        const size_t si = mx_scale_index(false, row, idx >> 3, D / MXK);
                        mx.s_hi[si] = (unsigned char)bh;
                        mx.s_lo[si] = (unsigned char)bl;
                    }
                }
            }
            if (out_f32) reinterpret_cast<float4*>(out_f32 + (size_t)row * D)[idx] = make_float4(o0, o1, o2, o3);
        }
    }
    // Outlier columns (oracle/outlier_budget.py): a checkpoint's LayerNorm output has a few channels that run
    // 10 - 100x hotter than the rest (large gamma), and an element's f16 rounding error is relative to ITS magnitude -- those few
    // columns carry most of the operand error of the qkv / lin1 products.  Their hi + lo split rides as 64 more K columns of the SAME
    // GEMM launch: row[D + j] = lo of column oc_idx[j] (meets W_hi[:, oc_idx[j]] in the weight's extension), row[D + 32 + j] = its hi
    // (meets W_lo[:, oc_idx[j]]); unused slots are zero on both sides.
    if (n_oc > 0 && out_et && lane < 32) {
        uint16_t hi = 0, lo = 0;
        if (lane < n_oc) {
            const int c = oc_idx[lane];
            const float o = (X[(size_t)src * D + c] - mean) * rstd * gamma[c] + beta[c];
            hi = ET<PREC>::from_float(o);
            lo = ET<PREC>::from_float(o - ET<PREC>::to_float(hi));
        }
        out_et[(size_t)row * ld_out + D + lane] = lo;
        out_et[(size_t)row * ld_out + D + 32 + lane] = hi;
    }
    // the rest of a padded row behind the extension: zero, so that a reader which walks whole rows (the side GEMM of lin2's
    // outlier columns, engine.hip) meets nothing stale there
    if (n_oc > 0 && out_et)
        for (int c = D + 64 + lane; c < ld_out; c += 64) out_et[(size_t)row * ld_out + c] = 0;
}

// Weight side of the outlier-column extension: ext[r][j] = ET(W[r][idx[j]]) (hi: meets the operand's lo), ext[r][32 + j] = ET(W - hi)
// (lo: meets the operand's hi), zeros in unused slots; written at column K of a row of stride ld (K + 64 <= ld).
template <int PREC>
__global__ __launch_bounds__(256) void outlier_weight_ext_kernel(const float* __restrict__ W, int N, int K, const int* __restrict__ idx, int n_oc,
                                                                  uint16_t* __restrict__ out, int ld, int col0) {
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
    if (r >= N) return;
    uint16_t hi = 0, lo = 0;
    if (j < n_oc) {
        const float w = W[(size_t)r * K + idx[j]];
        hi = ET<PREC>::from_float(w);
        lo = ET<PREC>::from_float(w - ET<PREC>::to_float(hi));
    }
    out[(size_t)r * ld + col0 + j] = hi;
    out[(size_t)r * ld + col0 + 32 + j] = lo;
}

// Side weights of lin2's outlier columns (hidden units idx2[0 .. n2)): rows j < n2 of out [32][Ks] = row idx2[j] of lin1's weight in the
// operand type, followed -- when lin1 itself carries outlier columns idx1[0 .. n1) -- by the same 64-column extension its own launch
// reads (hi at D + t, lo at D + 32 + t), zeros up to Ks; rows >= n2 are zero.  bias_out [32] = lin1's bias at idx2, zeros behind.
template <int PREC>
__global__ __launch_bounds__(256) void outlier_side_weight_kernel(const float* __restrict__ W1, const float* __restrict__ b1, int D,
                                                                   const int* __restrict__ idx2, int n2, const int* __restrict__ idx1, int n1,
                                                                   uint16_t* __restrict__ out, int Ks, float* __restrict__ bias_out) {
    const int j = blockIdx.x;                         // 0 .. 31
    const float* src = j < n2 ? W1 + (size_t)idx2[j] * D : nullptr;
    for (int k = threadIdx.x; k < Ks; k += 256) {
        uint16_t v = 0;
        if (src) {
            if (k < D) v = ET<PREC>::from_float(src[k]);
            else if (k < D + 64) {
                const int t = (k - D) & 31;
                if (t < n1) {
                    const float w = src[idx1[t]];
                    const uint16_t hi = ET<PREC>::from_float(w);
                    v = (k - D) < 32 ? hi : ET<PREC>::from_float(w - ET<PREC>::to_float(hi));
                }
            }
        }
        out[(size_t)j * Ks + k] = v;
    }
    if (threadIdx.x == 0 && bias_out) bias_out[j] = src ? b1[idx2[j]] : 0.f;
}

// A_x of proj: row r = lo of the attention output's outlier columns | their hi
__global__ __launch_bounds__(256) void outlier_gather_kernel(const uint16_t* __restrict__ hi, const uint16_t* __restrict__ lo, int D,
                                                             const int* __restrict__ idx, int n_oc, uint16_t* __restrict__ out, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (r >= rows) return;
    uint16_t v = 0;
    if ((j & 31) < n_oc) v = (j < 32 ? lo : hi)[(size_t)r * D + idx[j & 31]];
    out[(size_t)r * 64 + j] = v;
}

// A_x of lin2: GELU(lin1) of the <= 32 outlier hidden units BEFORE its rounding, recomputed from the LayerNorm output -- a skinny GEMM
// Y [M][K] (row stride lda) x Ws [32][K]^T + bias, fp32 accumulate on v_mfma_f32_16x16x32, exact-erf GELU, hi + lo split, written as
// lo | hi into out [M][64].  HBM-bound on the one pass over Y (92 MB at 8 tiles); one wave = 16 rows, both operands straight from
// global memory (the 86 KB of side weights stay in the L2 / L1), eight waves per SIMD hide the latency.  M % 64 == 0, K % 32 == 0.
template <int PREC>
__global__ __launch_bounds__(256) void outlier_side_gemm_kernel(const uint16_t* __restrict__ Y, int lda, const uint16_t* __restrict__ Ws,
                                                                 const float* __restrict__ bias, int K, uint16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const size_t row0 = (size_t)blockIdx.x * 64 + wave * 16;
    const uint16_t* a = Y + (row0 + fr) * (size_t)lda + fq * 8;
    const uint16_t* b0 = Ws + (size_t)fr * K + fq * 8;
    const uint16_t* b1 = b0 + (size_t)16 * K;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < K; k += 32) {
        const uint4 fa = *reinterpret_cast<const uint4*>(a + k);
        const uint4 f0 = *reinterpret_cast<const uint4*>(b0 + k);
        const uint4 f1 = *reinterpret_cast<const uint4*>(b1 + k);
        // D = A B: A = 16 rows of Y (lane: row fr, k 8 fq ..), B[k][n] = Ws[n][k] (lane: n = fr, k 8 fq ..); D[m = 4 fq + i][n = fr]
        if (PREC == PREC_F16) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa), __builtin_bit_cast(f16x8_t, f0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa), __builtin_bit_cast(f16x8_t, f1), acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, f0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, f1), acc1, 0, 0, 0);
        }
    }
    const float bs0 = bias[fr], bs1 = bias[16 + fr];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint16_t* o = out + (row0 + 4 * fq + i) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float x = (t ? acc1[i] : acc0[i]) + (t ? bs1 : bs0);
            const float g = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
            const uint16_t hi = ET<PREC>::from_float(g);
            o[16 * t + fr] = ET<PREC>::from_float(g - ET<PREC>::to_float(hi));
            o[32 + 16 * t + fr] = hi;
        }
    }
}

// Squared L2 norms of the columns (col_sq[K]) and of the rows (row_sq[N]) of an fp32 matrix W[N][K]: what the weights-only outlier rule
// scores columns with (engine.hip pick_outlier_columns).  Load-time only, and DETERMINISTIC (no atomics: which columns sit above the threshold
// must not depend on the order in which blocks happen to finish): a thread walks its column top to bottom, a wave sums its row in the
// fixed order of wave_sum.
// 64 columns per block, 16 row phases per column (rows p, p + 16, ...), the 16 partial sums added in phase order: deterministic,
// and 16 x 4 times the threads of the one-thread-per-column form (1.0 ms -> ~0.07 ms per ViT-H weight; 128 launches per engine load)
__global__ __launch_bounds__(1024) void weight_col_norms_kernel(const float* __restrict__ W, int N, int K, float* __restrict__ col_sq) {
    __shared__ float part[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < K)
        for (int r = ph; r < N; r += 16) {
            const float w = W[(size_t)r * K + c];
            acc += w * w;
        }
    part[ph][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ph == 0 && c < K) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][threadIdx.x];
        col_sq[c] = t;
    }
}
__global__ __launch_bounds__(256) void weight_row_norms_kernel(const float* __restrict__ W, int N, int K, float* __restrict__ row_sq) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= N) return;
    float acc = 0.f;
    for (int c = lane; c < K; c += 64) {
        const float w = W[(size_t)r * K + c];
        acc += w * w;
    }
    acc = wave_sum(acc);
    if (lane == 0) row_sq[r] = acc;
}

// -----------------------------------------------------------------------------------------
// shared attention pieces
// -----------------------------------------------------------------------------------------
#define LOG2E_F 1.4426950408889634f

// key (row) index of accumulator register r for lane-half hh within a 32x32 tile
__device__ __forceinline__ constexpr int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// Q fragments straight from global: lane (q = l&31, hh = l>>5) holds Q[q][16*ks + 8*hh .. +7].
template <int KS>
__device__ __forceinline__ void load_q_frags(const uint16_t* qrow /* &Q[q][0] or nullptr */, int hh,
                                             uint4 (&qf)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = qrow ? *reinterpret_cast<const uint4*>(qrow + 16 * ks + 8 * hh) : make_uint4(0u, 0u, 0u, 0u);
    }
}

// One 32-row tile of  M * Q^T : rows come from an LDS matrix [rows][STR] (ET, row stride STR elements, HD of them used).
// STR = HD + 8 (an ODD number of 16-byte chunks per row) makes the ds_read_b128 fragment reads conflict-free: the 16 lanes
// of a lane group read one chunk column of 16 rows that are distinct mod 16, and (row * odd + c) mod 16 is then a permutation
// of the sixteen 16-byte slots of the bank row.  With STR = HD (an even chunk count: 10 for HD = 80, 8 for HD = 64) only
// every second slot is reachable: 2-way conflicts at HD = 80 (PMC round 3: half of the LDS-active cycles of the global
// kernel), 8-way at HD = 64.
template <int PREC, int HD, int STR = HD>
__device__ __forceinline__ f32x16_t tile_times_qT(const uint16_t* lds_rows /* row 0 of the tile */,
                                                  int lane, const uint4 (&qf)[HD / 16]) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const uint16_t* p = lds_rows + (lane & 31) * STR + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(p + 16 * ks);
        acc = ET<PREC>::mfma32(a, qf[ks], acc);
    }
    return acc;
}

// Same, rows = rows [row0, row0+32) of an fp32 rel-pos table in GLOBAL memory (rows >= nrows are
// zero): the table is only touched once per block, so it is not worth an LDS copy.
template <int PREC, int HD>
__device__ __forceinline__ f32x16_t table_times_qT(const float* __restrict__ tab, int row0, int nrows, int lane,
                                                   const uint4 (&qf)[HD / 16]) {
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row = row0 + (lane & 31);
    const bool ok = row < nrows;
    const float* p = tab + (size_t)(ok ? row : 0) * HD + 8 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
        if (ok) {
            x = *reinterpret_cast<const float4*>(p + 16 * ks);
            y = *reinterpret_cast<const float4*>(p + 16 * ks + 4);
        }
        uint4 a;
        a.x = pack2<PREC>(x.x, x.y); a.y = pack2<PREC>(x.z, x.w);
        a.z = pack2<PREC>(y.x, y.y); a.w = pack2<PREC>(y.z, y.w);
        acc = ET<PREC>::mfma32(a, qf[ks], acc);
    }
    return acc;
}

// V^T fragment for PV MFMA `u` of a 32-key tile: lane (dd = l&31, hh) holds keys
// base + 4hh + {0..3} and base + 8 + 4hh + {0..3}, base = 16u, of row d = dd.
__device__ __forceinline__ uint4 load_vt_frag(const uint16_t* vt_row /* &Vt[d][tile key 0] */, int u, int hh) {
    const uint2 lo = *reinterpret_cast<const uint2*>(vt_row + 16 * u + 4 * hh);
    const uint2 hi = *reinterpret_cast<const uint2*>(vt_row + 16 * u + 8 + 4 * hh);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// The same fragment from a V^T tile whose keys are stored PERMUTED inside every group of 16 (vt_pack_kernel: position p holds
// key (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1), i.e. 0-3, 8-11, 4-7, 12-15): the eight keys of lane half hh are then 16
// contiguous bytes -- one ds_read_b128 instead of two ds_read_b64, and with a row stride of an odd number of 16-byte chunks
// the 16 rows of a lane group fall on 16 different slots (no bank conflict; two 8-byte reads of one half could never be
// better than 2-way: 32 lanes on the 16 eight-byte positions of one parity).
__device__ __forceinline__ uint4 load_vt_frag_perm(const uint16_t* vt_row /* &Vt[d][tile position 0] */, int u, int hh) {
    return *reinterpret_cast<const uint4*>(vt_row + 16 * u + 8 * hh);
}

template <int PREC>
__device__ __forceinline__ uint4 pack_p(const f32x16_t& p, int u) {   // probabilities in [0,1]: no saturation
    uint4 o;
    o.x = pack2_fast<PREC>(p[8 * u + 0], p[8 * u + 1]);
    o.y = pack2_fast<PREC>(p[8 * u + 2], p[8 * u + 3]);
    o.z = pack2_fast<PREC>(p[8 * u + 4], p[8 * u + 5]);
    o.w = pack2_fast<PREC>(p[8 * u + 6], p[8 * u + 7]);
    return o;
}

// word w (0..3) of a uint4
__device__ __forceinline__ uint32_t word_of(const uint4& v, int w) { return w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w; }
// V^T staging: element e (0..7) of two adjacent keys' 16-byte d-chunks -> one word {v0[e], v1[e]}
__device__ __forceinline__ uint32_t pair_elem(const uint4& v0, const uint4& v1, int e) {
    return __builtin_amdgcn_perm(word_of(v1, e >> 1), word_of(v0, e >> 1), (e & 1) ? 0x07060302u : 0x05040100u);
}

__device__ __forceinline__ uint4 sel4(bool c, const uint4& a, const uint4& b) {
    return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one online-softmax step on a 32-key S^T tile held in `S` (already = log2-domain logits WITHOUT the
// per-tile constant `bh2`): updates running max / sum, rescales O only when some lane's max moved,
// leaves the probabilities in S.
// Lanes l and l + 32 own the two key halves of one query.  v_permlane32_swap exchanges the halves of two registers
// inside the VALU (no LDS round trip, unlike ds_bpermute): with both operands = v, every lane ends up with its own and
// its partner's value.
__device__ __forceinline__ float xhalf_partner(float v) {
#ifdef SAMRS_XHALF_SHFL      // A/B build: the LDS-based exchange this replaced
    return __shfl_xor(v, 32, 64);
#endif
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

// l_run is a PER-LANE partial row sum (this lane's key half): both lanes of a query scale it by the same alpha, so
// the two halves are only added once, after the last tile (xhalf_partner), instead of once per tile.
// SUM = false: the caller gets the row sum out of the PV product instead (a V^T row of ones, see ones_row_sum).
template <int DT, bool SUM = true>
__device__ __forceinline__ void online_softmax_step(f32x16_t* S, int ntiles, float mx, float bh2, float& m_run, float& l_run,
                                                    f32x16_t (&O)[DT]) {
    mx = fmaxf(mx, xhalf_partner(mx));
    const float m_new = fmaxf(m_run, mx + bh2);
    if (__any(m_new != m_run)) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
    }
    m_run = m_new;
    const float off = m_new - bh2;
    float sum = 0.f;
    for (int a = 0; a < ntiles; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(S[a][r] - off);
            S[a][r] = p;
            if (SUM) sum += p;
        }
    l_run += sum;
}

// When HD is not a multiple of 32 the last d tile of the PV product has spare rows.  With V^T row HD set to ONE, row HD
// of O^T accumulates sum_k P[q][k] of the ROUNDED probabilities -- the weights PV really applies -- through the same
// alpha rescaling as O, and the 16 (32) adds per key tile of an explicit row sum disappear.  Row HD sits in register
// `orr` of the lanes of half `oh`; the other half fetches it from its partner lane.
template <int HD, int DT>
__device__ __forceinline__ float ones_row_sum(const f32x16_t (&O)[DT], int hh) {
    constexpr int dl = HD % 32, oh = (dl >> 2) & 1, orr = (dl & 3) + 4 * (dl >> 3);
    static_assert(dl != 0 && acc_row(orr, oh) == dl, "a spare row is needed");
    const float mine = O[DT - 1][orr], other = xhalf_partner(mine);
    return hh == oh ? mine : other;
}

// Attention output of one query row and head as MXFP4 (LO == 2: the proj GEMM's A operands in "lo_format" 4; gemm.hip
// gemm_et_mx_kernel).  The K axis of that GEMM is padded per head to DT * 32 (80 -> 96), so the DT blocks of this (row, head)
// belong to this lane pair alone: lanes l and l + 32 hold 16 values each of block dt (d = 32 dt + 8 g + 4 hh + e), exchange
// their maxima once per block and tensor, and write their 16 codes as ONE 8-byte store -- the block stores its elements in the
// order p = 16 hh + 4 g + e, the same permutation the engine applies to the proj weights (mx4_pack_kernel PERM; the block scale
// is shared, so any order both operands agree on gives the same product).  hi = the ET value the plain path stores, lo = the
// fp32 remainder.  Rows d >= HD of the last block (incl. the ones-row that carries the softmax sum) are padding: zero.
template <int PREC, int HD, int DT>
__device__ __forceinline__ void store_attention_row_mx(const f32x16_t (&O)[DT], float inv, uint16_t* __restrict__ orow, int hh,
                                                       const MxOut& mx, size_t row, int head, int heads) {
    const int nblk = heads * DT, nst4 = nblk * 32 / MXK;
    unsigned char* qh = mx.q_hi + row * (size_t)(nblk * 16) + (size_t)head * (DT * 16);
    unsigned char* ql = mx.q_lo + row * (size_t)(nblk * 16) + (size_t)head * (DT * 16);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        float h[16], l[16];
        float ah = 0.f, al = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            const bool real = d0 < HD;                       // HD % 8 == 0: a group of four is real or padding as a whole
            uint2 o = make_uint2(0u, 0u);
            if (real) {
                o.x = pack2_fast<PREC>(O[dt][4 * g + 0] * inv, O[dt][4 * g + 1] * inv);
                o.y = pack2_fast<PREC>(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(orow + d0) = o;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float hv = ET<PREC>::to_float((uint16_t)(((e & 2) ? o.y : o.x) >> (16 * (e & 1))));
                h[4 * g + e] = hv;
                l[4 * g + e] = real ? O[dt][4 * g + e] * inv - hv : 0.f;
                ah = fmaxf(ah, fabsf(hv));
                al = fmaxf(al, fabsf(l[4 * g + e]));
            }
        }
        ah = fmaxf(ah, xhalf_partner(ah));
        al = fmaxf(al, xhalf_partner(al));
        const int bh = mx_scale_byte(ah), bl = mx_scale_byte(al);
        const uint2 ch = make_uint2(fp4_pack8(h, bh), fp4_pack8(h + 8, bh)), cl = make_uint2(fp4_pack8(l, bl), fp4_pack8(l + 8, bl));
        *reinterpret_cast<uint2*>(qh + dt * 16 + 8 * hh) = ch;
        *reinterpret_cast<uint2*>(ql + dt * 16 + 8 * hh) = cl;
        if (hh == 0) {
            const size_t si = mx_scale_index(false, (int)row, head * DT + dt, nst4);
            mx.s_hi[si] = (unsigned char)bh;
            mx.s_lo[si] = (unsigned char)bl;
        }
    }
}

// =========================================================================================
// windowed attention.  Work item = (window, head); 8 waves, wave w < 7 owns the 32-query strip w
// (7 strips cover the 196 window tokens), keys are visited tile by tile with an online softmax.
// qkv rows are in plain TOKEN order [img][y][x]; the kernel does the window partition itself.  A
// window position that falls in the bottom/right padding is a zero token after norm1 in the
// reference (image_encoder.py:168-172,256-259), so its k / v are exactly the qkv BIAS: the kernel
// reads them from `qkv_bias` instead of having the GEMM grind through 17.6 % padding rows.  Padding
// tokens are valid keys (not masked); padding queries are dropped (image_encoder.py:287-288).
//
// PERSISTENT: the launch has one block per CU and a block walks items b, b + gridDim.x, ...  The
// kernel was latency-bound (59 % of wave cycles waiting at 2 waves / SIMD, PMC): per item a chain
// global load -> LDS -> barrier -> compute.  Now the Q / K / V loads of item i+1 are issued right
// after item i's K / V^T have been written to LDS and stay in flight (in registers) during the whole
// compute of item i.  vmcnt retires loads in order, so nothing else may load from global memory in
// between: the rel-pos tables are converted into LDS once per block.
//
// Round 2 (s_memtime phase timing, tools/win_timeline.py, profiles/r02_window_attention_phase_timing.txt):
//   * the issue of the next item's 13 loads per lane was 19 % of an item: a branch + an LDS read of the bias + a wait
//     per load (padding positions).  Now every load is issued, from clamped coordinates, and padding slots are
//     replaced when the registers go to LDS (12 %); the loads are further spread over the first five key tiles of the
//     computing waves instead of one burst from all 8 waves into the TA queue (3 %);
//   * a key tile is two window rows (28 keys + 4 zero keys) so the rel-pos bias of accumulator register r does not
//     depend on the tile: S*c2 + BW[r] is one fma, the row term RH[2t + row] merges with the softmax offset into the
//     one add in front of exp2;
//   * HD = 80: the spare V^T row 80 holds ones, the PV product delivers the row sum (ones_row_sum).
// 166 -> 145 us per launch (8 tiles, ViT-H) on its own; in the tile loop the step time did not move (the loop runs at
// the socket power limit).
// Round 4, measured and NOT kept: K rows padded to 88 elements (an odd number of 16-byte chunks: conflict-free fragment
// reads, as in the global kernel).  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.28 -> 0.165, but the padded addressing costs
// registers (the kernel sits at 256: RH had to move to LDS and the K slot coordinates be recomputed per item) and the launch
// went 145.3 -> 154.4 us: the item is bound by its barrier / staging phases, not by LDS cycles (profiles/r04_attention_lds.txt).
// =========================================================================================
template <int HD>
struct WinCfg {
    // A key tile = TWO window rows (28 keys) + 4 zero keys: 7 tiles as with a dense packing, but the rel-pos bias of
    // accumulator register r is then the same function of (r, lane half) in every tile (see the tile loop).
    static constexpr int WS = 14, N = WS * WS, TK = 2 * WS, NT = N / TK, NP = NT * 32;
    static constexpr int DT = (HD + 31) / 32;                          // d tiles for PV
    static constexpr bool ONES = SAMRS_WIN_ONES && DT * 32 > HD;       // a spare V^T row carries the row sum
    __device__ static constexpr int lds_key(int key) { return key + (32 - TK) * (key / TK); }   // key -> K row / V^T column
    static constexpr int VSTR = 228;                                   // V^T row stride (elements)
    static constexpr int NW = 8, THREADS = NW * 64;
    static constexpr int SSTR = 33;                                    // scratch row stride (floats)
    static constexpr int K_BYTES = NP * HD * 2;
    static constexpr int VT_BYTES = DT * 32 * VSTR * 2;
    static constexpr int S_BYTES = NW * 32 * SSTR * 4;
    static constexpr int TAB_BYTES = 2 * 32 * HD * 2;                  // rel_h | rel_w as ET, 32 rows each (27 used)
    static constexpr int MAX_D = 1280;                                 // ViT-H width (heads * HD)
    static constexpr int BIAS_BYTES = 2 * MAX_D * 2;                   // k | v bias of every head as ET
    static constexpr int LDS_BYTES = K_BYTES + VT_BYTES + S_BYTES + TAB_BYTES + BIAS_BYTES;
};

// PIPE (round 5; tools/win_timeline.py per wave, profiles/r05_window_attention_per_wave.txt): of ~17.2 k cycles per item the 7-tile
// loop takes 8.1 k on the four older waves and 10.9 k on the three younger ones that share their SIMDs -- 780 cycles per (wave, tile)
// for 352 cycles of matrix-pipe time: QK^T (five MFMAs chained on one accumulator), the softmax VALU block and PV run strictly one
// after the other inside a wave, and two such waves do not fill each other's gaps.  PIPE = 1 software-pipelines the loop INSIDE the
// wave: the QK^T chain of tile t + 1 is issued in front of the softmax of tile t (its MFMAs execute under that VALU block and under the
// chain's own dependency latency), S of two tiles is live at once; the row terms RH[14] move from registers to the wave's LDS scratch
// (two ds_read_b32 per tile) to pay for the second S tile.  Same products, same accumulation order: bit-identical output.
template <int PREC, int HD, int LO = 0 /* 0: out only; 1: + its f16 split remainder (out_lo); 2: + hi / lo as MXFP4 (mx) */, int PIPE = 0>
__global__ __launch_bounds__(512) void window_attention_kernel(
    const uint16_t* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ rel_h,
    const float* __restrict__ rel_w, uint16_t* __restrict__ out, int grid, int heads, int n_items,
    uint16_t* __restrict__ out_lo = nullptr /* LO == 1: the split remainder of out (reference-grade mode) */, MxOut mx = MxOut(),
    uint32_t lo_heads = 0xffffffffu /* LO == 1: bit h set = head h writes its remainder (the outlier extension of proj needs the
                                       heads that hold its columns only: engine.hip EncBlock::oc_heads) */) {
    using C = WinCfg<HD>;
    constexpr int KS = HD / 16;
    constexpr bool PL = (PIPE & 1) != 0;          // software-pipelined tile loop
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* Ks = reinterpret_cast<uint16_t*>(smem);
    uint16_t* Vt = reinterpret_cast<uint16_t*>(smem + C::K_BYTES);
    float* Scr = reinterpret_cast<float*>(smem + C::K_BYTES + C::VT_BYTES);
    uint16_t* Tab = reinterpret_cast<uint16_t*>(smem + C::K_BYTES + C::VT_BYTES + C::S_BYTES);
    uint16_t* Bia = Tab + 2 * 32 * HD;     // [2][D]: k bias | v bias (the k / v of a padding token)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, ql = lane & 31;
    const int D = heads * HD;
    const int nw = (grid + C::WS - 1) / C::WS;
    const size_t img_rows = (size_t)grid * grid;

    // once per block: rel-pos tables and k / v biases -> LDS as ET (table rows >= 27 zero); zero what the
    // per-item staging never writes: K rows 196..223 (tile padding), all of V^T (its key columns >= 196 and
    // rows >= HD stay zero)
    for (int i = tid; i < 2 * 32 * HD; i += C::THREADS) {
        const int t = i / (32 * HD), r = (i / HD) % 32, d = i % HD;
        const float v = r < 2 * C::WS - 1 ? (t ? rel_w : rel_h)[r * HD + d] : 0.f;
        Tab[i] = ET<PREC>::from_float(v);
    }
    for (int i = tid; i < 2 * D; i += C::THREADS) Bia[i] = ET<PREC>::from_float(qkv_bias[D + i]);
    for (int i = tid; i < C::NT * (32 - C::TK) * HD; i += C::THREADS)      // rows 28..31 of every key tile
        Ks[((i / HD) / (32 - C::TK) * 32 + C::TK + (i / HD) % (32 - C::TK)) * HD + i % HD] = 0;
    for (int i = tid; i < C::DT * 32 * C::VSTR / 2; i += C::THREADS) reinterpret_cast<uint32_t*>(Vt)[i] = 0u;
    if (C::ONES) {                        // V^T row HD = 1: the PV product's spare row accumulates the softmax row sum
        __syncthreads();
        for (int i = tid; i < C::NP; i += C::THREADS) Vt[HD * C::VSTR + i] = ET<PREC>::from_float(1.0f);
    }

    constexpr int CH = HD / 8;  // 16-byte chunks per row
    constexpr int NKC = C::N * CH, NVI = (C::N / 2) * CH;        // real rows / key pairs only
    constexpr int PK = (NKC + C::THREADS - 1) / C::THREADS, PV = (NVI + C::THREADS - 1) / C::THREADS;
    const int q = 32 * (wave < C::NT ? wave : 0) + ql;   // query index inside the window (wave 7 only stages)
    const int qc = q < C::N ? q : C::N - 1;
    const int qh = qc / C::WS, qw = qc % C::WS;

    // All global loads of an item (this wave's Q fragments, this thread's K chunks and V key pairs) are
    // issued back to back; n_* describe the item they belong to.
    // PIPE: the next item's Q fragments are loaded straight into qf during the LAST key tile (QK^T of that tile was issued one
    // tile earlier, so qf is dead by then): no second set of 20 registers (qfn) is held through the tile loop, which is what
    // lets two S tiles be live at once without spilling (spills here reload through scratch = VMEM, and their vmcnt(0) would
    // drain the prefetch that is in flight)
    uint4 qf[KS], qfn[PL ? 1 : KS], kreg[PK], v0reg[PV], v1reg[PV];
    auto& qdst = *[&]() { if constexpr (PL) return &qf; else return &qfn; }();
    int n_qoff = -1;                      // element offset of this lane's query row inside its image, -1 = dropped
    uint32_t n_pad = 0;                   // bit i: K slot i; bit PK + 2i (+1): V slot i key 0 (1) is a padding token
    int n_im = 0, n_head = 0;
    // A padding position still issues its load, from the nearest real token (clamped coordinates), and is replaced by
    // the bias chunk when the registers are written to LDS: the issue sequence is straight-line, 13 loads back to back
    // (the earlier form branched per load and waited on an LDS read of the bias before each one: 19 % of the item,
    // profiles/r02_window_attention_phase_timing.txt).
#define WIN_TOK(r_, off_, pad_)                                                                                      \
    const int ty_##off_ = y0_ + (r_) / C::WS, tx_##off_ = x0_ + (r_) % C::WS;                                        \
    const bool pad_ = ty_##off_ >= grid || tx_##off_ >= grid;                                                        \
    const int off_ = (min(ty_##off_, grid - 1) * grid + min(tx_##off_, grid - 1)) * (3 * D)
    // The issue is split into a head (item decode) and 5 groups of loads: a computing wave spreads the groups over its
    // first key tiles (a burst of 13 loads per lane from all 8 waves at once fills the TA queue and every wave then
    // blocks in issue: 12 % of the item); wave 7, which only stages, issues everything at once.
    const uint16_t* n_base = qkv;
    int n_y0 = 0, n_x0 = 0;
#define WIN_HEAD(it_)                                                                                                \
    do {                                                                                                             \
        const int wi_ = (it_) / heads;                                                                               \
        n_head = (it_) % heads;                                                                                      \
        const int win_ = wi_ % (nw * nw);                                                                            \
        n_im = wi_ / (nw * nw);                                                                                      \
        n_y0 = (win_ / nw) * C::WS;                                                                                  \
        n_x0 = (win_ % nw) * C::WS;                                                                                  \
        n_base = qkv + (size_t)n_im * img_rows * (3 * D) + n_head * HD;                                              \
        n_pad = 0u;                                                                                                  \
    } while (0)
#define WIN_LOAD_Q()                                                                                                 \
    do {                                                                                                             \
        const int y0_ = n_y0, x0_ = n_x0;                                                                            \
        WIN_TOK(qc, qo_, qp_);                                                                                       \
        n_qoff = (q < C::N && !qp_) ? qo_ : -1;                                                                      \
        _Pragma("unroll") for (int ks_ = 0; ks_ < KS; ++ks_)                                                         \
            qdst[ks_] = *reinterpret_cast<const uint4*>(n_base + qo_ + 16 * ks_ + 8 * hh);                           \
    } while (0)
    // Round 5 (the tile loop is bound by instruction ISSUE: profiles/r05_attention_pipelining.txt): which window token a load slot of
    // this thread addresses -- slot -> chunk -> row r -> (r / 14, r % 14) and the chunk inside the row: two to three divisions by
    // constants per load, ~310 integer instructions per item and thread, 24 % of a key tile's instruction stream -- does not depend on
    // the item.  It is decoded ONCE, into 12-bit fields (4 bits window row | 4 bits window column | 4 bits chunk), two slots per
    // register (three registers in all: hoisting the decoded values themselves spills, the kernel sits at 252 VGPRs); per item a slot
    // costs two field extracts instead.  Addresses only: bit-identical output.
    auto slot_field = [](int r, int ch) -> uint32_t { return (uint32_t)(r / C::WS) | ((uint32_t)(r % C::WS) << 4) | ((uint32_t)ch << 8); };
    uint32_t kpk[(PK + 1) / 2], vpk[(PV + 1) / 2];
#pragma unroll
    for (int i = 0; i < (PK + 1) / 2; ++i) kpk[i] = 0u;
#pragma unroll
    for (int i = 0; i < (PV + 1) / 2; ++i) vpk[i] = 0u;
#pragma unroll
    for (int i = 0; i < PK; ++i) {
        const int c = tid + i * C::THREADS;
        if (c < NKC) kpk[i >> 1] |= slot_field(c / CH, c % CH) << (12 * (i & 1));
    }
#pragma unroll
    for (int i = 0; i < PV; ++i) {
        const int c = tid + i * C::THREADS;
        if (c < NVI) vpk[i >> 1] |= slot_field(2 * (c % (C::N / 2)), c / (C::N / 2)) << (12 * (i & 1));      // key 2 kp; key 2 kp + 1 = the next column
    }
#define WIN_TOK2(dy_, dx_, off_, pad_)                                                                               \
    const int ty_##off_ = y0_ + (dy_), tx_##off_ = x0_ + (dx_);                                                      \
    const bool pad_ = ty_##off_ >= grid || tx_##off_ >= grid;                                                        \
    const int off_ = (min(ty_##off_, grid - 1) * grid + min(tx_##off_, grid - 1)) * (3 * D)
#define WIN_LOAD_K(i_)                                                                                               \
    do {                                                                                                             \
        const int y0_ = n_y0, x0_ = n_x0;                                                                            \
        if (tid + (i_) * C::THREADS < NKC) {                                                                         \
            uint32_t f_ = kpk[(i_) >> 1];                                                                            \
            asm volatile("" : "+v"(f_));          /* decode per item: hoisted out of the item loop the fields spill */ \
            f_ >>= 12 * ((i_) & 1);                                                                                  \
            WIN_TOK2((int)(f_ & 15u), (int)((f_ >> 4) & 15u), ko_, kp_);                                             \
            n_pad |= (uint32_t)kp_ << (i_);                                                                          \
            kreg[i_] = *reinterpret_cast<const uint4*>(n_base + ko_ + D + (int)((f_ >> 8) & 15u) * 8);               \
        }                                                                                                            \
    } while (0)
#define WIN_LOAD_V(i_)                                                                                               \
    do {                                                                                                             \
        const int y0_ = n_y0, x0_ = n_x0;                                                                            \
        if (tid + (i_) * C::THREADS < NVI) {                                                                         \
            uint32_t f_ = vpk[(i_) >> 1];                                                                            \
            asm volatile("" : "+v"(f_));                                                                             \
            f_ >>= 12 * ((i_) & 1);                                                                                  \
            const int dy_ = (int)(f_ & 15u), dx_ = (int)((f_ >> 4) & 15u), ch_ = (int)((f_ >> 8) & 15u);             \
            WIN_TOK2(dy_, dx_, vo0_, vp0_);                                                                          \
            WIN_TOK2(dy_, dx_ + 1, vo1_, vp1_);       /* 2 kp is even and a window row holds 14 tokens: same row */   \
            n_pad |= ((uint32_t)vp0_ | ((uint32_t)vp1_ << 1)) << (PK + 2 * (i_));                                   \
            v0reg[i_] = *reinterpret_cast<const uint4*>(n_base + vo0_ + 2 * D + ch_ * 8);                            \
            v1reg[i_] = *reinterpret_cast<const uint4*>(n_base + vo1_ + 2 * D + ch_ * 8);                            \
        }                                                                                                            \
    } while (0)
    // group g of the issue: 0 = Q, 1 .. = K slots two at a time, then V slots one at a time
    constexpr int NGK = (PK + 1) / 2, NGRP = 1 + NGK + PV;
    static_assert(NGRP <= C::NT, "issue groups are spread over the key tiles");
#define WIN_GROUP(g_)                                                                                                \
    do {                                                                                                             \
        if ((g_) == 0) { if constexpr (!PL) WIN_LOAD_Q(); }                                                        \
        else if ((g_) <= NGK) {                                                                                      \
            WIN_LOAD_K(2 * ((g_) - 1));                                                                              \
            if (2 * ((g_) - 1) + 1 < PK) WIN_LOAD_K(2 * ((g_) - 1) + 1 < PK ? 2 * ((g_) - 1) + 1 : 0);               \
        } else WIN_LOAD_V((g_) - 1 - NGK < PV ? (g_) - 1 - NGK : 0);                                                 \
    } while (0)
#define WIN_ISSUE(it_)                                                                                               \
    do {                                                                                                             \
        WIN_HEAD(it_);                                                                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < NGRP; ++g_) WIN_GROUP(g_);                                           \
        if constexpr (PL) WIN_LOAD_Q();                                                                              \
    } while (0)

    __syncthreads();                      // bias table is read by WIN_ISSUE
    // PIPE & 2: static priority for the younger half of the block.  Waves w and w + 4 share a SIMD; at equal priority the older
    // one wins every arbitration (tile loop 8.1 k cycles against 10.9 k, then 4.7 k at the item barrier waiting for its partner).
    // One s_setprio for waves 4 .. 6, outside every loop (MI355X_MICROARCH.md "two waves per SIMD" item 4); the condition is
    // wave-uniform by construction (readfirstlane), as s_setprio ignores EXEC.
    if constexpr ((PIPE & 2) != 0) {
        if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    }
    int it = blockIdx.x;
#ifdef WIN_TIMING
    unsigned long long wph[6] = {0, 0, 0, 0, 0, 0}, wprev = __builtin_amdgcn_s_memtime();
    const unsigned long long wstart = wprev;
    int n_done = 0;
#define WIN_STAMP(i_) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); wph[i_] += tn_ - wprev; wprev = tn_; }
#else
#define WIN_STAMP(i_)
#endif
    if (it < n_items) WIN_ISSUE(it);
    for (; it < n_items; it += gridDim.x) {
        // ---- stage this item's K (row-major) and V^T; the previous item's readers are done -------------
        __syncthreads();
        WIN_STAMP(0)
        if (__any(n_pad != 0u)) {             // only the bottom / right windows: k / v of a padding token = the bias
#pragma unroll
            for (int i = 0; i < PK; ++i) {
                const uint4 kb = *reinterpret_cast<const uint4*>(Bia + n_head * HD + ((tid + i * C::THREADS) % CH) * 8);
                kreg[i] = sel4((n_pad >> i) & 1u, kb, kreg[i]);
            }
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                const uint4 vb = *reinterpret_cast<const uint4*>(Bia + D + n_head * HD + ((tid + i * C::THREADS) / (C::N / 2)) % CH * 8);
                v0reg[i] = sel4((n_pad >> (PK + 2 * i)) & 1u, vb, v0reg[i]);
                v1reg[i] = sel4((n_pad >> (PK + 2 * i + 1)) & 1u, vb, v1reg[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < PK; ++i) {
            const int c = tid + i * C::THREADS;
            if (c < NKC) *reinterpret_cast<uint4*>(Ks + C::lds_key(c / CH) * HD + (c % CH) * 8) = kreg[i];
        }
        // V^T[d][key]: a thread takes one 8-wide d chunk of TWO adjacent keys and writes eight 4-byte
        // words; consecutive lanes take consecutive key pairs -> consecutive banks.
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            const int c = tid + i * C::THREADS;
            if (c < NVI) {
                const int kp = c % (C::N / 2), ch = c / (C::N / 2);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<uint32_t*>(Vt + (ch * 8 + e) * C::VSTR + C::lds_key(2 * kp)) = pair_elem(v0reg[i], v1reg[i], e);
            }
        }
        if constexpr (!PL) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = qfn[ks];
        }
        const int qoff = n_qoff;
        const bool qin = qoff >= 0;           // a real token (padding / tile-padding queries are dropped)
        const int im = n_im, head = n_head;
        WIN_STAMP(1)
        __syncthreads();
        WIN_STAMP(2)
        // ---- next item's loads: in flight during the whole compute below --------------------------------
        const bool has_next = it + (int)gridDim.x < n_items;
        if (has_next) {
            if (wave >= C::NT) WIN_ISSUE(it + (int)gridDim.x);
            else WIN_HEAD(it + (int)gridDim.x);
        }
        WIN_STAMP(3)
#ifdef WIN_TIMING
        ++n_done;
#endif
        if (wave >= C::NT) continue;          // wave 7 only stages

        // ---- decomposed rel-pos: RH[j] = q . rel_h[qh - j + 13], RW[j] = q . rel_w[qw - j + 13] -----
        // (log2 domain; image_encoder.py:325-361 uses the UNSCALED q)
        float* scr = Scr + wave * 32 * C::SSTR + ql * C::SSTR;
        // bias of accumulator register r (tile-local key c = acc_row(r, hh): window row 2t + c / 14, column c % 14):
        //   RH[2t + c / 14] + BW[r],  BW[r] = RW[c % 14]  (-inf for the 4 zero keys c >= 28)
        float RH[PL ? 1 : C::WS], BW[16];
        // PIPE: the row terms stay in the scratch row of this query, pre-multiplied by log2(e), as rhs[j] = RH[j] (j = 0 .. 13);
        // the column-term transpose goes through the upper half of the same row (floats 16 .. 31 + 32: the row is 33 floats and the
        // table product has 27 live rows, so the two uses are kept apart by writing tw to a second pass AFTER RH has been rebuilt)
        float* rhs = scr;
        {
            const f32x16_t th = tile_times_qT<PREC, HD>(Tab, lane, qf);
            if constexpr (PL) {
                // the column term first (its scratch use ends before the row terms are parked there for the whole tile loop)
                const f32x16_t tw = tile_times_qT<PREC, HD>(Tab + 32 * HD, lane, qf);
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[acc_row(r, hh)] = tw[r];
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = acc_row(r, hh);
                    BW[r] = c < C::TK ? scr[qw - c % C::WS + C::WS - 1] * LOG2E_F : -INFINITY;
                }
                wave_lds_sync();
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[acc_row(r, hh)] = th[r] * LOG2E_F;       // row k of the table product, log2 domain
                wave_lds_sync();
                RH[0] = 0.f;
            } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[acc_row(r, hh)] = th[r];
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < C::WS; ++j) RH[j] = scr[qh - j + C::WS - 1] * LOG2E_F;
            wave_lds_sync();
            const f32x16_t tw = tile_times_qT<PREC, HD>(Tab + 32 * HD, lane, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[acc_row(r, hh)] = tw[r];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = acc_row(r, hh);
                BW[r] = c < C::TK ? scr[qw - c % C::WS + C::WS - 1] * LOG2E_F : -INFINITY;
            }
            }
        }

        const float c2 = rsqrtf((float)HD) * LOG2E_F;
        f32x16_t O[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
#ifdef WIN_TIMING
        asm volatile("" :: "v"(RH[0]), "v"(BW[15]));
#endif
        WIN_STAMP(4)

        f32x16_t Sn;
        if constexpr (PL) Sn = tile_times_qT<PREC, HD>(Ks, lane, qf);
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            if constexpr (PL) {
                if (t + 1 < NGRP && has_next) WIN_GROUP(t + 1 < NGRP ? t + 1 : 0);     // K / V groups over tiles 0 .. NGRP - 2
                if (t == C::NT - 1 && has_next) WIN_LOAD_Q();                           // qf is dead from here on (see above)
            } else {
                if (t < NGRP && has_next) WIN_GROUP(t);
            }
            f32x16_t S;
            float rh0, rh1;
            if constexpr (PL) {
                S = Sn;
                // row terms of this tile: RH[2t] = table row qh - 2t + 13, RH[2t + 1] = the row below it (LDS, wave-private)
                rh0 = rhs[qh - 2 * t + C::WS - 1];
                rh1 = rhs[qh - 2 * t + C::WS - 2];
                if (t + 1 < C::NT) Sn = tile_times_qT<PREC, HD>(Ks + (t + 1) * 32 * HD, lane, qf);    // under the softmax below
            } else {
                S = tile_times_qT<PREC, HD>(Ks + t * 32 * HD, lane, qf);
                rh0 = RH[2 * t];
                rh1 = RH[2 * t + 1];
            }
            // registers 0..5 hold keys of window row 2t, 8..15 of row 2t + 1, 6 and 7 of row 2t + hh (c = 10, 11 | 14, 15)
            const float rhx = hh ? rh1 : rh0;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = S[r] * c2 + BW[r];
            float m0 = fmaxf(fmaxf(S[0], S[1]), fmaxf(S[2], S[3]));
            m0 = fmaxf(m0, fmaxf(S[4], S[5]));
            const float mxx = fmaxf(S[6], S[7]);
            float m1 = fmaxf(fmaxf(S[8], S[9]), fmaxf(S[10], S[11]));
            m1 = fmaxf(m1, fmaxf(fmaxf(S[12], S[13]), fmaxf(S[14], S[15])));
            float mx = fmaxf(fmaxf(m0 + rh0, mxx + rhx), m1 + rh1);
            mx = fmaxf(mx, xhalf_partner(mx));
            const float m_new = fmaxf(m_run, mx);
            if (__any(m_new != m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[dt][r] *= alpha;
            }
            m_run = m_new;
            const float o0 = rh0 - m_new, o1 = rh1 - m_new, ox = rhx - m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = __builtin_amdgcn_exp2f(S[r] + (r < 6 ? o0 : r < 8 ? ox : o1));
            if (!C::ONES) {
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += S[r];
                l_run += sum;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint4 pb = pack_p<PREC>(S, u);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const uint4 va = load_vt_frag(Vt + (dt * 32 + ql) * C::VSTR + 32 * t, u, hh);
                    O[dt] = ET<PREC>::mfma32(va, pb, O[dt]);
                }
            }
        }
        if constexpr (C::ONES) l_run = ones_row_sum<HD, C::DT>(O, hh);
        else l_run += xhalf_partner(l_run);

#ifdef WIN_TIMING
        asm volatile("" :: "v"(O[0][0]), "v"(O[C::DT - 1][15]), "v"(l_run), "v"(qin), "v"(im), "v"(head));
        WIN_STAMP(5)
        continue;
#endif
        // write: token (q) -> un-partitioned row; drop window padding (image_encoder.py:287-288)
        if (qin) {
            const float inv = 1.0f / l_run;
            const size_t orow_i = (size_t)im * img_rows + (size_t)(qoff / (3 * D));
            const size_t obase = orow_i * D + head * HD;
            uint16_t* orow = out + obase;
            if constexpr (LO == 2) store_attention_row_mx<PREC, HD, C::DT>(O, inv, orow, hh, mx, orow_i, head, heads);
            else
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 32 * dt + 8 * g + 4 * hh;
                    if (d0 < HD) {
                        uint2 o;
                        o.x = pack2_fast<PREC>(O[dt][4 * g + 0] * inv, O[dt][4 * g + 1] * inv);
                        o.y = pack2_fast<PREC>(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv);
                        *reinterpret_cast<uint2*>(orow + d0) = o;
                        if constexpr (LO == 1) {
                            if ((lo_heads >> (head & 31)) & 1u) {          // wave-uniform (head comes from the block's item)
                                uint2 h, l;
                                split2_pack<PREC>(O[dt][4 * g + 0] * inv, O[dt][4 * g + 1] * inv, h.x, l.x);
                                split2_pack<PREC>(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv, h.y, l.y);
                                *reinterpret_cast<uint2*>(out_lo + obase + d0) = l;
                            }
                        }
                    }
                }
        }
    }
#ifdef WIN_TIMING
    if (lane == 0) {          // every wave reports (round 5: the per-wave imbalance is what the first barrier's wait consists of)
        unsigned long long* tp = reinterpret_cast<unsigned long long*>(out) + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) tp[i] = wph[i];
        tp[6] = __builtin_amdgcn_s_memtime() - wstart;
        tp[7] = n_done;
    }
#endif
#undef WIN_ISSUE
#undef WIN_GROUP
#undef WIN_LOAD_V
#undef WIN_LOAD_K
#undef WIN_LOAD_Q
#undef WIN_HEAD
#undef WIN_TOK
#undef WIN_TOK2
}

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset; destination = LDS byte address
// m0v (wave-uniform) + 16 * lane.  Inline asm: hipcc must not see the load in its waitcnt bookkeeping (it would drain
// it before every ds_read); the caller waits with an explicit s_waitcnt vmcnt.  Inactive lanes write nothing.
__device__ __forceinline__ void glds16_sbase(uint32_t voff, const void* sbase, uint32_t m0v) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(m0v)
        : "memory");
}

// V of the global-attention blocks, transposed once per block into vt[img][head][d][token] (ET), so that the attention
// kernel can DMA its V^T tiles ([d][64 keys], 128 contiguous bytes per row) straight into LDS: a DMA cannot transpose.
// Block = (image, head, 64-token tile): 64 x HD in, HD x 64 out through LDS.  ~2 x 84 MB per 8-image block, HBM-bound.
template <int HD>
__global__ __launch_bounds__(256) void vt_pack_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ vt, int heads,
                                                      int ntok) {
    __shared__ uint16_t tile[64][HD + 2];
    const int tiles = ntok / 64;
    const int t = blockIdx.x % tiles, P = blockIdx.x / tiles;
    const int head = P % heads, im = P / heads;
    const int D = heads * HD;
    const uint16_t* src = qkv + ((size_t)im * ntok + (size_t)t * 64) * (3 * D) + 2 * D + head * HD;
    constexpr int CH = HD / 8;
    for (int c = threadIdx.x; c < 64 * CH; c += 256) {
        const int r = c / CH, ch = c % CH;
        const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)r * (3 * D) + ch * 8);
        uint32_t* p = reinterpret_cast<uint32_t*>(&tile[r][ch * 8]);       // rows are 4-byte aligned (HD + 2 even)
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
    __syncthreads();
    uint16_t* dst = vt + (size_t)P * HD * ntok + (size_t)t * 64;
    for (int c = threadIdx.x; c < HD * 8; c += 256) {                       // (d, 8-position chunk)
        // chunk c8 of a 64-key tile holds keys 16 (c8 / 2) + 4 (c8 % 2) + {0,1,2,3, 8,9,10,11}: the key order of the PV
        // MFMA's B fragment (pack_p), so that the attention kernel reads a lane's eight keys with ONE 16-byte LDS read
        // (load_vt_frag_perm).  The layout is private to vt_pack_kernel / global_attention_kernel.
        const int d = c / 8, c8 = c % 8, k0 = 16 * (c8 >> 1) + 4 * (c8 & 1);
        uint4 o;
        o.x = tile[k0 + 0][d] | ((uint32_t)tile[k0 + 1][d] << 16);
        o.y = tile[k0 + 2][d] | ((uint32_t)tile[k0 + 3][d] << 16);
        o.z = tile[k0 + 8][d] | ((uint32_t)tile[k0 + 9][d] << 16);
        o.w = tile[k0 + 10][d] | ((uint32_t)tile[k0 + 11][d] << 16);
        *reinterpret_cast<uint4*>(dst + (size_t)d * ntok + c8 * 8) = o;
    }
}

// =========================================================================================
// global attention (grid x grid tokens, flash-style online softmax).
// Block = 128 queries of one (image, head): 4 waves x 32-query strips.  A strip lies inside one
// image row half, so qh is wave-uniform and qw = qw0 + lane.  Keys are streamed one image row
// (64 keys) per tile: kh = tile index, kw = position in the tile.  80 KB of LDS and <= 256
// registers -> two blocks (two waves per SIMD) per CU.
// =========================================================================================
template <int HD, int NW = 4>
struct GlbCfg {
    static constexpr int G = 64, KT = 64;                 // grid side, keys per tile
    static constexpr int DT = (HD + 31) / 32;
    static constexpr int VSTR = KT + 8;                   // 72 el = 144 B = 9 DMA chunks per d row (8 data + 1 pad): an odd chunk
                                                          // count, conflict-free for the 16-byte reads of load_vt_frag_perm
    static constexpr int KSTR = HD + 8;                   // K row stride: HD / 8 data chunks + 1 pad chunk (odd: see tile_times_qT)
    static constexpr int K_BYTES = KT * KSTR * 2;
    static constexpr int VT_BYTES = DT * 32 * VSTR * 2;
    // round 4: the two wave groups of a block run a third of a tile apart (see the main loop), which needs V^T triple-buffered
    static constexpr int NKB = 2, NVB = SAMRS_GLB_SKEW ? 3 : 2;
    static constexpr int KV_BYTES = NKB * K_BYTES + NVB * VT_BYTES;
    static constexpr int SSTR = 33;
    static constexpr int SCR_BYTES = NW * 32 * SSTR * 4;         // setup scratch, aliases the K/V buffers
    static constexpr int UNION_BYTES = KV_BYTES > SCR_BYTES ? KV_BYTES : SCR_BYTES;
    static constexpr int RH_STR = 65;
    static constexpr int RH_BYTES = NW * 32 * RH_STR * 4;
    static constexpr int LDS_BYTES = UNION_BYTES + RH_BYTES;
};

// NW waves = NW x 32 queries per block.  NW = 8 (one block per CU, 113 KiB of LDS): the K / V^T tiles are staged once for
// 256 queries, which halves the LDS-DMA pieces each wave has to push through the CU's texture-address queue per tile
// (measured: ~150 cycles of issue time per piece and wave; 22 pieces per tile and block).
// PIPE (round 5): as in window_attention_kernel, the QK^T products of tile kt + 1 are issued in front of the softmax of tile kt, inside
// the same wave (two S tiles live: 32 more registers; 168 + 32 fit the 256 of two waves per SIMD).  K(kt + 1) must then be in LDS when
// iteration kt starts, so the K pieces run TWO tiles ahead and the V^T pieces one -- still two buffers each: K(kt) is dead once S(kt)
// sits in registers, i.e. from the barrier that closes iteration kt - 1.  Same products in the same order: bit-identical.
template <int PREC, int HD, int NW, int LO = 0 /* as for window_attention_kernel */, int PIPE = 0>
__global__ __launch_bounds__(64 * NW, 2) void global_attention_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ vt, const float* __restrict__ rel_h,
    const float* __restrict__ rel_w, uint16_t* __restrict__ out, int heads,
    uint16_t* __restrict__ out_lo = nullptr /* LO == 1: the split remainder of out */, MxOut mx = MxOut()) {
    using C = GlbCfg<HD, NW>;
    constexpr int KS = HD / 16;
    constexpr int QPB = 32 * NW, NTH = 64 * NW;     // queries / threads per block
    constexpr int G = C::G, NTOK = G * G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Scr = reinterpret_cast<float*>(smem);                              // setup only
    // buffer b: K tile at b*(K_BYTES+VT_BYTES), V^T tile right after it (computed, not an array
    // of pointers: runtime-indexed arrays end up in scratch)
    auto Kb = [&](int b) { return reinterpret_cast<uint16_t*>(smem + b * C::K_BYTES); };                       // b in [0, NKB)
    auto Vb = [&](int b) { return reinterpret_cast<uint16_t*>(smem + C::NKB * C::K_BYTES + b * C::VT_BYTES); };   // b in [0, NVB)
    float* RH = reinterpret_cast<float*>(smem + C::UNION_BYTES);             // [NW][32][RH_STR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, ql = lane & 31;
#ifdef GLB_TIMING
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    // 1-D grid.  All 32 query blocks of one (image, head) re-read the same 1.3 MB of K/V, so put
    // them on ONE XCD (hardware: block b -> XCD b % 8) to keep that working set in its 4 MB L2.
    const int QB = NTOK / QPB;
    int qb, P;
    {
        const int L = blockIdx.x, npairs = gridDim.x / QB;
        if (npairs % 8 == 0) {
            const int xcd = L % 8, idx = L / 8;
            qb = idx % QB;
            P = (idx / QB) * 8 + xcd;
        } else {
            qb = L % QB;
            P = L / QB;
        }
    }
    const int head = P % heads, im = P / heads;
    const int D = heads * HD;
    const uint16_t* base = qkv + (size_t)im * NTOK * (3 * D) + head * HD;

    const int q = qb * QPB + wave * 32 + ql;
    const int qh = q / G;                       // wave-uniform
    const int qw0 = (qb * QPB + wave * 32) % G; // 0 or 32
    uint4 qf[KS];
    load_q_frags<KS>(base + (size_t)q * (3 * D), hh, qf);

    // ---- setup: decomposed rel-pos terms (log2 domain) ---------------------------------------
    //  RH[q][kh] = q . rel_h[qh - kh + 63]   -> LDS (one value per key tile)
    //  rw[a][r]  = q . rel_w[qw - kw + 63], kw = 32a + acc_row(r, hh)  -> 32 registers
    float* rh = RH + wave * 32 * C::RH_STR + ql * C::RH_STR;
    float* scr = Scr + wave * 32 * C::SSTR + ql * C::SSTR;
    float rw[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // T_h'[i'][q] = Q[q] . rel_h[qh + i'],  i' = 63 - kh
        const f32x16_t th = table_times_qT<PREC, HD>(rel_h, qh + 32 * t, 2 * G - 1, lane, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r) rh[(G - 1) - (32 * t + acc_row(r, hh))] = th[r] * LOG2E_F;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        // T_w'[i'][q] = Q[q] . rel_w[qw0 + i'],  i' = ql - kw + 63  in [0, 94]
        const f32x16_t tt = table_times_qT<PREC, HD>(rel_w, qw0 + 32 * t, 2 * G - 1, lane, qf);
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[acc_row(r, hh)] = tt[r];
        wave_lds_sync();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ip = ql - (32 * a + acc_row(r, hh)) + (G - 1);
                const float v = scr[ip & 31] * LOG2E_F;
                if (t == 0) rw[a][r] = 0.f;
                rw[a][r] = ((ip >> 5) == t) ? v : rw[a][r];
            }
        wave_lds_sync();
    }
    __syncthreads();  // everyone is done with the scratch before K/V tiles overwrite it

    // ---- main loop over key tiles -----------------------------------------------------------
    // K / V staging by LDS-DMA (global_load_lds_dwordx4: global -> LDS without a register hop).  Measured on the
    // register-staged version (s_memtime stamps): 24 % of a tile went into ISSUING the 7 loads per thread (the CU's
    // texture-address unit serialises the 56 wave-loads of its 8 waves at ~25 cycles each) and 37 % into waiting for
    // them + permuting V into V^T + 19 LDS stores per thread, i.e. 61 % staging against 36 % QK^T / softmax / PV.
    // Now: the K tile comes straight from qkv (per-lane source addresses, rows of 2 HD bytes), the V^T tile from the
    // pre-transposed copy vt[img][head][d][token] written by vt_pack_kernel; 22 one-KiB pieces per tile and block
    // instead of 28 loads + 76 LDS stores, no staging registers, nothing to wait for until the end of the tile.
    constexpr int CH = HD / 8, CHP = C::KSTR / 8;   // data chunks / chunks incl. the pad chunk per K row
    static_assert(CHP == CH + 1 && (CHP & 1), "K rows: an odd number of 16-byte chunks");
    constexpr int NCH = C::KT * CHP;                // 16-byte chunks of a K tile, row-major [key][KSTR]
    constexpr int KPC = (NCH + 63) / 64;            // K pieces (704 chunks -> 11)
    constexpr int VCH = HD * 9;                     // chunks of a V^T tile: HD rows x (8 data + 1 pad)
    constexpr int VPC = (VCH + 63) / 64;            // V pieces (720 -> 12, the last one partial)
    constexpr int NPC = KPC + VPC;                  // pieces per tile; wave w takes pieces w, w + NW, ...
    constexpr int PPW = (NPC + NW - 1) / NW;
    const uint16_t* vth = vt + (size_t)P * HD * NTOK;                       // this (image, head): [HD][NTOK]
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // per-lane source byte offsets of this wave's pieces inside a tile (K: relative to the tile's first key row of
    // qkv's K third; V: relative to vt row 0 at the tile's first token); ~0u = lane idle in that piece
    uint32_t poff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pc = wave + NW * i;
        uint32_t o = ~0u;
        if (pc < KPC) {
            const int L = pc * 64 + lane;
            // the pad chunk re-reads the row's last data chunk (a DMA lane cannot be skipped inside a row without breaking
            // the 16-bytes-per-lane destination pattern; its content is never read)
            if (L < NCH) o = (uint32_t)((L / CHP) * (3 * D) + min(L % CHP, CH - 1) * 8) * 2u;
        } else if (pc < NPC) {
            const int L = (pc - KPC) * 64 + lane;
            if (L < VCH) { const int d = L / 9, c = (L % 9) < 8 ? (L % 9) : 7; o = (uint32_t)(d * NTOK + c * 8) * 2u; }
        }
        poff[i] = o;
    }
#define GLB_DMA(kt_, kb_, vb_) GLB_DMA_RANGE(kt_, kb_, vb_, 0, PPW)
    // PIPE: this wave's pieces [i0_, i1_) with the K pieces taken from tile ktk_ (-> K buffer kb_) and the V^T pieces from tile ktv_
    // (-> V^T buffer vb_); a negative tile index skips that kind
#define GLB_DMA_KV(ktk_, kb_, ktv_, vb_, i0_, i1_)                                                           \
    _Pragma("unroll") for (int i_ = (i0_); i_ < (i1_) && i_ < PPW; ++i_) {                                   \
        const int pc_ = __builtin_amdgcn_readfirstlane(wave) + NW * i_;                                       \
        if (pc_ < NPC && poff[i_] != ~0u) {                                                                  \
            const bool isk_ = pc_ < KPC;                                                                     \
            if (isk_ ? (ktk_) >= 0 : (ktv_) >= 0) {                                                          \
                const uint16_t* sb_ = isk_ ? base + (size_t)(ktk_) * C::KT * (3 * D) + D : vth + (size_t)(ktv_) * C::KT; \
                const uint32_t dst_ = lds_base + (isk_ ? (uint32_t)(kb_) * C::K_BYTES + (uint32_t)pc_ * 1024u        \
                                                       : (uint32_t)(C::NKB * C::K_BYTES) + (uint32_t)(vb_) * C::VT_BYTES + (uint32_t)(pc_ - KPC) * 1024u); \
                glds16_sbase(poff[i_], sb_, __builtin_amdgcn_readfirstlane(dst_));                           \
            }                                                                                                \
        }                                                                                                    \
    }
    // pieces [i0_, i1_) of this wave: the issue of a piece waits for a slot in the CU's texture-address queue (44 pieces
    // per tile time from the two resident blocks), so the pieces of the next tile are spread over the tile instead of
    // being queued in front of its first MFMA
#define GLB_DMA_RANGE(kt_, kb_, vb_, i0_, i1_)                                                               \
    _Pragma("unroll") for (int i_ = (i0_); i_ < (i1_) && i_ < PPW; ++i_) {                                   \
        const int pc_ = __builtin_amdgcn_readfirstlane(wave) + NW * i_;                                       \
        if (pc_ < NPC && poff[i_] != ~0u) {                                                                  \
            const bool isk_ = pc_ < KPC;                                                                     \
            const uint16_t* sb_ = isk_ ? base + (size_t)(kt_) * C::KT * (3 * D) + D : vth + (size_t)(kt_) * C::KT; \
            const uint32_t dst_ = lds_base + (isk_ ? (uint32_t)(kb_) * C::K_BYTES + (uint32_t)pc_ * 1024u            \
                                                   : (uint32_t)(C::NKB * C::K_BYTES) + (uint32_t)(vb_) * C::VT_BYTES + (uint32_t)(pc_ - KPC) * 1024u); \
            glds16_sbase(poff[i_], sb_, __builtin_amdgcn_readfirstlane(dst_));                               \
        }                                                                                                    \
    }
    // rows d >= HD of V^T must be zero in both buffers (only when HD is not a multiple of 32); the DMA never touches them
    constexpr bool ONES = SAMRS_GLB_ONES && C::DT * 32 > HD;   // ... except row HD = 1: the softmax row sum rides on the PV product
    if (C::DT * 32 > HD) {
        for (int b = 0; b < C::NVB; ++b)
            for (int i = tid; i < (C::DT * 32 - HD) * C::VSTR; i += NTH)
                Vb(b)[HD * C::VSTR + i] = (ONES && i < C::VSTR) ? ET<PREC>::from_float(1.0f) : (uint16_t)0;
    }
    GLB_DMA(0, 0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float c2 = rsqrtf((float)HD) * LOG2E_F;
    f32x16_t O[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = NTOK / C::KT;
#ifdef GLB_TIMING
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#define GLB_STAMP(i_) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tph[i_] += tn_ - tprev; tprev = tn_; }
#else
#define GLB_STAMP(i_)
#endif
    // Optional schedule (SAMRS_GLB_SKEW = 1, off by default: measured no faster): the two wave groups of the block -- waves w and w + NW / 2 share a SIMD -- run a third of
    // a tile apart, as the pair-stage GEMM's groups do.  With every wave in the same phase (round 3: one __syncthreads per tile
    // re-aligned them) a SIMD's two waves wanted the matrix pipe at the same time (QK^T, PV) and the VALU at the same time (softmax:
    // 32 exp2 per lane and tile at quarter rate), 3350 cycles per tile where the two units' work is 1400 + 2050.  Now group 1 takes
    // its block barrier BETWEEN its softmax and its PV product, group 0 at the end of the tile: the barrier counts match, group 1's
    // PV(t) runs beside group 0's QK^T(t + 1), and one group's softmax beside the other's MFMAs.  Buffer hand-over by the same
    // barrier: tile t + 1 is fetched during tile t into K buffer (t + 1) & 1 and V^T buffer (t + 1) % 3 -- group 1 still reads
    // V^T(t) after barrier t while group 0 already fetches tile t + 2, hence three V^T buffers; K(t) is only read before it.
    if constexpr (PIPE != 0) {
        static_assert(!SAMRS_GLB_SKEW, "the pipelined loop keeps every wave in one phase");
        // tile 0 is in K buffer 0 / V^T buffer 0 (prologue above); K(1) -> K buffer 1 must have landed before iteration 0 starts
        if (nkt > 1) { GLB_DMA_KV(1, 1, -1, 0, 0, PPW) }
        f32x16_t Sn[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) Sn[a] = tile_times_qT<PREC, HD, C::KSTR>(Kb(0) + a * 32 * C::KSTR, lane, qf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            // K(kt + 2) -> K buffer kt & 1 (K(kt) is in registers), V^T(kt + 1) -> V^T buffer (kt + 1) & 1 (V^T(kt - 1) was last read
            // before the barrier that closed iteration kt - 1); a third of the pieces now, a third after QK^T, a third after the softmax
            const int ktk = kt + 2 < nkt ? kt + 2 : -1, ktv = kt + 1 < nkt ? kt + 1 : -1;
            GLB_DMA_KV(ktk, buf, ktv, buf ^ 1, 0, (PPW + 2) / 3)
            const float bh2 = rh[kt];
            f32x16_t S[2];
            float mx = -INFINITY;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                S[a] = Sn[a];
                if (kt + 1 < nkt) Sn[a] = tile_times_qT<PREC, HD, C::KSTR>(Kb(buf ^ 1) + a * 32 * C::KSTR, lane, qf);   // under the VALU below
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = S[a][r] * c2 + rw[a][r];
                    S[a][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            // issue order of the block above: hipcc otherwise emits the ten MFMAs of QK^T(kt + 1) back to back in front of the 64 scale /
            // max VALU instructions of tile kt -- in order, the wave then sits in MFMA issue for 320 cycles and the overlap is lost.
            // One MFMA, its successor's K fragment read, six VALU, ten times.
#pragma unroll
            for (int i = 0; i < 2 * KS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
            // the scale / max block must not be sunk below the (branchy) DMA issue that follows: it would leave the MFMAs' basic block
            asm volatile("" :: "v"(mx), "v"(S[0][0]), "v"(S[1][15]));
            GLB_DMA_KV(ktk, buf, ktv, buf ^ 1, (PPW + 2) / 3, 2 * ((PPW + 2) / 3))
            online_softmax_step<C::DT, !ONES>(S, 2, mx, bh2, m_run, l_run, O);
            GLB_DMA_KV(ktk, buf, ktv, buf ^ 1, 2 * ((PPW + 2) / 3), PPW)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint4 pb = pack_p<PREC>(S[a], u);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) {
                        const uint4 va = load_vt_frag_perm(Vb(buf) + (dt * 32 + ql) * C::VSTR + 32 * a, u, hh);
                        O[dt] = ET<PREC>::mfma32(va, pb, O[dt]);
                    }
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of K(kt + 2) / V^T(kt + 1) have landed
            __syncthreads();                                        // ... everybody's; every read of K(kt + 1) and V^T(kt) is done
        }
    }
    const int grp = SAMRS_GLB_SKEW ? (wave >= NW / 2) : 0;
    int vb = 0;                                                   // V^T buffer of tile kt
    for (int kt = 0; PIPE == 0 && kt < nkt; ++kt) {
        const int buf = kt & 1;
        const int vb1 = SAMRS_GLB_SKEW ? (vb == 2 ? 0 : vb + 1) : (buf ^ 1);
        const bool more = kt + 1 < nkt;
        // group 1 waits for its pieces in the MIDDLE of the tile (before its barrier): it issues them one slot earlier than group 0,
        // so that its softmax still runs between the last issue and the wait
        if (more) { if (grp) { GLB_DMA_RANGE(kt + 1, buf ^ 1, vb1, 0, 2 * ((PPW + 2) / 3)) } else { GLB_DMA_RANGE(kt + 1, buf ^ 1, vb1, 0, (PPW + 2) / 3) } }
        const float bh2 = rh[kt];  // RH[q][kh = kt], constant over the tile
        GLB_STAMP(0)

        f32x16_t S[2];
        float mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            S[a] = tile_times_qT<PREC, HD, C::KSTR>(Kb(buf) + a * 32 * C::KSTR, lane, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = S[a][r] * c2 + rw[a][r];
                S[a][r] = v;
                mx = fmaxf(mx, v);
            }
        }
#ifdef GLB_TIMING
        asm volatile("" :: "v"(mx), "v"(S[0][15]), "v"(S[1][15]));
#endif
        GLB_STAMP(1)
        if (more) { if (grp) { GLB_DMA_RANGE(kt + 1, buf ^ 1, vb1, 2 * ((PPW + 2) / 3), PPW) } else { GLB_DMA_RANGE(kt + 1, buf ^ 1, vb1, (PPW + 2) / 3, 2 * ((PPW + 2) / 3)) } }
        online_softmax_step<C::DT, !ONES>(S, 2, mx, bh2, m_run, l_run, O);
#ifdef GLB_TIMING
        asm volatile("" :: "v"(S[0][0]), "v"(S[1][15]), "v"(l_run));
#endif
        GLB_STAMP(2)
        if (more && !grp) { GLB_DMA_RANGE(kt + 1, buf ^ 1, vb1, 2 * ((PPW + 2) / 3), PPW) }
        if (SAMRS_GLB_SKEW && grp == 1) {                           // group 1's rendezvous: its pieces of tile kt+1 have landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");                          // no LDS read of the PV product may move above the barrier
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint4 pb = pack_p<PREC>(S[a], u);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const uint4 va = load_vt_frag_perm(Vb(SAMRS_GLB_SKEW ? vb : buf) + (dt * 32 + ql) * C::VSTR + 32 * a, u, hh);
                    O[dt] = ET<PREC>::mfma32(va, pb, O[dt]);
                }
            }
#ifdef GLB_TIMING
        asm volatile("" :: "v"(O[0][0]), "v"(O[C::DT - 1][15]));
#endif
        GLB_STAMP(3)
        if (!SAMRS_GLB_SKEW) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt+1 have landed
            GLB_STAMP(4)
            __syncthreads();                                        // ... everybody's; and buffer `buf` is free again
        } else if (grp == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GLB_STAMP(4)
            __builtin_amdgcn_s_barrier();                           // group 0's rendezvous, at the end of its tile
            asm volatile("" ::: "memory");
        }
        GLB_STAMP(5)
        vb = vb1;
    }

#ifdef GLB_TIMING
    if (lane == 0) {          // every wave reports (round 5: the per-wave imbalance is what the first barrier's wait consists of)
        unsigned long long* tp = reinterpret_cast<unsigned long long*>(out) + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) tp[i] = tph[i];
        tp[6] = __builtin_amdgcn_s_memtime() - tstart;
        tp[7] = tstart - t_entry;      // setup: Q fragments, rel-pos tables, first tile
    }
    asm volatile("" :: "v"(O[0][0]), "v"(O[C::DT - 1][15]), "v"(l_run));
    return;
#endif
    if constexpr (ONES) l_run = ones_row_sum<HD, C::DT>(O, hh);
    else l_run += xhalf_partner(l_run);
    const float inv = 1.0f / l_run;
    const size_t obase = ((size_t)im * NTOK + q) * D + head * HD;
    uint16_t* orow = out + obase;
    if constexpr (LO == 2) store_attention_row_mx<PREC, HD, C::DT>(O, inv, orow, hh, mx, (size_t)im * NTOK + q, head, heads);
    else
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * hh;
            if (d0 < HD) {
                uint2 o;
                o.x = pack2_fast<PREC>(O[dt][4 * g + 0] * inv, O[dt][4 * g + 1] * inv);
                o.y = pack2_fast<PREC>(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(orow + d0) = o;
                if constexpr (LO == 1) {
                    uint2 h, l;
                    split2_pack<PREC>(O[dt][4 * g + 0] * inv, O[dt][4 * g + 1] * inv, h.x, l.x);
                    split2_pack<PREC>(O[dt][4 * g + 2] * inv, O[dt][4 * g + 3] * inv, h.y, l.y);
                    *reinterpret_cast<uint2*>(out_lo + obase + d0) = l;
                }
            }
        }
}

// -----------------------------------------------------------------------------------------
// neck: im2col for the 3x3 / pad 1 conv on a channels-last [img][g][g][C] ET tensor.
// A[row = (img, y, x)][k = (ky*3 + kx)*C + c]; one thread per 16-byte chunk.
// -----------------------------------------------------------------------------------------
__global__ void neck_im2col_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ A, int n_images,
                                   int grid, int C) {
    const int cpr = 9 * C / 8;  // chunks per output row
    const long total = (long)n_images * grid * grid * cpr;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ch = t % cpr;
    const long row = t / cpr;
    const int tap = (ch * 8) / C, c0 = (ch * 8) % C;
    const int x = row % grid, y = (row / grid) % grid;
    const long im = row / ((long)grid * grid);
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (yy >= 0 && yy < grid && xx >= 0 && xx < grid)
        v = *reinterpret_cast<const uint4*>(in + ((im * grid + yy) * grid + xx) * C + c0);
    *reinterpret_cast<uint4*>(A + row * (9L * C) + ch * 8) = v;
}

// [tokens][C] <-> [C][tokens] fp32 (embedding hand-over in the reference's NCHW layout)
__global__ void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

template <typename K>
hipError_t set_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

hipError_t launch_outlier_weight_ext(int prec, const float* W, int N, int K, const int* idx, int n_oc, void* out, int ld, int col0, hipStream_t s) {
    if (!W || !out || N < 1 || K < 1 || n_oc < 0 || n_oc > 32 || col0 < 0 || ld < col0 + 64 || (n_oc && !idx)) return hipErrorInvalidValue;
    if (prec == PREC_BF16) outlier_weight_ext_kernel<PREC_BF16><<<(N + 7) / 8, 256, 0, s>>>(W, N, K, idx, n_oc, (uint16_t*)out, ld, col0);
    else outlier_weight_ext_kernel<PREC_F16><<<(N + 7) / 8, 256, 0, s>>>(W, N, K, idx, n_oc, (uint16_t*)out, ld, col0);
    return hipGetLastError();
}
hipError_t launch_outlier_side_weight(int prec, const float* W1, const float* b1, int D, const int* idx2, int n2, const int* idx1, int n1,
                                      void* out, int Ks, float* bias_out, hipStream_t s) {
    if (!W1 || !b1 || !out || !idx2 || n2 < 1 || n2 > 32 || n1 < 0 || n1 > 32 || (n1 && !idx1) || Ks < D + (n1 ? 64 : 0)) return hipErrorInvalidValue;
    if (prec == PREC_BF16) outlier_side_weight_kernel<PREC_BF16><<<32, 256, 0, s>>>(W1, b1, D, idx2, n2, idx1, n1, (uint16_t*)out, Ks, bias_out);
    else outlier_side_weight_kernel<PREC_F16><<<32, 256, 0, s>>>(W1, b1, D, idx2, n2, idx1, n1, (uint16_t*)out, Ks, bias_out);
    return hipGetLastError();
}
hipError_t launch_outlier_gather(const void* hi, const void* lo, int D, const int* idx, int n_oc, void* out, int rows, hipStream_t s) {
    if (!hi || !lo || !idx || !out || n_oc < 1 || n_oc > 32 || rows < 1) return hipErrorInvalidValue;
    outlier_gather_kernel<<<(rows + 3) / 4, 256, 0, s>>>((const uint16_t*)hi, (const uint16_t*)lo, D, idx, n_oc, (uint16_t*)out, rows);
    return hipGetLastError();
}
hipError_t launch_outlier_side_gemm(int prec, const void* Y, int lda, const void* Ws, const float* bias, int M, int K, void* out, hipStream_t s) {
    if (!Y || !Ws || !bias || !out || M < 64 || M % 64 || K < 32 || K % 32 || lda < K || lda % 8) return hipErrorInvalidValue;
    if (prec == PREC_BF16) outlier_side_gemm_kernel<PREC_BF16><<<M / 64, 256, 0, s>>>((const uint16_t*)Y, lda, (const uint16_t*)Ws, bias, K, (uint16_t*)out);
    else outlier_side_gemm_kernel<PREC_F16><<<M / 64, 256, 0, s>>>((const uint16_t*)Y, lda, (const uint16_t*)Ws, bias, K, (uint16_t*)out);
    return hipGetLastError();
}

hipError_t launch_weight_norms(const float* W, int N, int K, float* col_sq, float* row_sq, hipStream_t s) {
    if (!W || N < 1 || K < 1) return hipErrorInvalidValue;
    if (col_sq) weight_col_norms_kernel<<<(K + 63) / 64, 1024, 0, s>>>(W, N, K, col_sq);
    if (row_sq) weight_row_norms_kernel<<<(N + 3) / 4, 256, 0, s>>>(W, N, K, row_sq);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_patch_im2col(int prec, const uint8_t* img, void* A, int n_images, int in_h, int in_w,
                               int grid, int patch, hipStream_t s, void* A_lo) {
    if (patch % 8) return hipErrorInvalidValue;
    const long total = (long)n_images * grid * grid * patch * (patch / 8);
    const int blocks = (int)((total + 255) / 256);
    if (prec == PREC_BF16)
        patch_im2col_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(img, (uint16_t*)A, (uint16_t*)A_lo, n_images, in_h, in_w, grid, patch);
    else
        patch_im2col_kernel<PREC_F16><<<blocks, 256, 0, s>>>(img, (uint16_t*)A, (uint16_t*)A_lo, n_images, in_h, in_w, grid, patch);
    return hipGetLastError();
}

// ---- operand-range check (engine option "range_check") ---------------------------------------------
// Every conversion to f16 on this path SATURATES at +-65504 (common.h ET<PREC_F16>::from_float / pack2) instead of producing
// inf -- silently.  The weights this repo is tested on are N(0, sigma) draws; a checkpoint's activations may not be (a few
// channels of a ViT's residual stream and MLP hidden layer are known to run 100 - 1000x the rest).  Checking inside the
// producers would put two VALU instructions per element pair into the GEMM epilogues the loop is bound by, so the check is a
// pass of its own, off by default: it counts the elements of an operand tensor whose magnitude is the largest finite value
// (0x7bff: what the saturating conversion writes; a legitimate 65504 is not distinguishable and not plausible) or inf / nan.
// bf16 (max finite 0x7f7f = 3.4e38, same range as fp32) can only ever show inf / nan.
static __global__ __launch_bounds__(256) void range_scan_kernel(const uint4* __restrict__ x, long n16, uint32_t limit /* 0x7bff | 0x7f80 */,
                                                         unsigned long long* __restrict__ counter, int row16 /* live 16-byte units per row */,
                                                         int ld16 /* row stride in 16-byte units */) {
    unsigned cnt = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        // rows with a stride: only the first row16 units of every ld16 are operand values (the pad of a padded row is never read by a
        // GEMM and may hold values of an earlier pass)
        const uint4 v = x[row16 == ld16 ? i : (i / row16) * ld16 + i % row16];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt += ((w[k] & 0x7fffu) >= limit) + (((w[k] >> 16) & 0x7fffu) >= limit);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(counter, (unsigned long long)cnt);
}
hipError_t launch_range_scan(int prec, const void* x, long n, unsigned long long* counter, hipStream_t s, int cols, int ld) {
    // n elements in all; cols / ld (optional): rows of `cols` live elements stored with a stride of `ld` (n = rows * cols)
    if (cols <= 0) { cols = 8; ld = 8; }
    if (n % 8 || !x || !counter || cols % 8 || ld % 8 || ld < cols || n % cols) return hipErrorInvalidValue;
    const long n16 = n / 8;
    long blocks = (n16 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    range_scan_kernel<<<(int)blocks, 256, 0, s>>>(reinterpret_cast<const uint4*>(x), n16, prec == PREC_F16 ? 0x7bffu : 0x7f80u, counter, cols / 8, ld / 8);
    return hipGetLastError();
}

hipError_t launch_convert(int prec, const float* in, void* out, long n, hipStream_t s, void* out_lo) {
    if (n % 4) return hipErrorInvalidValue;
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256);
    if (prec == PREC_BF16) convert_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(in, (uint16_t*)out, (uint16_t*)out_lo, n4);
    else convert_kernel<PREC_F16><<<blocks, 256, 0, s>>>(in, (uint16_t*)out, (uint16_t*)out_lo, n4);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------
// LayerNorm folded into the GEMM behind it (ViT-H encoder blocks; image_encoder.py:168,177 + 227 / common.py:25):
//     LN(x) W^T + b = rstd (x (W diag(gamma))^T - mean c) + (b + W beta),   c_n = sum_k (W diag(gamma))_nk
// ln_fold_weight_kernel prepares one output row n per block, once at load: Wf = ET(W diag(gamma)), c_n summed over the
// ROUNDED Wf (so that the mean term cancels exactly what the MFMA adds up), bias_f in double.
// -----------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             uint16_t* __restrict__ Wf, float* __restrict__ cvec,
                                                             float* __restrict__ bias_f, int K) {
    __shared__ double red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    double cs = 0.0, bs = 0.0;
    for (int k = tid; k < K; k += 256) {
        const float w = W[(size_t)n * K + k];
        const uint16_t u = ET<PREC>::from_float(w * gamma[k]);
        Wf[(size_t)n * K + k] = u;
        cs += (double)ET<PREC>::to_float(u);
        bs += (double)w * (double)beta[k];
    }
    red[0][tid] = cs;
    red[1][tid] = bs;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        cvec[n] = (float)red[0][0];
        bias_f[n] = (float)((double)(bias ? bias[n] : 0.f) + red[1][0]);
    }
}

hipError_t launch_ln_fold_weight(int prec, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf,
                                 float* cvec, float* bias_f, int N, int K, hipStream_t s) {
    if (N < 1 || K < 1) return hipErrorInvalidValue;
    if (prec == PREC_BF16) ln_fold_weight_kernel<PREC_BF16><<<N, 256, 0, s>>>(W, gamma, beta, bias, (uint16_t*)Wf, cvec, bias_f, K);
    else ln_fold_weight_kernel<PREC_F16><<<N, 256, 0, s>>>(W, gamma, beta, bias, (uint16_t*)Wf, cvec, bias_f, K);
    return hipGetLastError();
}

// Entry of the folded path (the residual stream as the patch-embed GEMM left it): Xh = ET(X) and, per row, the (mean, M2) pairs
// of eight 160-element groups -- the same partials the residual GEMMs emit (gemm.hip, epilogue_m32<STATS>; the merge does not
// care WHICH 160 elements form a group).  One wave per row of 1280: lane l holds float4 l, l + 64, ... (20 values), a group =
// 8 consecutive lanes, two-pass inside the group.
template <int PREC>
__global__ __launch_bounds__(256) void rowstats_convert_kernel(const float* __restrict__ X, uint16_t* __restrict__ Xh,
                                                               float2* __restrict__ stats, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(X + (size_t)row * 1280);
    float4 v[5];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        v[i] = xr[lane + 64 * i];
        sm += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        uint2 o;
        o.x = pack2<PREC>(v[i].x, v[i].y);
        o.y = pack2<PREC>(v[i].z, v[i].w);
        reinterpret_cast<uint2*>(Xh + (size_t)row * 1280)[lane + 64 * i] = o;
    }
    sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
    const float gm = sm * (1.0f / 160.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float a = v[i].x - gm, b = v[i].y - gm, c = v[i].z - gm, d = v[i].w - gm;
        q += (a * a + b * b) + (c * c + d * d);
    }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    if ((lane & 7) == 0) stats[(size_t)row * 8 + (lane >> 3)] = make_float2(gm, q);
}

// (mean, M2) partials of eight 160-element groups of a row -> (rstd, -rstd mean): Chan's pairwise update in a fixed order
// (no E[x^2] - mean^2 cancellation; bit-reproducible).  One thread per row, 64 bytes in, 8 bytes out.
__global__ __launch_bounds__(256) void ln_rowstat_kernel(const float4* __restrict__ stats, float2* __restrict__ rowstat, int rows, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float4 q[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) q[h] = stats[(size_t)row * 4 + h];
    float mean = q[0].x, m2 = q[0].y;
#pragma unroll
    for (int g = 1; g < 8; ++g) {
        const float gm = (g & 1) ? q[g >> 1].z : q[g >> 1].x, gq = (g & 1) ? q[g >> 1].w : q[g >> 1].y;
        const float d = gm - mean;
        mean += d * (1.0f / (g + 1));
        m2 += gq + d * d * (160.0f * g / (g + 1));
    }
    const float rstd = 1.0f / sqrtf(m2 * (1.0f / 1280.0f) + eps);
    rowstat[row] = make_float2(rstd, -rstd * mean);
}

hipError_t launch_ln_rowstat(const float* stats, float* rowstat, int rows, float eps, hipStream_t s) {
    if (rows < 1 || !stats || !rowstat) return hipErrorInvalidValue;
    ln_rowstat_kernel<<<(rows + 255) / 256, 256, 0, s>>>(reinterpret_cast<const float4*>(stats), reinterpret_cast<float2*>(rowstat), rows, eps);
    return hipGetLastError();
}

hipError_t launch_rowstats_convert(int prec, const float* X, void* Xh, float* stats, int rows, int D, hipStream_t s) {
    if (D != 1280 || rows < 1) return hipErrorInvalidValue;
    const int blocks = (rows + 3) / 4;
    if (prec == PREC_BF16) rowstats_convert_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(X, (uint16_t*)Xh, reinterpret_cast<float2*>(stats), rows);
    else rowstats_convert_kernel<PREC_F16><<<blocks, 256, 0, s>>>(X, (uint16_t*)Xh, reinterpret_cast<float2*>(stats), rows);
    return hipGetLastError();
}

hipError_t launch_layernorm(int prec, const float* X, const float* gamma, const float* beta, float eps,
                            void* out_et, float* out_f32, int rows_out, int D, int window_mode, int grid,
                            int window, hipStream_t s, void* out_lo, void* mx_q_hi, void* mx_q_lo, void* mx_s_hi, void* mx_s_lo, int ld_out,
                            const int* oc_idx, int n_oc) {
    if (D % 4 || D > LN_MAXV * 256) return hipErrorInvalidValue;
    if (ld_out == 0) ld_out = D;
    if (ld_out < D || ld_out % 4 || (ld_out != D && out_lo)) return hipErrorInvalidValue;      // (the MX rows and out_lo stay dense)
    if (n_oc < 0 || n_oc > 32 || (n_oc && (!oc_idx || !out_et || window_mode || ld_out < D + 64))) return hipErrorInvalidValue;
    MxOut mx;
    if (mx_q_hi) {       // MX outputs: whole stages per row, rows in plain order, an ET output to take hi from, no partial lane passes
        if (!mx_q_lo || !mx_s_hi || !mx_s_lo || !out_et || window_mode || D % MXK) return hipErrorInvalidValue;
        mx.q_hi = (unsigned char*)mx_q_hi; mx.q_lo = (unsigned char*)mx_q_lo; mx.s_hi = (unsigned char*)mx_s_hi; mx.s_lo = (unsigned char*)mx_s_lo;
    }
    const int blocks = (rows_out + 3) / 4;
    if (prec == PREC_BF16)
        layernorm_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(X, gamma, beta, eps, (uint16_t*)out_et, out_f32, rows_out, D, window_mode, grid, window, (uint16_t*)out_lo, mx, ld_out, oc_idx, n_oc);
    else
        layernorm_kernel<PREC_F16><<<blocks, 256, 0, s>>>(X, gamma, beta, eps, (uint16_t*)out_et, out_f32, rows_out, D, window_mode, grid, window, (uint16_t*)out_lo, mx, ld_out, oc_idx, n_oc);
    return hipGetLastError();
}

template <int PREC, int HD, int LO = 0>
static hipError_t launch_win(const void* qkv, const float* qb, const float* rh, const float* rw, void* out, int n_images,
                             int grid, int heads, hipStream_t s, void* out_lo = nullptr, MxOut mx = MxOut(), uint32_t lo_heads = 0xffffffffu) {
    using C = WinCfg<HD>;
    // SAMRS_WIN_PIPE=1 / 3: the software-pipelined tile loop (/ + static priority for the younger waves); A/B runs
    // measured (profiles/r05_attention_pipelining.txt): no faster, and the rel-pos row terms take one more rounding on their way through
    // the scratch (the un-pipelined loop contracts RH[j] = t * log2(e) into the add that consumes it): default OFF
#ifdef SAMRS_EXPERIMENTS        // make EXPERIMENTS=1: the pipelined flavours exist (tools/attn_bench.py A/B runs)
    static const int pipe = [] { const char* v = getenv("SAMRS_WIN_PIPE"); return v ? atoi(v) & 3 : 0; }();
    auto k = pipe == 3 ? window_attention_kernel<PREC, HD, LO, 3> : pipe == 2 ? window_attention_kernel<PREC, HD, LO, 2>
           : pipe == 1 ? window_attention_kernel<PREC, HD, LO, 1> : window_attention_kernel<PREC, HD, LO, 0>;
#else
    auto k = window_attention_kernel<PREC, HD, LO, 0>;
#endif
    HIP_CHECK_RET(set_lds(k, C::LDS_BYTES));
    if (heads * HD > C::MAX_D) return hipErrorInvalidValue;
    const int nw = (grid + C::WS - 1) / C::WS;
    const int n_items = n_images * nw * nw * heads;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    dim3 g(n_items < n_cu ? n_items : n_cu), b(C::THREADS);      // persistent: one block per CU (LDS-limited)
    k<<<g, b, C::LDS_BYTES, s>>>((const uint16_t*)qkv, qb, rh, rw, (uint16_t*)out, grid, heads, n_items, (uint16_t*)out_lo, mx, lo_heads);
    return hipGetLastError();
}

hipError_t launch_window_attention(int prec, const void* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, void* out,
                                   int n_images, int grid, int window, int heads, int head_dim, hipStream_t s, void* out_lo,
                                   void* mx_q_hi, void* mx_q_lo, void* mx_s_hi, void* mx_s_lo, uint32_t lo_heads) {
    if (window != 14) return hipErrorInvalidValue;
    if (heads > 32) lo_heads = 0xffffffffu;            // the mask has one bit per head
    MxOut mx;
    if (mx_q_hi) {
        if (!mx_q_lo || !mx_s_hi || !mx_s_lo || (heads * ((head_dim + 31) / 32) * 32) % MXK) return hipErrorInvalidValue;
        mx.q_hi = (unsigned char*)mx_q_hi; mx.q_lo = (unsigned char*)mx_q_lo; mx.s_hi = (unsigned char*)mx_s_hi; mx.s_lo = (unsigned char*)mx_s_lo;
    }
#define WIN_CALL(P, H)                                                                                                          \
    return mx_q_hi ? launch_win<P, H, 2>(qkv, qkv_bias, rel_h, rel_w, out, n_images, grid, heads, s, nullptr, mx)               \
         : out_lo ? launch_win<P, H, 1>(qkv, qkv_bias, rel_h, rel_w, out, n_images, grid, heads, s, out_lo, MxOut(), lo_heads)  \
                  : launch_win<P, H, 0>(qkv, qkv_bias, rel_h, rel_w, out, n_images, grid, heads, s)
    if (prec == PREC_BF16) {
        if (head_dim == 64) WIN_CALL(PREC_BF16, 64);
        if (head_dim == 80) WIN_CALL(PREC_BF16, 80);
    } else if (prec == PREC_F16) {
        if (head_dim == 64) WIN_CALL(PREC_F16, 64);
        if (head_dim == 80) WIN_CALL(PREC_F16, 80);
    }
#undef WIN_CALL
    return hipErrorInvalidValue;
}

// waves per block of the global attention kernel: 8 (256 queries, one block per CU) by default, 4 = the two-blocks-per-CU shape
static int g_glb_waves = [] { const char* v = getenv("SAMRS_GLB_WAVES"); return (v && atoi(v) == 4) ? 4 : 8; }();

template <int PREC, int HD, int NW, int LO = 0>
static hipError_t launch_glb_nw(const void* qkv, const float* rh, const float* rw, void* out, int n_images,
                                int heads, void* vt_ws, hipStream_t s, void* out_lo = nullptr, MxOut mx = MxOut()) {
    using C = GlbCfg<HD, NW>;
    constexpr int NTOK = C::G * C::G;
    // SAMRS_GLB_PIPE=1: the software-pipelined tile loop (bit-identical; measured 2 - 3 % SLOWER: default off)
#ifdef SAMRS_EXPERIMENTS
    static const bool pipe = [] { const char* v = getenv("SAMRS_GLB_PIPE"); return v ? atoi(v) != 0 : false; }();
    auto k = pipe ? global_attention_kernel<PREC, HD, NW, LO, 1> : global_attention_kernel<PREC, HD, NW, LO, 0>;
#else
    auto k = global_attention_kernel<PREC, HD, NW, LO, 0>;
#endif
    HIP_CHECK_RET(set_lds(k, C::LDS_BYTES));
    dim3 g((NTOK / (32 * NW)) * heads * n_images), b(64 * NW);
    k<<<g, b, C::LDS_BYTES, s>>>((const uint16_t*)qkv, (const uint16_t*)vt_ws, rh, rw, (uint16_t*)out, heads, (uint16_t*)out_lo, mx);
    return hipGetLastError();
}

template <int PREC, int HD>
static hipError_t launch_glb(const void* qkv, const float* rh, const float* rw, void* out, int n_images,
                             int heads, void* vt_ws, hipStream_t s, void* out_lo, MxOut mx) {
    constexpr int NTOK = GlbCfg<HD>::G * GlbCfg<HD>::G;
    vt_pack_kernel<HD><<<n_images * heads * (NTOK / 64), 256, 0, s>>>((const uint16_t*)qkv, (uint16_t*)vt_ws, heads, NTOK);
    HIP_CHECK_RET(hipGetLastError());
    if (mx.q_hi) return launch_glb_nw<PREC, HD, 8, 2>(qkv, rh, rw, out, n_images, heads, vt_ws, s, nullptr, mx);
    if (out_lo) return launch_glb_nw<PREC, HD, 8, 1>(qkv, rh, rw, out, n_images, heads, vt_ws, s, out_lo);
    if (g_glb_waves == 4) return launch_glb_nw<PREC, HD, 4>(qkv, rh, rw, out, n_images, heads, vt_ws, s);
    return launch_glb_nw<PREC, HD, 8>(qkv, rh, rw, out, n_images, heads, vt_ws, s);
}

hipError_t launch_global_attention(int prec, const void* qkv, const float* rel_h, const float* rel_w, void* out,
                                   int n_images, int grid, int heads, int head_dim, void* vt_ws, hipStream_t s, void* out_lo,
                                   void* mx_q_hi, void* mx_q_lo, void* mx_s_hi, void* mx_s_lo) {
    if (grid != 64 || !vt_ws) return hipErrorInvalidValue;
    MxOut mx;
    if (mx_q_hi) {
        if (!mx_q_lo || !mx_s_hi || !mx_s_lo || (heads * ((head_dim + 31) / 32) * 32) % MXK) return hipErrorInvalidValue;
        mx.q_hi = (unsigned char*)mx_q_hi; mx.q_lo = (unsigned char*)mx_q_lo; mx.s_hi = (unsigned char*)mx_s_hi; mx.s_lo = (unsigned char*)mx_s_lo;
    }
    if (prec == PREC_BF16) {
        if (head_dim == 64) return launch_glb<PREC_BF16, 64>(qkv, rel_h, rel_w, out, n_images, heads, vt_ws, s, out_lo, mx);
        if (head_dim == 80) return launch_glb<PREC_BF16, 80>(qkv, rel_h, rel_w, out, n_images, heads, vt_ws, s, out_lo, mx);
    } else if (prec == PREC_F16) {
        if (head_dim == 64) return launch_glb<PREC_F16, 64>(qkv, rel_h, rel_w, out, n_images, heads, vt_ws, s, out_lo, mx);
        if (head_dim == 80) return launch_glb<PREC_F16, 80>(qkv, rel_h, rel_w, out, n_images, heads, vt_ws, s, out_lo, mx);
    }
    return hipErrorInvalidValue;
}

// reference-grade mode: the fp32 result of the three-pass lin1 product -> exact-erf GELU -> hi + lo operands of lin2
template <int PREC>
__global__ void gelu_split_kernel(const float* __restrict__ in, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    const float2_t a = gelu_erf2(float2_t{v.x, v.y}), b = gelu_erf2(float2_t{v.z, v.w});
    uint2 h, l;
    split2_pack<PREC>(a.x, a.y, h.x, l.x);
    split2_pack<PREC>(b.x, b.y, h.y, l.y);
    reinterpret_cast<uint2*>(hi)[i] = h;
    reinterpret_cast<uint2*>(lo)[i] = l;
}
hipError_t launch_gelu_split(int prec, const float* in, void* hi, void* lo, long n, hipStream_t s) {
    if (n % 4 || !hi || !lo) return hipErrorInvalidValue;
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256);
    if (prec == PREC_BF16) gelu_split_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(in, (uint16_t*)hi, (uint16_t*)lo, n4);
    else gelu_split_kernel<PREC_F16><<<blocks, 256, 0, s>>>(in, (uint16_t*)hi, (uint16_t*)lo, n4);
    return hipGetLastError();
}

hipError_t launch_neck_im2col(const void* in, void* A, int n_images, int grid, int C, hipStream_t s) {
    if (C % 8) return hipErrorInvalidValue;
    const long total = (long)n_images * grid * grid * (9 * C / 8);
    neck_im2col_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>((const uint16_t*)in, (uint16_t*)A, n_images, grid, C);
    return hipGetLastError();
}

hipError_t launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t s) {
    dim3 g((cols + 31) / 32, (rows + 31) / 32), b(32, 8);
    transpose_f32_kernel<<<g, b, 0, s>>>(in, out, rows, cols);
    return hipGetLastError();
}
