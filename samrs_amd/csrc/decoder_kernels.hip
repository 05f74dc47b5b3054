// decoder_kernels.hip -- prompt encoder, two-way transformer glue, upscaler tail, postprocess,
// painting (gfx950).  The GEMM-shaped parts run through gemm.hip; everything here is the
// HBM-/latency-bound remainder, fp32 except where it feeds an MFMA GEMM.
//
// Reference semantics (paths under Generate Dataset/segment_anything/):
//   prompt tokens / PE : modeling/prompt_encoder.py:73-100,176-219 ; mask_decoder.py:127-129
//   mask prompt embed  : modeling/prompt_encoder.py:51-59,102-105
//   attention          : modeling/transformer.py:218-240 (softmax(QK^T / sqrt(d)) V)
//   upscaler tail      : modeling/mask_decoder.py:53-59,154-167
//   postprocess        : modeling/sam.py:133-162, predictor.py:242-243
//   painting / stats   : Generate Dataset/main_sam_hbox_semantic.py:195-206, statistic.py:15-21
#include "common.h"
#include "kernels.h"

namespace {

constexpr float TWO_PI = 6.283185307179586f;

// random-Fourier positional encoding of a point in [0,1]^2 (prompt_encoder.py:190-197):
// channel ch < 128 -> sin(2pi * ((2c-1) . G[:, ch])), ch >= 128 -> cos(... G[:, ch-128])
__device__ __forceinline__ float pe_channel(const float* __restrict__ gauss, float cx01, float cy01, int ch, int half) {
    const int f = ch < half ? ch : ch - half;
    const float vx = 2.0f * cx01 - 1.0f, vy = 2.0f * cy01 - 1.0f;
    const float a = TWO_PI * (vx * gauss[f] + vy * gauss[half + f]);
    return ch < half ? sinf(a) : cosf(a);
}

// grid: (T, n_prompts); block: C threads (one per channel)
__global__ void prompt_tokens_kernel(PromptParams p, float* __restrict__ tokens, float* __restrict__ tokens2, int T, int C) {
    const int j = blockIdx.x, b = blockIdx.y, ch = threadIdx.x;
    const int half = C / 2;
    const bool has_pts = p.point_coords != nullptr;
    const bool has_box = p.boxes != nullptr;
    const int npt = has_pts ? p.n_points + (has_box ? 0 : 1) : 0;
    float v;
    if (j == 0) {
        v = p.iou_token[ch];
    } else if (j < 5) {
        v = p.mask_tokens[(j - 1) * C + ch];
    } else if (j < 5 + npt) {
        const int k = j - 5;
        int label = -1;
        float x = 0.f, y = 0.f;
        if (k < p.n_points) {
            label = p.point_labels[b * p.n_points + k];
            x = p.point_coords[(b * p.n_points + k) * 2 + 0] + 0.5f;
            y = p.point_coords[(b * p.n_points + k) * 2 + 1] + 0.5f;
        }
        if (label == -1) {
            v = p.not_a_point[ch];
        } else {
            v = pe_channel(p.gauss, x / p.img_size, y / p.img_size, ch, half);
            if (label == 0) v += p.point_emb[0][ch];
            else if (label == 1) v += p.point_emb[1][ch];
        }
    } else {
        const int corner = j - 5 - npt;  // 0 or 1
        const float x = p.boxes[b * 4 + 2 * corner + 0] + 0.5f;
        const float y = p.boxes[b * 4 + 2 * corner + 1] + 0.5f;
        v = pe_channel(p.gauss, x / p.img_size, y / p.img_size, ch, half) + p.point_emb[2 + corner][ch];
    }
    tokens[((size_t)b * T + j) * C + ch] = v;
    if (tokens2) tokens2[((size_t)b * T + j) * C + ch] = v;      // the queries start as a copy of the tokens (transformer.py:87)
}

// grid: g*g blocks; block: C threads.  pe[(y*g + x)][ch] at ((x+.5)/g, (y+.5)/g)
__global__ void dense_pe_kernel(const float* __restrict__ gauss, float* __restrict__ pe, int g, int C) {
    const int t = blockIdx.x, ch = threadIdx.x;
    const int y = t / g, x = t % g;
    pe[(size_t)t * C + ch] = pe_channel(gauss, ((float)x + 0.5f) / (float)g, ((float)y + 0.5f) / (float)g, ch, C / 2);
}

// mask prompt -> dense embedding.  Block = 256 threads = 16 tokens x 16 mid channels in phase 1,
// 256 output channels in phase 2.
__global__ __launch_bounds__(256) void mask_embed_kernel(MaskEmbedParams p, const float* __restrict__ mask_in,
                                                         float* __restrict__ dense, int grid) {
    __shared__ float act2[16][16];
    const int t = threadIdx.x;
    const int b = blockIdx.y;
    const int tok0 = blockIdx.x * 16;
    {
        const int tl = t >> 4, c2 = t & 15;
        const int tok = tok0 + tl;
        const int ty = tok / grid, tx = tok % grid;
        const int S = 4 * grid;
        const float* in = mask_in + (size_t)b * S * S + (size_t)(4 * ty) * S + 4 * tx;
        float out2 = p.b3[c2];
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {
                float a[4];
                float mean = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float s = p.b0[c];
#pragma unroll
                    for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 2; ++kx)
                            s += p.w0[c * 4 + ky * 2 + kx] * in[(size_t)(2 * sy + ky) * S + 2 * sx + kx];
                    a[c] = s;
                    mean += s;
                }
                mean *= 0.25f;
                float var = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) var += (a[c] - mean) * (a[c] - mean);
                const float rstd = 1.0f / sqrtf(var * 0.25f + 1e-6f);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float g = gelu_erf((a[c] - mean) * rstd * p.ln1w[c] + p.ln1b[c]);
                    out2 += p.w3[((c2 * 4 + c) * 2 + sy) * 2 + sx] * g;
                }
            }
        // LayerNorm2d over the 16 channels = the 16 lanes of this token's group
        float s = out2;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float mean = s * (1.0f / 16.0f);
        float d = (out2 - mean) * (out2 - mean);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
        const float rstd = 1.0f / sqrtf(d * (1.0f / 16.0f) + 1e-6f);
        act2[tl][c2] = gelu_erf((out2 - mean) * rstd * p.ln4w[c2] + p.ln4b[c2]);
    }
    __syncthreads();
    float w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) w[c] = p.w6[t * 16 + c];
    const float bb = p.b6[t];
    for (int tl = 0; tl < 16; ++tl) {
        float s = bb;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += w[c] * act2[tl][c];
        dense[((size_t)b * grid * grid + tok0 + tl) * 256 + t] = s;
    }
}

template <int PREC>
__global__ void make_keys_kernel(const float* __restrict__ emb, const float* __restrict__ dense,
                                 const float* __restrict__ vec, float* __restrict__ out_f32,
                                 uint16_t* __restrict__ out_et, long per_batch4, long total4, int C4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const long r = i % per_batch4;
    float4 e = reinterpret_cast<const float4*>(emb)[r];
    const float4 d = dense ? reinterpret_cast<const float4*>(dense)[i] : reinterpret_cast<const float4*>(vec)[r % C4];
    e.x += d.x; e.y += d.y; e.z += d.z; e.w += d.w;
    reinterpret_cast<float4*>(out_f32)[i] = e;
    uint2 o;
    o.x = pack2<PREC>(e.x, e.y);
    o.y = pack2<PREC>(e.z, e.w);
    reinterpret_cast<uint2*>(out_et)[i] = o;
}

__global__ void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

// token self attention, one block per prompt.  Thread (h, i) -> query token i of head h.
constexpr int TOK_MAX = 16;
template <int DH>
__global__ __launch_bounds__(256) void token_self_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ o, int T,
                                                              int C, int heads) {
    extern __shared__ float sm[];  // q | k | v, each [T][C]
    float* sq = sm;
    float* sk = sm + T * C;
    float* sv = sm + 2 * T * C;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < T * C; i += blockDim.x) {
        sq[i] = q[(size_t)b * T * C + i];
        sk[i] = k[(size_t)b * T * C + i];
        sv[i] = v[(size_t)b * T * C + i];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)DH);
    for (int w = threadIdx.x; w < heads * T; w += blockDim.x) {
        const int h = w / T, i = w % T;
        float qi[DH], acc[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) {
            qi[c] = sq[i * C + h * DH + c];
            acc[c] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < T; ++j) {
            const float* kr = sk + j * C + h * DH;
            const float* vr = sv + j * C + h * DH;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) s += qi[c] * kr[c];
            s *= scale;
            const float mn = fmaxf(m, s);
            const float corr = expf(m - mn), pj = expf(s - mn);
            l = l * corr + pj;
#pragma unroll
            for (int c = 0; c < DH; ++c) acc[c] = acc[c] * corr + pj * vr[c];
            m = mn;
        }
        const float inv = 1.0f / l;
        float* orow = o + ((size_t)b * T + i) * C + h * DH;
#pragma unroll
        for (int c = 0; c < DH; ++c) orow[c] = acc[c] * inv;
    }
}

// tokens -> image attention (head dim 16, Ci = 128 = 16 chunks of 8 channels), flash-decoding style.
// The T (<= a dozen) prompt tokens of one prompt attend to all image keys; the work is reading K and
// V once, so the layout is chosen for the loads: a wave instruction fetches 4 consecutive key rows x
// 256 contiguous bytes (lane = key-in-group * 16 + channel chunk), a lane keeps the running softmax
// of T2I_TG tokens for its (head, half-head) chunk, the two halves of a head meet in a DPP add.
// Keys are split over gridDim.x blocks x 4 waves; a block merges its waves through LDS and writes
// one partial (m, l, acc) per (token, chunk); t2i_merge_kernel combines the splits in a fixed order.
//   grid (splits, n_prompts, ceil(T / T2I_TG)), 256 threads.
constexpr int T2I_TG = 8;            // tokens per pass
constexpr int T2I_REC = 10;          // floats per (token, chunk) partial: m, l, acc[8]
constexpr float T2I_NEG = -1.0e30f;  // finite "-inf": exp2(NEG - NEG) = 1 with l = acc = 0 stays harmless

template <int PREC>
__global__ __launch_bounds__(256, 2) void t2i_partial_kernel(const float* __restrict__ qp, const uint16_t* __restrict__ kp,
                                                          const uint16_t* __restrict__ vp, int ld, long bstride,
                                                          float* __restrict__ part, int T, int tokens, int Ci, int kpw) {
    __shared__ float sm[4][T2I_TG][16][T2I_REC];
    const int split = blockIdx.x, b = blockIdx.y, t0 = blockIdx.z * T2I_TG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = lane & 15, kq = lane >> 4;
    float q[T2I_TG][8], acc[T2I_TG][8], m[T2I_TG], l[T2I_TG];
#pragma unroll
    for (int t = 0; t < T2I_TG; ++t) {
        m[t] = T2I_NEG;
        l[t] = 0.f;
        const bool tv = t0 + t < T;
        const float* qr = qp + ((size_t)b * T + (tv ? t0 + t : 0)) * Ci + ch * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            q[t][c] = tv ? qr[c] * (0.25f * 1.44269504088896340736f) : 0.f;   // 1/sqrt(16), exp2 domain
            acc[t][c] = 0.f;
        }
    }
    const int kbeg = (split * 4 + wave) * kpw;
    const int kend = kbeg + kpw < tokens ? kbeg + kpw : tokens;
    const uint16_t* kb = kp + (size_t)b * bstride * ld + ch * 8;
    const uint16_t* vb = vp + (size_t)b * bstride * ld + ch * 8;
    constexpr int U = 2;    // key groups (of 4 keys) per iteration; the next iteration's loads are already in flight
    uint4 kk[U], vv[U], kn[U], vn[U];
#define T2I_LOAD(dk_, dv_, k0_)                                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < U; ++u_) {                                           \
        const int key_ = (k0_) + 4 * u_ + kq;                                                    \
        dk_[u_] = dv_[u_] = make_uint4(0u, 0u, 0u, 0u);                                          \
        if (key_ < kend) {                                                                       \
            dk_[u_] = *reinterpret_cast<const uint4*>(kb + (size_t)key_ * ld);                   \
            dv_[u_] = *reinterpret_cast<const uint4*>(vb + (size_t)key_ * ld);                   \
        }                                                                                        \
    }
    T2I_LOAD(kk, vv, kbeg)
    for (int k0 = kbeg; k0 < kend; k0 += 4 * U) {
        T2I_LOAD(kn, vn, k0 + 4 * U)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = k0 + 4 * u + kq < kend;      // same for both halves of a head (lane ^ 1)
            const uint32_t kw[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w};
            const uint32_t vw[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
            float kf[8], vf[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                kf[2 * c] = ET<PREC>::to_float((uint16_t)(kw[c] & 0xffffu));
                kf[2 * c + 1] = ET<PREC>::to_float((uint16_t)(kw[c] >> 16));
                vf[2 * c] = ET<PREC>::to_float((uint16_t)(vw[c] & 0xffffu));
                vf[2 * c + 1] = ET<PREC>::to_float((uint16_t)(vw[c] >> 16));
            }
#pragma unroll
            for (int t = 0; t < T2I_TG; ++t) {
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) sc += q[t][c] * kf[c];
                // other half of the head = lane ^ 1: a DPP quad_perm [1,0,3,2] move, not a ds_bpermute round trip
                sc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sc), 0xB1, 0xF, 0xF, true));
                sc = valid ? sc : T2I_NEG;
                const float mn = fmaxf(m[t], sc);
                const float corr = __builtin_amdgcn_exp2f(m[t] - mn);
                const float pj = valid ? __builtin_amdgcn_exp2f(sc - mn) : 0.f;
                l[t] = l[t] * corr + pj;
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[t][c] = acc[t][c] * corr + pj * vf[c];
                m[t] = mn;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { kk[u] = kn[u]; vv[u] = vn[u]; }
    }
#undef T2I_LOAD
    // merge the 4 key phases of the wave (lane bits 4, 5), then the 4 waves through LDS
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
        for (int t = 0; t < T2I_TG; ++t) {
            const float mo = __shfl_xor(m[t], off, 64), lo = __shfl_xor(l[t], off, 64);
            const float mn = fmaxf(m[t], mo);
            const float fs = __builtin_amdgcn_exp2f(m[t] - mn), fo = __builtin_amdgcn_exp2f(mo - mn);
            l[t] = l[t] * fs + lo * fo;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[t][c] = acc[t][c] * fs + __shfl_xor(acc[t][c], off, 64) * fo;
            m[t] = mn;
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < T2I_TG; ++t) {
            float* r = sm[wave][t][ch];
            r[0] = m[t]; r[1] = l[t];
#pragma unroll
            for (int c = 0; c < 8; ++c) r[2 + c] = acc[t][c];
        }
    }
    __syncthreads();
    if (threadIdx.x < T2I_TG * 16) {
        const int t = threadIdx.x >> 4, c16 = threadIdx.x & 15;
        float mm = sm[0][t][c16][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) mm = fmaxf(mm, sm[w][t][c16][0]);
        float ll = 0.f, a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __builtin_amdgcn_exp2f(sm[w][t][c16][0] - mm);
            ll += sm[w][t][c16][1] * f;
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] += sm[w][t][c16][2 + c] * f;
        }
        float* r = part + ((((size_t)b * gridDim.z + blockIdx.z) * gridDim.x + split) * (T2I_TG * 16) + threadIdx.x) * T2I_REC;
        r[0] = mm; r[1] = ll;
#pragma unroll
        for (int c = 0; c < 8; ++c) r[2 + c] = a[c];
    }
}

// grid (n_prompts, token groups), T2I_TG * 16 threads = (token, chunk): combine the key splits
__global__ __launch_bounds__(T2I_TG * 16) void t2i_merge_kernel(const float* __restrict__ part, float* __restrict__ o,
                                                                int splits, int T, int Ci) {
    const int b = blockIdx.x, tg = blockIdx.y;
    const int t = tg * T2I_TG + (threadIdx.x >> 4), ch = threadIdx.x & 15;
    if (t >= T) return;
    const float* base = part + (((size_t)b * gridDim.y + tg) * splits * (T2I_TG * 16) + threadIdx.x) * T2I_REC;
    const size_t sstride = (size_t)(T2I_TG * 16) * T2I_REC;
    float mm = T2I_NEG;
    for (int s = 0; s < splits; ++s) mm = fmaxf(mm, base[s * sstride]);
    float ll = 0.f, a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        const float* r = base + s * sstride;
        const float f = __builtin_amdgcn_exp2f(r[0] - mm);
        ll += r[1] * f;
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] += r[2 + c] * f;
    }
    const float inv = 1.0f / ll;
    float* orow = o + ((size_t)b * T + t) * Ci + ch * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) orow[c] = a[c] * inv;
}

// image -> tokens attention (head dim 16).  Thread = (image token, head); keys/values are the T
// prompt tokens (LDS).  grid (tokens/32, n_prompts).  The 8 heads of a token are 8 adjacent lanes that read
// DIFFERENT 16-float slices of a k / v row: at the natural stride they fall on 2 banks (PMC: 74 % of the LDS
// cycles were conflicts), so the LDS copy pads every head slice to I2T_HS = 20 floats -> the eight
// ds_read_b128 of a wave quarter cover all 32 banks.
constexpr int I2T_HS = 20;
template <int PREC>
__global__ __launch_bounds__(256) void i2t_attention_kernel(const uint16_t* __restrict__ qi, int ld, long bstride,
                                                            const float* __restrict__ kt, const float* __restrict__ vt,
                                                            uint16_t* __restrict__ out, int T, int tokens, int Ci) {
    constexpr int HD = 16;
    extern __shared__ float sm[];  // kt | vt, each [T][Ci]
    const int heads = Ci / HD;                  // 8
    const int RS = heads * I2T_HS;              // padded row stride
    float* sk = sm;
    float* sv = sm + T * RS;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < T * Ci; i += 256) {
        const int j = i / Ci, c = i % Ci;
        sk[j * RS + (c / HD) * I2T_HS + c % HD] = kt[(size_t)b * T * Ci + i];
        sv[j * RS + (c / HD) * I2T_HS + c % HD] = vt[(size_t)b * T * Ci + i];
    }
    __syncthreads();
    const int tok = blockIdx.x * (256 / heads) + threadIdx.x / heads;
    const int h = threadIdx.x % heads;
    if (tok >= tokens) return;
    const uint16_t* qrow = qi + ((size_t)b * bstride + tok) * ld + h * HD;
    const uint4 q0 = *reinterpret_cast<const uint4*>(qrow);
    const uint4 q1 = *reinterpret_cast<const uint4*>(qrow + 8);
    const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    float qf[HD];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        qf[2 * c] = ET<PREC>::to_float((uint16_t)(qw[c] & 0xffffu)) * 0.25f;
        qf[2 * c + 1] = ET<PREC>::to_float((uint16_t)(qw[c] >> 16)) * 0.25f;
    }
    float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    for (int j = 0; j < T; ++j) {
        float kr[HD], vr[HD];
#pragma unroll
        for (int c4 = 0; c4 < HD / 4; ++c4) {
            const float4 k4 = *reinterpret_cast<const float4*>(sk + j * RS + h * I2T_HS + 4 * c4);
            const float4 v4 = *reinterpret_cast<const float4*>(sv + j * RS + h * I2T_HS + 4 * c4);
            kr[4 * c4] = k4.x; kr[4 * c4 + 1] = k4.y; kr[4 * c4 + 2] = k4.z; kr[4 * c4 + 3] = k4.w;
            vr[4 * c4] = v4.x; vr[4 * c4 + 1] = v4.y; vr[4 * c4 + 2] = v4.z; vr[4 * c4 + 3] = v4.w;
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s += qf[c] * kr[c];
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), pj = __expf(s - mn);
        l = l * corr + pj;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] = acc[c] * corr + pj * vr[c];
        m = mn;
    }
    const float inv = 1.0f / l;
    uint4 o0, o1;
    o0.x = pack2<PREC>(acc[0] * inv, acc[1] * inv);   o0.y = pack2<PREC>(acc[2] * inv, acc[3] * inv);
    o0.z = pack2<PREC>(acc[4] * inv, acc[5] * inv);   o0.w = pack2<PREC>(acc[6] * inv, acc[7] * inv);
    o1.x = pack2<PREC>(acc[8] * inv, acc[9] * inv);   o1.y = pack2<PREC>(acc[10] * inv, acc[11] * inv);
    o1.z = pack2<PREC>(acc[12] * inv, acc[13] * inv); o1.w = pack2<PREC>(acc[14] * inv, acc[15] * inv);
    uint16_t* orow = out + ((size_t)b * tokens + tok) * Ci + h * HD;
    *reinterpret_cast<uint4*>(orow) = o0;
    *reinterpret_cast<uint4*>(orow + 8) = o1;
}

// ---- fused image -> tokens attention + output projection + residual + LayerNorm ------------------------
// transformer.py:176-181 (TwoWayAttentionBlock step 4):  keys = norm4(keys + out_proj(attn(q = keys + pe, k, v)))
// Everything is local to an image token: a 128-vector of attention output, 256 projected channels, a LayerNorm
// over those 256.  Done as three kernels the 134 MB fp32 key tensor makes four trips through HBM (GEMM
// read-modify-write, LN read + write) plus the attention output's round trip; here it makes two.
// Block = 4 waves, 32 image tokens per group, several groups per block:
//   * attention: thread = (token, head) exactly as in i2t_attention_kernel; its 16 outputs go to an LDS A tile
//     [32][128] in the operand type -- SPLIT: as hi + lo (two tiles), so that the projection sees them at ~2^-22;
//   * projection: wave w owns output columns 64 w .. +63; its W fragments (4 n-tiles x 4 k-steps; SPLIT: hi and lo)
//     stay in registers for the whole block, A fragments come from the LDS tile(s), `mfma_16x16x32`, ascending k;
//     SPLIT = three MFMAs per step (hi hi + lo hi + hi lo): the out-projection operand rounding was 163 of the 899
//     class-map pixels the round-2 engine lost at ViT-H (oracle/error_budget.py: "only dec.oi");
//   * + bias + residual (old keys, or the shared image embedding in layer 0), LayerNorm over the row: the 4 lane
//     quarters of a wave hold 64 of its 256 values, the four waves exchange partial sums through LDS (two-pass
//     statistics), then write fp32 keys, their ET copy and (last layer) the split remainder of that copy.
constexpr int I2TF_ROWS = 32;          // image tokens per group (two MFMA row tiles)
constexpr int I2TF_AST = 136;          // A-tile row stride (ET): 272 B = 68 words = 4 (mod 32)

template <int PREC, bool SPLIT>
__global__ __launch_bounds__(256, 2) void i2t_fused_kernel(const uint16_t* __restrict__ qi, int ld, long q_bstride,
                                                        const float* __restrict__ kt, const float* __restrict__ vt,
                                                        const uint16_t* __restrict__ w, const uint16_t* __restrict__ w_lo,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ resid, long r_bstride,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                        float* __restrict__ outF, uint16_t* __restrict__ outE,
                                                        uint16_t* __restrict__ outE_lo, int T, int tokens, int groups_per_block) {
    constexpr int HD = 16, HEADS = 8, CI = 128, CO = 256, NWV = 4;
    constexpr int KRS = HEADS * I2T_HS;                       // padded k / v row stride (floats)
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    float* sk = reinterpret_cast<float*>(smraw);              // [T][KRS]
    float* sv = sk + T * KRS;
    float* sp = sv + T * KRS;                                 // bias | gamma | beta, [3][256]
    float* ex = sp + 3 * CO;                                  // [2 stats][4 waves][32 rows]
    uint16_t* At = reinterpret_cast<uint16_t*>(ex + 2 * NWV * I2TF_ROWS);   // [32][I2TF_AST] (SPLIT: hi, then lo)
    uint16_t* Al = At + I2TF_ROWS * I2TF_AST;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.y;
    for (int i = tid; i < T * CI; i += 256) {
        const int j = i / CI, c = i % CI;
        sk[j * KRS + (c / HD) * I2T_HS + c % HD] = kt[(size_t)b * T * CI + i];
        sv[j * KRS + (c / HD) * I2T_HS + c % HD] = vt[(size_t)b * T * CI + i];
    }
    for (int i = tid; i < CO; i += 256) { sp[i] = bias[i]; sp[CO + i] = gamma[i]; sp[2 * CO + i] = beta[i]; }
    uint4 wf[4][4], wl[SPLIT ? 4 : 1][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const size_t o = (size_t)(wave * 64 + i * 16 + fr) * CI + ks * 32 + fq * 8;
            wf[i][ks] = *reinterpret_cast<const uint4*>(w + o);
            if constexpr (SPLIT) wl[i][ks] = *reinterpret_cast<const uint4*>(w_lo + o);
        }
    __syncthreads();

    const int tr = tid >> 3, h = tid & 7;                     // attention role: token-in-group, head
    const int g0 = blockIdx.x * groups_per_block;
    // this thread's query slice of the NEXT group is loaded while the current group is projected / normalised
    const uint16_t* qbase = qi + (size_t)b * q_bstride * ld + (size_t)tr * ld + h * HD;
    uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0;
    if (g0 * I2TF_ROWS < tokens) {
        q0 = *reinterpret_cast<const uint4*>(qbase + (size_t)g0 * I2TF_ROWS * ld);
        q1 = *reinterpret_cast<const uint4*>(qbase + (size_t)g0 * I2TF_ROWS * ld + 8);
    }
    for (int g = g0; g < g0 + groups_per_block; ++g) {
        const int row0 = g * I2TF_ROWS;
        if (row0 >= tokens) break;
        // ---- residual loads first: they fly during the attention math -------------------------------------
        float4 rs[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rs[mt][i] = *reinterpret_cast<const float4*>(resid + ((size_t)b * r_bstride + row0 + mt * 16 + fr) * CO + wave * 64 + i * 16 + 4 * fq);
        // ---- attention of (token tr, head h) over the T prompt tokens ------------------------------------------
        {
            const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            float qf[HD];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                qf[2 * c] = ET<PREC>::to_float((uint16_t)(qw[c] & 0xffffu)) * 0.25f;
                qf[2 * c + 1] = ET<PREC>::to_float((uint16_t)(qw[c] >> 16)) * 0.25f;
            }
            float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
            for (int c = 0; c < HD; ++c) acc[c] = 0.f;
            for (int j = 0; j < T; ++j) {
                float kr[HD], vr[HD];
#pragma unroll
                for (int c4 = 0; c4 < HD / 4; ++c4) {
                    const float4 k4 = *reinterpret_cast<const float4*>(sk + j * KRS + h * I2T_HS + 4 * c4);
                    const float4 v4 = *reinterpret_cast<const float4*>(sv + j * KRS + h * I2T_HS + 4 * c4);
                    kr[4 * c4] = k4.x; kr[4 * c4 + 1] = k4.y; kr[4 * c4 + 2] = k4.z; kr[4 * c4 + 3] = k4.w;
                    vr[4 * c4] = v4.x; vr[4 * c4 + 1] = v4.y; vr[4 * c4 + 2] = v4.z; vr[4 * c4 + 3] = v4.w;
                }
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) sc += qf[c] * kr[c];
                const float mn = fmaxf(m, sc);
                const float corr = __expf(m - mn), pj = __expf(sc - mn);
                l = l * corr + pj;
#pragma unroll
                for (int c = 0; c < HD; ++c) acc[c] = acc[c] * corr + pj * vr[c];
                m = mn;
            }
            const float inv = 1.0f / l;
            uint32_t oh[8], ol[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) split2_pack<PREC>(acc[2 * c] * inv, acc[2 * c + 1] * inv, oh[c], ol[c]);
            *reinterpret_cast<uint4*>(At + tr * I2TF_AST + h * HD) = make_uint4(oh[0], oh[1], oh[2], oh[3]);
            *reinterpret_cast<uint4*>(At + tr * I2TF_AST + h * HD + 8) = make_uint4(oh[4], oh[5], oh[6], oh[7]);
            if constexpr (SPLIT) {
                *reinterpret_cast<uint4*>(Al + tr * I2TF_AST + h * HD) = make_uint4(ol[0], ol[1], ol[2], ol[3]);
                *reinterpret_cast<uint4*>(Al + tr * I2TF_AST + h * HD + 8) = make_uint4(ol[4], ol[5], ol[6], ol[7]);
            }
        }
        __syncthreads();
        if (g + 1 < g0 + groups_per_block && (g + 1) * I2TF_ROWS < tokens) {
            q0 = *reinterpret_cast<const uint4*>(qbase + (size_t)(g + 1) * I2TF_ROWS * ld);
            q1 = *reinterpret_cast<const uint4*>(qbase + (size_t)(g + 1) * I2TF_ROWS * ld + 8);
        }
        // ---- projection: 2 x 16 rows x 64 columns per wave, K = 128 ------------------------------------------------
        float v[2][4][4];
        float s1[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            uint4 af[4], al[SPLIT ? 4 : 1];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                af[ks] = *reinterpret_cast<const uint4*>(At + (mt * 16 + fr) * I2TF_AST + ks * 32 + fq * 8);
                if constexpr (SPLIT) al[ks] = *reinterpret_cast<const uint4*>(Al + (mt * 16 + fr) * I2TF_AST + ks * 32 + fq * 8);
            }
            s1[mt] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4_t a = {0.f, 0.f, 0.f, 0.f};
                if constexpr (SPLIT) {      // the small terms first, the leading one on top of them
                    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) c = ET<PREC>::mfma16(wf[i][ks], al[ks], c);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) c = ET<PREC>::mfma16(wl[i][ks], af[ks], c);
                    a = c;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a = ET<PREC>::mfma16(wf[i][ks], af[ks], a);
                const float4 bb = *reinterpret_cast<const float4*>(sp + wave * 64 + i * 16 + 4 * fq);
                v[mt][i][0] = (a[0] + bb.x) + rs[mt][i].x; v[mt][i][1] = (a[1] + bb.y) + rs[mt][i].y;
                v[mt][i][2] = (a[2] + bb.z) + rs[mt][i].z; v[mt][i][3] = (a[3] + bb.w) + rs[mt][i].w;
                s1[mt] += (v[mt][i][0] + v[mt][i][1]) + (v[mt][i][2] + v[mt][i][3]);
            }
        }
        // ---- LayerNorm over the 256 values of a row: lane quarters, then the four waves ---------------------------
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            s1[mt] += __shfl_xor(s1[mt], 16, 64);
            s1[mt] += __shfl_xor(s1[mt], 32, 64);
            if (fq == 0) ex[wave * I2TF_ROWS + mt * 16 + fr] = s1[mt];
        }
        __syncthreads();
        float mean[2], s2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int r = mt * 16 + fr;
            mean[mt] = ((ex[r] + ex[I2TF_ROWS + r]) + (ex[2 * I2TF_ROWS + r] + ex[3 * I2TF_ROWS + r])) * (1.0f / CO);
            s2[mt] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) { v[mt][i][r4] -= mean[mt]; s2[mt] += v[mt][i][r4] * v[mt][i][r4]; }
            s2[mt] += __shfl_xor(s2[mt], 16, 64);
            s2[mt] += __shfl_xor(s2[mt], 32, 64);
            if (fq == 0) ex[NWV * I2TF_ROWS + wave * I2TF_ROWS + r] = s2[mt];
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int r = mt * 16 + fr;
            const float* e2 = ex + NWV * I2TF_ROWS;
            const float rstd = 1.0f / sqrtf(((e2[r] + e2[I2TF_ROWS + r]) + (e2[2 * I2TF_ROWS + r] + e2[3 * I2TF_ROWS + r])) * (1.0f / CO) + eps);
            const size_t orow = (size_t)b * tokens + row0 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = wave * 64 + i * 16 + 4 * fq;
                const float4 gm = *reinterpret_cast<const float4*>(sp + CO + n), bt = *reinterpret_cast<const float4*>(sp + 2 * CO + n);
                const float y0 = v[mt][i][0] * rstd * gm.x + bt.x, y1 = v[mt][i][1] * rstd * gm.y + bt.y;
                const float y2 = v[mt][i][2] * rstd * gm.z + bt.z, y3 = v[mt][i][3] * rstd * gm.w + bt.w;
                if (outF) *reinterpret_cast<float4*>(outF + orow * CO + n) = make_float4(y0, y1, y2, y3);
                uint2 o, lo;
                split2_pack<PREC>(y0, y1, o.x, lo.x);
                split2_pack<PREC>(y2, y3, o.y, lo.y);
                *reinterpret_cast<uint2*>(outE + orow * CO + n) = o;
                if (outE_lo) *reinterpret_cast<uint2*>(outE_lo + orow * CO + n) = lo;
            }
        }
        // the next group's A-tile / ex writes come after this group's last barrier and its reads above
    }
}

// rows of 256 floats = 4 groups of 64: LayerNorm2d(64) (eps) + GELU per group -> ET.
// One wave per row; lane holds 4 consecutive values; a group = 16 lanes.
template <int PREC>
__global__ __launch_bounds__(256) void group_ln_gelu_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            uint16_t* __restrict__ out, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 v = reinterpret_cast<const float4*>(in + row * 256)[lane];
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / 64.0f);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    float q = (a * a + b * b) + (c * c + d * d);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + eps);
    const int c0 = (lane & 15) * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c0);
    const float4 bt = *reinterpret_cast<const float4*>(beta + c0);
    uint2 o;
    const float2_t g0 = gelu_erf2(float2_t{a * rstd * g.x + bt.x, b * rstd * g.y + bt.y});
    const float2_t g1 = gelu_erf2(float2_t{c * rstd * g.z + bt.z, d * rstd * g.w + bt.w});
    o.x = pack2<PREC>(g0.x, g0.y);
    o.y = pack2<PREC>(g1.x, g1.y);
    reinterpret_cast<uint2*>(out + row * 256)[lane] = o;
}

// low[b][c][Y][X] = hyper[b][sel0 + c] . up2[b][pixel(Y, X)][0..31]
// up2 rows are ordered (token y, x)(dy, dx)(dy2, dx2); Y = 4y + 2dy + dy2, X = 4x + 2dx + dx2.
template <int PREC>
__global__ __launch_bounds__(256) void mask_product_kernel(const uint16_t* __restrict__ up2, const float* __restrict__ hyper,
                                                           float* __restrict__ low, int grid, int n_mask_tokens,
                                                           int sel0, int n_sel) {
    __shared__ float hy[4][32];
    const int b = blockIdx.y;
    if (threadIdx.x < n_sel * 32)
        hy[threadIdx.x / 32][threadIdx.x % 32] = hyper[((size_t)b * n_mask_tokens + sel0 + threadIdx.x / 32) * 32 + threadIdx.x % 32];
    __syncthreads();
    const int S = 4 * grid;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int Y = pix / S, X = pix % S;
    const int y = Y >> 2, dy = (Y >> 1) & 1, dy2 = Y & 1;
    const int x = X >> 2, dx = (X >> 1) & 1, dx2 = X & 1;
    const size_t prow = (((size_t)b * grid * grid + (size_t)y * grid + x) * 4 + dy * 2 + dx) * 4 + dy2 * 2 + dx2;
    const uint4* src = reinterpret_cast<const uint4*>(up2 + prow * 32);
    float u[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint4 w = src[i];
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u[i * 8 + 2 * e] = ET<PREC>::to_float((uint16_t)(ww[e] & 0xffffu));
            u[i * 8 + 2 * e + 1] = ET<PREC>::to_float((uint16_t)(ww[e] >> 16));
        }
    }
    for (int c = 0; c < n_sel; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += hy[c][k] * u[k];
        low[(((size_t)b * n_sel + c) * S + Y) * S + X] = s;
    }
}

// ---- fused second transposed conv + GELU + hypernetwork product -------------------------------------
// mask_decoder.py:57-59,154-167:  up2 = GELU(ConvT2x2(u1)) ; masks = hyper . up2   (32 channels).
// The ConvT is a per-pixel GEMM with K = 64: rows = (prompt, token, sub-pixel 1) of `u1` [rows][64] ET,
// columns = sub-pixel 2 x 32 channels (W = [128][64], repacked at load).  K is so short that a GEMM kernel
// is all epilogue, and the 128-wide result would be written (134 MB for 32 prompts) only to be read back by
// the 32-channel dot.  Here one wave owns 16 rows at a time: W (64 registers) and the bias stay resident, the
// A fragments stream straight from global memory (16 rows x 128 contiguous bytes), GELU and the dot run in
// fp32 on the accumulators, the 4 partial sums of a row meet across the lane quarters, and lane quarter q
// writes low[b][c][Y][X] of sub-pixel 2 = q.   grid (row blocks per prompt, n_prompts), 256 threads.
constexpr int U2_GROUPS_PER_WAVE = 16;          // 16-row groups per wave -> 1024 rows per block

// SPLIT: `u1` is the fp32 output of the first transposed conv ([rows][64] floats) and is split into hi + lo operand
// fragments in registers, `w_lo` is the remainder of the weight split: three MFMAs per k-step, operand error ~2^-22
// (oracle/error_budget.py "only dec.up2": 395 of the 899 class-map pixels of the round-2 engine at ViT-H).
template <int PREC, int NSEL, bool SPLIT>
__global__ __launch_bounds__(256) void upscale2_mask_kernel(const void* __restrict__ u1v, const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ w_lo,
                                                            const float* __restrict__ bias, const float* __restrict__ hyper,
                                                            float* __restrict__ low, int grid, int n_mask_tokens, int sel0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.y;
    const int rows_per_prompt = grid * grid * 4;
    const int S = 4 * grid;
    uint4 wf[8][2], wl[SPLIT ? 8 : 1][2];
    float bv[8][4], hy[NSEL][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wf[i][ks] = *reinterpret_cast<const uint4*>(w + (i * 16 + fr) * 64 + ks * 32 + fq * 8);
            if constexpr (SPLIT) wl[i][ks] = *reinterpret_cast<const uint4*>(w_lo + (i * 16 + fr) * 64 + ks * 32 + fq * 8);
        }
        const float4 t = *reinterpret_cast<const float4*>(bias + i * 16 + 4 * fq);
        bv[i][0] = t.x; bv[i][1] = t.y; bv[i][2] = t.z; bv[i][3] = t.w;
    }
    // this lane's channels: c = (i & 1) * 16 + 4 fq + r  (the same for every sub-pixel 2 = i >> 1)
#pragma unroll
    for (int c = 0; c < NSEL; ++c) {
        const float* h = hyper + ((size_t)b * n_mask_tokens + sel0 + c) * 32;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float4 t = *reinterpret_cast<const float4*>(h + half * 16 + 4 * fq);
            hy[c][half * 4 + 0] = t.x; hy[c][half * 4 + 1] = t.y; hy[c][half * 4 + 2] = t.z; hy[c][half * 4 + 3] = t.w;
        }
    }
    const int row0 = (blockIdx.x * 4 + wave) * (16 * U2_GROUPS_PER_WAVE);     // first row (within the prompt) of this wave
    // fragments of a 16-row group: lane (fr, fq) holds columns 32 ks + 8 fq .. +7 of row fr (ET: 16 bytes, fp32: 32 bytes)
    const uint16_t* abase = reinterpret_cast<const uint16_t*>(u1v) + ((size_t)b * rows_per_prompt + row0 + fr) * 64 + fq * 8;
    const float* fbase = reinterpret_cast<const float*>(u1v) + ((size_t)b * rows_per_prompt + row0 + fr) * 64 + fq * 8;
    uint4 a0 = make_uint4(0u, 0u, 0u, 0u), a1 = a0;
    float4 f[SPLIT ? 4 : 1];
    if constexpr (SPLIT) {
        f[0] = *reinterpret_cast<const float4*>(fbase); f[1] = *reinterpret_cast<const float4*>(fbase + 4);
        f[2] = *reinterpret_cast<const float4*>(fbase + 32); f[3] = *reinterpret_cast<const float4*>(fbase + 36);
    } else {
        a0 = *reinterpret_cast<const uint4*>(abase); a1 = *reinterpret_cast<const uint4*>(abase + 32);
    }
    for (int g = 0; g < U2_GROUPS_PER_WAVE; ++g) {
        uint4 c0 = a0, c1 = a1, l0 = make_uint4(0u, 0u, 0u, 0u), l1 = l0;
        if constexpr (SPLIT) {
            split2_pack<PREC>(f[0].x, f[0].y, c0.x, l0.x); split2_pack<PREC>(f[0].z, f[0].w, c0.y, l0.y);
            split2_pack<PREC>(f[1].x, f[1].y, c0.z, l0.z); split2_pack<PREC>(f[1].z, f[1].w, c0.w, l0.w);
            split2_pack<PREC>(f[2].x, f[2].y, c1.x, l1.x); split2_pack<PREC>(f[2].z, f[2].w, c1.y, l1.y);
            split2_pack<PREC>(f[3].x, f[3].y, c1.z, l1.z); split2_pack<PREC>(f[3].z, f[3].w, c1.w, l1.w);
        }
        if (g + 1 < U2_GROUPS_PER_WAVE) {        // next group's fragments fly during this group's math
            if constexpr (SPLIT) {
                const float* nb = fbase + (size_t)(g + 1) * 16 * 64;
                f[0] = *reinterpret_cast<const float4*>(nb); f[1] = *reinterpret_cast<const float4*>(nb + 4);
                f[2] = *reinterpret_cast<const float4*>(nb + 32); f[3] = *reinterpret_cast<const float4*>(nb + 36);
            } else {
                a0 = *reinterpret_cast<const uint4*>(abase + (size_t)(g + 1) * 16 * 64);
                a1 = *reinterpret_cast<const uint4*>(abase + (size_t)(g + 1) * 16 * 64 + 32);
            }
        }
        float part[NSEL][4];
#pragma unroll
        for (int c = 0; c < NSEL; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[c][q] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {              // the small terms first, the leading one on top of them
                acc = ET<PREC>::mfma16(wf[i][0], l0, acc);
                acc = ET<PREC>::mfma16(wf[i][1], l1, acc);
                acc = ET<PREC>::mfma16(wl[i][0], c0, acc);
                acc = ET<PREC>::mfma16(wl[i][1], c1, acc);
            }
            acc = ET<PREC>::mfma16(wf[i][0], c0, acc);
            acc = ET<PREC>::mfma16(wf[i][1], c1, acc);
            const float2_t g01 = gelu_erf2(float2_t{acc[0] + bv[i][0], acc[1] + bv[i][1]});
            const float2_t g23 = gelu_erf2(float2_t{acc[2] + bv[i][2], acc[3] + bv[i][3]});
#pragma unroll
            for (int c = 0; c < NSEL; ++c) {
                const float* h = hy[c] + (i & 1) * 4;
                part[c][i >> 1] += h[0] * g01.x + h[1] * g01.y + h[2] * g23.x + h[3] * g23.y;
            }
        }
        // row of this lane: (token t, sub-pixel 1); quarter fq keeps sub-pixel 2 = fq
        const int row = row0 + g * 16 + fr;
        const int t = row >> 2, s1 = row & 3;
        const int y = t / grid, x = t % grid;
        const int Y = 4 * y + 2 * (s1 >> 1) + (fq >> 1), X = 4 * x + 2 * (s1 & 1) + (fq & 1);
#pragma unroll
        for (int c = 0; c < NSEL; ++c) {
            float mine = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = part[c][q];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                mine = (q == fq) ? v : mine;
            }
            low[(((size_t)b * NSEL + c) * S + Y) * S + X] = mine;
        }
    }
}

// ---- postprocess ------------------------------------------------------------------------
// PyTorch upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0,
// scale = in/out; out = wy0*(wx0*v00 + wx1*v01) + wy1*(wx0*v10 + wx1*v11).
struct Lin {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lin lin_coord(int dst, float scale, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lin r;
    r.i0 = (int)src;
    r.i0 = r.i0 < in_size - 1 ? r.i0 : in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.w1 = src - (float)r.i0;
    r.w0 = 1.0f - r.w1;
    return r;
}
// stage 1: value of the img_size^2 upsampled map at integer (Y1, X1), from the 256^2 logits
__device__ __forceinline__ float stage1(const float* __restrict__ low, int LS, float s1, int Y1, int X1) {
    const Lin ly = lin_coord(Y1, s1, LS), lx = lin_coord(X1, s1, LS);
    const float* r0 = low + (size_t)ly.i0 * LS;
    const float* r1 = low + (size_t)ly.i1 * LS;
    return ly.w0 * (lx.w0 * r0[lx.i0] + lx.w1 * r0[lx.i1]) + ly.w1 * (lx.w0 * r1[lx.i0] + lx.w1 * r1[lx.i1]);
}

// One thread = 4 horizontally adjacent output pixels.  grid (ceil(W/4 * H / 256), n_masks).
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ low_all, int LS, int in_h, int in_w,
                                                          int H, int W, int img_size, int return_logits,
                                                          void* __restrict__ out) {
    const int W4 = (W + 3) / 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)W4 * H) return;
    const int Y = (int)(t / W4), X0 = (int)(t % W4) * 4;
    const int mi = blockIdx.y;
    const float* low = low_all + (size_t)mi * LS * LS;
    const float s1 = (float)LS / (float)img_size;
    const bool identity = (H == in_h) && (W == in_w);
    float v[4];
    if (identity && 4 * LS == img_size && X0 + 3 < W) {
        // the case of every 1024^2 tile: scale exactly 1/4, so output columns 4k, 4k+1 interpolate between the same two
        // logits columns, and 4k+2, 4k+3 between the next pair; one row pair serves all four.  Same expression per
        // pixel as stage1 (bit-identical), 8 loads and 5 coordinate computations instead of 16 and 8.
        const Lin ly = lin_coord(Y, s1, LS);
        const float* r0 = low + (size_t)ly.i0 * LS;
        const float* r1 = low + (size_t)ly.i1 * LS;
        const Lin la = lin_coord(X0, s1, LS), lb = lin_coord(X0 + 1, s1, LS), lc = lin_coord(X0 + 2, s1, LS), ld = lin_coord(X0 + 3, s1, LS);
        const float a00 = r0[la.i0], a01 = r0[la.i1], a10 = r1[la.i0], a11 = r1[la.i1];
        const float c00 = r0[lc.i0], c01 = r0[lc.i1], c10 = r1[lc.i0], c11 = r1[lc.i1];
        v[0] = ly.w0 * (la.w0 * a00 + la.w1 * a01) + ly.w1 * (la.w0 * a10 + la.w1 * a11);
        v[1] = ly.w0 * (lb.w0 * a00 + lb.w1 * a01) + ly.w1 * (lb.w0 * a10 + lb.w1 * a11);
        v[2] = ly.w0 * (lc.w0 * c00 + lc.w1 * c01) + ly.w1 * (lc.w0 * c10 + lc.w1 * c11);
        v[3] = ly.w0 * (ld.w0 * c00 + ld.w1 * c01) + ly.w1 * (ld.w0 * c10 + ld.w1 * c11);
    } else if (identity) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (X0 + e < W) ? stage1(low, LS, s1, Y, X0 + e) : 0.f;
    } else {
        const float sy = (float)in_h / (float)H, sx = (float)in_w / (float)W;
        const Lin ly = lin_coord(Y, sy, in_h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (X0 + e < W) {
                const Lin lx = lin_coord(X0 + e, sx, in_w);
                const float a = lx.w0 * stage1(low, LS, s1, ly.i0, lx.i0) + lx.w1 * stage1(low, LS, s1, ly.i0, lx.i1);
                const float b = lx.w0 * stage1(low, LS, s1, ly.i1, lx.i0) + lx.w1 * stage1(low, LS, s1, ly.i1, lx.i1);
                v[e] = ly.w0 * a + ly.w1 * b;
            } else {
                v[e] = 0.f;
            }
        }
    }
    const size_t obase = ((size_t)mi * H + Y) * W + X0;
    if (return_logits) {
        float* o = reinterpret_cast<float*>(out) + obase;
        for (int e = 0; e < 4 && X0 + e < W; ++e) o[e] = v[e];
    } else {
        uint8_t* o = reinterpret_cast<uint8_t*>(out) + obase;
        if ((W & 3) == 0) {
            const uint32_t pk = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 1u << 8 : 0u) | (v[2] > 0.f ? 1u << 16 : 0u) |
                                (v[3] > 0.f ? 1u << 24 : 0u);
            *reinterpret_cast<uint32_t*>(o) = pk;
        } else {
            for (int e = 0; e < 4 && X0 + e < W; ++e) o[e] = v[e] > 0.f ? 1 : 0;
        }
    }
}

// ---- painting -----------------------------------------------------------------------------
// seg[p] = label of the LAST mask (in box order) covering p; untouched otherwise.
__global__ void paint_kernel(const uint8_t* __restrict__ masks, const int32_t* __restrict__ labels, int n, long hw,
                             uint8_t* __restrict__ seg) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    for (int j = n - 1; j >= 0; --j) {
        if (masks[(size_t)j * hw + p]) {
            seg[p] = (uint8_t)labels[j];
            return;
        }
    }
}
// areas[j] += popcount(mask j); grid (chunks, n)
__global__ void area_kernel(const uint8_t* __restrict__ masks, long hw, unsigned long long* __restrict__ areas) {
    const int j = blockIdx.y;
    const uint8_t* m = masks + (size_t)j * hw;
    unsigned int cnt = 0;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long)gridDim.x * blockDim.x) cnt += m[p] ? 1u : 0u;
    float c = (float)cnt;  // <= hw / (gridDim*blockDim) + 1, exact in fp32
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0 && c > 0.f) atomicAdd(&areas[j], (unsigned long long)c);
}
// Both of the above in ONE pass over the masks with 16-byte loads (the two kernels read every mask byte by byte, twice:
// 26 + 58 us for 32 masks of 1024^2 against ~10 us for the 32 MiB at HBM speed).  A thread owns 16 adjacent pixels:
// for j = n-1 .. 0 it counts the non-zero bytes of mask j (area) and paints the pixels no later mask has claimed.
// Integer arithmetic only: seg and areas are bit-identical to paint_kernel + area_kernel.
constexpr int PA_CHUNK = 64;          // masks per LDS counter chunk
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) {      // 0x80 in every byte of w that is not 0
    return (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
}
__global__ __launch_bounds__(256) void paint_area_kernel(const uint8_t* __restrict__ masks, const int32_t* __restrict__ labels,
                                                         int n, long hw, uint8_t* __restrict__ seg,
                                                         unsigned long long* __restrict__ areas) {
    __shared__ unsigned int cnt_s[PA_CHUNK];
    const long p0 = ((long)blockIdx.x * 256 + threadIdx.x) * 16;
    const bool in = p0 < hw;                              // hw % 16 == 0 (launcher)
    uint32_t done[4] = {0u, 0u, 0u, 0u}, res[4] = {0u, 0u, 0u, 0u};
    for (int jhi = n; jhi > 0; jhi -= PA_CHUNK) {
        const int jlo = jhi > PA_CHUNK ? jhi - PA_CHUNK : 0;
        if (threadIdx.x < PA_CHUNK) cnt_s[threadIdx.x] = 0u;
        __syncthreads();
        for (int j = jhi - 1; j >= jlo; --j) {
            unsigned int cnt = 0;
            if (in) {
                const uint4 m = *reinterpret_cast<const uint4*>(masks + (size_t)j * hw + p0);
                const uint32_t w[4] = {m.x, m.y, m.z, m.w};
                const uint32_t lab = (uint32_t)(uint8_t)labels[j] * 0x01010101u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t t = nonzero_bytes(w[k]);
                    cnt += __popc(t);
                    const uint32_t mm = (t >> 7) * 0xFFu, fresh = mm & ~done[k];
                    res[k] = (res[k] & ~fresh) | (fresh & lab);
                    done[k] |= mm;
                }
            }
            float c = wave_sum((float)cnt);               // <= 1024, exact
            if ((threadIdx.x & 63) == 0 && c > 0.f) atomicAdd(&cnt_s[j - jlo], (unsigned int)c);
        }
        __syncthreads();
        if (threadIdx.x < jhi - jlo && cnt_s[threadIdx.x]) atomicAdd(&areas[jlo + threadIdx.x], (unsigned long long)cnt_s[threadIdx.x]);
        __syncthreads();
    }
    if (in && (done[0] | done[1] | done[2] | done[3])) {
        uint4* sp = reinterpret_cast<uint4*>(seg + p0);
        uint4 o = *sp;
        o.x = (o.x & ~done[0]) | res[0]; o.y = (o.y & ~done[1]) | res[1];
        o.z = (o.z & ~done[2]) | res[2]; o.w = (o.w & ~done[3]) | res[3];
        *sp = o;
    }
}
// ---- instance drivers: keep the mask with the highest predicted IoU --------------------------------------------------------
// main_sam_*_mask_instance.py with multimask_output=True: of the nsel masks of an object keep the one with the highest
// predicted IoU (first maximum, like torch.argmax), its quality and its area.  One pass over the kept mask with 16-byte
// loads: copy + popcount.  grid (chunks, n); integer arithmetic only.
__global__ __launch_bounds__(256) void select_best_kernel(const uint8_t* __restrict__ masks, const float* __restrict__ iou, int nsel,
                                                          long hw, uint8_t* __restrict__ out, float* __restrict__ quality,
                                                          unsigned long long* __restrict__ areas) {
    const int j = blockIdx.y;
    int best = 0;
    float bq = iou[(size_t)j * nsel];
    for (int c = 1; c < nsel; ++c) {
        const float q = iou[(size_t)j * nsel + c];
        if (q > bq) { bq = q; best = c; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) quality[j] = bq;
    const uint8_t* src = masks + ((size_t)j * nsel + best) * hw;
    uint8_t* dst = out + (size_t)j * hw;
    unsigned int cnt = 0;
    const bool vec = (hw % 16 == 0) && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    if (vec) {
        const long n16 = hw / 16;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
            const uint4 m = reinterpret_cast<const uint4*>(src)[i];
            reinterpret_cast<uint4*>(dst)[i] = m;
            cnt += __popc(nonzero_bytes(m.x)) + __popc(nonzero_bytes(m.y)) + __popc(nonzero_bytes(m.z)) + __popc(nonzero_bytes(m.w));
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < hw; i += (long)gridDim.x * 256) {
            const uint8_t m = src[i];
            dst[i] = m;
            cnt += m ? 1u : 0u;
        }
    }
    const float c = wave_sum((float)cnt);               // <= 64 * 16 * iterations: keep it exact -> few iterations per thread
    if ((threadIdx.x & 63) == 0 && c > 0.f) atomicAdd(&areas[j], (unsigned long long)c);
}

__global__ void class_stats_kernel(const unsigned long long* __restrict__ areas, const int32_t* __restrict__ labels, int n,
                                   unsigned long long* __restrict__ cpix, unsigned long long* __restrict__ cins, int n_classes) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int j = 0; j < n; ++j) {
        const int l = labels[j];
        if (areas[j] > 0 && l >= 0 && l < n_classes) {   // statistic.py:18-21 (area > 0 only)
            if (cpix) cpix[l] += areas[j];
            if (cins) cins[l] += 1ull;
        }
    }
}

// ---- PIL-compatible separable resample pass (next-row N3) ------------------------------------
// Pillow's ImagingResample{Horizontal,Vertical}_8bpc: out = clip8((2^21 + sum_k in[k0 + k] * coef[k]) >> 22)
// with per-output-index bounds (first input index, tap count) and 22-bit fixed-point coefficients
// computed on the host exactly like Pillow's precompute_coeffs / normalize_coeffs_8bpc.
// `horizontal`: in [rows, in_len, 3] -> out [rows, out_len, 3]; else in [in_len, cols, 3] -> out [out_len, cols, 3].
__global__ void resample_pass_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int32_t* __restrict__ bounds,
                                     const int32_t* __restrict__ coef, int ksize, int in_len, int out_len, int other,
                                     int horizontal) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)out_len * other * 3;
    if (t >= total) return;
    const int c = t % 3;
    long u = t / 3;
    int o, q;            // o: index along the resampled axis, q: index along the other axis
    if (horizontal) { o = u % out_len; q = u / out_len; } else { q = u % other; o = u / other; }
    const int k0 = bounds[2 * o], kn = bounds[2 * o + 1];
    const int32_t* k = coef + (size_t)o * ksize;
    int ss = 1 << 21;
    for (int i = 0; i < kn; ++i) {
        const size_t idx = horizontal ? ((size_t)q * in_len + (k0 + i)) * 3 + c : ((size_t)(k0 + i) * other + q) * 3 + c;
        ss += (int)in[idx] * k[i];
    }
    ss >>= 22;
    ss = ss < 0 ? 0 : (ss > 255 ? 255 : ss);
    out[t] = (uint8_t)ss;
}

// ---- rotated-box mask prompt (N3: replaces cv2.fillPoly / resize / copyMakeBorder) ---------------------
// `Generate Dataset/main_sam_rbox_mask_instance.py:125-141`: polygon -> +-1000 mask at the image size ->
// bilinear resize to the ResizeLongestSide shape -> pad to img_size^2 with -1000 -> bilinear resize to 256^2.
// One thread per OUTPUT pixel composes the two resizes on the fly (2x2 taps of 2x2 taps = 16 polygon tests),
// so no intermediate image exists.  Arithmetic follows OpenCV's published algorithms (see oracle/rbox_prompt.py
// for the citations): float32 tap weights from double coordinates, double accumulation, horizontal pass before
// vertical pass; polygon = FillEdgeCollection spans (16.16 fixed point, ceil(left) .. floor(right)) united with
// the 8-connected LineIterator walk of every edge, here in closed form:
//   after i major steps the walk has taken  floor((2 * dminor * i + dmajor - 1) / (2 * dmajor))  minor steps.
// FP contraction is off: the products of the second stage are not exact, an fma would round differently.
constexpr int RBOX_MAXV = 8;
struct RboxPoly {
    // fill rule of FillEdgeCollection's spans: 0 = OpenCV <= 4.5.1 (ceil(x_left) .. floor(x_right)), 1 = OpenCV >= 4.5.2 (round half up on
    // both sides: the edge x gets XY_ONE >> 1 added).  The boundary lines are the same under both.
    int rule;
    int x[RBOX_MAXV], y[RBOX_MAXV];
    long long ex[RBOX_MAXV], edx[RBOX_MAXV];      // per edge (v-1 -> v): x at y0 (16.16), dx per scanline
    int ey0[RBOX_MAXV], ey1[RBOX_MAXV];           // scanline range [y0, y1); y0 == y1 for horizontal edges
    int nv;
};

__device__ __forceinline__ bool rbox_on_line(int px, int py, int x1, int y1, int x2, int y2) {
    if (x2 < x1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }     // leftToRight
    const int dx = x2 - x1, sdy = y2 - y1, dy = sdy < 0 ? -sdy : sdy, ys = sdy < 0 ? -1 : 1;
    if (dy > dx) {                               // y major
        const int i = (py - y1) * ys;
        if (i < 0 || i > dy) return false;
        return px == x1 + (int)((2LL * dx * i + dy - 1) / (2LL * dy));
    }
    const int i = px - x1;
    if (i < 0 || i > dx) return false;
    if (dx == 0) return py == y1;                // a single point
    return py == y1 + ys * (int)((2LL * dy * i + dx - 1) / (2LL * dx));
}

__device__ bool rbox_inside(const RboxPoly& P, int px, int py) {
    bool in = false;
    long long xs[RBOX_MAXV];
    int na = 0;
    for (int e = 0; e < P.nv; ++e) {
        const int a = e == 0 ? P.nv - 1 : e - 1;
        if (rbox_on_line(px, py, P.x[a], P.y[a], P.x[e], P.y[e])) in = true;
        if (P.ey0[e] <= py && py < P.ey1[e]) {
            const long long xe = P.ex[e] + (long long)(py - P.ey0[e]) * P.edx[e];
            int k = na++;
            while (k > 0 && xs[k - 1] > xe) { xs[k] = xs[k - 1]; --k; }      // insertion sort (<= 8 entries)
            xs[k] = xe;
        }
    }
    for (int k = 0; k + 1 < na; k += 2) {
        const long long xa = P.rule ? (xs[k] + 32768) >> 16 : (xs[k] + 65535) >> 16;
        const long long xb = P.rule ? (xs[k + 1] + 32768) >> 16 : xs[k + 1] >> 16;
        if (xa <= px && px <= xb) in = true;
    }
    return in;
}

struct RTap { int i0, i1; float w0, w1; };
__device__ __forceinline__ RTap rbox_tap(int d, int n_in, int n_out) {
#pragma clang fp contract(off)
    const double inv = (double)n_out / (double)n_in;
    const double scale = 1.0 / inv;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_in - 1) { f = 0.f; s = n_in - 1; }
    RTap t;
    t.i0 = s;
    t.i1 = s + 1 < n_in ? s + 1 : n_in - 1;
    t.w0 = 1.0f - f;
    t.w1 = f;
    return t;
}

// grid (out*out / 256, n_boxes); pts int32 [n][nv][2] (x, y)
__global__ __launch_bounds__(256) void rbox_prompt_kernel(const int32_t* __restrict__ pts, int nv, int h, int w, int th, int tw,
                                                          int img_size, int out_size, float* __restrict__ out, int fill_rule) {
#pragma clang fp contract(off)
    __shared__ RboxPoly P;
    const int b = blockIdx.y;
    if (threadIdx.x == 0) {
        P.nv = nv;
        P.rule = fill_rule;
        for (int e = 0; e < nv; ++e) { P.x[e] = pts[((size_t)b * nv + e) * 2]; P.y[e] = pts[((size_t)b * nv + e) * 2 + 1]; }
        for (int e = 0; e < nv; ++e) {
            const int a = e == 0 ? nv - 1 : e - 1;
            const long long x0 = P.x[a], y0 = P.y[a], x1 = P.x[e], y1 = P.y[e];
            if (y0 == y1) { P.ey0[e] = P.ey1[e] = 0; P.ex[e] = P.edx[e] = 0; continue; }
            P.edx[e] = ((x1 - x0) << 16) / (y1 - y0);                 // C++ division: truncates toward zero
            if (y0 < y1) { P.ey0[e] = (int)y0; P.ey1[e] = (int)y1; P.ex[e] = x0 << 16; }
            else { P.ey0[e] = (int)y1; P.ey1[e] = (int)y0; P.ex[e] = x1 << 16; }
        }
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= out_size * out_size) return;
    const int oy = pix / out_size, ox = pix % out_size;
    const RTap by = rbox_tap(oy, img_size, out_size), bx = rbox_tap(ox, img_size, out_size);
    const int Ys[2] = {by.i0, by.i1}, Xs[2] = {bx.i0, bx.i1};
    double pv[2][2];
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) {
            const int Y = Ys[a], X = Xs[c];
            if (Y >= th || X >= tw) { pv[a][c] = -1000.0; continue; }      // copyMakeBorder(value = -1000)
            const RTap ay = rbox_tap(Y, h, th), ax = rbox_tap(X, w, tw);
            const double m00 = rbox_inside(P, ax.i0, ay.i0) ? 1000.0 : -1000.0;
            const double m01 = rbox_inside(P, ax.i1, ay.i0) ? 1000.0 : -1000.0;
            const double m10 = rbox_inside(P, ax.i0, ay.i1) ? 1000.0 : -1000.0;
            const double m11 = rbox_inside(P, ax.i1, ay.i1) ? 1000.0 : -1000.0;
            const double r0 = m00 * (double)ax.w0 + m01 * (double)ax.w1;      // horizontal pass
            const double r1 = m10 * (double)ax.w0 + m11 * (double)ax.w1;
            pv[a][c] = r0 * (double)ay.w0 + r1 * (double)ay.w1;              // vertical pass
        }
    const double t0 = pv[0][0] * (double)bx.w0 + pv[0][1] * (double)bx.w1;
    const double t1 = pv[1][0] * (double)bx.w0 + pv[1][1] * (double)bx.w1;
    out[(size_t)b * out_size * out_size + pix] = (float)(t0 * (double)by.w0 + t1 * (double)by.w1);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
hipError_t launch_resample_pass(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* coef, int ksize,
                                int in_len, int out_len, int other, int horizontal, hipStream_t s) {
    const long total = (long)out_len * other * 3;
    resample_pass_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(in, out, bounds, coef, ksize, in_len, out_len, other, horizontal);
    return hipGetLastError();
}

hipError_t launch_prompt_tokens(const PromptParams& p, float* tokens, float* tokens2, int T, hipStream_t s) {
    dim3 g(T, p.n_prompts), b(256);
    prompt_tokens_kernel<<<g, b, 0, s>>>(p, tokens, tokens2, T, 256);
    return hipGetLastError();
}
hipError_t launch_dense_pe(const float* gauss, float* pe, int grid, hipStream_t s) {
    dense_pe_kernel<<<grid * grid, 256, 0, s>>>(gauss, pe, grid, 256);
    return hipGetLastError();
}
hipError_t launch_mask_embed(const MaskEmbedParams& p, const float* mask_in, float* dense, int n, int grid, hipStream_t s) {
    dim3 g(grid * grid / 16, n);
    mask_embed_kernel<<<g, 256, 0, s>>>(p, mask_in, dense, grid);
    return hipGetLastError();
}
hipError_t launch_make_keys(int prec, const float* emb, const float* dense, const float* vec, float* out_f32,
                            void* out_et, int n_batches, int tokens, int C, hipStream_t s) {
    const long per4 = (long)tokens * C / 4, tot4 = per4 * n_batches;
    const int blocks = (int)((tot4 + 255) / 256);
    if (prec == PREC_BF16)
        make_keys_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(emb, dense, vec, out_f32, (uint16_t*)out_et, per4, tot4, C / 4);
    else
        make_keys_kernel<PREC_F16><<<blocks, 256, 0, s>>>(emb, dense, vec, out_f32, (uint16_t*)out_et, per4, tot4, C / 4);
    return hipGetLastError();
}
hipError_t launch_add_f32(const float* a, const float* b, float* out, long n, hipStream_t s) {
    if (n % 4) return hipErrorInvalidValue;
    add_f32_kernel<<<(int)((n / 4 + 255) / 256), 256, 0, s>>>(a, b, out, n / 4);
    return hipGetLastError();
}
hipError_t launch_token_self_attn(const float* q, const float* k, const float* v, float* o, int n, int T, int C,
                                  int heads, hipStream_t s) {
    if (C / heads != 32 || T > TOK_MAX) return hipErrorInvalidValue;
    token_self_attn_kernel<32><<<n, 256, 3 * T * C * sizeof(float), s>>>(q, k, v, o, T, C, heads);
    return hipGetLastError();
}
size_t t2i_workspace_floats(int n, int T) {
    return (size_t)n * ((T + T2I_TG - 1) / T2I_TG) * T2I_MAX_SPLITS * (T2I_TG * 16) * T2I_REC;
}
hipError_t launch_t2i_attention(int prec, const float* qp, const void* kp, const void* vp, int ld, long bstride,
                                float* o, float* workspace, int n, int T, int tokens, int Ci, int heads, hipStream_t s) {
    if (Ci != 128 || heads != 8 || T > TOK_MAX || !workspace) return hipErrorInvalidValue;
    // keys per wave: a multiple of 16 (4 groups of 4 keys per iteration), at most T2I_MAX_SPLITS blocks of 4 waves
    int splits = (tokens + 255) / 256;
    splits = splits < 1 ? 1 : (splits > T2I_MAX_SPLITS ? T2I_MAX_SPLITS : splits);
    const int kpw = ((tokens + splits * 4 - 1) / (splits * 4) + 15) / 16 * 16;
    const int tgs = (T + T2I_TG - 1) / T2I_TG;
    dim3 g(splits, n, tgs), b(256);
    const uint16_t* k = (const uint16_t*)kp;
    const uint16_t* v = (const uint16_t*)vp;
    if (prec == PREC_BF16) t2i_partial_kernel<PREC_BF16><<<g, b, 0, s>>>(qp, k, v, ld, bstride, workspace, T, tokens, Ci, kpw);
    else t2i_partial_kernel<PREC_F16><<<g, b, 0, s>>>(qp, k, v, ld, bstride, workspace, T, tokens, Ci, kpw);
    t2i_merge_kernel<<<dim3(n, tgs), T2I_TG * 16, 0, s>>>(workspace, o, splits, T, Ci);
    return hipGetLastError();
}
hipError_t launch_i2t_attention(int prec, const void* qi, int ld, long bstride, const float* kt, const float* vt,
                                void* out, int n, int T, int tokens, int Ci, int heads, hipStream_t s) {
    if (Ci / heads != 16 || 256 % heads) return hipErrorInvalidValue;
    const int per = 256 / heads;
    dim3 g((tokens + per - 1) / per, n), b(256);
    const size_t sh = 2 * (size_t)T * heads * I2T_HS * sizeof(float);
    if (prec == PREC_BF16)
        i2t_attention_kernel<PREC_BF16><<<g, b, sh, s>>>((const uint16_t*)qi, ld, bstride, kt, vt, (uint16_t*)out, T, tokens, Ci);
    else
        i2t_attention_kernel<PREC_F16><<<g, b, sh, s>>>((const uint16_t*)qi, ld, bstride, kt, vt, (uint16_t*)out, T, tokens, Ci);
    return hipGetLastError();
}
hipError_t launch_i2t_fused(int prec, const void* qi, int ld, long q_bstride, const float* kt, const float* vt, const void* w,
                            const void* w_lo, const float* bias, const float* resid, long r_bstride, const float* gamma,
                            const float* beta, float eps, float* outF, void* outE, void* outE_lo, int n, int T, int tokens, int Ci,
                            int C, hipStream_t s) {
    if (Ci != 128 || C != 256 || T < 1 || T > TOK_MAX || tokens % I2TF_ROWS) return hipErrorInvalidValue;
    const int groups = tokens / I2TF_ROWS;
    int gpb = 4;                                   // groups per block: W fragments (64 / 128 KB per block) are loaded once
    while (gpb > 1 && groups % gpb) gpb >>= 1;
    dim3 g(groups / gpb, n), b(256);
    const bool split = w_lo != nullptr;
    const size_t sh = (size_t)(2 * T * 8 * I2T_HS + 3 * 256 + 2 * 4 * I2TF_ROWS) * sizeof(float) +
                      (size_t)(split ? 2 : 1) * I2TF_ROWS * I2TF_AST * 2;
#define I2TF_LAUNCH(P, S)                                                                                                       \
    i2t_fused_kernel<P, S><<<g, b, sh, s>>>((const uint16_t*)qi, ld, q_bstride, kt, vt, (const uint16_t*)w, (const uint16_t*)w_lo, \
                                           bias, resid, r_bstride, gamma, beta, eps, outF, (uint16_t*)outE, (uint16_t*)outE_lo, T, \
                                           tokens, gpb)
    if (prec == PREC_BF16) { if (split) I2TF_LAUNCH(PREC_BF16, true); else I2TF_LAUNCH(PREC_BF16, false); }
    else { if (split) I2TF_LAUNCH(PREC_F16, true); else I2TF_LAUNCH(PREC_F16, false); }
#undef I2TF_LAUNCH
    return hipGetLastError();
}
hipError_t launch_group_ln_gelu(int prec, const float* in, const float* gamma, const float* beta, float eps,
                                void* out, long rows, int groups, int gsize, hipStream_t s) {
    if (groups != 4 || gsize != 64) return hipErrorInvalidValue;
    const int blocks = (int)((rows + 3) / 4);
    if (prec == PREC_BF16) group_ln_gelu_kernel<PREC_BF16><<<blocks, 256, 0, s>>>(in, gamma, beta, eps, (uint16_t*)out, rows);
    else group_ln_gelu_kernel<PREC_F16><<<blocks, 256, 0, s>>>(in, gamma, beta, eps, (uint16_t*)out, rows);
    return hipGetLastError();
}
hipError_t launch_mask_product(int prec, const void* up2, const float* hyper, float* low, int n, int grid,
                               int n_mask_tokens, int sel0, int n_sel, hipStream_t s) {
    if (n_sel > 4 || n_sel < 1) return hipErrorInvalidValue;
    const int S = 4 * grid;
    dim3 g(S * S / 256, n), b(256);
    if (prec == PREC_BF16) mask_product_kernel<PREC_BF16><<<g, b, 0, s>>>((const uint16_t*)up2, hyper, low, grid, n_mask_tokens, sel0, n_sel);
    else mask_product_kernel<PREC_F16><<<g, b, 0, s>>>((const uint16_t*)up2, hyper, low, grid, n_mask_tokens, sel0, n_sel);
    return hipGetLastError();
}
hipError_t launch_upscale2_masks(int prec, const void* u1, const void* w, const void* w_lo, const float* bias, const float* hyper,
                                 float* low, int n, int grid, int n_mask_tokens, int sel0, int n_sel, hipStream_t s) {
    const int rows_per_block = 4 * 16 * U2_GROUPS_PER_WAVE;
    if ((grid * grid * 4) % rows_per_block || (n_sel != 1 && n_sel != 3)) return hipErrorInvalidValue;
    dim3 g(grid * grid * 4 / rows_per_block, n), b(256);
    const uint16_t* ww = (const uint16_t*)w;
    const uint16_t* wl = (const uint16_t*)w_lo;
#define U2_LAUNCH(P, NS, SP) upscale2_mask_kernel<P, NS, SP><<<g, b, 0, s>>>(u1, ww, wl, bias, hyper, low, grid, n_mask_tokens, sel0)
#define U2_SEL(P, SP) do { if (n_sel == 1) U2_LAUNCH(P, 1, SP); else U2_LAUNCH(P, 3, SP); } while (0)
    if (prec == PREC_BF16) { if (w_lo) U2_SEL(PREC_BF16, true); else U2_SEL(PREC_BF16, false); }
    else { if (w_lo) U2_SEL(PREC_F16, true); else U2_SEL(PREC_F16, false); }
#undef U2_SEL
#undef U2_LAUNCH
    return hipGetLastError();
}
hipError_t launch_postprocess(const float* low, int n_masks, int in_h, int in_w, int orig_h, int orig_w,
                              int img_size, int return_logits, void* out, hipStream_t s) {
    const long work = (long)((orig_w + 3) / 4) * orig_h;
    dim3 g((unsigned)((work + 255) / 256), n_masks), b(256);
    postprocess_kernel<<<g, b, 0, s>>>(low, img_size / 4, in_h, in_w, orig_h, orig_w, img_size, return_logits, out);
    return hipGetLastError();
}
hipError_t launch_paint(const uint8_t* masks, const int32_t* labels, int n, int h, int w, uint8_t* seg,
                        unsigned long long* areas, unsigned long long* class_pixels,
                        unsigned long long* class_instances, int n_classes, hipStream_t s) {
    const long hw = (long)h * w;
    const bool fused = seg && areas && hw % 16 == 0 && (((uintptr_t)masks | (uintptr_t)seg) & 15) == 0;
    if (seg && !fused) paint_kernel<<<(int)((hw + 255) / 256), 256, 0, s>>>(masks, labels, n, hw, seg);
    if (areas) {
        HIP_CHECK_RET(hipMemsetAsync(areas, 0, sizeof(unsigned long long) * n, s));
        if (fused) {
            paint_area_kernel<<<(int)((hw / 16 + 255) / 256), 256, 0, s>>>(masks, labels, n, hw, seg, areas);
        } else {
            dim3 g(64, n);
            area_kernel<<<g, 256, 0, s>>>(masks, hw, areas);
        }
        if (class_pixels || class_instances)
            class_stats_kernel<<<1, 64, 0, s>>>(areas, labels, n, class_pixels, class_instances, n_classes);
    }
    return hipGetLastError();
}

hipError_t launch_select_best(const uint8_t* masks, const float* iou, int n, int nsel, int h, int w, uint8_t* out, float* quality,
                              unsigned long long* areas, hipStream_t s) {
    if (n < 1 || nsel < 1 || h < 1 || w < 1) return hipErrorInvalidValue;
    const long hw = (long)h * w;
    HIP_CHECK_RET(hipMemsetAsync(areas, 0, sizeof(unsigned long long) * n, s));
    // <= 2^24 / 1024 iterations per thread keeps the fp32 wave sum exact; 64 chunks of 256 threads x 16 bytes cover 1024^2 in 4
    int chunks = (int)((hw / 16 + 255) / 256);
    chunks = chunks < 1 ? 1 : (chunks > 64 ? 64 : chunks);
    select_best_kernel<<<dim3(chunks, n), 256, 0, s>>>(masks, iou, nsel, hw, out, quality, areas);
    return hipGetLastError();
}

hipError_t launch_rbox_prompt(const int32_t* pts, int n, int nv, int h, int w, int th, int tw, int img_size, int out_size,
                              float* out, hipStream_t s, int fill_rule) {
    if (n < 1 || nv < 3 || nv > RBOX_MAXV || h < 1 || w < 1 || th < 1 || tw < 1 || th > img_size || tw > img_size || out_size < 1 ||
        fill_rule < 0 || fill_rule > 1)
        return hipErrorInvalidValue;
    dim3 g((out_size * out_size + 255) / 256, n), b(256);
    rbox_prompt_kernel<<<g, b, 0, s>>>(pts, nv, h, w, th, tw, img_size, out_size, out, fill_rule);
    return hipGetLastError();
}
