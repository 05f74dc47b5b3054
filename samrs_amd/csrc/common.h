// common.h -- shared device helpers for libsamrs_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define SAMRS_WAVE 64

enum { PREC_BF16 = 0, PREC_F16 = 1 };

// ---------------------------------------------------------------------------------------------
// MFMA operand element type ("ET"): bf16 or f16, carried around as raw uint16 bit patterns so
// that LDS / global traffic is type-agnostic 16-byte vectors.
// ---------------------------------------------------------------------------------------------
template <int PREC>
struct ET;

template <>
struct ET<PREC_BF16> {
    static __device__ __forceinline__ uint16_t from_float(float f) {
        __bf16 b = (__bf16)f;  // RNE, v_cvt_pk_bf16_f32 on gfx950
        return __builtin_bit_cast(uint16_t, b);
    }
    static __device__ __forceinline__ float to_float(uint16_t u) {
        return __builtin_bit_cast(float, (uint32_t)u << 16);
    }
    static __device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                       __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};

template <>
struct ET<PREC_F16> {
    static __device__ __forceinline__ uint16_t from_float(float f) {
        // saturate instead of producing inf: fp16 max is 65504
        f = __builtin_fminf(__builtin_fmaxf(f, -65504.0f), 65504.0f);
        _Float16 h = (_Float16)f;  // RNE
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ float to_float(uint16_t u) {
        return (float)__builtin_bit_cast(_Float16, u);
    }
    static __device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16_t mfma32(uint4 a, uint4 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                      __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bfloat2_t __attribute__((ext_vector_type(2)));

// two floats -> one packed ET word, RNE, NO saturation (v_cvt_pk_{f16,bf16}_f32): for values known
// to be in range (softmax probabilities, normalised attention outputs).
template <int PREC>
__device__ __forceinline__ uint32_t pack2_fast(float lo, float hi) {
    const float2_t f = {lo, hi};
    if (PREC == PREC_F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, half2_t));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bfloat2_t));
}
// general version: f16 saturates at +-65504 instead of overflowing to inf
template <int PREC>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if (PREC == PREC_F16) {
        lo = __builtin_amdgcn_fmed3f(lo, -65504.0f, 65504.0f);
        hi = __builtin_amdgcn_fmed3f(hi, -65504.0f, 65504.0f);
    }
    return pack2_fast<PREC>(lo, hi);
}

// Two-term split of an fp32 value into MFMA operands: v ~= hi + lo with hi = ET(v), lo = ET(v - hi).  A product of two
// split operands as three MFMAs (hi hi + lo hi + hi lo, fp32 accumulate) carries ~2^-22 relative operand error instead of
// 2^-11; used where the error budget (oracle/error_budget.py) says a rounding point is expensive in
// mask pixels and cheap in FLOPs: patch embed, neck, and the decoder's out-projection / upscaler operands.
template <int PREC>
__device__ __forceinline__ void split2_pack(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack2<PREC>(a, b);
    const float ha = ET<PREC>::to_float((uint16_t)(hi & 0xffffu)), hb = ET<PREC>::to_float((uint16_t)(hi >> 16));
    // saturating as well: where |v| exceeds the f16 range hi = +-65504 and v - hi can exceed it too (the raw residual
    // stream in front of the neck is the one operand without a LayerNorm); an inf in lo would poison the accumulator.
    // The 2^-22 claim holds for 2^-14 * 2^11 <= |v| <= 65504 in f16 (lo = a multiple of hi's ulp / 2^11 must still be a
    // normal or subnormal f16: below |v| ~ 2^-3 the lo term loses bits one by one and is 0 under |v| ~ 2^-14); bf16 has the
    // fp32 exponent range and no such floor.
    lo = pack2<PREC>(a - ha, b - hb);
}

// ---------------------------------------------------------------------------------------------
// MXFP4 operands of the split products' correction terms (gemm.hip gemm_et_mx_kernel): e2m1 codes, one E8M0 scale per
// 32 elements, the scale bytes stored in the order the GEMM's lanes read them.  Shared by the GEMM, the generic pack
// kernel and the producers that emit the format themselves (LayerNorm).
// ---------------------------------------------------------------------------------------------
constexpr int MXK = 256;                       // k per MX stage (128 bytes of fp4 per row: the f16 stage's geometry)
constexpr int MX_SA_BYTES = 2048, MX_SB_BYTES = 4096;      // scale tile of one (256-row A tile | 320-row B tile, stage)

// ---- fp4 (e2m1) quantisation of one value that has been divided by its block scale: the nearest point of
// {0, 0.5, 1, 1.5, 2, 3, 4, 6} (ties to even; |v| > 6 saturates, which can only hit the block maximum: the scale puts it in [4, 8)),
// exactly what oracle/sam_oracle.py split_fp8_lo._q8 computes for fmt="e2m1" ----
__device__ __forceinline__ uint32_t fp4_code(float v) {
    const float a = fabsf(v);
    const float step = a >= 4.f ? 2.f : a >= 2.f ? 1.f : 0.5f;
    const float q = fminf(rintf(a / step) * step, 6.f);
    const int idx = (int)(q * 2.f);                            // 0 1 2 3 4 6 8 12
    const uint32_t code = idx <= 4 ? (uint32_t)idx : (uint32_t)(idx >> 2) + 4u;
    return code | (v < 0.f ? 8u : 0u);
}
// E8M0 exponent of a block: floor(log2(amax)) - 2 (OCP MX: emax of e2m1 is 2), as a biased byte; amax = 0 / subnormal -> 2^-127
__device__ __forceinline__ int mx_scale_byte(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);            // biased fp32 exponent = floor(log2) + 127 for normal values
    return e > 2 ? e - 2 : 0;
}
__device__ __forceinline__ float mx_inv_scale(int byte) {                  // 2^-(byte - 127), exact
    return __uint_as_float((uint32_t)(254 - byte) << 23);                  // byte in [0, 252]: exponent field 254 - byte in [2, 254]
}
// byte offset of the scale of (row r, padded-k block b) in the tiled scale tensors described above
__device__ __forceinline__ size_t mx_scale_index(bool is_b, int r, int b, int nst4) {
    const int st = b >> 3, kh = (b >> 2) & 1, fq = b & 3;
    if (!is_b) {
        const int tile = r >> 8, wm = (r >> 7) & 1, j = (r >> 4) & 7, fr = r & 15;
        return ((size_t)tile * nst4 + st) * MX_SA_BYTES + (size_t)(((kh * 2 + wm) * 64 + fq * 16 + fr) * 8 + j);
    }
    const int tile = r / 320, rr = r % 320, wn = rr / 80, i = (rr % 80) >> 4, fr = rr & 15;
    return ((size_t)tile * nst4 + st) * MX_SB_BYTES + (size_t)(((kh * 4 + wn) * 64 + fq * 16 + fr) * 8 + i);
}

// four / eight values of one block -> packed fp4 codes (element 0 in the low nibble), given the block's E8M0 scale byte.
// SAMRS_FP4_HWCVT = 1: gfx950's v_cvt_scalef32_pk_fp4_f32 (two values per instruction; tools/mx_probe.hip checks on the device
// that it is bit-identical with fp4_code(v / scale) incl. ties and saturation); 0: the software quantiser (~15 VALU per value:
// +75 us per windowed-attention launch when that kernel emits its MX rows with it).
#ifndef SAMRS_FP4_HWCVT
#define SAMRS_FP4_HWCVT 1
#endif
__device__ __forceinline__ uint32_t fp4_pack4(float a, float b, float c, float d, int scale_byte) {
#if SAMRS_FP4_HWCVT
    const float sc = __uint_as_float((uint32_t)scale_byte << 23);              // 2^(byte - 127); byte 0: the hardware reads the exponent field
    uint32_t r = 0u;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, a, b, sc, 0);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, c, d, sc, 1);
    return r & 0xffffu;
#else
    const float inv = mx_inv_scale(scale_byte);
    return fp4_code(a * inv) | (fp4_code(b * inv) << 4) | (fp4_code(c * inv) << 8) | (fp4_code(d * inv) << 12);
#endif
}
__device__ __forceinline__ uint32_t fp4_pack8(const float* v, int scale_byte) {
#if SAMRS_FP4_HWCVT
    const float sc = __uint_as_float((uint32_t)scale_byte << 23);
    uint32_t r = 0u;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[0], v[1], sc, 0);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[2], v[3], sc, 1);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[4], v[5], sc, 2);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(r, v[6], v[7], sc, 3);
    return r;
#else
    return fp4_pack4(v[0], v[1], v[2], v[3], scale_byte) | (fp4_pack4(v[4], v[5], v[6], v[7], scale_byte) << 16);
#endif
}

// optional MX outputs of a producer kernel (all null: none)
struct MxOut {
    unsigned char *q_hi = nullptr, *q_lo = nullptr, *s_hi = nullptr, *s_lo = nullptr;
};

// value of lane l ^ 16 / l ^ 32 (v_permlane16_swap / v_permlane32_swap: VALU cross-lane moves, no LDS crossbar)
__device__ __forceinline__ float lane_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return (threadIdx.x & 16) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

// ---------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// exact-erf GELU (nn.GELU default; Generate Dataset/segment_anything/modeling/common.py:18-26).
// erfc(z) for z >= 0 from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-epsilon class):
//   erfc(z) = t*(a1 + t*(a2 + t*(a3 + t*(a4 + t*a5)))) * exp(-z^2),  t = 1/(1 + p z)
// and gelu(x) = x - 0.5*x*erfc(z) for x >= 0, 0.5*x*erfc(z) for x < 0, z = |x|/sqrt(2): no
// cancellation in the negative tail, ~12 VALU ops instead of libm erff's two-branch ~35.
// Written two elements wide on ext-vector floats so that the polynomial runs on v_pk_fma_f32 /
// v_pk_mul_f32 (2 fp32 per lane per issue), with the raw v_rcp_f32 / v_exp_f32 (1 ulp, no range
// fix-ups: 1 + p z is in [1, inf) and the exp2 argument is <= 0, where underflow to 0 is the right
// answer).  The GEMM epilogues that apply it are VALU-bound on exactly this function.
//   |h| = 0.5 |x| erfc(z),  gelu(x) = max(x, 0) - |h|      (x >= 0: x - h;  x < 0: h)
__device__ __forceinline__ float2_t gelu_erf2(float2_t x) {
    const float2_t ax = {fabsf(x.x), fabsf(x.y)};
    const float2_t z = ax * 0.70710678118654752440f;
    const float2_t d = z * 0.3275911f + 1.0f;
    const float2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const float2_t poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float2_t arg = z * (z * -1.44269504088896340736f);          // -z^2 log2(e)
    const float2_t ex = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
    const float2_t h = (z * 0.70710678118654752440f) * poly * ex;
    const float2_t r = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return r - h;
}
// The same function for values that are ROUNDED TO THE MFMA OPERAND TYPE right away (the ET-output GEMM epilogues: lin1 of every
// encoder block, image_encoder.py:177-180 / common.py:18-26 -- 168 M elements per launch, 16 % of the dominant kernel with the
// form above, executed while the block's matrix pipe idles).  erf by Abramowitz-Stegun 7.1.28,
//     erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16,  |error| <= 3e-7,
// with the 1/sqrt(2) of z = |x| / sqrt(2) folded into the coefficients: six packed fmas, ONE v_rcp_f32 per element (no exp2),
// four packed squarings, and   gelu(x) = 0.5 (x + |x| erf(z)) = 0.5 (x + u (1 - r^16))   -- three packed ops, no max.
// 19 VALU issues per element pair against 25 (two transcendentals per element instead of four).  Absolute error vs the
// exact function <= 9e-7 (fp32 emulation over [-12, 12], tests/test_host_logic.py::test_gelu_as28_error_budget), 1.2e-7 rms
// under N(0, 1) -- three orders of magnitude under the f16 rounding that follows (1.4e-4 rms); the x < 0 tail is absolute-,
// not relative-accurate (x + u (1 - r^16) cancels), which is all a value about to be rounded to f16 and summed by lin2 needs.
// P overflows to inf for |x| > 3e6, r = 0, and the result is max(x, 0) as it should be.  SAMRS_GELU_AS28=0: the form above (A/B).
#ifndef SAMRS_GELU_AS28
#define SAMRS_GELU_AS28 1
#endif
__device__ __forceinline__ float2_t gelu_erf2_et(float2_t x) {
#if SAMRS_GELU_AS28
    const float2_t u = {fabsf(x.x), fabsf(x.y)};
    float2_t p = u * 5.38297490493278e-06f + 4.889063711743802e-05f;
    p = p * u + 3.8003574445610866e-05f;
    p = p * u + 0.0032776263542473316f;
    p = p * u + 0.02114100567996502f;
    p = p * u + 0.04986734688282013f;
    p = p * u + 1.0f;
    const float2_t r = {__builtin_amdgcn_rcpf(p.x), __builtin_amdgcn_rcpf(p.y)};
    const float2_t r2 = r * r, r4 = r2 * r2, r8 = r4 * r4, r16 = r8 * r8;
    const float2_t s = 1.0f - r16;
    return (u * s + x) * 0.5f;
#else
    return gelu_erf2(x);
#endif
}
__device__ __forceinline__ float gelu_erf(float x) {
    const float2_t y = gelu_erf2(float2_t{x, x});
    return y.x;
}

// 16-byte store with the non-temporal hint when STREAM is set: for an output that is larger than the 256 MB Infinity Cache
// it would pass through -- the MLP hidden tensor, 336 MB per lin1 launch at 8 tiles of ViT-H -- so that it does not displace what the next
// kernels find there (the fp32 residual stream, the LayerNorm output).  A hint only: the same bytes reach memory.
// (A compile-time switch: behind a run-time flag LLVM merges the two stores into one and drops the hint.)
typedef unsigned int nt_u32x4_t __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ void store16_stream(void* p, const uint4& v) {
    if constexpr (STREAM) __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4_t, v), reinterpret_cast<nt_u32x4_t*>(p));
    else *reinterpret_cast<uint4*>(p) = v;
}

// XCD-aware, bijective remap of a linear block id (cdna_hip_programming.md T1): the hardware
// dispatches block b to XCD b % 8; give every XCD a contiguous chunk of the logical grid so
// that neighbouring tiles (which share operand panels) hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int q = nwg / NX, r = nwg % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

#define HIP_CHECK_RET(expr)                                                     \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return _e;                                        \
    } while (0)
