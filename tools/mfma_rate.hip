// Sustained MFMA rate per operand format on THIS part under its power cap: f16 16x16x32 / 32x32x16 against the block-scaled
// f8f6f4 forms (fp8 e4m3, fp6 e2m3, fp4 e2m1), register-resident operands; three operand arms: random bits, all zero, and the
// GEMM's own statistics (A ~ N(0, 1) LayerNorm outputs, B ~ U(+-1/sqrt(1280)) weights, as f16 / as their fp8 / fp6 / fp4 codes).
// Round-4 fix: the round-3 version picked its operand registers with a RUNTIME index (a[(it + j) & 3]): the compiler turned that
// into v_cndmask chains, ~10 VALU instructions per MFMA, and the "pipe-only ceiling" it reported (1.05 PFLOP/s, below what a full
// hipBLASLt GEMM reaches) was the VALU rate of that selection code, not the matrix pipe ("weak" 6).  The loop is now
// unrolled by four so that every operand register is named at compile time: the inner loop is MFMAs and nothing else.
// Question behind it: the two correction terms of the operand split tolerate fp6 / fp4 operands
// (oracle/error_budget.py plans8 / plans9) -- do those MFMAs really run at 2x / 4x the f16 rate when the socket is power-limited?
//     hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// VAR: 0 f16 16x16x32, 1 f16 32x32x16, 2 scaled 16x16x128 fp8, 3 ... fp6 (e2m3), 4 ... fp4, 5 scaled 32x32x64 fp8, 6 ... fp6, 7 ... fp4
template <int VAR>
__global__ __launch_bounds__(256) void rate_kernel(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    v8i a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // the host prepares the words (random bits with the exponent MSB of every byte / half cleared: no Inf / NaN; zeros;
            // or encoded N(0, 1) / weight-like values): A operands from the first half of the table, B from the second
            a[i][j] = (int)seed[(tid * 32 + i * 8 + j) & 0x7fff];
            b[i][j] = (int)seed[0x8000 + ((tid * 32 + i * 8 + j) & 0x7fff)];
        }
    if constexpr (VAR == 0 || (VAR >= 2 && VAR <= 4)) {
        v4f acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int it4 = 0; it4 < iters; it4 += 4)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const v8i x = a[(it + j) & 3], y = b[(it + 3 * j) & 3];
                if constexpr (VAR == 0) {
                    const v8h xa = __builtin_bit_cast(v8h, __builtin_shufflevector(x, x, 0, 1, 2, 3));
                    const v8h yb = __builtin_bit_cast(v8h, __builtin_shufflevector(y, y, 0, 1, 2, 3));
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xa, yb, acc[j], 0, 0, 0);
                } else {
                    constexpr int F = VAR == 2 ? 0 : VAR == 3 ? 2 : 4;       // 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(x, y, acc[j], F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
        out[tid] = s;
    } else {
        v16f acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it4 = 0; it4 < iters; it4 += 4)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v8i x = a[(it + j) & 3], y = b[(it + 3 * j) & 3];
                if constexpr (VAR == 1) {
                    const v8h xa = __builtin_bit_cast(v8h, __builtin_shufflevector(x, x, 0, 1, 2, 3));
                    const v8h yb = __builtin_bit_cast(v8h, __builtin_shufflevector(y, y, 0, 1, 2, 3));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa, yb, acc[j], 0, 0, 0);
                } else {
                    constexpr int F = VAR == 5 ? 0 : VAR == 6 ? 2 : 4;
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc[j], F, F, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
        out[tid] = s;
    }
}

template <int VAR>
static void run(const char* name, double flop_per_mfma, int mfma_per_iter, const uint32_t* seed, float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    rate_kernel<VAR><<<blocks, 256>>>(seed, out, iters / 8);                    // warm-up
    CHECK(hipDeviceSynchronize());
    double best = 0, sum = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        rate_kernel<VAR><<<blocks, 256>>>(seed, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double tf = flop_per_mfma * mfma_per_iter * (double)iters * blocks * 4 / (ms * 1e-3) / 1e12;
        sum += tf; if (tf > best) best = tf;
    }
    printf("%-34s mean %8.1f TFLOP/s  best %8.1f\n", name, sum / reps, best);
}

// ---- host-side encoders for the "GEMM statistics" arm ---------------------------------------------------------------------
static uint16_t f16_bits(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static double gauss(uint64_t& st) {          // Box-Muller on a xorshift stream
    auto u = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return ((st >> 11) + 1) * (1.0 / 9007199254740993.0); };
    return sqrt(-2.0 * log(u())) * cos(6.283185307179586 * u());
}
// nearest code of a small float format (sign + ebits + mbits, bias, no Inf / NaN codes used), value pre-divided by the block scale
static uint32_t small_code(double v, int ebits, int mbits, int bias) {
    const uint32_t sign = v < 0 ? 1u : 0u;
    double av = fabs(v), best = 1e300; uint32_t bc = 0;
    const int ncode = 1 << (ebits + mbits);
    for (int c = 0; c < ncode; ++c) {
        const int e = c >> mbits, m = c & ((1 << mbits) - 1);
        const double val = e == 0 ? ldexp((double)m, 1 - bias - mbits) : ldexp(1.0 + m / (double)(1 << mbits), e - bias);
        if (ebits == 4 && c >= ncode - 1) continue;                     // e4m3: the all-ones code is NaN
        if (fabs(val - av) < best) { best = fabs(val - av); bc = (uint32_t)c; }
    }
    return (sign << (ebits + mbits)) | bc;
}
// fill n words with element codes of `bits` bits each (16 = f16), values drawn by `draw`, scaled into the format's range
template <typename F>
static void fill_codes(uint32_t* w, int n, int bits, F draw, double scale) {
    for (int i = 0; i < n; ++i) {
        uint32_t word = 0;
        if (bits == 6) {                      // 16 fp6 values in three words: handled as a 96-bit group by the caller's layout; here: pack what fits
            for (int e = 0; e < 5; ++e) word |= small_code(draw() / scale, 2, 3, 1) << (6 * e);
        } else if (bits == 16) {
            word = f16_bits((float)draw()) | ((uint32_t)f16_bits((float)draw()) << 16);
        } else {
            for (int e = 0; e < 32 / bits; ++e)
                word |= (bits == 8 ? small_code(draw() / scale, 4, 3, 7) : small_code(draw() / scale, 2, 1, 1)) << (bits * e);
        }
        w[i] = word;
    }
}

int main() {
    const int blocks = 256 * 8, iters = 8192;
    uint32_t* h = (uint32_t*)malloc(65536 * 4);
    uint32_t *seed; float* out;
    CHECK(hipMalloc(&seed, 65536 * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    uint64_t st = 0x9e3779b97f4a7c15ull;
    // arm 0: random bits (worst-case toggling; exponent MSBs cleared so that nothing is Inf / NaN); arm 1: zeros;
    // arm 2..5: the GEMM's operand statistics encoded as f16 / fp8 / fp6 / fp4 (A ~ N(0,1): first half, B ~ U(+-0.028): second half)
    for (int arm = 0; arm < 6; ++arm) {
        const char* names[6] = {"random bits", "all zero", "N(0,1) x U(+-1/sqrt(K)) as f16", "... as fp8 e4m3 codes", "... as fp6 e2m3 codes", "... as fp4 e2m1 codes"};
        if (arm == 0) for (int i = 0; i < 65536; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (uint32_t)(st >> 16) & 0xbfbfbfbfu; }
        else if (arm == 1) memset(h, 0, 65536 * 4);
        else {
            const int bits = arm == 2 ? 16 : arm == 3 ? 8 : arm == 4 ? 6 : 4;
            uint64_t s2 = 0x123456789abcdefull;
            auto ga = [&]() { return gauss(s2); };
            auto ub = [&]() { s2 ^= s2 << 13; s2 ^= s2 >> 7; s2 ^= s2 << 17; return ((double)(s2 >> 11) / 9007199254740992.0 * 2.0 - 1.0) * 0.028; };
            // block scales: N(0,1) |max| ~ 4 -> top binade of the format; weights 0.028 likewise
            const double sa = bits == 8 ? 4.0 / 256.0 : bits == 6 ? 4.0 / 4.0 : bits == 4 ? 4.0 / 4.0 : 1.0;
            const double sb = bits == 8 ? 0.028 / 256.0 : bits == 6 ? 0.028 / 4.0 : bits == 4 ? 0.028 / 4.0 : 1.0;
            fill_codes(h, 32768, bits, ga, sa);
            fill_codes(h + 32768, 32768, bits, ub, sb);
        }
        CHECK(hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice));
        printf("---- operands: %s\n", names[arm]);
        if (arm <= 2) {
            run<0>("f16 16x16x32", 2.0 * 16 * 16 * 32, 8, seed, out, blocks, iters);
            run<1>("f16 32x32x16", 2.0 * 32 * 32 * 16, 4, seed, out, blocks, iters);
        }
        if (arm <= 1 || arm == 3) { run<2>("scaled 16x16x128 fp8 (e4m3)", 2.0 * 16 * 16 * 128, 8, seed, out, blocks, iters / 2);
                                    run<5>("scaled 32x32x64 fp8 (e4m3)", 2.0 * 32 * 32 * 64, 4, seed, out, blocks, iters / 2); }
        if (arm <= 1 || arm == 4) { run<3>("scaled 16x16x128 fp6 (e2m3)", 2.0 * 16 * 16 * 128, 8, seed, out, blocks, iters / 2);
                                    run<6>("scaled 32x32x64 fp6 (e2m3)", 2.0 * 32 * 32 * 64, 4, seed, out, blocks, iters / 2); }
        if (arm <= 1 || arm == 5) { run<4>("scaled 16x16x128 fp4 (e2m1)", 2.0 * 16 * 16 * 128, 8, seed, out, blocks, iters / 2);
                                    run<7>("scaled 32x32x64 fp4 (e2m1)", 2.0 * 32 * 32 * 64, 4, seed, out, blocks, iters / 2); }
    }
    return 0;
}
