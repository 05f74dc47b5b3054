// Sustained MFMA rate per operand format on THIS part under its power cap: f16 16x16x32 / 32x32x16 against the block-scaled
// f8f6f4 forms (fp8 e4m3, fp6 e2m3, fp4 e2m1), register-resident operands, random vs all-zero bits.
// Question behind it: the two correction terms of the operand split tolerate fp6 / fp4 operands
// (oracle/error_budget.py plans8 / plans9) -- do those MFMAs really run at 2x / 4x the f16 rate when the socket is power-limited?
//     hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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
            // bit patterns: masked so that no f16 / fp8 lane is Inf / NaN (exponent MSB cleared in every byte / half)
            a[i][j] = (int)(seed[(tid * 64 + i * 8 + j) & 0xffff] & 0xbfbfbfbfu);
            b[i][j] = (int)(seed[(tid * 64 + 32 + i * 8 + j) & 0xffff] & 0xbfbfbfbfu);
        }
    if constexpr (VAR == 0 || (VAR >= 2 && VAR <= 4)) {
        v4f acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
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
        for (int it = 0; it < iters; ++it) {
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

int main() {
    const int blocks = 256 * 8, iters = 8192;
    uint32_t* h = (uint32_t*)malloc(65536 * 4);
    uint32_t *seed, *zero; float* out;
    CHECK(hipMalloc(&seed, 65536 * 4)); CHECK(hipMalloc(&zero, 65536 * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < 65536; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = (uint32_t)(st >> 16); }
    CHECK(hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(zero, 0, 65536 * 4));
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t* s = pass ? zero : seed;
        printf("---- operands: %s\n", pass ? "all zero" : "random bits");
        run<0>("f16 16x16x32", 2.0 * 16 * 16 * 32, 8, s, out, blocks, iters);
        run<1>("f16 32x32x16", 2.0 * 32 * 32 * 16, 4, s, out, blocks, iters);
        run<2>("scaled 16x16x128 fp8 (e4m3)", 2.0 * 16 * 16 * 128, 8, s, out, blocks, iters / 2);
        run<3>("scaled 16x16x128 fp6 (e2m3)", 2.0 * 16 * 16 * 128, 8, s, out, blocks, iters / 2);
        run<4>("scaled 16x16x128 fp4 (e2m1)", 2.0 * 16 * 16 * 128, 8, s, out, blocks, iters / 2);
        run<5>("scaled 32x32x64 fp8 (e4m3)", 2.0 * 32 * 32 * 64, 4, s, out, blocks, iters / 2);
        run<6>("scaled 32x32x64 fp6 (e2m3)", 2.0 * 32 * 32 * 64, 4, s, out, blocks, iters / 2);
        run<7>("scaled 32x32x64 fp4 (e2m1)", 2.0 * 32 * 32 * 64, 4, s, out, blocks, iters / 2);
    }
    return 0;
}
