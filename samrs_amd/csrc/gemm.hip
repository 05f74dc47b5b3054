// gemm.hip -- MFMA GEMMs for the SAM hot path (gfx950).
//
//   gemm_et  : C[M,N] = A[M,K] * B[N,K]^T (+bias[n]) (+add2d[m % period, n]) (+C) (GELU)
//              A, B in the MFMA operand type (bf16 / f16), fp32 accumulate.
//              Replaces every nn.Linear / 1x1 conv / ConvTranspose-as-GEMM on the image side:
//              qkv / proj (modeling/image_encoder.py:227,238), MLP lin1/lin2 (common.py:25-26),
//              patch embed (image_encoder.py:387-395), neck (image_encoder.py:88-104), decoder
//              image-side projections (transformer.py:203-206) and the upscaler
//              (mask_decoder.py:53-59).
//   gemm_f32 : exact-fp32 GEMM on v_mfma_f32_16x16x4_f32 for the token side of the decoder
//              (a few hundred rows; precision matters more than rate there).
//
// gemm_et structure (v1): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64
// sub-tile as 4x4 MFMA 16x16x32 tiles; operands staged global -> VGPR -> LDS with a one-tile
// register prefetch (issue loads for tile t+1 before computing tile t, write them after); LDS
// rows are 128 B with an XOR swizzle of the 16-byte chunk index so that ds_read_b128 fragment
// reads are bank-conflict free.  MFMA operands are swapped (first = weight fragment, second =
// activation fragment) so that each lane ends up with 4 CONSECUTIVE output columns of one row:
// the epilogue then uses 8-byte (ET) / 16-byte (fp32) vector accesses and float4 bias loads.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;

// physical 16-byte chunk of logical chunk c in row r of a [rows][64] ET tile
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((r >> 1) & 7); }

// LDS-DMA helper: 16 bytes per lane, global (per-lane address) -> LDS (wave-uniform base + lane*16).
__device__ __forceinline__ void glds16(const uint16_t* gsrc, uint16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int PREC, bool OUT_F32, bool GELU, bool GLDS>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_et_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period,
    int M, int N, int K, int accumulate) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2][2][BM * BK];  // [buf][A|B] 64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = N / BN;
    const int nwg = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // --- staging map: thread t moves chunks (row = (t>>3) + 32*i, chunk = t&7), i = 0..3 ---
    const int ld_row = tid >> 3;
    const int ld_chunk = tid & 7;
    const uint16_t* gA = A + (size_t)(m0 + ld_row) * K + ld_chunk * 8;
    const uint16_t* gB = B + (size_t)(n0 + ld_row) * K + ld_chunk * 8;

    // Staging registers.  Plain macros (not lambdas / conditionals): anything that makes the
    // compiler see these arrays through a pointer puts them in scratch.
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GEMM_LOAD_TILE(kt_)                                                                   \
    do {                                                                                       \
        const size_t koff_ = (size_t)(kt_) * BK;                                               \
        ra0 = *reinterpret_cast<const uint4*>(gA + koff_);                                     \
        ra1 = *reinterpret_cast<const uint4*>(gA + (size_t)32 * K + koff_);                    \
        ra2 = *reinterpret_cast<const uint4*>(gA + (size_t)64 * K + koff_);                    \
        ra3 = *reinterpret_cast<const uint4*>(gA + (size_t)96 * K + koff_);                    \
        rb0 = *reinterpret_cast<const uint4*>(gB + koff_);                                     \
        rb1 = *reinterpret_cast<const uint4*>(gB + (size_t)32 * K + koff_);                    \
        rb2 = *reinterpret_cast<const uint4*>(gB + (size_t)64 * K + koff_);                    \
        rb3 = *reinterpret_cast<const uint4*>(gB + (size_t)96 * K + koff_);                    \
    } while (0)
    // swz(r + 32*i, c) == swz(r, c) ^ ... is NOT row-invariant, so compute the 4 offsets once.
    int st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ld_row + 32 * i;
        st_off[i] = r * BK + swz(r, ld_chunk) * 8;
    }
#define GEMM_STORE_TILE(buf_)                                                                 \
    do {                                                                                       \
        uint16_t* la_ = &lds[buf_][0][0];                                                      \
        uint16_t* lb_ = &lds[buf_][1][0];                                                      \
        *reinterpret_cast<uint4*>(la_ + st_off[0]) = ra0;                                      \
        *reinterpret_cast<uint4*>(la_ + st_off[1]) = ra1;                                      \
        *reinterpret_cast<uint4*>(la_ + st_off[2]) = ra2;                                      \
        *reinterpret_cast<uint4*>(la_ + st_off[3]) = ra3;                                      \
        *reinterpret_cast<uint4*>(lb_ + st_off[0]) = rb0;                                      \
        *reinterpret_cast<uint4*>(lb_ + st_off[1]) = rb1;                                      \
        *reinterpret_cast<uint4*>(lb_ + st_off[2]) = rb2;                                      \
        *reinterpret_cast<uint4*>(lb_ + st_off[3]) = rb3;                                      \
    } while (0)

    f32x4_t acc[4][4];  // [n-tile i][m-tile j]; D[i_local = n][j_local = m]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    // GLDS staging (cdna_hip_programming.md section 5, rule 21): the DMA writes LDS linearly
    // (wave-uniform base + lane*16 B), so wave w's i-th instruction fills rows 32i + 8w .. +7
    // (lane l -> row +(l>>3), physical chunk l&7) and the swizzle goes on the SOURCE address:
    // the lane fetches logical chunk (l&7) ^ swz(row).  swz(row) does not depend on i.
    const int g_row = 8 * wave + (lane >> 3);
    const int g_chunk = (lane & 7) ^ ((g_row >> 1) & 7);
    const uint16_t* gAg = A + (size_t)(m0 + g_row) * K + g_chunk * 8;
    const uint16_t* gBg = B + (size_t)(n0 + g_row) * K + g_chunk * 8;
#define GEMM_GLDS_TILE(kt_, buf_)                                                              \
    do {                                                                                       \
        const size_t koff_ = (size_t)(kt_) * BK;                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                     \
            glds16(gAg + (size_t)(32 * i_) * K + koff_, &lds[buf_][0][(32 * i_ + 8 * wave) * BK]); \
            glds16(gBg + (size_t)(32 * i_) * K + koff_, &lds[buf_][1][(32 * i_ + 8 * wave) * BK]); \
        }                                                                                      \
    } while (0)

    if (GLDS) {
        GEMM_GLDS_TILE(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        GEMM_LOAD_TILE(0);
        GEMM_STORE_TILE(0);
    }
    __syncthreads();

    const int fr = lane & 15;   // fragment row within a 16-row tile
    const int fq = lane >> 4;   // k-quarter (8 elements) within a 32-wide k-substep

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (GLDS) {
            if (kt + 1 < nk) GEMM_GLDS_TILE(kt + 1, buf ^ 1);
        } else {
            // unconditional prefetch (the last iteration re-reads the last tile; nobody consumes it)
            GEMM_LOAD_TILE(kt + 1 < nk ? kt + 1 : kt);
        }

        const uint16_t* la = &lds[buf][0][0];
        const uint16_t* lb = &lds[buf][1][0];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fa[4], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wm * 64 + j * 16 + fr;
                fa[j] = *reinterpret_cast<const uint4*>(la + r * BK + swz(r, ks * 4 + fq) * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wn * 64 + i * 16 + fr;
                fb[i] = *reinterpret_cast<const uint4*>(lb + r * BK + swz(r, ks * 4 + fq) * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]);
        }

        if (GLDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile has landed in LDS
        } else {
            GEMM_STORE_TILE(buf ^ 1);
        }
        __syncthreads();
    }

    // --- epilogue: lane holds C[m][n..n+3], m = m0 + wm*64 + j*16 + (lane&15),
    //                                      n = n0 + wn*64 + i*16 + 4*(lane>>4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn * 64 + i * 16 + 4 * fq;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + fr;
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y;
            float v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (add2d) {
                const float4 e = *reinterpret_cast<const float4*>(
                    add2d + (size_t)(m % add2d_period) * N + n);
                v0 += e.x; v1 += e.y; v2 += e.z; v3 += e.w;
            }
            if (GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
            if (OUT_F32) {
                float* C = reinterpret_cast<float*>(Cv) + (size_t)m * N + n;
                if (accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(C);
                    v0 += o.x; v1 += o.y; v2 += o.z; v3 += o.w;
                }
                *reinterpret_cast<float4*>(C) = make_float4(v0, v1, v2, v3);
            } else {
                uint16_t* C = reinterpret_cast<uint16_t*>(Cv) + (size_t)m * N + n;
                uint2 o;
                o.x = pack2<PREC>(v0, v1);
                o.y = pack2<PREC>(v2, v3);
                *reinterpret_cast<uint2*>(C) = o;
            }
        }
    }
}

int g_gemm_variant = 1;   // 0 = register-staged, 1 = LDS-DMA (global_load_lds)

template <int PREC, bool GLDS>
hipError_t launch_gemm_prec(const void* A, const void* B, void* C, const float* bias,
                            const float* add2d, int period, int M, int N, int K, bool out_f32,
                            bool gelu, bool accumulate, hipStream_t s) {
    dim3 grid((M / BM) * (N / BN)), block(GEMM_THREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu)
            gemm_et_kernel<PREC, true, true, GLDS><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else
            gemm_et_kernel<PREC, true, false, GLDS><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    } else {
        if (gelu)
            gemm_et_kernel<PREC, false, true, GLDS><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else
            gemm_et_kernel<PREC, false, false, GLDS><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// exact fp32 GEMM: C[M,N] = A[M,K] * W[N,K]^T + bias, optional ReLU / accumulate.
// 64x64 block tile, 4 waves (2x2), wave tile 32x32 = 2x2 MFMA 16x16x4 (f32 in, f32 acc).
// M, N arbitrary (bounds-checked), K % 16 == 0.  lda / ldc in elements; W is dense [N][K].
// ---------------------------------------------------------------------------------------------
constexpr int FM = 64, FN = 64, FK = 16;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ C, int ldc, int M, int N,
                                                       int K, int relu, int accumulate) {
    // +1 padding: fragment reads walk rows at fixed k -> stride 17 floats is conflict-free
    __shared__ float sa[FM][FK + 1];
    __shared__ float sw[FN][FK + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FM, n0 = blockIdx.x * FN;

    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // staging: 64 rows x 16 floats = 256 float4; thread t -> row t>>2, float4 index t&3
    const int sr = tid >> 2, sc = (tid & 3) * 4;
    for (int k0 = 0; k0 < K; k0 += FK) {
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vw = va;
        if (m0 + sr < M) va = *reinterpret_cast<const float4*>(A + (size_t)(m0 + sr) * lda + k0 + sc);
        if (n0 + sr < N) vw = *reinterpret_cast<const float4*>(W + (size_t)(n0 + sr) * K + k0 + sc);
        __syncthreads();  // previous iteration's fragment reads are done
        sa[sr][sc + 0] = va.x; sa[sr][sc + 1] = va.y; sa[sr][sc + 2] = va.z; sa[sr][sc + 3] = va.w;
        sw[sr][sc + 0] = vw.x; sw[sr][sc + 1] = vw.y; sw[sr][sc + 2] = vw.z; sw[sr][sc + 3] = vw.w;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < FK; kk += 4) {
            // 16x16x4: first operand lane l = P[i = l&15][k = l>>4]; second = Q[k = l>>4][j = l&15]
            float fa[2], fw[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fa[j] = sa[wm * 32 + j * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 2; ++i) fw[i] = sw[wn * 32 + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[i_local = n][j_local = m]: lane holds n = base + 4*(lane>>4) + r, m = base + (lane&15)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 32 + j * 16 + (lane & 15);
            if (m >= M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 32 + i * 16 + 4 * (lane >> 4) + r;
                if (n >= N) continue;
                float v = acc[i][j][r] + (bias ? bias[n] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                float* c = C + (size_t)m * ldc + n;
                if (accumulate) v += *c;
                *c = v;
            }
        }
}

}  // namespace

hipError_t launch_gemm_et(int prec, const void* A, const void* B, void* C, const float* bias,
                          const float* add2d, int add2d_period, int M, int N, int K, bool out_f32,
                          bool gelu, bool accumulate, hipStream_t s) {
    if (M % BM || N % BN || K % BK || M <= 0 || N <= 0 || K <= 0) return hipErrorInvalidValue;
    if (add2d && add2d_period <= 0) return hipErrorInvalidValue;
    const bool glds = g_gemm_variant == 1;
    if (prec == PREC_BF16)
        return glds ? launch_gemm_prec<PREC_BF16, true>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s)
                    : launch_gemm_prec<PREC_BF16, false>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
    if (prec == PREC_F16)
        return glds ? launch_gemm_prec<PREC_F16, true>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s)
                    : launch_gemm_prec<PREC_F16, false>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
    return hipErrorInvalidValue;
}

void set_gemm_variant(int v) { g_gemm_variant = v; }

hipError_t launch_gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C,
                           int ldc, int M, int N, int K, bool relu, bool accumulate, hipStream_t s) {
    if (K % FK || M <= 0 || N <= 0 || (lda % 4) || (K % 4)) return hipErrorInvalidValue;
    dim3 grid((N + FN - 1) / FN, (M + FM - 1) / FM), block(256);
    gemm_f32_kernel<<<grid, block, 0, s>>>(A, lda, W, bias, C, ldc, M, N, K, relu ? 1 : 0, accumulate ? 1 : 0);
    return hipGetLastError();
}
