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
#include <cstdlib>

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

template <int PREC, bool OUT_F32, bool GELU, bool GLDS, int GROUP_M>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_et_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period,
    int M, int N, int K, int accumulate) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2][2][BM * BK];  // [buf][A|B] 64 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // Tile order.  (1) xcd_remap gives every XCD a contiguous range of logical ids (blocks that
    // run together share one 4 MB L2).  (2) Inside that range ids walk GROUP_M tile rows fastest,
    // then tile columns: the ~64 blocks resident on an XCD form a ~8x8 super-tile, so each A and B
    // panel fetched into L2 is reused 8 times instead of streaming the whole weight matrix through
    // L2 once per tile row.
    const int tiles_n = N / BN, tiles_m = M / BM;
    const int nwg = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    if (GROUP_M > 1) {
        const int per_group = GROUP_M * tiles_n;
        const int group = bid / per_group, first_m = group * GROUP_M;
        const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        const int in_g = bid - group * per_group;
        tile_m = first_m + in_g % gsz;
        tile_n = in_g / gsz;
    } else {
        tile_m = bid / tiles_n;
        tile_n = bid % tiles_n;
    }
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
            if (GELU) { const float2_t g01 = gelu_erf2(float2_t{v0, v1}), g23 = gelu_erf2(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; }
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


__device__ __forceinline__ void wave_lds_sync_g() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// Coalesced epilogue.  In the MFMA accumulator layout a lane owns 4 consecutive columns of ONE row,
// so a direct store instruction scatters 16 rows x 32-byte pieces: measured (ablation)
// that costs a third of the whole GEMM.  Instead each wave bounces its 64-column sub-tile through
// its private slice of the (now idle) LDS ring and writes / read-modify-writes FULL 128- or 256-byte
// row segments with 16 bytes per lane.  bias + GELU are applied before the bounce (per-lane column
// known), the 2-D addend and the residual accumulate after it (coalesced loads).
//   acc[i][j]: n-tile i (16 cols), m-tile j (16 rows); NJ m-tiles per wave, JC of them per pass.
// ---------------------------------------------------------------------------------------------
// DRAIN (persistent kernels): the caller has LDS-DMA pieces of its NEXT tile in flight; they are retired (vmcnt(0)) right
// before this wave's first global store, i.e. after the bias loads, the VALU work and the first LDS bounce have covered
// their latency, and before any store joins the queue -- so the stores themselves are never waited for.
template <int PREC, bool OUT_F32, int GELU /* 0 none, 1 fp32-epsilon erf, 2 the cheaper erf (ET output) */, int NJ, int JC, int NI = 4, bool GLN = false,
          bool DRAIN = false, bool STREAM = false /* ET output: non-temporal stores (common.h store16_stream) */>
__device__ __forceinline__ void epilogue_coalesced(f32x4_t (&acc)[NI][NJ], unsigned char* scr /* wave-private */,
                                                   void* __restrict__ Cv, const float* __restrict__ bias,
                                                   const float* __restrict__ add2d, int add2d_period, int N,
                                                   int m_base /* first row of the wave tile */,
                                                   int n_base /* first column of the wave tile */, int accumulate,
                                                   int lane, const float* __restrict__ pre2d = nullptr /* ET output: 2-D addend
                                                   applied before the bounce, loaded element by element (keeps registers low) */) {
    const int fr = lane & 15, fq = lane >> 4;
    // padded row stride: the smallest >= the 16*NI-column row with (words % 32) == 4, which spreads the 16
    // rows of an accumulator column block over all banks (NI = 4: 272 / 144 bytes)
    constexpr int ROWW = 16 * NI * (OUT_F32 ? 4 : 2) / 4;
    constexpr int RS = (((ROWW - 4 + 31) / 32) * 32 + 4) * 4;
    constexpr int TS = 16 * RS;                                     // one m-tile
    constexpr int CPR = 16 * NI * (OUT_F32 ? 4 : 2) / 16;           // 16-byte chunks per row segment
    constexpr int NCH = 16 * CPR, NPASS = (NCH + 63) / 64;
    float4 bv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
        bv[i] = bias ? *reinterpret_cast<const float4*>(bias + n_base + i * 16 + 4 * fq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j0 = 0; j0 < NJ; j0 += JC) {
        // fp32 output: the residual (old C) and the 2-D addend of this pass do not depend on the accumulators --
        // issue their loads FIRST, so that they are in flight during the LDS bounce instead of after it
        float4 res[OUT_F32 ? JC : 1][OUT_F32 ? NPASS : 1];
        if (OUT_F32) {
#pragma unroll
            for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                for (int h = 0; h < NPASS; ++h) {
                    const int idx = lane + 64 * h;
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (NCH % 64 == 0 || idx < NCH) {
                        const int m = m_base + (j0 + jj) * 16 + idx / CPR, n = n_base + (idx % CPR) * 4;
                        if (accumulate) r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Cv) + (size_t)m * N + n);
                        if (add2d && !GLN) {      // GLN: `add2d` carries gamma | beta, not a 2-D addend
                            const float4 e = *reinterpret_cast<const float4*>(add2d + (size_t)(m % add2d_period) * N + n);
                            r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w;
                        }
                    }
                    res[jj][h] = r;
                }
        }
        if (j0) wave_lds_sync_g();
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
            const int j = j0 + jj;
            // GLN: LayerNorm2d over the wave's 64 columns (= one channel group of the transposed-conv output,
            // mask_decoder.py:55-56, eps 1e-6) before the GELU.  A row's 64 values sit in the 4 lane quarters x
            // 16 registers; two-pass statistics like the stand-alone kernel.  `add2d` carries gamma[64] | beta[64].
            float gmean = 0.f, grstd = 1.f;
            if (GLN) {
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    s1 += ((acc[i][j][0] + bv[i].x) + (acc[i][j][1] + bv[i].y)) + ((acc[i][j][2] + bv[i].z) + (acc[i][j][3] + bv[i].w));
                s1 += __shfl_xor(s1, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                gmean = s1 * (1.0f / 64.0f);
                float s2 = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const float a0 = acc[i][j][0] + bv[i].x - gmean, a1 = acc[i][j][1] + bv[i].y - gmean;
                    const float a2 = acc[i][j][2] + bv[i].z - gmean, a3 = acc[i][j][3] + bv[i].w - gmean;
                    s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
                s2 += __shfl_xor(s2, 16, 64);
                s2 += __shfl_xor(s2, 32, 64);
                grstd = 1.0f / sqrtf(s2 * (1.0f / 64.0f) + 1e-6f);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
                float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                if (pre2d) {
                    const float4 e = *reinterpret_cast<const float4*>(pre2d + (size_t)((m_base + j * 16 + fr) % add2d_period) * N +
                                                                      n_base + i * 16 + 4 * fq);
                    v0 += e.x; v1 += e.y; v2 += e.z; v3 += e.w;
                }
                if (GLN) {
                    const float4 gm = *reinterpret_cast<const float4*>(add2d + i * 16 + 4 * fq);
                    const float4 bt = *reinterpret_cast<const float4*>(add2d + 64 + i * 16 + 4 * fq);
                    v0 = (v0 - gmean) * grstd * gm.x + bt.x; v1 = (v1 - gmean) * grstd * gm.y + bt.y;
                    v2 = (v2 - gmean) * grstd * gm.z + bt.z; v3 = (v3 - gmean) * grstd * gm.w + bt.w;
                }
                if constexpr (GELU == 2 && !OUT_F32) { const float2_t g01 = gelu_erf2_et(float2_t{v0, v1}), g23 = gelu_erf2_et(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; }
                else if constexpr (GELU != 0) { const float2_t g01 = gelu_erf2(float2_t{v0, v1}), g23 = gelu_erf2(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; }
                unsigned char* p = scr + jj * TS + fr * RS;
                if (OUT_F32) {
                    *reinterpret_cast<float4*>(p + (i * 16 + 4 * fq) * 4) = make_float4(v0, v1, v2, v3);
                } else {
                    uint2 o;
                    o.x = pack2<PREC>(v0, v1);
                    o.y = pack2<PREC>(v2, v3);
                    *reinterpret_cast<uint2*>(p + (i * 16 + 4 * fq) * 2) = o;
                }
            }
        }
        wave_lds_sync_g();
        if (DRAIN && j0 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
            const int j = j0 + jj;
            if (OUT_F32) {
                // 16 rows x (64 NI) B in 16-byte chunks, consecutive lanes on consecutive chunks of a row
#pragma unroll
                for (int h = 0; h < NPASS; ++h) {
                    const int idx = lane + 64 * h;
                    if (NCH % 64 != 0 && idx >= NCH) break;
                    const int row = idx / CPR, ch = idx % CPR;
                    float4 v = *reinterpret_cast<const float4*>(scr + jj * TS + row * RS + ch * 16);
                    const int m = m_base + j * 16 + row, n = n_base + ch * 4;
                    const float4 o = res[OUT_F32 ? jj : 0][OUT_F32 ? h : 0];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    float* C = reinterpret_cast<float*>(Cv) + (size_t)m * N + n;
                    *reinterpret_cast<float4*>(C) = v;
                }
            } else {
                // 16 rows x (32 NI) B
#pragma unroll
                for (int h = 0; h < NPASS; ++h) {
                    const int idx = lane + 64 * h;
                    if (NCH % 64 != 0 && idx >= NCH) break;
                    const int row = idx / CPR, ch = idx % CPR;
                    const uint4 v = *reinterpret_cast<const uint4*>(scr + jj * TS + row * RS + ch * 16);
                    uint16_t* C = reinterpret_cast<uint16_t*>(Cv) + (size_t)(m_base + j * 16 + row) * N + n_base + ch * 8;
                    store16_stream<STREAM>(C, v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ET-output epilogue of the 256x320 kernel (NI = 5).  A wave's sub-tile is 80 columns = 160 bytes per row:
// stored per wave, the row segments straddle 64-byte sectors and the generic code spills.  Here the two waves
// that sit side by side (wn = 2p, 2p+1: 160 columns = 320 bytes, 64-byte aligned) bounce through ONE shared
// LDS buffer and each stores half of the rows as full 320-byte segments (20 lanes x 16 B per row).
// Block-wide barriers: all 8 waves are past the main loop here (the stagger has been re-aligned).
//   acc[i][j]: n-tile i (16 cols), m-tile j (16 rows); 8 m-tiles, 4 per pass.
// ---------------------------------------------------------------------------------------------
// FOLD (LayerNorm folded into this GEMM, see gemm_et_x64p_kernel): the value is rstd_m * acc + (-rstd_m mean_m) * cvec_n +
// bias_n with (rstd_m, -rstd_m mean_m) = rowstat[m] (global, written by ln_rowstat_kernel) instead of acc + bias_n.
// MXO (round 4): the output is also the A operand of a GEMM whose lo terms run on MXFP4 (lin1 -> lin2 in the all-split mode): besides
// the ET row segments the epilogue writes hi / lo of every value as fp4 codes + E8M0 scales on a K axis padded per WAVE TILE (80 -> 96
// columns: a 32-element block never straddles two waves; mx4_pack_kernel perm 2 lays the weights out the same way).  The 16 columns of
// n-tile i of a row sit in the 4 lanes fq = 0..3 (4 each): block 0 = n-tiles 0 | 1, block 1 = 2 | 3, block 2 = 4 | zeros; block maxima
// by two cross-lane swaps (l ^ 16, l ^ 32), codes by v_cvt_scalef32_pk_fp4_f32, one dword store per lane, block and tensor (position
// 8 fq + 4 (i & 1) + e of the block holds column 16 (i & 1) + 4 fq + e).
// GELU: 0 none, 1 the fp32-epsilon erf (common.h gelu_erf2), 2 the cheaper erf of the 1x-rate mode (gelu_erf2_et)
template <int PREC, int GELU, int JC = 4, bool DRAIN = false, bool FOLD = false, bool MXO = false, bool STREAM = false /* non-temporal ET stores */>
__device__ __forceinline__ void epilogue_pair_et(f32x4_t (&acc)[5][8], unsigned char* lds, void* __restrict__ Cv,
                                                 const float* __restrict__ bias, const float* __restrict__ pre2d, int period, int N,
                                                 int m_base /* wave tile row 0 */, int n_pair /* first column of the pair */,
                                                 int wm, int wn, int lane, const float2* __restrict__ rowstat = nullptr /* FOLD: [M] */,
                                                 const float* __restrict__ cvec = nullptr /* FOLD */, MxOut mxo = MxOut()) {
    constexpr int RS = 400;                 // 320 data bytes + pad: 100 words = 4 (mod 32)
    constexpr int TS = 16 * RS;
    const int fr = lane & 15, fq = lane >> 4, half = wn & 1;
    unsigned char* scr = lds + (wm * 2 + (wn >> 1)) * (JC * TS);
    float4 bv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
        bv[i] = bias ? *reinterpret_cast<const float4*>(bias + n_pair + half * 80 + i * 16 + 4 * fq) : make_float4(0.f, 0.f, 0.f, 0.f);
    // FOLD: (rstd, -rstd mean) of this lane's rows, one pass ahead (global, 8 bytes per row: L2 hits next to the bias loads)
    float2 rs_cur[JC], rs_nxt[JC];
    if constexpr (FOLD) {
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) rs_cur[jj] = rowstat[m_base + jj * 16 + fr];
    }
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += JC) {
        if (j0) __syncthreads();
        if constexpr (FOLD) {
            if (j0 + JC < 8) {
#pragma unroll
                for (int jj = 0; jj < JC; ++jj) rs_nxt[jj] = rowstat[m_base + (j0 + JC + jj) * 16 + fr];
            }
        }
#pragma unroll
        for (int jj = 0; jj < JC; ++jj) {
            const int j = j0 + jj;
            float2 rs = make_float2(1.f, 0.f);
            if constexpr (FOLD) rs = rs_cur[jj];
            float mxh[MXO ? 24 : 1], mxl[MXO ? 24 : 1];          // MXO: hi / lo of this row's 20 values (+ 4 zeros of the padded n-tile)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float v0 = acc[i][j][0] + bv[i].x, v1 = acc[i][j][1] + bv[i].y;
                float v2 = acc[i][j][2] + bv[i].z, v3 = acc[i][j][3] + bv[i].w;
                if constexpr (FOLD) {
                    const float4 cv = *reinterpret_cast<const float4*>(cvec + n_pair + half * 80 + i * 16 + 4 * fq);
                    v0 = fmaf(rs.x, acc[i][j][0], fmaf(rs.y, cv.x, bv[i].x)); v1 = fmaf(rs.x, acc[i][j][1], fmaf(rs.y, cv.y, bv[i].y));
                    v2 = fmaf(rs.x, acc[i][j][2], fmaf(rs.y, cv.z, bv[i].z)); v3 = fmaf(rs.x, acc[i][j][3], fmaf(rs.y, cv.w, bv[i].w));
                }
                if (pre2d) {
                    const float4 e = *reinterpret_cast<const float4*>(pre2d + (size_t)((m_base + j * 16 + fr) % period) * N + n_pair +
                                                                      half * 80 + i * 16 + 4 * fq);
                    v0 += e.x; v1 += e.y; v2 += e.z; v3 += e.w;
                }
                if constexpr (GELU == 2) { const float2_t g01 = gelu_erf2_et(float2_t{v0, v1}), g23 = gelu_erf2_et(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; }
                else if constexpr (GELU != 0) { const float2_t g01 = gelu_erf2(float2_t{v0, v1}), g23 = gelu_erf2(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; }
                uint2 o;
                o.x = pack2<PREC>(v0, v1);
                o.y = pack2<PREC>(v2, v3);
                *reinterpret_cast<uint2*>(scr + jj * TS + fr * RS + half * 160 + (i * 16 + 4 * fq) * 2) = o;
                if constexpr (MXO) {
                    const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hv = ET<PREC>::to_float((uint16_t)(((e & 2) ? o.y : o.x) >> (16 * (e & 1))));
                        mxh[4 * i + e] = hv;
                        mxl[4 * i + e] = vv[e] - hv;
                    }
                }
            }
            if constexpr (MXO) {
#pragma unroll
                for (int e = 20; e < 24; ++e) { mxh[e] = 0.f; mxl[e] = 0.f; }
                const int row = m_base + j * 16 + fr;
                const int wt = (n_pair + half * 80) / 80, nblk = (N / 80) * 3;
                const size_t rowb = (size_t)row * (size_t)(nblk * 16);
#pragma unroll
                for (int b3 = 0; b3 < 3; ++b3) {
                    float ah = 0.f, al = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ah = fmaxf(ah, fabsf(mxh[8 * b3 + e])); al = fmaxf(al, fabsf(mxl[8 * b3 + e])); }
                    ah = fmaxf(ah, lane_xor16(ah)); al = fmaxf(al, lane_xor16(al));
                    ah = fmaxf(ah, lane_xor32(ah)); al = fmaxf(al, lane_xor32(al));
                    const int bh = mx_scale_byte(ah), bl = mx_scale_byte(al);
                    const size_t at = rowb + (size_t)(wt * 3 + b3) * 16 + 4 * fq;
                    *reinterpret_cast<uint32_t*>(mxo.q_hi + at) = fp4_pack8(mxh + 8 * b3, bh);
                    *reinterpret_cast<uint32_t*>(mxo.q_lo + at) = fp4_pack8(mxl + 8 * b3, bl);
                    if (fq == 0) {
                        const size_t si = mx_scale_index(false, row, wt * 3 + b3, nblk * 32 / MXK);
                        mxo.s_hi[si] = (unsigned char)bh;
                        mxo.s_lo[si] = (unsigned char)bl;
                    }
                }
            }
        }
        if (DRAIN && j0 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // this wave: rows 8*half .. +7 of each of the JC m-tiles = JC x 8 rows x 20 chunks = 2.5 JC passes of 64 lanes
#pragma unroll
        for (int h = 0; h < JC * 160 / 64; ++h) {
            const int idx = lane + 64 * h;
            const int jj = idx / 160, rem = idx % 160;
            const int row = 8 * half + rem / 20, ch = rem % 20;
            const uint4 v = *reinterpret_cast<const uint4*>(scr + jj * TS + row * RS + ch * 16);
            uint16_t* C = reinterpret_cast<uint16_t*>(Cv) + (size_t)(m_base + (j0 + jj) * 16 + row) * N + n_pair + ch * 8;
            store16_stream<STREAM>(C, v);
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) rs_cur[jj] = rs_nxt[jj];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shared geometry of the 256x128x64 staggered kernel below (its lock-step round-1 flavour was removed): block tile, 512 threads = 8 waves (4 along M x 2 along N, each
// a 64x64 sub-tile exactly as above), THREE-stage LDS ring filled by LDS-DMA with a COUNTED
// s_waitcnt vmcnt: the loads of tile t+2 are issued before computing tile t and only tile t+1 is
// waited for at the (raw) barrier, so DMA traffic stays in flight across barriers instead of
// draining every K step (rocprofv3 on the 2-stage kernel: 47 % of wave cycles parked in
// s_waitcnt/s_barrier, MFMA busy 30 %).  144 KiB LDS -> one block (8 waves) per CU.
// ---------------------------------------------------------------------------------------------
constexpr int PBM = 256, PBN = 128, PTHREADS = 512, PSTAGES = 3;
constexpr int PSTAGE_ELEMS = (PBM + PBN) * BK;          // A rows then B rows, 64 ET each
constexpr int P_GLDS_PER_TILE = (PBM + PBN) * BK * 2 / (PTHREADS * 16);   // 6 per thread

// LDS-DMA issued from inline asm so that hipcc's s_waitcnt insertion does not know about it (it
// would otherwise put vmcnt(0) in front of every ds_read).  M0 = LDS byte address of the wave's
// 1 KiB destination; M0 is compiler-reserved, so it is saved / restored inside the statement, and
// the SALU-write -> LDS-DMA hazard gets its s_nop (cdna_hip_programming.md 5.7).  The caller
// counts vmcnt by hand.
template <int OFF>
__device__ __forceinline__ void glds16_asm(const uint16_t* gsrc, uint32_t wave_lds_base) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_add_u32 m0, %2, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(wave_lds_base), "n"(OFF)
        : "memory", "scc");
}

// one K tile (A: 4 x 64 rows, B: 2 x 64 rows) into ring stage `stage_`: six DMA pieces per wave
#define PIPE_ISSUE(kt_, stage_)                                                                 \
    do {                                                                                         \
        const size_t koff_ = (size_t)(kt_) * BK;                                                 \
        constexpr int SB_ = (stage_) * PSTAGE_ELEMS * 2;                                         \
        glds16_asm<SB_ + 0 * 64 * BK * 2>(gAg + koff_, wave_lds_base);                           \
        glds16_asm<SB_ + 1 * 64 * BK * 2>(gAg + rs64 + koff_, wave_lds_base);                    \
        glds16_asm<SB_ + 2 * 64 * BK * 2>(gAg + 2 * rs64 + koff_, wave_lds_base);                \
        glds16_asm<SB_ + 3 * 64 * BK * 2>(gAg + 3 * rs64 + koff_, wave_lds_base);                \
        glds16_asm<SB_ + PBM * BK * 2 + 0 * 64 * BK * 2>(gBg + koff_, wave_lds_base);            \
        glds16_asm<SB_ + PBM * BK * 2 + 1 * 64 * BK * 2>(gBg + rs64 + koff_, wave_lds_base);     \
    } while (0)

// ---------------------------------------------------------------------------------------------
// gemm_et_stag_kernel: the tile / ring / DMA described above, and the two waves that share
// a SIMD (waves w and w+4) run STAGGERED by one barrier interval.  Every K step is cut into four
// segments separated by raw barriers -- L0 (fragment reads, k 0..31 + DMA issue), C0 (16 MFMAs),
// L1 (reads, k 32..63), C1 (16 MFMAs) -- and group 1 (waves 4..7) executes one extra barrier up
// front, so while group 0 is in a C segment group 1 is in an L segment and vice versa: the matrix
// pipe of every SIMD always has one wave feeding it while its partner waits on LDS (rocprofv3 on the
// lock-step kernel: 48 % of wave cycles parked, MFMA busy 37 %).  s_setprio(1) around the MFMA
// segments lets the computing wave win issue arbitration (cdna_hip_programming.md T5).
//
// Ring safety with the stagger (interval numbering 4*kt + {0,1,2,3} for group 0, +1 for group 1):
//  WAR  the slot of tile kt-1 is last read by group 1 in interval 4kt-1; its refill (tile kt+2) is
//       issued in an L0 segment, i.e. interval 4kt (group 0) / 4kt+1 (group 1): after that barrier.
//  RAW  tile kt+1 is first read in interval 4kt+4; every wave waits vmcnt for its own share at the
//       end of BOTH its L1 and C1 segments (interval 4kt+3 for either group) and then passes the
//       barrier that opens interval 4kt+4.
// Both groups execute exactly 2 + 4*nk barriers.
// ---------------------------------------------------------------------------------------------
template <int PREC, bool OUT_F32, bool GELU>
__global__ __launch_bounds__(PTHREADS) void gemm_et_stag_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period,
    int M, int N, int K, int accumulate) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[PSTAGES * PSTAGE_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int grp = wave >> 2;
    const int wm = wave >> 1, wn = wave & 1;

    constexpr int GROUP = 8;
    const int tiles_n = N / PBN, tiles_m = M / PBM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GROUP * tiles_n;
    const int group = bid / per_group, first_m = group * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = bid - group * per_group;
    const int tile_m = first_m + in_g % gsz, tile_n = in_g / gsz;
    const int m0 = tile_m * PBM, n0 = tile_n * PBN;

    const int g_row = 8 * wave + (lane >> 3);
    const int g_chunk = (lane & 7) ^ ((g_row >> 1) & 7);
    const uint16_t* gAg = A + (size_t)(m0 + g_row) * K + g_chunk * 8;
    const uint16_t* gBg = B + (size_t)(n0 + g_row) * K + g_chunk * 8;
    const uint32_t wave_lds_base = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * (8 * BK * 2));
    const size_t rs64 = (size_t)64 * K;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    PIPE_ISSUE(0, 0);
    if (nk > 1) {
        PIPE_ISSUE(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_GLDS_PER_TILE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one interval behind

    const int fr = lane & 15, fq = lane >> 4;
#define STAG_LOAD(S_, KS_)                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
        const int r = wm * 64 + j * 16 + fr;                                                      \
        fa[j] = *reinterpret_cast<const uint4*>(lds + (S_) * PSTAGE_ELEMS + r * BK + swz(r, (KS_) * 4 + fq) * 8); \
    }                                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        const int r = wn * 64 + i * 16 + fr;                                                      \
        fb[i] = *reinterpret_cast<const uint4*>(lds + (S_) * PSTAGE_ELEMS + PBM * BK + r * BK + swz(r, (KS_) * 4 + fq) * 8); \
    }
#define STAG_MMA()                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]); \
    __builtin_amdgcn_s_setprio(0);
#define STAG_WAIT_NEXT(kt_)                                                                       \
    if ((kt_) + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_GLDS_PER_TILE) : "memory");    \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define STAG_STEP(kt_, S_)                                                                        \
    if ((kt_) < nk) {                                                                             \
        uint4 fa[4], fb[4];                                                                       \
        /* L0 */                                                                                  \
        if ((kt_) + 2 < nk) PIPE_ISSUE((kt_) + 2, ((S_) + 2) % PSTAGES);                          \
        STAG_LOAD(S_, 0)                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        __builtin_amdgcn_s_barrier();                                                             \
        /* C0 */                                                                                  \
        STAG_MMA()                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        __builtin_amdgcn_s_barrier();                                                             \
        /* L1 */                                                                                  \
        STAG_LOAD(S_, 1)                                                                          \
        STAG_WAIT_NEXT(kt_)                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                        \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        __builtin_amdgcn_s_barrier();                                                             \
        /* C1 */                                                                                  \
        STAG_MMA()                                                                                \
        STAG_WAIT_NEXT(kt_)                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        __builtin_amdgcn_s_barrier();                                                             \
    }
    for (int kt = 0; kt < nk; kt += PSTAGES) {
        STAG_STEP(kt, 0)
        STAG_STEP(kt + 1, 1)
        STAG_STEP(kt + 2, 2)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // group 0 finishes one interval early
    // every wave is past its last ring read here -> the ring is free scratch (18 KiB per wave)
    {
        unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + wave * (PSTAGES * PSTAGE_ELEMS * 2 / 8);
        // add2d (if any) only comes with fp32 output in the engine; ET output ignores accumulate
        if (!OUT_F32 && add2d) {
            // rare combination (ET out + 2-D addend): fold the addend in before the bounce
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * 64 + j * 16 + fr, n = n0 + wn * 64 + i * 16 + 4 * fq;
                    const float4 e = *reinterpret_cast<const float4*>(add2d + (size_t)(m % add2d_period) * N + n);
                    acc[i][j][0] += e.x; acc[i][j][1] += e.y; acc[i][j][2] += e.z; acc[i][j][3] += e.w;
                }
        }
        epilogue_coalesced<PREC, OUT_F32, GELU, 4, 4>(acc, scr, Cv, bias, OUT_F32 ? add2d : nullptr, add2d_period, N,
                                                      m0 + wm * 64, n0 + wn * 64, accumulate, lane);
    }
}

template <int PREC>
hipError_t launch_gemm_stag(const void* A, const void* B, void* C, const float* bias, const float* add2d, int period,
                            int M, int N, int K, bool out_f32, bool gelu, bool accumulate, hipStream_t s) {
    dim3 grid((M / PBM) * (N / PBN)), block(PTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_stag_kernel<PREC, true, true><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else gemm_et_stag_kernel<PREC, true, false><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    } else {
        if (gelu) gemm_et_stag_kernel<PREC, false, true><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else gemm_et_stag_kernel<PREC, false, false><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    }
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// gemm_et_big_kernel: 256x256x32 block tile, 8 waves as 2 (M) x 4 (N), wave tile 128x64
// (8 x 4 MFMA 16x16x32 tiles, one full K=32 step per MFMA), FOUR-stage LDS ring (4 x 32 KiB) filled
// by LDS-DMA three tiles ahead, wave groups staggered by one barrier interval (L | C segments).
// Versus the 256x128x64 kernel: 4 instead of 6 DMA pieces and 12 instead of 16 ds_read_b128 per 32
// MFMAs, and 2 instead of 4 barriers -- the per-CU vector-memory / LDS traffic per FLOP is what
// capped that kernel.
//
// LDS rows are 64 B (32 ET): four rows per 256-B bank row, so the ds_read_b128 lane groups
// (MI355X_MICROARCH.md, LDS table) need the 16-byte chunk index XORed with g(row>>2),
// g = [0,2,3,1]: checked conflict-free for all four lane groups.
// ---------------------------------------------------------------------------------------------
constexpr int QBM = 256, QBN = 256, QBK = 32, QSTAGES = 4, QTHREADS = 512;
constexpr int WBN = 320;     // NI = 5 flavour: 256x320 tile (wave tile 128x80), 4 x 36 KiB ring

__device__ __forceinline__ int qswz(int r, int c) { return c ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3); }

// ABL (ablation bits, timing experiments only; results are garbage when non-zero):
//   1 = no DMA, 2 = no fragment reads, 4 = no MFMA, 8 = no barriers
// NI = n-tiles (16 columns) per wave: 4 -> 256x256 block tile, 5 -> 256x320.  The wide flavour exists for
// the wave quantisation: N = 1280 / 3840 / 5120 at M = 32768 give 512 / 1536 / 2048 tiles = exactly
// 2 / 6 / 8 rounds over 256 CUs (256x256: 2.5 / 7.5 / 10), with 40 MFMAs per 13 fragment reads and
// 4.5 DMA pieces per wave step.  Its 20 B pieces do not divide by 8 waves: waves 0-3 (= stagger group
// 0) issue a third B piece, so the counted vmcnt waits are per group.
template <int PREC, bool OUT_F32, bool GELU, int ABL = 0, int NI = 4>
__global__ __launch_bounds__(QTHREADS) void gemm_et_big_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period,
    int M, int N, int K, int accumulate) {
    constexpr int QBN = 64 * NI;
    constexpr int QSTAGE_ELEMS = (QBM + QBN) * QBK;                // NI = 4: 16384 ET = 32 KiB
    __shared__ __attribute__((aligned(16))) uint16_t lds[QSTAGES * QSTAGE_ELEMS];   // 128 / 144 KiB, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // waves w and w+4 share a SIMD -> different groups
    const int wm = wave >> 2, wn = wave & 3;   // wave tile rows wm*128.., cols wn*(16 NI)..

    constexpr int GROUP = 8;
    const int tiles_n = N / QBN, tiles_m = M / QBM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GROUP * tiles_n;
    const int group = bid / per_group, first_m = group * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = bid - group * per_group;
    const int tile_m = first_m + in_g % gsz, tile_n = in_g / gsz;
    const int m0 = tile_m * QBM, n0 = tile_n * QBN;

    // DMA map: one piece (wave instruction) = 1 KiB = 16 rows x 64 B; lane l -> row l>>2, physical
    // chunk l&3, source chunk (l&3) ^ g(row>>2).  Round i of 8 waves covers rows 128*i + 16*w .. +15.
    const int g_row = 16 * wave + (lane >> 2);
    const int g_chunk = qswz(g_row, lane & 3);
    const uint16_t* gAg = A + (size_t)(m0 + g_row) * K + g_chunk * 8;
    const uint16_t* gBg = B + (size_t)(n0 + g_row) * K + g_chunk * 8;
    const uint32_t wave_lds_base = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * (16 * QBK * 2));
    const size_t rs128 = (size_t)128 * K;
#define BIG_ISSUE(kt_, stage_)                                                                  \
    do {                                                                                         \
        const size_t koff_ = (size_t)(kt_) * QBK;                                                \
        constexpr int SB_ = (stage_) * QSTAGE_ELEMS * 2;                                         \
        glds16_asm<SB_ + 0>(gAg + koff_, wave_lds_base);                                         \
        glds16_asm<SB_ + 128 * QBK * 2>(gAg + rs128 + koff_, wave_lds_base);                     \
        glds16_asm<SB_ + QBM * QBK * 2>(gBg + koff_, wave_lds_base);                             \
        glds16_asm<SB_ + QBM * QBK * 2 + 128 * QBK * 2>(gBg + rs128 + koff_, wave_lds_base);     \
        if (NI == 5 && grp == 0)                                                                 \
            glds16_asm<SB_ + QBM * QBK * 2 + 256 * QBK * 2>(gBg + 2 * rs128 + koff_, wave_lds_base); \
    } while (0)

    f32x4_t acc[NI][8];    // [n-tile i][m-tile j]; D[i_local = n][j_local = m]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = K / QBK;
    // prologue: tiles 0..2 in flight, tile 0 landed
    if (!(ABL & 1)) {
    BIG_ISSUE(0, 0);
    if (nk > 1) BIG_ISSUE(1, 1);
    if (nk > 2) BIG_ISSUE(2, 2);
    }
    // outstanding-DMA budget: this wave's pieces per tile (4, or 5 for group 0 of the wide flavour)
#define BIG_VMCNT(tiles_)                                                                        \
    do {                                                                                         \
        if (NI == 5 && grp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((tiles_) * 5) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((tiles_) * 4) : "memory");                 \
    } while (0)
    if (nk > 2) BIG_VMCNT(2);
    else if (nk > 1) BIG_VMCNT(1);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1 && !(ABL & 8)) __builtin_amdgcn_s_barrier();          // stagger

    const int fr = lane & 15, fq = lane >> 4;
    // fragment byte-free offsets (elements) inside a stage; rows are fixed per lane, only the stage moves
    int offA[8], offB[NI];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = wm * 128 + j * 16 + fr; offA[j] = r * QBK + qswz(r, fq) * 8; }
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int r = wn * (16 * NI) + i * 16 + fr; offB[i] = QBM * QBK + r * QBK + qswz(r, fq) * 8; }

    // wait so that tile kt+1 has landed; tiles kt+2 / kt+3 may stay in flight
#define BIG_WAIT(kt_)                                                                            \
    if ((kt_) + 3 < nk) BIG_VMCNT(2);                                                            \
    else if ((kt_) + 2 < nk) BIG_VMCNT(1);                                                       \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define BIG_STEP(kt_, S_)                                                                        \
    if ((kt_) < nk) {                                                                            \
        uint4 fa[8], fb[NI];                                                                      \
        /* L: refill the slot freed last interval, read this step's fragments */                 \
        if (!(ABL & 1)) { if ((kt_) + 3 < nk) BIG_ISSUE((kt_) + 3, ((S_) + 3) % QSTAGES); }      \
        if (!(ABL & 2)) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                           \
            fb[i] = *reinterpret_cast<const uint4*>(lds + (S_) * QSTAGE_ELEMS + offB[i]);        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
            fa[j] = *reinterpret_cast<const uint4*>(lds + (S_) * QSTAGE_ELEMS + offA[j]);        \
        } else {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) { fb[i] = make_uint4(lane, i, (kt_), 1); asm volatile("" : "+v"(fb[i].x)); } \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) { fa[j] = make_uint4(lane, j, (kt_), 2); asm volatile("" : "+v"(fa[j].x)); } \
        }                                                                                        \
        BIG_WAIT(kt_)                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();                                            \
        /* C: 32 MFMAs */                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                           \
        if (!(ABL & 4)) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < NI; ++i)                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                        \
                acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]);                           \
        } else {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(fb[i].x), "v"(fb[i].w)); \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(fa[j].x), "v"(fa[j].w)); \
        }                                                                                        \
        __builtin_amdgcn_s_setprio(0);                                                           \
        BIG_WAIT(kt_)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();                                            \
    }
    for (int kt = 0; kt < nk; kt += QSTAGES) {
        BIG_STEP(kt, 0)
        BIG_STEP(kt + 1, 1)
        BIG_STEP(kt + 2, 2)
        BIG_STEP(kt + 3, 3)
    }
    if (grp == 0 && !(ABL & 8)) __builtin_amdgcn_s_barrier();          // both groups: 2 + 2*nk barriers

    if (ABL & 16) {   // timing experiment: keep the accumulators alive, store nothing
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][3]));
        return;
    }
    {   // coalesced epilogue through the idle ring (16 KiB per wave)
        unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + wave * (QSTAGES * QSTAGE_ELEMS * 2 / 8);
        const float* pre2d = OUT_F32 ? nullptr : add2d;     // ET output: the 2-D addend goes in before the rounding
        if constexpr (NI == 5 && !OUT_F32) {
            epilogue_pair_et<PREC, GELU>(acc, reinterpret_cast<unsigned char*>(lds), Cv, bias, pre2d, add2d_period, N, m0 + wm * 128,
                                         n0 + (wn >> 1) * 160, wm, wn, lane);
        } else {
            epilogue_coalesced<PREC, OUT_F32, GELU, 8, OUT_F32 ? 2 : 4, NI>(acc, scr, Cv, bias, OUT_F32 ? add2d : nullptr, add2d_period,
                                                                             N, m0 + wm * 128, n0 + wn * (16 * NI), accumulate, lane, pre2d);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gemm_et_x64_kernel: the 256x256 / 256x320 staggered kernel re-staged in PAIR stages of 64 k.
//
// Why: the kernel above fetches a k-step of 32 ET = 64 bytes per row, i.e. HALF of a 128-byte cache line per
// row per LDS-DMA instruction, and asks for the other half one k-step (> 1 us) later, when the 32 KiB vector L1
// has long dropped the line: every line travels L2 -> L1 twice, and the per-CU L1 fill port (64 B/clk) is the
// measured ceiling of the fill path (DMA-only ablation: ~21-25 B/clk/CU useful).  Here ONE DMA
// instruction covers 8 rows x 128 bytes = 8 WHOLE lines (lanes 0-31: k-half 0, lanes 32-63: k-half 1 of the same
// rows), so each line crosses once.
//
// LDS image of a pair stage: 8-row blocks of 1 KiB = [k-half][8 rows][64 B]; inside a k-half the geometry
// (4 rows per 256-byte bank row, 16-byte chunk index XORed with g(row>>2)) is exactly the one of the kernel above,
// so its conflict-free ds_read_b128 pattern carries over unchanged.  Two pair stages = 128 / 144 KiB.
//
// Schedule (I_n = interval between consecutive block-wide raw barriers; group 1 = waves 4-7 runs one interval
// behind group 0, so that each SIMD always has one wave in its MFMA segment):
//     group 0, stage t:  I_4t   L: [issue DMA(t+1)] read fragments of k-half 0     group 1: C(t-1, half 1) [issue DMA(t+1)]
//                        I_4t+1 C: 8 NI MFMAs                                               L(t, half 0)
//                        I_4t+2 L: read fragments of k-half 1                               C(t, half 0)
//                        I_4t+3 C: 8 NI MFMAs, then vmcnt(0)                                L(t, half 1), then vmcnt(0)
// Stage t+1 goes into the buffer of stage t-1, whose last reads (group 1, I_4t-1) are complete at the barrier
// that opens I_4t; every wave retires its own pieces (vmcnt(0): one stage in flight per wave) before the barrier
// that closes I_4t+3, and the first read of stage t+1 comes after that barrier.  Each output element is accumulated
// over k in ascending 32-wide MFMA steps exactly like every other tile shape (bit-identical results).
// SPREAD: the DMA pieces are issued between the MFMAs of a C segment instead of as one burst.
// M % 256 == 0, N % (64 NI) == 0, K % 64 == 0.
// ---------------------------------------------------------------------------------------------
constexpr int XBK = 64;

// Start skew of the FIRST round of blocks (experiment / tuning knob, SAMRS_GEMM_SKEW=<xcd units>,<cu-group units> in
// 1024-cycle steps): all 256 CUs start a GEMM together and reach their epilogues together, so the output tile writes
// arrive as one HBM-bound burst per round while the matrix pipes idle.  Delaying the first block of XCD x by x units (and
// CU group g by g units) keeps the rounds of different XCDs out of phase for the whole launch: one XCD writes while the
// others compute.  Blocks of later rounds inherit the skew because each CU takes its next block when it finishes one.
static int g_x64_skew = [] {
    const char* v = getenv("SAMRS_GEMM_SKEW");
    if (!v) return 0;
    int a = 0, b = 0;
    sscanf(v, "%d,%d", &a, &b);
    return (a & 0xffff) | (b << 16);
}();

// LDS-DMA, scalar-base form: source = sbase (SGPR pair, wave-uniform) + voff (per-lane byte offset, one VGPR),
// destination = LDS byte address m0v (wave-uniform) + 16 * lane.  Issued from inline asm for the same reason as
// glds16_asm (hipcc must not see it in its waitcnt bookkeeping); the caller counts vmcnt by hand.
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t m0v) {
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

// SPLIT3 (reference-grade operand split, engine option bits 16 / 32): A = A_hi + A_lo, B = B_hi + B_lo, and the product
// A_lo B_hi^T + A_hi B_lo^T + A_hi B_hi^T is ONE pass over a K axis of three segments -- the accumulators stay in registers, no
// fp32 round trip through memory between the terms (the generic route: three accumulating launches).  Pair stage st of
// 3 K / 64 reads segment st / (K / 64): the only change to the main loop is where a stage's DMA source starts.
template <bool SPLIT3>
__device__ __forceinline__ const uint16_t* seg_src_a(const uint16_t* hi, const uint16_t* lo, int st, int nst1) {
    if constexpr (!SPLIT3) return hi + (size_t)st * XBK;
    const int seg = (st >= nst1) + (st >= 2 * nst1);
    return (seg == 0 ? lo : hi) + (size_t)(st - seg * nst1) * XBK;
}
template <bool SPLIT3>
__device__ __forceinline__ const uint16_t* seg_src_b(const uint16_t* hi, const uint16_t* lo, int st, int nst1) {
    if constexpr (!SPLIT3) return hi + (size_t)st * XBK;
    const int seg = (st >= nst1) + (st >= 2 * nst1);
    return (seg == 1 ? lo : hi) + (size_t)(st - seg * nst1) * XBK;
}

// LNT (round 5): the LayerNorm that FOLLOWS this GEMM (image_encoder.py:177 norm2 after proj, :168 norm1 of the next block after
// lin2) as a tail of the GEMM itself.  The GEMM writes the fp32 residual stream in 256 x 320 tiles, a LayerNorm row needs all N = 1280
// columns = the four tiles of a 256-row panel, which four CUs of one XCD compute side by side; the stand-alone LayerNorm then re-reads
// the 168 MB stream from HBM / MALL (53 us per launch, 64 launches per encoder pass = 6 % of the step for zero FLOPs).  Here every block,
// once its own stores are complete, publishes them (agent-scope release by one lane after a block barrier: cdna_hip_programming.md
// 6 G16) and bumps a counter of its row panel; the block that finds the other tiles_n - 1 already counted -- the LAST ARRIVER -- makes
// them visible to itself (agent-scope acquire) and normalises the panel's 256 rows with layernorm_kernel's arithmetic (bit-identical
// output).  Nobody ever waits for anybody (no spin, no ordering or placement assumption: whichever block arrives last does the
// work), the counter is reset by the last arriver for the next launch.  A variant that wrote the tiles with sc1 stores instead of
// the release fence was no faster and WRONG (plain loads on another CU still hit the L2 lines the residual read had allocated).
// N <= 1280 must be the LayerNorm width.
struct LnTail {
    const float* gamma = nullptr;
    const float* beta = nullptr;
    uint16_t* out = nullptr;           // [M][N] in the operand type
    unsigned int* counters = nullptr;  // [M / 256], zero before the first launch
    float eps = 0.f;
};

// EXT (round 6, the outlier-column extension of proj / lin2: engine.hip EncBlock::oc_*): ONE more pair stage behind the K axis whose
// 64 k come from two dense side operands A_x [M][64], B_x [N][64] (the hi + lo split of up to 32 outlier columns: A_lo | A_hi against
// B_hi | B_lo) -- the operand A is written by another kernel with its own row stride, so the extra columns cannot simply be appended
// to its rows as for qkv / lin1.  Only the DMA source of that one stage differs (row stride 64 instead of K); passed in the A_lo /
// B_lo parameters (EXT and SPLIT3 exclude each other).
template <int PREC, bool OUT_F32, bool GELU, int NI = 4, int MODE = 0, int ABL = 0, bool SPLIT3 = false, bool LNT = false, bool EXT = false>
__global__ __launch_bounds__(QTHREADS) void gemm_et_x64_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period,
    int M, int N, int K, int accumulate, int skew,
    const uint16_t* __restrict__ A_lo = nullptr, const uint16_t* __restrict__ B_lo = nullptr, LnTail ln = LnTail()) {
    static_assert(!LNT || (OUT_F32 && !GELU), "the LayerNorm tail follows the fp32 residual outputs");
    static_assert(!(EXT && SPLIT3), "the side operands of EXT travel in the A_lo / B_lo parameters");
    constexpr int XBN = 64 * NI;
    constexpr int XROWS = QBM + XBN;                       // 512 / 576 rows per stage
    constexpr int XSTAGE_ELEMS = XROWS * XBK;              // 64 / 72 KiB
    constexpr uint32_t XSB = XSTAGE_ELEMS * 2;             // stage bytes
    constexpr int NPIECE = 4 + NI;                         // DMA pieces per wave and stage (8 rows x 128 B each): 4 A + NI B
    constexpr bool SPREAD = (MODE & 1) != 0;               // DMA pieces issued between the MFMAs instead of as one burst
    // LIGHT: ONE block-wide barrier per pair stage (the one that hands a landed stage over) instead of four.  The
    // barriers between the L and C segments of a stage guard no LDS hazard -- they only keep the two waves of a
    // SIMD in complementary segments, at ~150 cycles of re-convergence per 640-cycle MFMA segment.  Without them
    // group 1 still runs half a stage behind group 0 (its rendezvous sits between L(t, 1) and C(t, 1)), so its
    // fragment reads fall into the other wave's MFMA segments by themselves.
    constexpr bool LIGHT = (MODE & 2) != 0;
    constexpr bool NOPRIO = (MODE & 4) != 0;               // experiment: no s_setprio around the MFMA segments
    // ABL (timing experiments only, results are garbage): 1 no DMA, 2 no fragment reads, 4 no MFMA, 16 no epilogue
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * XSTAGE_ELEMS];   // 128 / 144 KiB, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // waves w and w+4 share a SIMD -> different groups
    const int wm = wave >> 2, wn = wave & 3;   // wave tile rows wm*128.., cols wn*(16 NI)..

    constexpr int GROUP = 8;
    const int tiles_n = N / XBN, tiles_m = M / QBM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GROUP * tiles_n;
    const int group = bid / per_group, first_m = group * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = bid - group * per_group;
    const int tile_m = first_m + in_g % gsz, tile_n = in_g / gsz;
    const int m0 = tile_m * QBM, n0 = tile_n * XBN;

    // DMA map: piece q of this wave covers stage rows 64 q + 8 wave .. + 7 (q < 4: A rows, else B rows 64 (q - 4) + ..);
    // lane l -> k-half l>>5, row (l>>2)&7, physical chunk l&3 which holds source chunk (l&3) ^ g(row>>2) (g depends on
    // row mod 16 only, and 64 q = 0 mod 16).  The per-lane byte offset is the same for A and B (both have row stride K).
    const int prow = 8 * wave + ((lane >> 2) & 7);
    const uint32_t voff = ((uint32_t)prow * (uint32_t)K + (uint32_t)(lane >> 5) * 32u + (uint32_t)qswz(prow, lane & 3) * 8u) * 2u;
    const uint16_t* sA = A + (size_t)m0 * K;               // wave-uniform bases (SGPR pairs)
    const uint16_t* sB = B + (size_t)n0 * K;
    const uint16_t* sAl = SPLIT3 ? A_lo + (size_t)m0 * K : nullptr;
    const uint16_t* sBl = SPLIT3 ? B_lo + (size_t)n0 * K : nullptr;
    const int nst1 = K / XBK;                              // pair stages per K segment
    const size_t rs64 = (size_t)64 * K;
    // EXT: the side operands' tile bases and the per-lane offset on their 64-element rows
    const uint16_t* sAx = EXT ? A_lo + (size_t)m0 * XBK : nullptr;
    const uint16_t* sBx = EXT ? B_lo + (size_t)n0 * XBK : nullptr;
    const uint32_t voffx = ((uint32_t)prow * (uint32_t)XBK + (uint32_t)(lane >> 5) * 32u + (uint32_t)qswz(prow, lane & 3) * 8u) * 2u;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * 1024u);
    // piece q_ (literal) of pair stage st_ into the buffer at byte offset wr_
    // EXT: where the pieces of a stage come from is chosen ONCE per stage (X64_SEL: wave-uniform selects into xpa / xpb / xprs and one
    // v_cndmask for the lane offset), not by a branch in front of each of its nine pieces inside the MFMA stream (the first form: lin2
    // + 17 us for 1 / 80 more work).  The other instantiations never see these variables.
    const uint16_t *xpa = sA, *xpb = sB;
    size_t xprs = rs64;
    uint32_t xvo = voff;
#define X64_SEL(st_)                                                                                       \
    if constexpr (EXT) {                                                                                   \
        const bool xs_ = (st_) == nst1;                                                                    \
        xpa = xs_ ? sAx : sA + (size_t)(st_) * XBK;                                                        \
        xpb = xs_ ? sBx : sB + (size_t)(st_) * XBK;                                                        \
        xprs = xs_ ? (size_t)(64 * XBK) : rs64;                                                            \
        xvo = xs_ ? voffx : voff;                                                                          \
    }
#define X64_PIECE(st_, wr_, q_)                                                                            \
    if constexpr (!(ABL & 1)) {                                                                            \
        if constexpr (EXT)                                                                                 \
            glds16_s(xvo, ((q_) < 4 ? xpa + (size_t)(q_) * xprs : xpb + (size_t)((q_) - 4) * xprs),                                      \
                     lds0 + (wr_) + ((q_) < 4 ? (q_) * 8192u : (uint32_t)(QBM * XBK * 2) + ((q_) - 4) * 8192u));                         \
        else                                                                                               \
            glds16_s(voff, ((q_) < 4 ? seg_src_a<SPLIT3>(sA, sAl, (st_), nst1) + (size_t)(q_) * rs64                                    \
                                      : seg_src_b<SPLIT3>(sB, sBl, (st_), nst1) + (size_t)((q_) - 4) * rs64),                            \
                     lds0 + (wr_) + ((q_) < 4 ? (q_) * 8192u : (uint32_t)(QBM * XBK * 2) + ((q_) - 4) * 8192u));                         \
    }
#define X64_ISSUE(st_, wr_)                                                                                \
    do {                                                                                                   \
        X64_SEL(st_)                                                                                       \
        X64_PIECE(st_, wr_, 0); X64_PIECE(st_, wr_, 1); X64_PIECE(st_, wr_, 2); X64_PIECE(st_, wr_, 3);    \
        X64_PIECE(st_, wr_, 4); X64_PIECE(st_, wr_, 5); X64_PIECE(st_, wr_, 6); X64_PIECE(st_, wr_, 7);    \
        if constexpr (NPIECE == 9) X64_PIECE(st_, wr_, 8);                                                 \
    } while (0)

    f32x4_t acc[NI][8];    // [n-tile i][m-tile j]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nst = (SPLIT3 ? 3 : 1) * nst1 + (EXT ? 1 : 0);
    const int fr = lane & 15, fq = lane >> 4;
    // fragment BYTE offsets inside a stage for k-half 0; k-half 1 is +512
    uint32_t offA[8], offB[NI];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = wm * 128 + j * 16 + fr; offA[j] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = QBM + wn * (16 * NI) + i * 16 + fr;
        offB[i] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2;
    }
    const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);

    // ABL 64 (timing experiment): wave 0 records s_memtime stamps into the buffer passed as add2d: 32 x u64 per block
    // [0] entry, [1] stage 0 landed, [2 + t] end of stage t (t < 24), [26] loop done, [27] epilogue done, [28] HW_ID | XCC_ID << 32
    unsigned long long* tl = nullptr;
    if constexpr (ABL & 64) {
        tl = reinterpret_cast<unsigned long long*>(const_cast<float*>(add2d)) + (size_t)blockIdx.x * 32;
        if (tid == 0) {
            tl[0] = __builtin_amdgcn_s_memtime();
            tl[28] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                     ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        }
    }
    if (skew && blockIdx.x < 256) {            // first round only (one block per CU; block b starts on XCD b % 8)
        const int n = (int)(blockIdx.x & 7) * (skew & 0xffff) + (int)((blockIdx.x >> 3) & 3) * (skew >> 16);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 cycles
    }
    // prologue: stage 0 landed for everybody
    X64_ISSUE(0, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 64) { if (tid == 0) tl[1] = __builtin_amdgcn_s_memtime(); }
    if (grp == 1) {                            // I_0 of group 1: nothing to compute yet; its share of stage 1 goes out
        if (nst > 1) X64_ISSUE(1, XSB);
        __builtin_amdgcn_s_barrier();
    }

#define X64_READ(rd_, kh_)                                                                                 \
    if constexpr (!(ABL & 2)) {                                                                            \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                         \
        fb[i] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offB[i]);                     \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                          \
        fa[j] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offA[j]);                     \
    } else {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) { fb[i] = make_uint4(lane, i, (rd_), 1); asm volatile("" : "+v"(fb[i].x)); } \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) { fa[j] = make_uint4(lane, j, (rd_), 2); asm volatile("" : "+v"(fa[j].x)); } \
    }
    // 8 NI MFMAs of one k-half; when dma_ (wave-uniform) is set, this wave's pieces of stage st_ go out first (burst)
    // or one per m-tile row of MFMAs (SPREAD).  The accumulators never sit inside a conditional region.
#define X64_MFMA(dma_, st_, wr_)                                                                           \
    if (!SPREAD) { if (dma_) X64_ISSUE(st_, wr_); }                                                        \
    if (SPREAD) { X64_SEL(st_) }                                                                           \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
        if constexpr (!(ABL & 4)) {                                                                        \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]); \
        } else {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(fb[i].x), "v"(fb[i].w), "v"(fa[j].x), "v"(fa[j].w)); \
        }                                                                                                  \
        if (SPREAD) {                                                                                      \
            if (dma_) {                                                                                    \
                if (j == 0) X64_PIECE(st_, wr_, 0); if (j == 1) X64_PIECE(st_, wr_, 1);                    \
                if (j == 2) X64_PIECE(st_, wr_, 2); if (j == 3) X64_PIECE(st_, wr_, 3);                    \
                if (j == 4) X64_PIECE(st_, wr_, 4); if (j == 5) X64_PIECE(st_, wr_, 5);                    \
                if (j == 6) X64_PIECE(st_, wr_, 6);                                                        \
                if (j == 7) { X64_PIECE(st_, wr_, 7); if constexpr (NPIECE == 9) X64_PIECE(st_, wr_, 8); } \
            }                                                                                              \
        }                                                                                                  \
    }
    // intra-stage barrier (kept only in the four-barrier schedule)
#define X64_MIDBAR() if constexpr (!LIGHT) __builtin_amdgcn_s_barrier();

    uint32_t rd = 0;                           // byte offset of the buffer that holds stage t
    for (int t = 0; t < nst; ++t) {
        uint4 fa[8], fb[NI];
        const uint32_t wr = XSB - rd;
        const bool dma0 = (grp == 0) && (t + 1 < nst);     // group 0 feeds stage t+1 from I_4t
        const bool dma1 = (grp == 1) && (t + 2 < nst);     // group 1 feeds stage t+2 from its C(t, 1) = I_4(t+1)
        // LIGHT: rendezvous R_t -- group 0 arrives here (stage t-1 finished, its pieces of stage t landed), group 1
        // from between L(t-1, 1) and C(t-1, 1) (t = 0: from the prologue)
        if constexpr (LIGHT) { if (grp == 0) __builtin_amdgcn_s_barrier(); }
        // ---- L(t, 0) ----
        if (!SPREAD) { if (dma0) X64_ISSUE(t + 1, wr); }
        X64_READ(rd, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        X64_MIDBAR()
        // ---- C(t, 0) ----
        if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(1);
        X64_MFMA(SPREAD && dma0, t + 1, wr)
        if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        X64_MIDBAR()
        // ---- L(t, 1); group 1: its pieces of stage t+1 (issued one stage ago) must have landed ----
        X64_READ(rd, 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LIGHT) { if (grp == 1) __builtin_amdgcn_s_barrier(); } else __builtin_amdgcn_s_barrier();
        // ---- C(t, 1); group 1: buffer `rd` is free from here on (both groups have read k-half 1) ----
        if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(1);
        X64_MFMA(dma1, t + 2, rd)
        if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LIGHT) __builtin_amdgcn_s_barrier();
        if constexpr (ABL & 64) { if (tid == 0 && t < 24) tl[2 + t] = __builtin_amdgcn_s_memtime(); }
        rd = wr;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // both groups: same barrier count; every ring read is done

    if constexpr (ABL & 64) { if (tid == 0) tl[26] = __builtin_amdgcn_s_memtime(); }
    if constexpr (ABL & 16) {   // timing experiment: keep the accumulators alive, store nothing
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][3]));
        return;
    }
    {   // coalesced epilogue through the idle ring (16 / 18 KiB per wave), identical to the kernel above
        unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + wave * (2 * XSB / 8);
        const float* pre2d = OUT_F32 ? nullptr : add2d;     // ET output: the 2-D addend goes in before the rounding
        // ABL 32 (timing experiment): every block stores to the SAME 256 x 64 NI window -> the stores stay in the L2
        const int em0 = (ABL & 32) ? 0 : m0, en0 = (ABL & 32) ? 0 : n0;
        if constexpr (NI == 5 && !OUT_F32) {
            epilogue_pair_et<PREC, GELU>(acc, reinterpret_cast<unsigned char*>(lds), Cv, bias, pre2d, add2d_period, N, em0 + wm * 128,
                                         en0 + (wn >> 1) * 160, wm, wn, lane);
        } else {
            epilogue_coalesced<PREC, OUT_F32, GELU, 8, OUT_F32 ? 2 : 4, NI>(acc, scr, Cv, bias, OUT_F32 ? add2d : nullptr, add2d_period,
                                                                             N, em0 + wm * 128, en0 + wn * (16 * NI), accumulate, lane, pre2d);
        }
    }
    if constexpr (ABL & 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores acknowledged: what s_endpgm would wait for anyway
        if (tid == 0) tl[27] = __builtin_amdgcn_s_memtime();
    }
    if constexpr (LNT) {
        int* flag = reinterpret_cast<int*>(lds);               // the ring is idle: every wave is past its bounce after the barrier below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores of the tile have completed
        __syncthreads();                                       // ... everybody's
        if (threadIdx.x == 0) {
            // publish the tile (agent-scope release: L2 write-back), count it, and if it was the panel's last: make the other
            // tiles visible to this CU (agent-scope acquire).  cdna_hip_programming.md 6 G16: placement-independent.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned int old = __hip_atomic_fetch_add(ln.counters + tile_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned int)(tiles_n - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(ln.counters + tile_m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        // ---- last arriver: LayerNorm of rows m0 .. m0 + 255: wave w takes rows 32 w .. 32 w + 31, one row at a time with
        // layernorm_kernel's arithmetic, the next row's loads in flight under the current row's.  Everything the tail needs is
        // RE-DERIVED here from an opaque copy of threadIdx.x: the kernel sits at 256 VGPRs, and every value kept alive across the
        // epilogue for the tail's sake turned into scratch traffic inside the epilogue of EVERY block (versions 1 - 3: forty spill
        // instructions there, each with a vmcnt(0) that serialises the residual loads: +185 us per launch, profiles/r05_ln_tail.txt).
        int tid_t = threadIdx.x;
        asm volatile("" : "+v"(tid_t));
        const int lane_t = tid_t & 63, wave_t = __builtin_amdgcn_readfirstlane(tid_t >> 6);
        // Latency: by the time the last tile lands most of the panel has left the 4 MB L2 (a round of tiles writes 10 MB per XCD), a
        // row costs a ~3 us trip to the Infinity Cache / HBM, and with ONE row in flight per wave the tail took 100 us (version 4:
        // proj + tail 340 us against 134 + 53, rocprofv3).  So a ring of eight row buffers per wave: 64 rows = 320 KB in flight per CU.
        constexpr int LNV = 5, LNP = 8;                        // float4 per lane and row (N <= 1280); rows in flight per wave
        const int nv = N >> 2;
        const float* X = reinterpret_cast<const float*>(Cv);
        float4 ring[LNP][LNV];
        const int row0 = tile_m * QBM + wave_t * 32;
#define LNT_LOAD(slot_, row_)                                                                              \
        {                                                                                                  \
            const float4* xr_ = reinterpret_cast<const float4*>(X + (size_t)(row_) * N);                    \
            _Pragma("unroll") for (int i = 0; i < LNV; ++i) { const int idx = lane_t + 64 * i; if (idx < nv) ring[slot_][i] = xr_[idx]; } \
        }
#pragma unroll
        for (int sl = 0; sl < LNP; ++sl) LNT_LOAD(sl, row0 + sl)
        for (int g = 0; g < 32 / LNP; ++g) {
#pragma unroll
            for (int sl = 0; sl < LNP; ++sl) {
                const int row = row0 + g * LNP + sl;
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < LNV; ++i) {
                    const int idx = lane_t + 64 * i;
                    if (idx < nv) sm += (ring[sl][i].x + ring[sl][i].y) + (ring[sl][i].z + ring[sl][i].w);
                }
                const float mean = wave_sum(sm) / (float)N;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < LNV; ++i) {
                    const int idx = lane_t + 64 * i;
                    if (idx < nv) {
                        const float a = ring[sl][i].x - mean, b = ring[sl][i].y - mean, c = ring[sl][i].z - mean, d = ring[sl][i].w - mean;
                        q += (a * a + b * b) + (c * c + d * d);
                    }
                }
                const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + ln.eps);
#pragma unroll
                for (int i = 0; i < LNV; ++i) {
                    const int idx = lane_t + 64 * i;
                    if (idx < nv) {
                        const float4 g4 = reinterpret_cast<const float4*>(ln.gamma)[idx];
                        const float4 bt = reinterpret_cast<const float4*>(ln.beta)[idx];
                        const float o0 = (ring[sl][i].x - mean) * rstd * g4.x + bt.x;
                        const float o1 = (ring[sl][i].y - mean) * rstd * g4.y + bt.y;
                        const float o2 = (ring[sl][i].z - mean) * rstd * g4.z + bt.z;
                        const float o3 = (ring[sl][i].w - mean) * rstd * g4.w + bt.w;
                        uint2 o;
                        o.x = pack2<PREC>(o0, o1);
                        o.y = pack2<PREC>(o2, o3);
                        reinterpret_cast<uint2*>(ln.out + (size_t)row * N)[idx] = o;
                    }
                }
                if (g + 1 < 32 / LNP) LNT_LOAD(sl, row + LNP)       // this slot's next row goes out as soon as the slot is free
            }
        }
#undef LNT_LOAD
    }
#undef X64_PIECE
#undef X64_SEL
#undef X64_ISSUE
#undef X64_READ
#undef X64_MFMA
#undef X64_MIDBAR
}

// ---------------------------------------------------------------------------------------------
// gemm_et_x64p_kernel: the pair-stage 256x320 kernel (one barrier per stage, spread DMA) made PERSISTENT: one block per
// CU walks tiles L, L + gridDim.x, ...  Measured on the one-tile-per-block kernel (s_memtime stamps, tools/gemm_timeline.py,
// qkv shape): of ~90 k cycles per tile, 3.2 k pass before stage 0 has landed, 2.3 k between a block's end and its successor's
// first instruction, and the block's last ~2 k wait for store acknowledgements.  Here the next tile's stage 0 goes out
// (into ring buffer 0, free since the last rendezvous) BEFORE the epilogue, whose bounce scratch is confined to ring buffer
// 1; every wave retires those pieces just before its first store (epilogue DRAIN), so the stores are never waited for: the
// next tile's main loop starts right after the epilogue, with its stage 0 already in LDS.  Main loop = the kernel above.
// add2d is not supported (the encoder's four big GEMMs have none).
// ---------------------------------------------------------------------------------------------
//
// FOLD: the LayerNorm in front of this GEMM (image_encoder.py:168,177) folded into it.  A = the raw residual stream rounded to
// ET (written by the GEMM that produced it, gemm_et_m32_kernel<STATS>), B = W diag(gamma) rounded to ET, cvec_n = sum_k of
// that B row, bias_n = b_n + sum_k W_nk beta_k:
//     LN(x) W^T + b  =  rstd (x W'^T - mean cvec) + bias'
// rowstat[m] = (rstd, -rstd mean) comes from ln_rowstat_kernel (encoder_kernels.hip), which merges the per-row (mean, M2)
// partials of the LN_NS 160-column groups; only the ET epilogue reads it (8 bytes per row and lane, next to the bias loads).
constexpr int LN_NS = 8;                       // 1280 / 160: ViT-H only

// MXO (round 4; ET output, no FOLD / SPLIT3): the epilogue also writes the output as MXFP4 hi / lo rows (epilogue_pair_et<MXO>: lin1 of
// split 207, whose lin2 takes MXFP4 lo terms while lin1 itself takes none).  The four output pointers travel in their own MxOut
// parameter (round 5: they used to be smuggled through rowstat / cvec / A_lo / B_lo with const_casts); the flavours that do not
// use it never load it from the kernarg segment, so their code is what it was.
template <int PREC, bool OUT_F32, int GELU /* 0 none, 1 fp32-epsilon erf, 2 the 1x-rate mode's cheaper erf (ET output only) */, bool FOLD = false,
          bool SPLIT3 = false, bool MXO = false, bool STREAM = false /* plain ET flavours: non-temporal output stores (stream_hidden) */>
__global__ __launch_bounds__(QTHREADS) void gemm_et_x64p_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, int M, int N, int K, int accumulate,
    const float2* __restrict__ rowstat = nullptr, const float* __restrict__ cvec = nullptr,
    const uint16_t* __restrict__ A_lo = nullptr, const uint16_t* __restrict__ B_lo = nullptr, int split_from_n = 0,
    MxOut mxo = MxOut(), int ld = 0 /* row stride of A and B in elements, 0 = K (gemm_ld_ok: operands padded off the 2560-byte stride) */) {
    static_assert(!(FOLD && OUT_F32), "the folded LayerNorm feeds ET outputs only (qkv, lin1)");
    static_assert(!(FOLD && SPLIT3), "the split operands come from an explicit LayerNorm");
    static_assert(!(MXO && (FOLD || SPLIT3 || OUT_F32)), "MX rows go with the plain ET-output flavour");
    constexpr int NI = 5;
    constexpr int XBN = 64 * NI;
    constexpr int XROWS = QBM + XBN;
    constexpr int XSTAGE_ELEMS = XROWS * XBK;
    constexpr uint32_t XSB = XSTAGE_ELEMS * 2;
    constexpr int NPIECE = 4 + NI;
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * XSTAGE_ELEMS];   // 144 KiB, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 2, wn = wave & 3;

    constexpr int GROUP = 8;
    const int tiles_n = N / XBN, tiles_m = M / QBM, ntiles = tiles_n * tiles_m;
    const int per_group = GROUP * tiles_n;
#define X64P_TILE(L_, m_, n_)                                                                    \
    do {                                                                                         \
        const int bid_ = xcd_remap((L_), ntiles);                                                \
        const int group_ = bid_ / per_group, first_m_ = group_ * GROUP;                           \
        const int gsz_ = (tiles_m - first_m_) < GROUP ? (tiles_m - first_m_) : GROUP;            \
        const int in_g_ = bid_ - group_ * per_group;                                             \
        (m_) = (first_m_ + in_g_ % gsz_) * QBM;                                                  \
        (n_) = (in_g_ / gsz_) * XBN;                                                             \
    } while (0)

    const int LD = ld > 0 ? ld : K;                // operand row stride (elements)
    const int prow = 8 * wave + ((lane >> 2) & 7);
    const uint32_t voff = ((uint32_t)prow * (uint32_t)LD + (uint32_t)(lane >> 5) * 32u + (uint32_t)qswz(prow, lane & 3) * 8u) * 2u;
    const size_t rs64 = (size_t)64 * LD;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * 1024u);
    const uint16_t* sA;                            // wave-uniform bases of the tile being FED (SGPR pairs)
    const uint16_t* sB;
    const uint16_t* sAl = nullptr;                 // SPLIT3: the lo operands of that tile
    const uint16_t* sBl = nullptr;
    const int nst1 = K / XBK;                      // pair stages per K segment
    // SPLIT3 with split_from_n > 0: only output columns >= split_from_n take the lo terms (qkv: the v third -- q and k pass
    // through the softmax, error_budget.py plans4); the other tiles start at the hi x hi segment: stage base sb = 2 nst1
    int sb = 0;                                    // of the tile being FED
#define X64P_PIECE(st_, wr_, q_)                                                                           \
    glds16_s(voff, ((q_) < 4 ? seg_src_a<SPLIT3>(sA, sAl, (st_) + sb, nst1) + (size_t)(q_) * rs64          \
                             : seg_src_b<SPLIT3>(sB, sBl, (st_) + sb, nst1) + (size_t)((q_) - 4) * rs64),  \
             lds0 + (wr_) + ((q_) < 4 ? (q_) * 8192u : (uint32_t)(QBM * XBK * 2) + ((q_) - 4) * 8192u))
#define X64P_ISSUE(st_, wr_)                                                                               \
    do {                                                                                                   \
        X64P_PIECE(st_, wr_, 0); X64P_PIECE(st_, wr_, 1); X64P_PIECE(st_, wr_, 2); X64P_PIECE(st_, wr_, 3); \
        X64P_PIECE(st_, wr_, 4); X64P_PIECE(st_, wr_, 5); X64P_PIECE(st_, wr_, 6); X64P_PIECE(st_, wr_, 7); \
        X64P_PIECE(st_, wr_, 8);                                                                           \
    } while (0)

    const int fr = lane & 15, fq = lane >> 4;
    uint32_t offA[8], offB[NI];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = wm * 128 + j * 16 + fr; offA[j] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = QBM + wn * (16 * NI) + i * 16 + fr;
        offB[i] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2;
    }
    const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);

#define X64P_READ(rd_, kh_)                                                                                \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                         \
        fb[i] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offB[i]);                     \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                          \
        fa[j] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offA[j]);
#define X64P_MFMA(dma_, st_, wr_)                                                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]); \
        if (dma_) {                                                                                        \
            if (j == 0) X64P_PIECE(st_, wr_, 0); if (j == 1) X64P_PIECE(st_, wr_, 1);                      \
            if (j == 2) X64P_PIECE(st_, wr_, 2); if (j == 3) X64P_PIECE(st_, wr_, 3);                      \
            if (j == 4) X64P_PIECE(st_, wr_, 4); if (j == 5) X64P_PIECE(st_, wr_, 5);                      \
            if (j == 6) X64P_PIECE(st_, wr_, 6);                                                           \
            if (j == 7) { X64P_PIECE(st_, wr_, 7); X64P_PIECE(st_, wr_, 8); }                              \
        }                                                                                                  \
    }

    int L = blockIdx.x, m0, n0;
    X64P_TILE(L, m0, n0);
    sA = A + (size_t)m0 * LD;
    sB = B + (size_t)n0 * LD;
    if constexpr (SPLIT3) { sAl = A_lo + (size_t)m0 * LD; sBl = B_lo + (size_t)n0 * LD; sb = n0 >= split_from_n ? 0 : 2 * nst1; }
    X64P_ISSUE(0, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4_t acc[NI][8];
    for (;;) {
        const int nst = SPLIT3 ? 3 * nst1 - sb : nst1;       // stages of THIS tile (sb still is its base here)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (grp == 1) {                        // rendezvous R_0 of group 1; its share of stage 1 goes out first
            if (nst > 1) X64P_ISSUE(1, XSB);
            __builtin_amdgcn_s_barrier();
        }
        uint32_t rd = 0;
        for (int t = 0; t < nst; ++t) {
            uint4 fa[8], fb[NI];
            const uint32_t wr = XSB - rd;
            const bool dma0 = (grp == 0) && (t + 1 < nst);
            const bool dma1 = (grp == 1) && (t + 2 < nst);
            if (grp == 0) __builtin_amdgcn_s_barrier();                       // R_t
            X64P_READ(rd, 0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            X64P_MFMA(dma0, t + 1, wr)
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            X64P_READ(rd, 1)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 1) __builtin_amdgcn_s_barrier();                       // R_{t+1} of group 1
            __builtin_amdgcn_s_setprio(1);
            X64P_MFMA(dma1, t + 2, rd)
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            rd = wr;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();      // last rendezvous: every ring read of this tile is done

        // next tile: its stage 0 goes into ring buffer 0 now and lands under the epilogue
        const int Ln = L + (int)gridDim.x;
        const bool more = Ln < ntiles;
        int m1 = m0, n1 = n0;
        if (more) {
            X64P_TILE(Ln, m1, n1);
            sA = A + (size_t)m1 * LD;
            sB = B + (size_t)n1 * LD;
            if constexpr (SPLIT3) { sAl = A_lo + (size_t)m1 * LD; sBl = B_lo + (size_t)n1 * LD; sb = n1 >= split_from_n ? 0 : 2 * nst1; }
            X64P_ISSUE(0, 0u);
        }
        {   // epilogue of tile (m0, n0); bounce scratch = ring buffer 1 only (72 KiB: 9 KiB per wave / 18 KiB per pair)
            unsigned char* upper = reinterpret_cast<unsigned char*>(lds) + XSB;
            if constexpr (MXO) {
                epilogue_pair_et<PREC, GELU, 2, true, false, true>(acc, upper, Cv, bias, nullptr, 1, N, m0 + wm * 128, n0 + (wn >> 1) * 160, wm, wn,
                                                                   lane, nullptr, nullptr, mxo);
            } else if constexpr (!OUT_F32) {
                epilogue_pair_et<PREC, GELU, 2, true, FOLD, false, STREAM>(acc, upper, Cv, bias, nullptr, 1, N, m0 + wm * 128, n0 + (wn >> 1) * 160, wm, wn,
                                                                           lane, rowstat, cvec);
            } else {
                epilogue_coalesced<PREC, true, GELU != 0, 8, 1, NI, false, true>(acc, upper + wave * (XSB / 8), Cv, bias, nullptr, 1, N,
                                                                           m0 + wm * 128, n0 + wn * (16 * NI), accumulate, lane);
            }
        }
        if (!more) break;
        __builtin_amdgcn_s_barrier();          // scratch free again; stage 0 visible (each wave drained its pieces before its first store)
        L = Ln; m0 = m1; n0 = n1;
    }
#undef X64P_TILE
#undef X64P_PIECE
#undef X64P_ISSUE
#undef X64P_READ
#undef X64P_MFMA
}

// erf form of the ET-output GELU epilogue of the persistent 256x320 kernel (lin1 of ViT-H): 1 = fp32-epsilon class (A-S 7.1.26),
// 2 = the cheaper one (A-S 7.1.28, common.h gelu_erf2_et); set around an engine's launches (engine.hip run_encoder)
thread_local int tl_gelu_form = 1;
// Row stride (elements) of the A and B operands of the calling thread's next plain ET launches, 0 = K.  Round 5, late: at K = 1280 an
// operand row is 2560 B = ten 256-byte units, so the 256 rows a tile fetches per k-slice fall on half of the memory channels; stored with a
// stride of 1408 elements (eleven units: every channel) the same kernels run their main loops ~4 % faster (tools/gemm_bench.py STRIDE_PROBE:
// lin1 + GELU 432.2 us at K = 1280, 451.5 at K = 1408 with 10 % more stages).  Only the persistent ET kernels take a stride (gemm_ld_ok says
// whether a launch will run on one of them); the engine pads the LayerNorm output and keeps padded copies of the qkv / lin1 weights.
thread_local int tl_gemm_ld = 0;
// lin1's output -- the MLP hidden tensor, read once, by lin2 -- is written with NON-TEMPORAL stores when it is larger than the 256 MB
// Infinity Cache it would otherwise be allocated in (8 tiles of ViT-H: 336 MB per launch; SAMRS_NT_HIDDEN=0 / 1 forces the choice).
// Measured on MI355X, libraries alternated on one box (profiles/r05_nt_streams.txt): lin1 424 -> 408 us in situ (the write-allocates no
// longer push the A / B panels out of the L2s), lin2 / proj -7 us on average (the residual stream survives longer), the 8-tile step
// +1.1 ... +1.4 % on three boxes.  The same hint on q | k | v, the LayerNorm or the attention output LOSES 2 - 3.5 % (their consumers live
// on finding them cached), and on lin2's A-row loads it gives the gain back (four tile columns re-read those rows from the L2).
static bool stream_hidden(long M, long N) {
    static const int mode = [] { const char* v = getenv("SAMRS_NT_HIDDEN"); return v ? atoi(v) : -1; }();
    if (mode >= 0) return mode != 0;
    return M * N * 2 >= (256L << 20);
}
template <int PREC>
hipError_t launch_gemm_x64p(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, bool out_f32, bool gelu,
                            bool accumulate, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / WBN);
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    dim3 grid(ntiles < n_cu ? ntiles : n_cu), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_x64p_kernel<PREC, true, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
        else gemm_et_x64p_kernel<PREC, true, false><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
    } else {
        const bool so = gelu && stream_hidden(M, N);
        const int ld = tl_gemm_ld;
#define X64P_ARGS a, b, C, bias, M, N, K, acc, nullptr, nullptr, nullptr, nullptr, 0, MxOut(), ld
        if (gelu && tl_gelu_form == 2 && so) gemm_et_x64p_kernel<PREC, false, 2, false, false, false, true><<<grid, block, 0, s>>>(X64P_ARGS);
        else if (gelu && tl_gelu_form == 2) gemm_et_x64p_kernel<PREC, false, 2><<<grid, block, 0, s>>>(X64P_ARGS);
        else if (gelu && so) gemm_et_x64p_kernel<PREC, false, true, false, false, false, true><<<grid, block, 0, s>>>(X64P_ARGS);
        else if (gelu) gemm_et_x64p_kernel<PREC, false, true><<<grid, block, 0, s>>>(X64P_ARGS);
        else gemm_et_x64p_kernel<PREC, false, false><<<grid, block, 0, s>>>(X64P_ARGS);
#undef X64P_ARGS
    }
    return hipGetLastError();
}

// ET output (+ GELU) that is also written as MXFP4 hi / lo rows: the persistent kernel with the MX-row epilogue, no lo terms taken
template <int PREC>
hipError_t launch_gemm_x64p_mxo(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, bool gelu,
                                void* o4_hi, void* o4_lo, void* so_hi, void* so_lo, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / WBN);
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    dim3 grid(ntiles < n_cu ? ntiles : n_cu), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    MxOut mxo;
    mxo.q_hi = reinterpret_cast<unsigned char*>(o4_hi);
    mxo.q_lo = reinterpret_cast<unsigned char*>(o4_lo);
    mxo.s_hi = reinterpret_cast<unsigned char*>(so_hi);
    mxo.s_lo = reinterpret_cast<unsigned char*>(so_lo);
    if (gelu) gemm_et_x64p_kernel<PREC, false, true, false, false, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, 0, nullptr, nullptr, nullptr, nullptr, 0, mxo);
    else gemm_et_x64p_kernel<PREC, false, false, false, false, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, 0, nullptr, nullptr, nullptr, nullptr, 0, mxo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// gemm_et_m32_kernel: the pair-stage 256x320 kernel (same LDS image, same whole-line LDS-DMA map) on
// v_mfma_f32_32x32x16 with a SYMMETRIC, register-double-buffered schedule.
//
// Why another main loop.  Measured on the 16x16x32 kernels above: a 64-k stage takes ~3.5 k cycles
// against 2.56 k of matrix-pipe time; one 16x16x32 MFMA occupies the pipe for ~17 cycles (MI355X_MICROARCH.md: 4.2
// quad-cycles, i.e. 6 % above its nominal 16), leaves ~4 issue slots before the next one is due, and an LDS-DMA piece
// (m0 write + s_nop + the load + address SALU) does not fit into such a gap -- every piece opens a bubble in the issuing
// wave's MFMA stream that only the partner wave can fill, which is why that schedule needs the two wave groups half a
// stage apart and L / C segments.  A 32x32x16 MFMA runs 32 cycles at the full rate, reads HALF the operand registers
// per FLOP (2 x 512 elements for 32 K FLOP instead of 16 K) and leaves ~8 issue slots: one ds_read_b128 plus one DMA
// piece fit behind every MFMA.  So here every wave runs ONE uninterrupted MFMA stream; the fragments of k-step s+1 are
// read while the MFMAs of step s run (two register sets), the DMA pieces ride in the same gaps, and there is no load
// segment and no group asymmetry.  The two waves of a SIMD share its matrix pipe by the hardware's age arbitration.
//
// Wave layout 4 (M) x 2 (N): wave tile 64 x 160 = 2 x 5 MFMA tiles, 160 accumulator registers; a wave owns whole
// 320-byte (f16) / 640-byte (fp32) row segments of the output, so the epilogue bounces through a wave-private scratch
// without block barriers.  Per 16-k step and wave: 10 MFMAs, 7 ds_read_b128 (14.3 KiB per 32 k against 13.3 KiB for
// the 128 x 80 wave tile).
//
// Stage hand-over (one block barrier per 64-k stage, placed between steps 2 and 3 of the stage):
//     steps 0, 1, 2 of stage t : MFMAs | reads of the next step's fragments (buffer t & 1)
//     lgkmcnt(0) (every fragment of stage t is in registers), vmcnt(0) (this wave's pieces of stage t+1 have landed), barrier B_t
//     step 3 of stage t        : MFMAs | reads of step 0 of stage t+1 (other buffer) | DMA pieces of stage t+2 -> buffer t & 1
// so a stage is in flight for three steps (~2 k cycles).  Persistent: the stream of stages simply continues into the next
// tile -- at t = nst-2 the pieces issued are the next tile's stage 0 (they land under the last stage and the epilogue);
// the epilogue bounces through buffer 1 (free after B_nst-1; nst must be even), and the next tile's stage 1 goes out
// right after it.  K % 128 == 0, M % 256 == 0, N % 320 == 0, no 2-D addend.
// k is accumulated in 16-wide MFMA steps, so results are NOT bit-identical with the 16x16x32 kernels (same fp32
// accumulation error class; tests/test_kernels_gpu.py::test_gemm_m32_*).
// ---------------------------------------------------------------------------------------------
#ifdef SAMRS_EXPERIMENTS   // the round-2 32x32x16 kernels (m32, w4) and the LayerNorm fold built on them: measured slower, tools / A-B builds only (make EXPERIMENTS=1)
constexpr int M32_RS = 144;                   // scratch row stride (bytes): 128 data bytes + 16; 36 words = 4 (mod 32)

// Epilogue of one wave tile: acc[i][j] = 32 (n) x 32 (m) block, lane (m = l & 31, h = l >> 5) holds n = 8 g + 4 h + 0..3
// for g = 0..3 in registers 4 g .. 4 g + 3.  Bounce through `scr` (wave-private, 32 rows x M32_RS bytes) so that the global
// accesses are whole 128-byte row segments, 8 rows per instruction.
// STATS (fp32 output = the residual stream; the LayerNorm that follows is folded into the next GEMM, gemm_et_x64p_kernel<FOLD>):
// besides C the final values are written rounded to ET into Xh (same [M][N] shape), and stats[m][n_base / 160] receives
// (mean, M2 = sum of squared deviations) of the wave's 160 columns of row m.  In the read-out a row's 32 columns of a pass sit
// in 8 consecutive lanes: group mean / M2 by three xor-shuffles each (two-pass inside the group), merged into the running
// pair over the five passes by Chan's update -- no E[x^2] - mean^2 cancellation, fixed order, bit-reproducible.
template <int PREC, bool OUT_F32, bool GELU, bool STATS = false, int NJ = 2>
__device__ __forceinline__ void epilogue_m32(f32x16_t (&acc)[5][NJ], unsigned char* scr, void* __restrict__ Cv,
                                             const float* __restrict__ bias, int N, int m_base, int n_base, int wn,
                                             int accumulate, int lane, uint16_t* __restrict__ Xh = nullptr,
                                             float2* __restrict__ stats = nullptr) {
    const int l31 = lane & 31, h = lane >> 5;
    if constexpr (OUT_F32) {
        float* C = reinterpret_cast<float*>(Cv);
        bool first = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float rmean[4], rm2[4];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                // the residual (old C) does not depend on the accumulators: its loads go out first
                float4 res[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int idx = lane + 64 * p, row = idx >> 3, ch = idx & 7;
                    res[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (accumulate) res[p] = *reinterpret_cast<const float4*>(C + (size_t)(m_base + j * 32 + row) * N + n_base + i * 32 + ch * 4);
                }
                if (!first) wave_lds_sync_g();
                first = false;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n_base + i * 32 + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float v0 = acc[i][j][4 * g] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
                    float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
                    if (GELU) { const float2_t g01 = gelu_erf2(float2_t{v0, v1}), g23 = gelu_erf2(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; }
                    *reinterpret_cast<float4*>(scr + l31 * M32_RS + (8 * g + 4 * h) * 4) = make_float4(v0, v1, v2, v3);
                }
                wave_lds_sync_g();
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int idx = lane + 64 * p, row = idx >> 3, ch = idx & 7;
                    float4 v = *reinterpret_cast<const float4*>(scr + row * M32_RS + ch * 16);
                    v.x += res[p].x; v.y += res[p].y; v.z += res[p].z; v.w += res[p].w;
                    const size_t o = (size_t)(m_base + j * 32 + row) * N + n_base + i * 32 + ch * 4;
                    *reinterpret_cast<float4*>(C + o) = v;
                    if constexpr (STATS) {
                        uint2 e;
                        e.x = pack2<PREC>(v.x, v.y);
                        e.y = pack2<PREC>(v.z, v.w);
                        *reinterpret_cast<uint2*>(Xh + o) = e;
                        float sm = (v.x + v.y) + (v.z + v.w);
                        sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
                        const float gm = sm * (1.0f / 32.0f);
                        const float d0 = v.x - gm, d1 = v.y - gm, d2 = v.z - gm, d3 = v.w - gm;
                        float gq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                        gq += __shfl_xor(gq, 1, 64); gq += __shfl_xor(gq, 2, 64); gq += __shfl_xor(gq, 4, 64);
                        if (i == 0) { rmean[p] = gm; rm2[p] = gq; }
                        else {
                            const float d = gm - rmean[p];
                            rmean[p] += d * (1.0f / (i + 1));
                            rm2[p] += gq + d * d * (32.0f * i / (i + 1));
                        }
                    }
                }
            }
            if constexpr (STATS) {
                if ((lane & 7) == 0) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        stats[(size_t)(m_base + j * 32 + (lane >> 3) + 8 * p) * (N / 160) + n_base / 160] = make_float2(rmean[p], rm2[p]);
                }
            }
        }
    } else {
        uint16_t* C = reinterpret_cast<uint16_t*>(Cv);
        // one pass = n-tiles [I0, I0 + CNT): CNT * 64 bytes per row.  The wave's 320-byte row segment starts on a 128-byte
        // line for wn = 0 and in the middle of one for wn = 1, so the passes are grouped {0,1}{2,3}{4} / {0}{1,2}{3,4}: every
        // two-tile pass writes whole lines.
#define M32_ET_PASS(j_, I0, CNT)                                                                                     \
        do {                                                                                                         \
            wave_lds_sync_g();                                                                                       \
            _Pragma("unroll") for (int ii = 0; ii < (CNT); ++ii)                                                     \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                          \
                const int i = (I0) + ii;                                                                             \
                const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n_base + i * 32 + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f); \
                float v0 = acc[i][j_][4 * g] + bv.x, v1 = acc[i][j_][4 * g + 1] + bv.y;                              \
                float v2 = acc[i][j_][4 * g + 2] + bv.z, v3 = acc[i][j_][4 * g + 3] + bv.w;                          \
                if (GELU) { const float2_t g01 = gelu_erf2(float2_t{v0, v1}), g23 = gelu_erf2(float2_t{v2, v3}); v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y; } \
                uint2 o;                                                                                             \
                o.x = pack2<PREC>(v0, v1);                                                                           \
                o.y = pack2<PREC>(v2, v3);                                                                           \
                *reinterpret_cast<uint2*>(scr + l31 * M32_RS + (ii * 32 + 8 * g + 4 * h) * 2) = o;                   \
            }                                                                                                        \
            wave_lds_sync_g();                                                                                       \
            _Pragma("unroll") for (int p = 0; p < 2 * (CNT); ++p) {                                                  \
                const int idx = lane + 64 * p, row = idx / (4 * (CNT)), ch = idx % (4 * (CNT));                      \
                const uint4 v = *reinterpret_cast<const uint4*>(scr + row * M32_RS + ch * 16);                       \
                *reinterpret_cast<uint4*>(C + (size_t)(m_base + (j_) * 32 + row) * N + n_base + (I0) * 32 + ch * 8) = v; \
            }                                                                                                        \
        } while (0)
        if (wn == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) { M32_ET_PASS(j, 0, 2); M32_ET_PASS(j, 2, 2); M32_ET_PASS(j, 4, 1); }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) { M32_ET_PASS(j, 0, 1); M32_ET_PASS(j, 1, 2); M32_ET_PASS(j, 3, 2); }
        }
#undef M32_ET_PASS
    }
}

#endif  // SAMRS_EXPERIMENTS (epilogue_m32)
// LDS-DMA piece with M0 declared as clobbered instead of saved / restored (two SALU instructions less per piece)
__device__ __forceinline__ void glds16_m(uint32_t voff, const void* sbase, uint32_t m0v) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(m0v)
        : "memory", "m0");
}

#ifdef SAMRS_EXPERIMENTS
// SPREAD: 0 = the 9 DMA pieces of a stage all go out in step 3 (one behind each of the first 9 MFMAs); 1 = pieces 0-4 in
// step 3 and pieces 5-8 in step 0 of the following stage (fewer fillers per gap, one step less in flight)
template <int PREC, bool OUT_F32, bool GELU, int SPREAD, bool STATS = false>
__global__ __launch_bounds__(QTHREADS) void gemm_et_m32_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, int M, int N, int K, int accumulate,
    uint16_t* __restrict__ Xh = nullptr, float2* __restrict__ stats = nullptr) {
    static_assert(!STATS || (OUT_F32 && !GELU), "row statistics accompany the fp32 residual output only");
    constexpr int NI = 5, NJ = 2;
    constexpr int XBN = 32 * NI * 2;                       // 320
    constexpr int XROWS = QBM + XBN;
    constexpr int XSTAGE_ELEMS = XROWS * XBK;
    constexpr uint32_t XSB = XSTAGE_ELEMS * 2;             // 72 KiB per stage
    constexpr int NP3 = SPREAD ? 5 : 9;                    // pieces [0, NP3) ride in step 3, [NP3, 9) in the next step 0
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * XSTAGE_ELEMS];   // 144 KiB, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;   // wave tile rows wm*64.., cols wn*160..

    constexpr int GROUP = 8;
    const int tiles_n = N / XBN, tiles_m = M / QBM, ntiles = tiles_n * tiles_m;
    const int per_group = GROUP * tiles_n;
#define M32_TILE(L_, m_, n_)                                                                     \
    do {                                                                                         \
        const int bid_ = xcd_remap((L_), ntiles);                                                \
        const int group_ = bid_ / per_group, first_m_ = group_ * GROUP;                           \
        const int gsz_ = (tiles_m - first_m_) < GROUP ? (tiles_m - first_m_) : GROUP;            \
        const int in_g_ = bid_ - group_ * per_group;                                             \
        (m_) = (first_m_ + in_g_ % gsz_) * QBM;                                                  \
        (n_) = (in_g_ / gsz_) * XBN;                                                             \
    } while (0)

    // DMA map: identical to the 16x16x32 pair-stage kernels (piece q of a wave = stage rows 64 q + 8 wave .. + 7)
    const int prow = 8 * wave + ((lane >> 2) & 7);
    const uint32_t voff = ((uint32_t)prow * (uint32_t)K + (uint32_t)(lane >> 5) * 32u + (uint32_t)qswz(prow, lane & 3) * 8u) * 2u;
    const size_t rs64 = (size_t)64 * K;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * 1024u);
    // piece q_ (literal) of the stage whose A / B rows start at pa_ / pb_ (wave-uniform) into the buffer at byte offset wr_
#define M32_PIECE(pa_, pb_, wr_, q_)                                                                       \
    glds16_m(voff, ((q_) < 4 ? (pa_) + (size_t)(q_) * rs64 : (pb_) + (size_t)((q_) - 4) * rs64),          \
             lds0 + (wr_) + ((q_) < 4 ? (q_) * 8192u : (uint32_t)(QBM * XBK * 2) + ((q_) - 4) * 8192u))

    // fragment BYTE offsets inside a stage for k-step 0 of k-half 0 (16-byte chunk h of the row); step parity 1 = chunk
    // h + 2 = the physical chunk XOR 2 (byte offset XOR 32), k-half 1 = +512
    const int nst = K / XBK;
    const int l31 = lane & 31, hh = lane >> 5;
    uint32_t offA[NJ], offB[NI];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int r = wm * 64 + j * 32 + l31; offA[j] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, hh) * 8) * 2; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = QBM + wn * 160 + i * 32 + l31;
        offB[i] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, hh) * 8) * 2;
    }
    const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);

    // fragments of k-step s_ (0..3) of the stage in the buffer at byte offset rd_ -> register set fa / fb
#define M32_RDB(fb_, rd_, s_, i_) fb_[i_] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + ((s_) >> 1) * 512 + (offB[i_] ^ (((s_) & 1) * 32u)))
#define M32_RDA(fa_, rd_, s_, j_) fa_[j_] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + ((s_) >> 1) * 512 + (offA[j_] ^ (((s_) & 1) * 32u)))
    // one k-step: 10 MFMAs on the set (ca_, cb_); behind MFMA k one read of the NEXT step's set (na_, nb_) from (nrd_, ns_)
    // and DMA piece Q0_ + k (while < Q1_) of the stage (pa_, pb_) -> buffer wr_: D_ = 0 none, 1 always, 2 when dma_ (wave-
    // uniform) is set.  The accumulators never sit inside a conditional region.
#define M32_SLOT(k_, i_, j_, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)             \
        acc[i_][j_] = ET<PREC>::mfma32(cb_[i_], ca_[j_], acc[i_][j_]);                                     \
        if constexpr ((k_) < 5) { M32_RDB(nb_, nrd_, ns_, ((k_) < 5 ? (k_) : 0)); }                        \
        else if constexpr ((k_) < 7) { M32_RDA(na_, nrd_, ns_, ((k_) >= 5 && (k_) < 7 ? (k_) - 5 : 0)); }  \
        if constexpr ((D_) != 0 && (Q0_) + (k_) < (Q1_)) {                                                 \
            if ((D_) == 1 || (dma_)) M32_PIECE(pa_, pb_, wr_, ((Q0_) + (k_) < 9 ? (Q0_) + (k_) : 0));      \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);
#define M32_STEP(ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                         \
        M32_SLOT(0, 0, 0, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(1, 1, 0, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(2, 2, 0, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(3, 3, 0, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(4, 4, 0, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(5, 0, 1, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(6, 1, 1, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(7, 2, 1, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(8, 3, 1, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                \
        M32_SLOT(9, 4, 1, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)
    // one 64-k stage held in buffer rd.  Step 0 carries the late pieces [NP3, 9) of the stage that goes into the OTHER buffer
    // (D0_, d0_, a0_, b0_); step 3, after the hand-over barrier, the early pieces [0, NP3) of the stage that goes into THIS
    // buffer (D3_, d3_, a3_, b3_).
#define M32_STAGE(D0_, d0_, a0_, b0_, D3_, d3_, a3_, b3_)                                                  \
        {                                                                                                  \
            const uint32_t ot = XSB - rd;                                                                  \
            M32_STEP(fa0, fb0, fa1, fb1, rd, 1, D0_, d0_, NP3, 9, a0_, b0_, ot)                            \
            M32_STEP(fa1, fb1, fa0, fb0, rd, 2, 0, false, 0, 0, a0_, b0_, ot)                              \
            M32_STEP(fa0, fb0, fa1, fb1, rd, 3, 0, false, 0, 0, a0_, b0_, ot)                              \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                    \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            M32_STEP(fa1, fb1, fa0, fb0, ot, 0, D3_, d3_, 0, NP3, a3_, b3_, rd)                            \
            rd = ot;                                                                                       \
        }

    int L = blockIdx.x, m0, n0;
    M32_TILE(L, m0, n0);
    const uint16_t* sA = A + (size_t)m0 * K;       // rows of the tile being computed (wave-uniform: SGPR pairs)
    const uint16_t* sB = B + (size_t)n0 * K;
    M32_PIECE(sA, sB, 0u, 0); M32_PIECE(sA, sB, 0u, 1); M32_PIECE(sA, sB, 0u, 2); M32_PIECE(sA, sB, 0u, 3); M32_PIECE(sA, sB, 0u, 4);
    M32_PIECE(sA, sB, 0u, 5); M32_PIECE(sA, sB, 0u, 6); M32_PIECE(sA, sB, 0u, 7); M32_PIECE(sA, sB, 0u, 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16_t acc[NI][NJ];
    for (;;) {
        // the tile after this one (its stage 0 is fed from inside this tile's main loop)
        const int Ln = L + (int)gridDim.x;
        const bool more = Ln < ntiles;
        int m1 = m0, n1 = n0;
        if (more) M32_TILE(Ln, m1, n1);
        const uint16_t* nA = A + (size_t)m1 * K;
        const uint16_t* nB = B + (size_t)n1 * K;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {   // early pieces of stage 1 -> buffer 1 (the late ones ride in step 0 of stage 0)
            const uint16_t* a1 = sA + XBK;
            const uint16_t* b1 = sB + XBK;
            M32_PIECE(a1, b1, XSB, 0); M32_PIECE(a1, b1, XSB, 1); M32_PIECE(a1, b1, XSB, 2); M32_PIECE(a1, b1, XSB, 3); M32_PIECE(a1, b1, XSB, 4);
            if constexpr (!SPREAD) { M32_PIECE(a1, b1, XSB, 5); M32_PIECE(a1, b1, XSB, 6); M32_PIECE(a1, b1, XSB, 7); M32_PIECE(a1, b1, XSB, 8); }
        }
        uint4 fa0[NJ], fb0[NI], fa1[NJ], fb1[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) M32_RDB(fb0, 0u, 0, i);
#pragma unroll
        for (int j = 0; j < NJ; ++j) M32_RDA(fa0, 0u, 0, j);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t rd = 0;
        // stage t: step 0 completes stage t+1, step 3 starts stage t+2
        const uint16_t* pa = sA + XBK;
        const uint16_t* pb = sB + XBK;
        for (int t = 0; t + 2 < nst; ++t) {
            M32_STAGE(1, true, pa, pb, 1, true, pa + XBK, pb + XBK)
            pa += XBK;
            pb += XBK;
        }
        // t = nst-2 (buffer 0): completes the last stage; starts the NEXT tile's stage 0 in buffer 0 when there is one
        M32_STAGE(1, true, pa, pb, 2, more, nA, nB)
        // t = nst-1 (buffer 1): completes the next tile's stage 0
        M32_STAGE(2, more, nA, nB, 0, false, nA, nB)
        // all fragment reads of this tile are complete (B_nst-1); buffer 1 held its last stage -> scratch
        {
            unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + XSB + wave * (XSB / 8);
            epilogue_m32<PREC, OUT_F32, GELU, STATS>(acc, scr, Cv, bias, N, m0 + wm * 64, n0 + wn * 160, wn, accumulate, lane, Xh, stats);
        }
        if (!more) break;
        __builtin_amdgcn_s_barrier();              // every wave is done with its scratch before stage 1 of the next tile lands there
        L = Ln; m0 = m1; n0 = n1; sA = nA; sB = nB;
    }
#undef M32_TILE
#undef M32_PIECE
#undef M32_RDA
#undef M32_RDB
#undef M32_SLOT
#undef M32_STEP
#undef M32_STAGE
}

// persistent = one block per CU walks the tiles; otherwise one tile per block (same kernel: `more` is never true)
template <int PREC, int SPREAD>
hipError_t launch_gemm_m32(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, bool out_f32, bool gelu,
                           bool accumulate, bool persistent, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / WBN);
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    dim3 grid(persistent && ntiles > n_cu ? n_cu : ntiles), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_m32_kernel<PREC, true, true, SPREAD><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
        else gemm_et_m32_kernel<PREC, true, false, SPREAD><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
    } else {
        if (gelu) gemm_et_m32_kernel<PREC, false, true, SPREAD><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
        else gemm_et_m32_kernel<PREC, false, false, SPREAD><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
    }
    return hipGetLastError();
}
#endif  // SAMRS_EXPERIMENTS (gemm_et_m32_kernel)
// ---------------------------------------------------------------------------------------------
// gemm_et_w4_kernel: the symmetric 32x32x16 schedule above on FOUR waves with 512 registers each (one wave per SIMD).
//
// Why: the DVFS probe says the encoder GEMMs are limited by energy per FLOP, and the one structural knob on it
// is the number of bytes that cross LDS -> registers per FLOP, i.e. the perimeter of the WAVE tile: 8 waves x (128 + 80) rows
// of fragments per k-step for the 2 x 4 layout, 8 x (64 + 160) for the 4 x 2 layout of gemm_et_m32_kernel, 4 x (128 + 160) here
// (-31 % / -36 %).  A 128 x 160 wave tile needs 320 accumulator registers; hipcc keeps all MFMA results of a kernel either in
// VGPRs or in AGPRs (256 each), so the MFMAs are issued from inline asm: the 16 accumulator tiles of n-tiles 0-3 are constrained
// to AGPRs ("+a"), the 4 tiles of n-tile 4 to VGPRs ("+v"); everything else (fragments 2 x 36, addressing) fits the VGPR half.
// The compiler does not know these statements are MFMAs, so the two hazards it would pad are padded by hand (s_nop after the
// accumulators are zeroed and before the epilogue reads them).
//
// Stage image, DMA pieces (now 18 per wave and stage: rows 32 q + 8 wave .. + 7), hand-over barrier, persistent stage stream and
// epilogue are those of gemm_et_m32_kernel; pieces 0-9 ride in step 3 (slots 10-19, the reads use slots 0-8), pieces 10-17 in
// the next step 0.  With one wave per SIMD nothing covers a stall, so the single barrier per stage costs what it costs.
// ---------------------------------------------------------------------------------------------
constexpr int W4THREADS = 256;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#ifdef SAMRS_EXPERIMENTS
template <int PREC, bool AG>
__device__ __forceinline__ void mfma32_asm(f32x16_t& c, const uint4& a, const uint4& b) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
    if constexpr (PREC == PREC_F16) {
        if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    } else {
        if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    }
}

template <int PREC, bool OUT_F32, bool GELU>
__global__ __attribute__((amdgpu_flat_work_group_size(W4THREADS, W4THREADS), amdgpu_waves_per_eu(1, 1))) void gemm_et_w4_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, int M, int N, int K, int accumulate) {
    constexpr int NI = 5, NJ = 4;
    constexpr int XBN = 320;
    constexpr int XROWS = QBM + XBN;
    constexpr int XSTAGE_ELEMS = XROWS * XBK;
    constexpr uint32_t XSB = XSTAGE_ELEMS * 2;             // 72 KiB per stage
    constexpr int NP = 18, NP3 = 10;                       // pieces per wave and stage; [0, NP3) in step 3, the rest in step 0
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * XSTAGE_ELEMS];   // 144 KiB, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;   // wave tile rows wm*128.., cols wn*160..

    constexpr int GROUP = 8;
    const int tiles_n = N / XBN, tiles_m = M / QBM, ntiles = tiles_n * tiles_m;
    const int per_group = GROUP * tiles_n;
#define W4_TILE(L_, m_, n_)                                                                      \
    do {                                                                                         \
        const int bid_ = xcd_remap((L_), ntiles);                                                \
        const int group_ = bid_ / per_group, first_m_ = group_ * GROUP;                           \
        const int gsz_ = (tiles_m - first_m_) < GROUP ? (tiles_m - first_m_) : GROUP;            \
        const int in_g_ = bid_ - group_ * per_group;                                             \
        (m_) = (first_m_ + in_g_ % gsz_) * QBM;                                                  \
        (n_) = (in_g_ / gsz_) * XBN;                                                             \
    } while (0)

    // DMA map: piece q of this wave covers stage rows 32 q + 8 wave .. + 7 (q < 8: A rows, else B rows 32 (q - 8) + ..); lane
    // l -> k-half l>>5, row (l>>2)&7, physical chunk l&3 (same image and swizzle as the eight-wave kernels: 32 q = 0 mod 16)
    const int prow = 8 * wave + ((lane >> 2) & 7);
    const uint32_t voff = ((uint32_t)prow * (uint32_t)K + (uint32_t)(lane >> 5) * 32u + (uint32_t)qswz(prow, lane & 3) * 8u) * 2u;
    const size_t rs32 = (size_t)32 * K;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * 1024u);
#define W4_PIECE(pa_, pb_, wr_, q_)                                                                        \
    glds16_m(voff, ((q_) < 8 ? (pa_) + (size_t)(q_) * rs32 : (pb_) + (size_t)((q_) - 8) * rs32), lds0 + (wr_) + (q_) * 4096u)

    const int nst = K / XBK;
    const int l31 = lane & 31, hh = lane >> 5;
    uint32_t offA[NJ], offB[NI];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int r = wm * 128 + j * 32 + l31; offA[j] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, hh) * 8) * 2; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = QBM + wn * 160 + i * 32 + l31;
        offB[i] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, hh) * 8) * 2;
    }
    const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);

#define W4_RDB(fb_, rd_, s_, i_) fb_[i_] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + ((s_) >> 1) * 512 + (offB[i_] ^ (((s_) & 1) * 32u)))
#define W4_RDA(fa_, rd_, s_, j_) fa_[j_] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + ((s_) >> 1) * 512 + (offA[j_] ^ (((s_) & 1) * 32u)))
    // slot k (0..19) of a k-step: MFMA on accumulator tile (i = k % 5, j = k / 5) of the set (ca_, cb_); slots 0-8: one read of
    // the NEXT step's set; slots 10-19: DMA piece Q0_ + (k - 10) while < Q1_ (D_: 0 none, 1 always, 2 when dma_ is set)
#define W4_SLOT(k_, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        mfma32_asm<PREC, ((k_) % 5) < 4>(acc[(k_) % 5][(k_) / 5], cb_[(k_) % 5], ca_[(k_) / 5]);           \
        if constexpr ((k_) < 5) { W4_RDB(nb_, nrd_, ns_, ((k_) < 5 ? (k_) : 0)); }                         \
        else if constexpr ((k_) < 9) { W4_RDA(na_, nrd_, ns_, ((k_) >= 5 && (k_) < 9 ? (k_) - 5 : 0)); }   \
        if constexpr ((D_) != 0 && (k_) >= 10 && (Q0_) + (k_) - 10 < (Q1_)) {                              \
            if ((D_) == 1 || (dma_)) W4_PIECE(pa_, pb_, wr_, ((k_) >= 10 && (Q0_) + (k_) - 10 < NP ? (Q0_) + (k_) - 10 : 0)); \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);
#define W4_STEP(ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                          \
        W4_SLOT(0, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(1, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(2, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(3, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(4, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(5, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(6, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(7, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(8, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(9, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                       \
        W4_SLOT(10, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(11, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(12, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(13, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(14, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(15, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(16, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(17, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(18, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)                      \
        W4_SLOT(19, ca_, cb_, na_, nb_, nrd_, ns_, D_, dma_, Q0_, Q1_, pa_, pb_, wr_)
#define W4_STAGE(D0_, d0_, a0_, b0_, D3_, d3_, a3_, b3_)                                                   \
        {                                                                                                  \
            const uint32_t ot = XSB - rd;                                                                  \
            W4_STEP(fa0, fb0, fa1, fb1, rd, 1, D0_, d0_, NP3, NP, a0_, b0_, ot)                            \
            W4_STEP(fa1, fb1, fa0, fb0, rd, 2, 0, false, 0, 0, a0_, b0_, ot)                               \
            W4_STEP(fa0, fb0, fa1, fb1, rd, 3, 0, false, 0, 0, a0_, b0_, ot)                               \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                    \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            W4_STEP(fa1, fb1, fa0, fb0, ot, 0, D3_, d3_, 0, NP3, a3_, b3_, rd)                             \
            rd = ot;                                                                                       \
        }
#define W4_ISSUE_RANGE(pa_, pb_, wr_, LO, HI)                                                              \
    do {                                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < NP; ++q_)                                                  \
            if (q_ >= (LO) && q_ < (HI)) {                                                                 \
                glds16_m(voff, (q_ < 8 ? (pa_) + (size_t)q_ * rs32 : (pb_) + (size_t)(q_ - 8) * rs32), lds0 + (wr_) + q_ * 4096u); \
            }                                                                                              \
    } while (0)

    int L = blockIdx.x, m0, n0;
    W4_TILE(L, m0, n0);
    const uint16_t* sA = A + (size_t)m0 * K;
    const uint16_t* sB = B + (size_t)n0 * K;
    W4_ISSUE_RANGE(sA, sB, 0u, 0, NP);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16_t acc[NI][NJ];
    for (;;) {
        const int Ln = L + (int)gridDim.x;
        const bool more = Ln < ntiles;
        int m1 = m0, n1 = n0;
        if (more) W4_TILE(Ln, m1, n1);
        const uint16_t* nA = A + (size_t)m1 * K;
        const uint16_t* nB = B + (size_t)n1 * K;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        {
            const uint16_t* a1 = sA + XBK;
            const uint16_t* b1 = sB + XBK;
            W4_ISSUE_RANGE(a1, b1, XSB, 0, NP3);           // early pieces of stage 1 -> buffer 1
        }
        uint4 fa0[NJ], fb0[NI], fa1[NJ], fb1[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) W4_RDB(fb0, 0u, 0, i);
#pragma unroll
        for (int j = 0; j < NJ; ++j) W4_RDA(fa0, 0u, 0, j);
        // accumulator writes (VALU / v_accvgpr_write) before the first MFMA: same tie
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j == NJ - 1)
                asm volatile("s_nop 7" : "+a"(acc[0][j]), "+a"(acc[1][j]), "+a"(acc[2][j]), "+a"(acc[3][j]), "+v"(acc[4][j]));
            else
                asm volatile("" : "+a"(acc[0][j]), "+a"(acc[1][j]), "+a"(acc[2][j]), "+a"(acc[3][j]), "+v"(acc[4][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t rd = 0;
        const uint16_t* pa = sA + XBK;
        const uint16_t* pb = sB + XBK;
        for (int t = 0; t + 2 < nst; ++t) {
            W4_STAGE(1, true, pa, pb, 1, true, pa + XBK, pb + XBK)
            pa += XBK;
            pb += XBK;
        }
        W4_STAGE(1, true, pa, pb, 2, more, nA, nB)         // t = nst-2
        W4_STAGE(2, more, nA, nB, 0, false, nA, nB)        // t = nst-1
        // The last MFMA results must not be read for 18 wait states, and hipcc neither knows that the asm statements above are
        // MFMAs nor keeps register-only code behind a bare asm: the padding is tied to every accumulator tile by data dependence.
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j == 0)
                asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][j]), "+a"(acc[1][j]), "+a"(acc[2][j]), "+a"(acc[3][j]), "+v"(acc[4][j]));
            else
                asm volatile("" : "+a"(acc[0][j]), "+a"(acc[1][j]), "+a"(acc[2][j]), "+a"(acc[3][j]), "+v"(acc[4][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + XSB + wave * (XSB / 4);
            epilogue_m32<PREC, OUT_F32, GELU, false, NJ>(acc, scr, Cv, bias, N, m0 + wm * 128, n0 + wn * 160, wn, accumulate, lane);
        }
        if (!more) break;
        __builtin_amdgcn_s_barrier();
        L = Ln; m0 = m1; n0 = n1; sA = nA; sB = nB;
    }
#undef W4_TILE
#undef W4_PIECE
#undef W4_RDA
#undef W4_RDB
#undef W4_SLOT
#undef W4_STEP
#undef W4_STAGE
#undef W4_ISSUE_RANGE
}

template <int PREC>
hipError_t launch_gemm_w4(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, bool out_f32, bool gelu,
                          bool accumulate, bool persistent, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / WBN);
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    dim3 grid(persistent && ntiles > n_cu ? n_cu : ntiles), block(W4THREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_w4_kernel<PREC, true, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
        else gemm_et_w4_kernel<PREC, true, false><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
    } else {
        if (gelu) gemm_et_w4_kernel<PREC, false, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
        else gemm_et_w4_kernel<PREC, false, false><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc);
    }
    return hipGetLastError();
}

#endif  // SAMRS_EXPERIMENTS (gemm_et_w4_kernel)
// ---------------------------------------------------------------------------------------------
// gemm_et_w4x_kernel (round 5): the vendor's tile geometry on this file's LDS-DMA pair stages -- 256 x 256 x 64 macro tile, FOUR waves
// with 512 registers each, wave tile 128 x 128 on v_mfma_f32_16x16x32 (the form tools/mfma_rate.hip measures 24 % faster than
// 32x32x16 in the bare pipe; the round-2 four-wave kernel above was built on 32x32x16).  VERDICT r04 item 1a.
//
//   * 64 accumulator tiles of 16 x 16 = 256 registers, ALL in AGPRs (inline-asm MFMAs with "+a"); the first k-step of a tile
//     writes them with C = 0, so no accumulator is ever zeroed by 256 v_accvgpr_write;
//   * per 32-k step and wave: 64 MFMAs (1024 pipe cycles), 16 ds_read_b128 (8 A + 8 B fragments: 0.25 per MFMA against 0.325 for the
//     128 x 80 wave tile of the eight-wave kernels), into the OTHER of two fragment register sets (128 VGPRs);
//   * a 64-k pair stage = 512 rows x 128 B = 64 KiB = 64 whole-line DMA pieces, 16 per wave, spread over the second step of the stage
//     one behind every second MFMA (a piece = s_mov m0 + the load; the per-piece source offset sits in eight VGPRs, the stage base
//     in two SGPR pairs);
//   * ONE block barrier per stage, between its two steps: before it every wave waits for its own pieces of stage t + 1 (issued a
//     whole stage earlier) and has finished reading stage t (its k-half 1 fragments were read during step 0); after it stage t + 2
//     goes into the buffer of stage t.  The stage stream runs on across tiles (persistent, one block per CU): the next tile's
//     stages 0 and 1 land under the last stage and the epilogue, which bounces through 17 KiB of LDS of its own.
// Same LDS image, swizzle and k order as gemm_et_x64_kernel: bit-identical output.  M % 256 == 0, N % 256 == 0, K % 64 == 0 (round 6: an
// odd number of stages per tile is fine -- the buffer parity simply carries over to the next tile).
// ---------------------------------------------------------------------------------------------
template <int PREC, bool ZERO>
__device__ __forceinline__ void mfma16_asm(f32x4_t& c, const uint4& a, const uint4& b) {
    const u32x4_t av = __builtin_bit_cast(u32x4_t, a), bv = __builtin_bit_cast(u32x4_t, b);
    if constexpr (PREC == PREC_F16) {
        if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
    } else {
        if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(av), "v"(bv));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(av), "v"(bv));
    }
}

constexpr int W4X_BN = 256;
constexpr int W4X_ROWS = QBM + W4X_BN;                 // 512 stage rows
constexpr int W4X_STAGE_ELEMS = W4X_ROWS * XBK;        // 64 KiB per stage
constexpr int W4X_SCR_BYTES = 4 * 16 * 272;            // epilogue scratch: per wave one m-tile of 16 rows x 272 B (epilogue_coalesced, NI = 8, JC = 1)

// ET outputs only: the fp32 epilogue bounces 8 KiB per wave and m-tile, which the 160 KiB do not hold next to two 64 KiB stages (and the
// fp32-output shapes of this path, N = 1280, are 2.5 rounds of 256 x 256 tiles: nothing to gain there).
template <int PREC, bool OUT_F32, int GELU, bool STREAM = false /* non-temporal output stores (stream_hidden) */>
__global__ __attribute__((amdgpu_flat_work_group_size(W4THREADS, W4THREADS), amdgpu_waves_per_eu(1, 1))) void gemm_et_w4x_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, int M, int N, int K, int accumulate, int ld = 0 /* row stride of A and B in elements, 0 = K */) {
    static_assert(!OUT_F32, "ET outputs only (epilogue scratch: 16 rows x 272 B per wave)");
    constexpr int NI = 8, NJ = 8;
    constexpr uint32_t XSB = W4X_STAGE_ELEMS * 2;
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * W4X_STAGE_ELEMS + W4X_SCR_BYTES / 2];   // 128 KiB ring + 17 KiB scratch, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;               // wave tile rows wm*128.., cols wn*128..

    constexpr int GROUP = 8;
    const int tiles_n = N / W4X_BN, tiles_m = M / QBM, ntiles = tiles_n * tiles_m;
    const int per_group = GROUP * tiles_n;
#define W4X_TILE(L_, m_, n_)                                                                     \
    do {                                                                                         \
        const int bid_ = xcd_remap((L_), ntiles);                                                \
        const int group_ = bid_ / per_group, first_m_ = group_ * GROUP;                           \
        const int gsz_ = (tiles_m - first_m_) < GROUP ? (tiles_m - first_m_) : GROUP;            \
        const int in_g_ = bid_ - group_ * per_group;                                             \
        (m_) = (first_m_ + in_g_ % gsz_) * QBM;                                                  \
        (n_) = (in_g_ / gsz_) * W4X_BN;                                                          \
    } while (0)

    // DMA map: piece q of this wave covers stage rows 32 q + 8 wave .. + 7 (q < 8: A rows, else B rows 32 (q - 8) + ..); lane l ->
    // k-half l>>5, row (l>>2)&7, physical chunk l&3 (the image and swizzle of the eight-wave kernels: 32 q = 0 mod 16).  The byte
    // offset of (piece q & 7, lane) from the stage base is the same for A and B: eight VGPRs.
    const int LD = ld > 0 ? ld : K;                        // operand row stride (elements)
    const int prow = 8 * wave + ((lane >> 2) & 7);
    uint32_t voffq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        voffq[q] = ((uint32_t)(prow + 32 * q) * (uint32_t)LD + (uint32_t)(lane >> 5) * 32u + (uint32_t)qswz(prow, lane & 3) * 8u) * 2u;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * 1024u);
#define W4X_PIECE(pa_, pb_, wr_, q_) glds16_m(voffq[(q_) & 7], ((q_) < 8 ? (pa_) : (pb_)), lds0 + (wr_) + (q_) * 4096u)
#define W4X_ISSUE_ALL(pa_, pb_, wr_)                                                                       \
    do {                                                                                                   \
        W4X_PIECE(pa_, pb_, wr_, 0); W4X_PIECE(pa_, pb_, wr_, 1); W4X_PIECE(pa_, pb_, wr_, 2); W4X_PIECE(pa_, pb_, wr_, 3);     \
        W4X_PIECE(pa_, pb_, wr_, 4); W4X_PIECE(pa_, pb_, wr_, 5); W4X_PIECE(pa_, pb_, wr_, 6); W4X_PIECE(pa_, pb_, wr_, 7);     \
        W4X_PIECE(pa_, pb_, wr_, 8); W4X_PIECE(pa_, pb_, wr_, 9); W4X_PIECE(pa_, pb_, wr_, 10); W4X_PIECE(pa_, pb_, wr_, 11);   \
        W4X_PIECE(pa_, pb_, wr_, 12); W4X_PIECE(pa_, pb_, wr_, 13); W4X_PIECE(pa_, pb_, wr_, 14); W4X_PIECE(pa_, pb_, wr_, 15); \
    } while (0)

    const int nst = K / XBK;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment BYTE offsets inside a stage for k-half 0; k-half 1 is +512
    uint32_t offA[NJ], offB[NI];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int r = wm * 128 + j * 16 + fr; offA[j] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2; }
#pragma unroll
    for (int i = 0; i < NI; ++i) { const int r = QBM + wn * 128 + i * 16 + fr; offB[i] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2; }
    const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);
#define W4X_RDB(fb_, rd_, kh_, i_) fb_[i_] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offB[i_])
#define W4X_RDA(fa_, rd_, kh_, j_) fa_[j_] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offA[j_])
    // slot k (0 .. 63) of a 32-k step: MFMA on accumulator tile (i = k % 8, j = k / 8) of the set (ca_, cb_); slots 0 - 15: one read of
    // the NEXT step's set (na_, nb_) from buffer nrd_, k-half nkh_; D_ != 0: DMA piece (k - 16) / 2 behind the even slots 16 .. 46
    // when dma_ (wave-uniform) is set.  Z_: first step of a tile (C = 0).
#define W4X_SLOT(k_, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                          \
        mfma16_asm<PREC, Z_>(acc[(k_) % 8][(k_) / 8], cb_[(k_) % 8], ca_[(k_) / 8]);                       \
        if constexpr ((k_) < 8) { W4X_RDB(nb_, nrd_, nkh_, ((k_) < 8 ? (k_) : 0)); }                       \
        else if constexpr ((k_) < 16) { W4X_RDA(na_, nrd_, nkh_, ((k_) >= 8 && (k_) < 16 ? (k_) - 8 : 0)); } \
        if constexpr ((D_) != 0 && (k_) >= 16 && (k_) < 48 && ((k_) & 1) == 0) {                           \
            if (dma_) W4X_PIECE(pa_, pb_, wr_, ((k_) >= 16 && (k_) < 48 ? ((k_) - 16) / 2 : 0));           \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);
#define W4X_SLOT8(k_, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                         \
        W4X_SLOT((k_) + 0, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 1, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 2, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 3, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 4, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 5, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 6, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                    \
        W4X_SLOT((k_) + 7, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)
#define W4X_STEP(Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                              \
        W4X_SLOT8(0, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                          \
        W4X_SLOT8(8, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                          \
        W4X_SLOT8(16, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                         \
        W4X_SLOT8(24, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                         \
        W4X_SLOT8(32, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                         \
        W4X_SLOT8(40, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                         \
        W4X_SLOT8(48, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)                         \
        W4X_SLOT8(56, Z_, ca_, cb_, na_, nb_, nrd_, nkh_, D_, dma_, pa_, pb_, wr_)
    // one 64-k stage held in buffer rd; (da_, db_) = A / B rows of the stage that goes into this buffer two stages on (dma_: it exists)
#define W4X_STAGE(Z_, dma_, da_, db_)                                                                      \
        {                                                                                                  \
            const uint32_t ot = XSB - rd;                                                                  \
            W4X_STEP(Z_, fa0, fb0, fa1, fb1, rd, 1, 0, false, da_, db_, rd)                                \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                    \
            __builtin_amdgcn_s_barrier();                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            W4X_STEP(false, fa1, fb1, fa0, fb0, ot, 0, 1, dma_, da_, db_, rd)                              \
            rd = ot;                                                                                       \
        }

    int L = blockIdx.x, m0, n0;
    W4X_TILE(L, m0, n0);
    const uint16_t* sA = A + (size_t)m0 * LD;
    const uint16_t* sB = B + (size_t)n0 * LD;
    W4X_ISSUE_ALL(sA, sB, 0u);                             // stage 0 -> buffer 0
    {
        const uint16_t* a1 = sA + XBK;
        const uint16_t* b1 = sB + XBK;
        W4X_ISSUE_ALL(a1, b1, XSB);                        // stage 1 -> buffer 1 (lands under stage 0)
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // in-order retirement: the 16 pieces of stage 0 have landed
    __builtin_amdgcn_s_barrier();
    f32x4_t acc[NI][NJ];
    uint4 fa0[NJ], fb0[NI], fa1[NJ], fb1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) W4X_RDB(fb0, 0u, 0, i);
#pragma unroll
    for (int j = 0; j < NJ; ++j) W4X_RDA(fa0, 0u, 0, j);
    __builtin_amdgcn_sched_barrier(0);
    uint32_t rd = 0;
    for (;;) {
        const int Ln = L + (int)gridDim.x;
        const bool more = Ln < ntiles;
        int m1 = m0, n1 = n0;
        if (more) W4X_TILE(Ln, m1, n1);
        const uint16_t* nA = A + (size_t)m1 * LD;
        const uint16_t* nB = B + (size_t)n1 * LD;
        // (rd lives across tiles: with an EVEN number of stages per tile it is back at 0 here -- the round-5 behaviour, bit for bit --, with
        // an odd number (K = 1344: the K = 1280 operands + the 64-column outlier extension, engine.hip) the next tile starts in buffer 1)
        // stage t feeds stage t + 2 into its own buffer; past the end of this tile that is stage t + 2 - nst of the NEXT tile
        const uint16_t* pa = sA + 2 * XBK;
        const uint16_t* pb = sB + 2 * XBK;
        W4X_STAGE(true, true, pa, pb)                      // t = 0 (nst >= 4: stage 2 exists)
        for (int t = 1; t + 2 < nst; ++t) {
            pa += XBK;
            pb += XBK;
            W4X_STAGE(false, true, pa, pb)
        }
        W4X_STAGE(false, more, nA, nB)                     // t = nst - 2 (buffer 0): the next tile's stage 0
        {
            const uint16_t* a1 = nA + XBK;
            const uint16_t* b1 = nB + XBK;
            W4X_STAGE(false, more, a1, b1)                 // t = nst - 1 (buffer 1): the next tile's stage 1; its step 1 reads that tile's stage 0
        }
        // The last MFMA results must not be read for 18 wait states, and hipcc does not know the asm statements above are MFMAs: the
        // padding is tied to every accumulator tile by data dependence (eight statements, one row of tiles each).
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j == 0)
                asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][j]), "+a"(acc[1][j]), "+a"(acc[2][j]), "+a"(acc[3][j]), "+a"(acc[4][j]),
                             "+a"(acc[5][j]), "+a"(acc[6][j]), "+a"(acc[7][j]));
            else
                asm volatile("" : "+a"(acc[0][j]), "+a"(acc[1][j]), "+a"(acc[2][j]), "+a"(acc[3][j]), "+a"(acc[4][j]), "+a"(acc[5][j]),
                             "+a"(acc[6][j]), "+a"(acc[7][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + 2 * XSB + wave * (W4X_SCR_BYTES / 4);
            epilogue_coalesced<PREC, OUT_F32, GELU, NJ, 1, NI, false, false, STREAM>(acc, scr, Cv, bias, nullptr, 1, N, m0 + wm * 128, n0 + wn * 128, accumulate, lane);
        }
        if (!more) break;
        L = Ln; m0 = m1; n0 = n1; sA = nA; sB = nB;
    }
#undef W4X_TILE
#undef W4X_PIECE
#undef W4X_ISSUE_ALL
#undef W4X_RDA
#undef W4X_RDB
#undef W4X_SLOT
#undef W4X_SLOT8
#undef W4X_STEP
#undef W4X_STAGE
}

static bool w4x_ok(int M, int N, int K, const float* add2d, bool out_f32) {
    return M % QBM == 0 && N % W4X_BN == 0 && K % XBK == 0 && K >= 4 * XBK && !add2d && !out_f32;
}

template <int PREC>
hipError_t launch_gemm_w4x(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, bool out_f32, bool gelu,
                           bool accumulate, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / W4X_BN);
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    dim3 grid(ntiles > n_cu ? n_cu : ntiles), block(W4THREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) return hipErrorInvalidValue;
    const bool so = gelu && stream_hidden(M, N);
    const int ld = tl_gemm_ld;
    if (gelu && tl_gelu_form == 2 && so) gemm_et_w4x_kernel<PREC, false, 2, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, ld);
    else if (gelu && tl_gelu_form == 2) gemm_et_w4x_kernel<PREC, false, 2><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, ld);
    else if (gelu && so) gemm_et_w4x_kernel<PREC, false, 1, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, ld);
    else if (gelu) gemm_et_w4x_kernel<PREC, false, 1><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, ld);
    else gemm_et_w4x_kernel<PREC, false, 0><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, ld);
    return hipGetLastError();
}

#ifdef SAMRS_EXPERIMENTS
// producer of the folded LayerNorm: C = A B^T + bias + C (fp32), Xh = ET(C), stats[m][N / 160] = (mean, M2) per 160 columns
template <int PREC>
hipError_t launch_gemm_m32_stats(const void* A, const void* B, float* C, const float* bias, void* Xh, float* stats,
                                 int M, int N, int K, int spread, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / WBN);
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    dim3 grid(ntiles > n_cu ? n_cu : ntiles), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    if (spread) gemm_et_m32_kernel<PREC, true, false, 1, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, 1, reinterpret_cast<uint16_t*>(Xh), reinterpret_cast<float2*>(stats));
    else gemm_et_m32_kernel<PREC, true, false, 0, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, 1, reinterpret_cast<uint16_t*>(Xh), reinterpret_cast<float2*>(stats));
    return hipGetLastError();
}
// consumer: C (ET) = [GELU](rstd (Xh Wf^T - mean cvec) + bias_f) on the persistent pair-stage kernel
template <int PREC>
hipError_t launch_gemm_x64p_fold(const void* A, const void* B, void* C, const float* bias, const float* cvec, const float* rowstat,
                                 int M, int N, int K, bool gelu, hipStream_t s) {
    const int ntiles = (M / QBM) * (N / WBN);
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    dim3 grid(ntiles < n_cu ? ntiles : n_cu), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const float2* rs = reinterpret_cast<const float2*>(rowstat);
    if (gelu) gemm_et_x64p_kernel<PREC, false, true, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, 0, rs, cvec);
    else gemm_et_x64p_kernel<PREC, false, false, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, 0, rs, cvec);
    return hipGetLastError();
}
static bool m32_ok(int M, int N, int K, const float* add2d) {
    return M % QBM == 0 && N % WBN == 0 && K % (2 * XBK) == 0 && K >= 2 * XBK && !add2d;
}

#endif  // SAMRS_EXPERIMENTS (fold launchers, m32_ok)
template <int PREC, int NI, int MODE>
hipError_t launch_gemm_x64(const void* A, const void* B, void* C, const float* bias, const float* add2d, int period,
                           int M, int N, int K, bool out_f32, bool gelu, bool accumulate, hipStream_t s) {
    dim3 grid((M / QBM) * (N / (64 * NI))), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_x64_kernel<PREC, true, true, NI, MODE><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc, g_x64_skew);
        else gemm_et_x64_kernel<PREC, true, false, NI, MODE><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc, g_x64_skew);
    } else {
        if (gelu) gemm_et_x64_kernel<PREC, false, true, NI, MODE><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc, g_x64_skew);
        else gemm_et_x64_kernel<PREC, false, false, NI, MODE><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc, g_x64_skew);
    }
    return hipGetLastError();
}

// proj / lin2 with the outlier-column stage (gemm_et_x64_kernel EXT): C (fp32) += A B^T + A_x B_x^T + bias
template <int PREC>
hipError_t launch_gemm_x64_ext(const void* A, const void* B, const void* Ax, const void* Bx, void* C, const float* bias,
                               int M, int N, int K, hipStream_t s) {
    dim3 grid((M / QBM) * (N / WBN)), block(QTHREADS);
    gemm_et_x64_kernel<PREC, true, false, 5, 3, 0, false, false, true><<<grid, block, 0, s>>>(
        reinterpret_cast<const uint16_t*>(A), reinterpret_cast<const uint16_t*>(B), C, bias, nullptr, 0, M, N, K, 1, g_x64_skew,
        reinterpret_cast<const uint16_t*>(Ax), reinterpret_cast<const uint16_t*>(Bx));
    return hipGetLastError();
}

template <int PREC, int NI = 4>
hipError_t launch_gemm_big(const void* A, const void* B, void* C, const float* bias, const float* add2d, int period,
                           int M, int N, int K, bool out_f32, bool gelu, bool accumulate, hipStream_t s) {
    dim3 grid((M / QBM) * (N / (64 * NI))), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_big_kernel<PREC, true, true, 0, NI><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else gemm_et_big_kernel<PREC, true, false, 0, NI><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    } else {
        if (gelu) gemm_et_big_kernel<PREC, false, true, 0, NI><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else gemm_et_big_kernel<PREC, false, false, 0, NI><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    }
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// gemm_et_dual_kernel: the staggered 256x128 kernel with a 32-wide K step: three 24 KiB ring stages
// = 72 KiB of LDS and <= 128 VGPRs, so TWO blocks (16 waves) are resident per CU and one block's
// epilogue (HBM-write bound: up to a third of the single-block kernel, ablation in DESIGN.md)
// overlaps the other block's main loop.  Per K step and wave: 3 DMA pieces, 8 ds_read_b128,
// 16 MFMAs, two raw barriers (L | C segments, wave groups staggered by one interval).
// ---------------------------------------------------------------------------------------------
constexpr int DBM = 256, DBN = 128, DBK = 32, DSTAGES = 3, DTHREADS = 512;
constexpr int DSTAGE_ELEMS = (DBM + DBN) * DBK;               // 12288 ET = 24 KiB
constexpr int D_DMA_PER_TILE = (DBM + DBN) * DBK * 2 / (DTHREADS * 16);   // 3 per thread

// SPLIT: both operands come as hi + lo (two-term split, common.h) and the k loop runs three times over K:
// A_lo B_hi, A_hi B_lo, A_hi B_hi (small terms first), all into the same fp32 accumulators -- the product of the
// un-rounded operands to ~2^-22 at three times the MFMA work.  Used for the decoder's first transposed conv, whose
// operand rounding cost 376 of the 899 class-map pixels of the round-2 engine at ViT-H (oracle/error_budget.py).
template <int PREC, bool OUT_F32, bool GELU, bool STAG = true, bool GLN = false, bool SPLIT = false>
__global__ __launch_bounds__(DTHREADS, 4) void gemm_et_dual_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period,
    int M, int N, int K, int accumulate, const uint16_t* __restrict__ A_lo = nullptr, const uint16_t* __restrict__ B_lo = nullptr) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[DSTAGES * DSTAGE_ELEMS];   // 72 KiB, ONE object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 1, wn = wave & 1;

    constexpr int GROUP = 8;
    const int tiles_n = N / DBN, tiles_m = M / DBM;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GROUP * tiles_n;
    const int group = bid / per_group, first_m = group * GROUP;
    const int gsz = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const int in_g = bid - group * per_group;
    const int tile_m = first_m + in_g % gsz, tile_n = in_g / gsz;
    const int m0 = tile_m * DBM, n0 = tile_n * DBN;

    // DMA: piece = 16 rows x 64 B; lane l -> row l>>2, physical chunk l&3 (source chunk swizzled).
    // wave w: A rows 16w.. and 128+16w.., B rows 16w..
    const int g_row = 16 * wave + (lane >> 2);
    const int g_chunk = qswz(g_row, lane & 3);
    const uint16_t* gAg = A + (size_t)(m0 + g_row) * K + g_chunk * 8;
    const uint16_t* gBg = B + (size_t)(n0 + g_row) * K + g_chunk * 8;
    const uint16_t* gAl = SPLIT ? A_lo + (size_t)(m0 + g_row) * K + g_chunk * 8 : gAg;
    const uint16_t* gBl = SPLIT ? B_lo + (size_t)(n0 + g_row) * K + g_chunk * 8 : gBg;
    const uint32_t wave_lds_base = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * (16 * DBK * 2));
    const size_t rs128 = (size_t)128 * K;
    const int nk1 = K / DBK;                   // k tiles of one pass over K
#define DUAL_ISSUE(kt_, stage_)                                                                 \
    do {                                                                                         \
        const int ph_ = SPLIT ? (kt_) / nk1 : 2;       /* 0: A_lo B_hi, 1: A_hi B_lo, 2: A_hi B_hi */ \
        const size_t koff_ = (size_t)((kt_) - (SPLIT ? ph_ * nk1 : 0)) * DBK;                    \
        const uint16_t* pa_ = ph_ == 0 ? gAl : gAg;                                              \
        const uint16_t* pb_ = ph_ == 1 ? gBl : gBg;                                              \
        constexpr int SB_ = (stage_) * DSTAGE_ELEMS * 2;                                         \
        glds16_asm<SB_ + 0>(pa_ + koff_, wave_lds_base);                                         \
        glds16_asm<SB_ + 128 * DBK * 2>(pa_ + rs128 + koff_, wave_lds_base);                     \
        glds16_asm<SB_ + DBM * DBK * 2>(pb_ + koff_, wave_lds_base);                             \
    } while (0)

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = nk1 * (SPLIT ? 3 : 1);
    DUAL_ISSUE(0, 0);
    if (nk > 1) {
        DUAL_ISSUE(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D_DMA_PER_TILE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (STAG && grp == 1) __builtin_amdgcn_s_barrier();          // stagger

    const int fr = lane & 15, fq = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = wm * 64 + j * 16 + fr; offA[j] = r * DBK + qswz(r, fq) * 8; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wn * 64 + i * 16 + fr; offB[i] = DBM * DBK + r * DBK + qswz(r, fq) * 8; }

#define DUAL_WAIT(kt_)                                                                           \
    if ((kt_) + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D_DMA_PER_TILE) : "memory");     \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define DUAL_STEP(kt_, S_)                                                                       \
    if ((kt_) < nk) {                                                                            \
        uint4 fa[4], fb[4];                                                                      \
        if ((kt_) + 2 < nk) DUAL_ISSUE((kt_) + 2, ((S_) + 2) % DSTAGES);                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            fb[i] = *reinterpret_cast<const uint4*>(lds + (S_) * DSTAGE_ELEMS + offB[i]);        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                            \
            fa[j] = *reinterpret_cast<const uint4*>(lds + (S_) * DSTAGE_ELEMS + offA[j]);        \
        if (STAG) {                                                                              \
        DUAL_WAIT(kt_)                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_barrier();                                                            \
        }                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                        \
                acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]);                           \
        __builtin_amdgcn_s_setprio(0);                                                           \
        DUAL_WAIT(kt_)                                                                           \
        if (!STAG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_barrier();                                                            \
    }
    for (int kt = 0; kt < nk; kt += DSTAGES) {
        DUAL_STEP(kt, 0)
        DUAL_STEP(kt + 1, 1)
        DUAL_STEP(kt + 2, 2)
    }
    if (STAG && grp == 0) __builtin_amdgcn_s_barrier();          // both groups: 2 + 2*nk barriers
    {
        unsigned char* scr = reinterpret_cast<unsigned char*>(lds) + wave * (DSTAGES * DSTAGE_ELEMS * 2 / 8);   // 9 KiB
        if (!OUT_F32 && !GLN && add2d) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * 64 + j * 16 + fr, n = n0 + wn * 64 + i * 16 + 4 * fq;
                    const float4 e = *reinterpret_cast<const float4*>(add2d + (size_t)(m % add2d_period) * N + n);
                    acc[i][j][0] += e.x; acc[i][j][1] += e.y; acc[i][j][2] += e.z; acc[i][j][3] += e.w;
                }
        }
        epilogue_coalesced<PREC, OUT_F32, GELU, 4, OUT_F32 ? 2 : 4, 4, GLN>(acc, scr, Cv, bias, (OUT_F32 || GLN) ? add2d : nullptr,
                                                                             add2d_period, N, m0 + wm * 64, n0 + wn * 64, accumulate, lane);
    }
}

// C (ET) = GELU(LayerNorm2d_64(A B^T + bias)): the 64-column groups of N are normalised independently.
// A_lo / B_lo given: split-precision product (see the kernel), C is then FP32 (its consumer splits it again).
template <int PREC>
hipError_t launch_gemm_dual_gln(const void* A, const void* B, void* C, const float* bias, const float* gamma_beta,
                                int M, int N, int K, hipStream_t s, const void* A_lo = nullptr, const void* B_lo = nullptr) {
    dim3 grid((M / DBM) * (N / DBN)), block(DTHREADS);
    if (A_lo && B_lo)
        gemm_et_dual_kernel<PREC, true, true, true, true, true><<<grid, block, 0, s>>>(
            reinterpret_cast<const uint16_t*>(A), reinterpret_cast<const uint16_t*>(B), C, bias, gamma_beta, 1, M, N, K, 0,
            reinterpret_cast<const uint16_t*>(A_lo), reinterpret_cast<const uint16_t*>(B_lo));
    else
        gemm_et_dual_kernel<PREC, false, true, true, true><<<grid, block, 0, s>>>(
            reinterpret_cast<const uint16_t*>(A), reinterpret_cast<const uint16_t*>(B), C, bias, gamma_beta, 1, M, N, K, 0);
    return hipGetLastError();
}

template <int PREC, bool STAG = true>
hipError_t launch_gemm_dual(const void* A, const void* B, void* C, const float* bias, const float* add2d, int period,
                            int M, int N, int K, bool out_f32, bool gelu, bool accumulate, hipStream_t s) {
    dim3 grid((M / DBM) * (N / DBN)), block(DTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu) gemm_et_dual_kernel<PREC, true, true, STAG><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else gemm_et_dual_kernel<PREC, true, false, STAG><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    } else {
        if (gelu) gemm_et_dual_kernel<PREC, false, true, STAG><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else gemm_et_dual_kernel<PREC, false, false, STAG><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K = 256 streaming GEMM (the decoder's image-side projections: rows = prompts x 4096 keys, N = 256 / 384).
// With K this short a tiled GEMM is all prologue and epilogue (82 / 52 us for 167 / 134 MB, hipBLASLt 66 / 32): the
// operand that matters is A, read once, and C, written once.  So: a PERSISTENT block keeps its whole weight slice in
// registers (wave w: columns 64 w .. +63 x 256 k = 32 uint4 fragments), walks 128-row tiles of A that arrive by LDS-DMA
// into a double buffer (tile t+1 is in flight while tile t is multiplied and stored), and stores straight from the
// accumulators (a wave's 4 column blocks complete whole 128-byte lines of a row).  No B traffic after the first tile, one
// barrier per tile.  Same k order as every other kernel (ascending 32-wide steps from a zero accumulator), same
// epilogue arithmetic as the tiled path ((acc + add2d) + bias) -> bit-identical results.
// LDS image of a tile: row r = 32 16-byte chunks, chunk c of the row at position c ^ (r & 15): the fragment reads of a
// 16-row group (lane (fr, fq) reads chunk 4 ks + fq of row fr) are conflict-free, and a DMA piece is two whole rows.
// ---------------------------------------------------------------------------------------------
constexpr int K2_ROWS = 128, K2_K = 256, K2_TILE_BYTES = K2_ROWS * K2_K * 2;

// RS = row split: RS waves share a column slice (each keeps its own copy of the weight fragments) and take every RS-th
// 16-row group -- N = 256: 8 waves instead of 4, twice the loads in flight per CU for the L2-latency-bound 2-D addend.
template <int PREC, int NW, int RS>
__global__ __launch_bounds__(64 * NW * RS) void gemm_et_k256_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint16_t* __restrict__ C,
    const float* __restrict__ bias, const float* __restrict__ add2d, int add2d_period, int M) {
    constexpr int N = 64 * NW, K = K2_K;
    extern __shared__ __attribute__((aligned(16))) unsigned char k2_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0 .. NW RS - 1: DMA piece owner
    const int wave = wid % NW, part = wid / NW;                    // column slice, row-group phase
    const int fr = lane & 15, fq = lane >> 4;
    const int ntiles = M / K2_ROWS;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)k2_lds);
    constexpr int NWT = NW * RS;

    // DMA piece p (two rows 2p, 2p + 1): lane l fetches row 2p + (l >> 5), source chunk (l & 31) ^ (row & 15)
#define K2_ISSUE(tile_, buf_)                                                                                      \
    do {                                                                                                           \
        const uint16_t* sb_ = A + (size_t)(tile_) * K2_ROWS * K;                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < (64 + NWT - 1) / NWT; ++i_) {                                      \
            const int p_ = wid + NWT * i_;                                                                         \
            if (p_ < 64) {                                                                                         \
                const int row_ = 2 * p_ + (lane >> 5);                                                             \
                const uint32_t voff_ = (uint32_t)(row_ * K + (((lane & 31) ^ (row_ & 15)) << 3)) * 2u;             \
                glds16_s(voff_, sb_, lds0 + (uint32_t)(buf_) * K2_TILE_BYTES + (uint32_t)p_ * 1024u);              \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)

    int t = blockIdx.x, buf = 0;
    if (t < ntiles) K2_ISSUE(t, 0);
    // weight slice and bias of this wave: resident for the whole launch
    uint4 wf[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wf[i][ks] = *reinterpret_cast<const uint4*>(B + (size_t)(wave * 64 + i * 16 + fr) * K + ks * 32 + fq * 8);
    // the bias lives in LDS behind the two tile buffers (16 registers the pipelined loop needs elsewhere)
    constexpr int K2_PS = 144;                 // patch row stride (bytes): 128 + 16
    unsigned char* patch = k2_lds + 2 * K2_TILE_BYTES + 2048 + wid * (16 * K2_PS);
    float* bias_s = reinterpret_cast<float*>(k2_lds + 2 * K2_TILE_BYTES);
    for (int i = tid; i < N; i += 64 * NWT) bias_s[i] = bias ? bias[i] : 0.f;
#ifdef K2_TIMING
    unsigned long long kph[4] = {0, 0, 0, 0}, kprev = __builtin_amdgcn_s_memtime();
    const unsigned long long kstart = kprev;
    int kn = 0;
#define K2_STAMP(i_) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); kph[i_] += tn_ - kprev; kprev = tn_; }
#else
#define K2_STAMP(i_)
#endif
    constexpr int NG = (K2_ROWS / 16) / RS;                   // row groups per wave and tile
#define K2_AF(tile_, g_, ks_) (*reinterpret_cast<const uint4*>((tile_) + ((((g_) * 16 + fr) * 32 + (((ks_) * 4 + fq) ^ fr)) << 4)))
#define K2_ADDEND(dst_, row_)                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                               \
        dst_[i_] = add2d ? *reinterpret_cast<const float4*>(add2d + (size_t)((row_) % add2d_period) * N + wave * 64 + i_ * 16 + 4 * fq) \
                         : make_float4(0.f, 0.f, 0.f, 0.f)
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed
        K2_STAMP(0)
        __syncthreads();                                        // everybody's; and nobody reads the other buffer any more
        K2_STAMP(1)
        if (t + (int)gridDim.x < ntiles) K2_ISSUE(t + (int)gridDim.x, buf ^ 1);
        K2_STAMP(2)
        const unsigned char* tile = k2_lds + buf * K2_TILE_BYTES;
        // Software pipeline over this wave's NG row groups: a group's addend is loaded before its MFMAs, the next group's A
        // fragments replace the current ones k-step by k-step as the MFMAs consume them; the four column blocks advance
        // together (4 independent accumulator chains).
        uint4 af[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[ks] = K2_AF(tile, part, ks);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = part + gi * RS;
            const bool more = gi + 1 < NG;
            float4 e[4];                          // in flight during this group's MFMAs (no room for a second set)
            K2_ADDEND(e, t * K2_ROWS + g * 16 + fr);
            f32x4_t acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = ET<PREC>::mfma16(wf[i][ks], af[ks], acc[i]);
                if (more) af[ks] = K2_AF(tile, g + RS, ks);
            }
            // Bounce through a wave-private LDS patch: straight from the accumulators a store instruction would touch 16
            // cache lines with 32 bytes each (measured: ~320 cycles of issue time per store, 5 k of 9 k cycles per tile);
            // after the bounce a lane stores 16 bytes and an instruction covers 8 whole 128-byte lines.
            if (gi) wave_lds_sync_g();            // the previous group's reads of the patch are done
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 bb = *reinterpret_cast<const float4*>(bias_s + wave * 64 + i * 16 + 4 * fq);
                uint2 o;
                o.x = pack2<PREC>((acc[i][0] + e[i].x) + bb.x, (acc[i][1] + e[i].y) + bb.y);
                o.y = pack2<PREC>((acc[i][2] + e[i].z) + bb.z, (acc[i][3] + e[i].w) + bb.w);
                *reinterpret_cast<uint2*>(patch + fr * K2_PS + (i * 16 + 4 * fq) * 2) = o;
            }
            wave_lds_sync_g();
            uint16_t* cbase = C + (size_t)(t * K2_ROWS + g * 16) * N + wave * 64;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = h * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(patch + r * K2_PS + ch * 16);
                *reinterpret_cast<uint4*>(cbase + (size_t)r * N + ch * 8) = v;
            }
        }
        K2_STAMP(3)
#ifdef K2_TIMING
        ++kn;
#endif
    }
#ifdef K2_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wid == 0) {   // timing build: the first rows of C hold the stamps instead of results
        unsigned long long* tp = reinterpret_cast<unsigned long long*>(C) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 4; ++i) tp[i] = kph[i];
        tp[4] = __builtin_amdgcn_s_memtime() - kstart;
        tp[5] = kn;
    }
#endif
#undef K2_ISSUE
#undef K2_STAMP
#undef K2_AF
#undef K2_ADDEND
}

template <int PREC>
hipError_t launch_gemm_k256(const void* A, const void* B, void* C, const float* bias, const float* add2d, int period,
                            int M, int N, hipStream_t s) {
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int ntiles = M / K2_ROWS;
    dim3 grid(ntiles < n_cu ? ntiles : n_cu);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    constexpr int LDS = 2 * K2_TILE_BYTES + 2048 + 8 * 16 * 144;   // two tile buffers + the bias + 8 store patches
#define K2_LAUNCH(NW_, RS_)                                                                                          \
    do {                                                                                                             \
        auto k = gemm_et_k256_kernel<PREC, NW_, RS_>;                                                                \
        HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
        k<<<grid, 64 * NW_ * RS_, LDS, s>>>(a, b, (uint16_t*)C, bias, add2d, period, M);                             \
    } while (0)
    if (N == 256) K2_LAUNCH(4, 2);
    else if (N == 384) K2_LAUNCH(6, 1);
    else return hipErrorInvalidValue;
#undef K2_LAUNCH
    return hipGetLastError();
}
// shapes the streaming kernel takes: ET output without GELU / accumulate, K = 256, N = 256 or 384, whole 128-row tiles
static bool k256_ok(int M, int N, int K, bool out_f32, bool gelu, bool accumulate) {
    return !out_f32 && !gelu && !accumulate && K == K2_K && (N == 256 || N == 384) && M % K2_ROWS == 0;
}

thread_local int tl_gemm_variant = -1;   // per-engine override, set around an engine's launches (engine.hip GemmVariantScope)
int g_gemm_variant = 8;   // 0 reg-staged 128^2, 1 +LDS-DMA, 2 +grouped order, 3 reg+grouped, 4 256x128 3-stage pipe, 5 +staggered groups, 6 256x256, 7 2 blocks/CU, 8 auto(5|6|7)

template <int PREC, bool GLDS, int GROUP_M>
hipError_t launch_gemm_prec(const void* A, const void* B, void* C, const float* bias,
                            const float* add2d, int period, int M, int N, int K, bool out_f32,
                            bool gelu, bool accumulate, hipStream_t s) {
    dim3 grid((M / BM) * (N / BN)), block(GEMM_THREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    if (out_f32) {
        if (gelu)
            gemm_et_kernel<PREC, true, true, GLDS, GROUP_M><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else
            gemm_et_kernel<PREC, true, false, GLDS, GROUP_M><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    } else {
        if (gelu)
            gemm_et_kernel<PREC, false, true, GLDS, GROUP_M><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
        else
            gemm_et_kernel<PREC, false, false, GLDS, GROUP_M><<<grid, block, 0, s>>>(a, b, C, bias, add2d, period, M, N, K, acc);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// exact fp32 GEMM: C[M,N] = (A [+ A2])[M,K] * W[N,K]^T + bias, optional ReLU / accumulate, for up
// to F32_BATCH_MAX independent problems of one shape per launch (blockIdx.z picks the pointer set).
// Token side of the decoder: M is a few hundred rows, so the launch is latency- and parallelism-
// bound, not FLOP-bound.  A block of S waves owns one 32x32 output tile and splits K S ways;
// partial tiles are combined through LDS in a FIXED order, so the result is bit-reproducible
// (unlike a blockIdx-level split-K with fp32 atomics).
//
// No LDS staging of the operands: v_mfma_f32_16x16x4_f32 takes, per lane (i = lane&15, q = lane>>4),
// ONE element P[i][k_q] of the 16x4 slab, and a sum over k does not care which physical k sits in
// slot q as long as both operands agree.  So each lane loads a float4 = row i, columns 4q..4q+3 of
// a 16-wide K slab straight from global memory (16 rows x 64 contiguous bytes per instruction),
// and MFMA t of the slab consumes component t of both operands: slot q of MFMA t holds k = 4q + t.
// Up to F32_PF slabs are kept in flight per wave (a register ring).
// M, N arbitrary (bounds-checked), K % 16 == 0.  lda / ldc in elements; W is dense [N][K].
// ---------------------------------------------------------------------------------------------
constexpr int FM = 32, FN = 32, FKS = 16, F32_PF = 4;

template <int S>
__global__ __launch_bounds__(64 * S) void gemm_f32_kernel(F32Batch bt, int lda, int ldc, int M, int N, int K,
                                                          int relu, int accumulate) {
    __shared__ float red[S][FM][FN + 1];
    const int z = blockIdx.z;
    const float* __restrict__ A = bt.A[z];
    const float* __restrict__ A2 = bt.A2[z];
    const float* __restrict__ W = bt.W[z];
    const float* __restrict__ bias = bt.bias[z];
    float* __restrict__ C = bt.C[z];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = blockIdx.y * FM, n0 = blockIdx.x * FN;
    const int KW = K / S, NS = KW / FKS;
    const int r = lane & 15, q = lane >> 4;

    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    bool va[2], vw[2];
    const float *pa[2], *pa2[2], *pw[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        va[h] = m0 + 16 * h + r < M;
        vw[h] = n0 + 16 * h + r < N;
        const size_t ao = (size_t)(va[h] ? m0 + 16 * h + r : 0) * lda + wave * KW + 4 * q;
        pa[h] = A + ao;
        pa2[h] = A2 ? A2 + ao : nullptr;
        pw[h] = W + (size_t)(vw[h] ? n0 + 16 * h + r : 0) * K + wave * KW + 4 * q;
    }
    float4 xa[F32_PF][2], xw[F32_PF][2];
#define F32_LOAD(slot_, s_)                                                                        \
    if ((s_) < NS) {                                                                               \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                         \
            float4 a_ = make_float4(0.f, 0.f, 0.f, 0.f), w_ = a_;                                  \
            if (va[h_]) {                                                                          \
                a_ = *reinterpret_cast<const float4*>(pa[h_] + (s_) * FKS);                        \
                if (pa2[h_]) {                                                                     \
                    const float4 b_ = *reinterpret_cast<const float4*>(pa2[h_] + (s_) * FKS);      \
                    a_.x += b_.x; a_.y += b_.y; a_.z += b_.z; a_.w += b_.w;                        \
                }                                                                                  \
            }                                                                                      \
            if (vw[h_]) w_ = *reinterpret_cast<const float4*>(pw[h_] + (s_) * FKS);                \
            xa[slot_][h_] = a_; xw[slot_][h_] = w_;                                                \
        }                                                                                          \
    }
#pragma unroll
    for (int p = 0; p < F32_PF; ++p) F32_LOAD(p, p)
    for (int g = 0; g < NS; g += F32_PF) {
#pragma unroll
        for (int p = 0; p < F32_PF; ++p) {
            if (g + p < NS) {
                const float fa[2][4] = {{xa[p][0].x, xa[p][0].y, xa[p][0].z, xa[p][0].w},
                                        {xa[p][1].x, xa[p][1].y, xa[p][1].z, xa[p][1].w}};
                const float fw[2][4] = {{xw[p][0].x, xw[p][0].y, xw[p][0].z, xw[p][0].w},
                                        {xw[p][1].x, xw[p][1].y, xw[p][1].z, xw[p][1].w}};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[i][t], fa[j][t], acc[i][j], 0, 0, 0);
                F32_LOAD(p, g + p + F32_PF)
            }
        }
    }
#undef F32_LOAD
    // D[i_local = n][j_local = m]: lane holds n = base + 4*(lane>>4) + rr, m = base + (lane&15)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[wave][j * 16 + r][i * 16 + 4 * q + rr] = acc[i][j][rr];
    __syncthreads();
    for (int o = threadIdx.x; o < FM * FN; o += 64 * S) {
        const int ml = o >> 5, nl = o & 31;
        float v = red[0][ml][nl];
#pragma unroll
        for (int w = 1; w < S; ++w) v += red[w][ml][nl];     // fixed order: reproducible
        const int m = m0 + ml, n = n0 + nl;
        if (m < M && n < N) {
            float* c = C + (size_t)m * ldc + n;
            v += bias ? bias[n] : 0.f;
            if (relu) v = fmaxf(v, 0.f);
            if (accumulate) v += *c;
            *c = v;
        }
    }
}


// =============================================================================================================================
// MXFP4 lo terms (round 4).  The operand split  A B^T ~= A_hi B_hi^T + A_lo B_hi^T + A_hi B_lo^T  pays two extra f16 passes for
// corrections that are 2^-11 of the product.  Their operands tolerate very few mantissa bits (oracle/error_budget.py plans8 - 10:
// e2m1 lo terms leave 1.8e-1 of a plain f16 GEMM's error, and the C4 floor of the multimask mode holds with them), so here the
// two correction terms run on gfx950's block-scaled  v_mfma_scale_f32_16x16x128_f8f6f4  with FP4 (e2m1) operands and one E8M0
// scale per 32 k (OCP MX): 4x the f16 MFMA rate -- and, the reason for fp4 rather than fp8 / fp6, the SAME LDS geometry as the
// f16 stage: a 128-byte stage row holds 256 k of fp4 (two 64-byte k-halves of 128 k each) where it holds 64 k of f16, and a
// lane's fragment of a k-half is the same 16 bytes (32 fp4 values = its MX block) it is for f16.  So the pair-stage image, its
// XOR swizzle, the DMA map, the 13 fragment reads and the 40 MFMAs per k-half are those of gemm_et_x64_kernel; an MX stage
// differs in the MFMA opcode, in two 8-byte scale reads per k-half, and in six 1-KiB scale pieces per stage.  A correction
// segment over K is K / 256 stages against K / 64 for the f16 segment: the split product costs 1.5x a plain one (f16 lo: 3x).
//
// Operand tensors (written by mx4_pack_kernel, or by the producers themselves):
//   A4 / B4  [rows][Kp / 2] bytes: element k' of a row in nibble k' (low nibble first); Kp = padded K (a multiple of 256)
//   scales   E8M0 bytes, one per (row, 32-k' block), stored in the order the kernel's lanes read them:
//              A: [tile_m = r / 256][stage = b / 8][k-half = (b / 4) % 2][wm = (r / 128) % 2][fq = b % 4][fr = r % 16][j = (r / 16) % 8]
//              B: [tile_n = n / 320][stage][k-half][wn = (n / 80) % 4][fq][fr = n % 16][i = (n % 80) / 16, 8 slots]
//            i.e. 2 KiB / 4 KiB per (tile, stage), DMA'd as they are; lane (fr, fq) of wave (wm, wn) reads 8 bytes per operand and
//            k-half, byte j / i = the scale of its row in m-tile j / n-tile i (op_sel picks the byte).
// K' may be K with each group of G elements padded to GP (heads of 80 padded to 96 for the proj GEMM: no MX block straddles two
// heads, so the attention kernels can emit the blocks of their own head); the f16 segment keeps the plain K.
// =============================================================================================================================
constexpr int MX_S_BYTES = MX_SA_BYTES + MX_SB_BYTES;          // MXK, MX_SA_BYTES, MX_SB_BYTES and the quantisers: common.h

struct MxOperands {
    const unsigned char *a4_lo, *a4_hi, *b4_hi, *b4_lo;      // fp4 data
    const unsigned char *sa_lo, *sa_hi, *sb_hi, *sb_lo;      // scale tiles
    int Kp;                                                  // padded K of the MX segments
    int split_from_n;                                        // output columns below this take no lo terms (ET outputs; 0 = all do)
};

typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef int mx_v4i __attribute__((ext_vector_type(4)));

template <int OPA, int OPB>
__device__ __forceinline__ f32x4_t mfma_mx4(const uint4& a, const uint4& b, f32x4_t c, int sa, int sb) {
    const mx_v4i xa = {(int)a.x, (int)a.y, (int)a.z, (int)a.w}, xb = {(int)b.x, (int)b.y, (int)b.z, (int)b.w};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(xa, xa, 0, 1, 2, 3, -1, -1, -1, -1),
                                                            __builtin_shufflevector(xb, xb, 0, 1, 2, 3, -1, -1, -1, -1), c,
                                                            4 /* fp4 e2m1 */, 4, OPA, sa, OPB, sb);
}

template <int PREC, bool OUT_F32, bool GELU = false, bool MXO = false>
__global__ __launch_bounds__(QTHREADS) void gemm_et_mx_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv, const float* __restrict__ bias,
    int M, int N, int K, int accumulate, MxOperands mx, MxOut mxo = MxOut(),
    int ld = 0 /* row stride of A and B in elements, 0 = K */,
    int ext = 0 /* round 6: the rows carry the 64-column outlier extension behind their K live columns (engine.hip EncBlock::oc_*): the tiles
                   that take NO lo terms (n0 < split_from_n: q and k of the v-third split) read it as one more f16 stage; the tiles that do
                   take them are covered by their fp4 correction segments, which span every column */) {
    constexpr int NI = 5;
    constexpr int XBN = 64 * NI;
    constexpr int XROWS = QBM + XBN;
    constexpr int XSTAGE_ELEMS = XROWS * XBK;
    constexpr uint32_t XSB = XSTAGE_ELEMS * 2;             // stage bytes (f16 and MX alike: rows x 128 B)
    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * XSTAGE_ELEMS + MX_S_BYTES];   // ring (144 KiB) + 2 x 6 KiB of scales
    const int LD = ld > 0 ? ld : K;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 2, wn = wave & 3;

    // PERSISTENT as gemm_et_x64p_kernel: a block walks tiles L, L + gridDim.x, ...; the next tile's stage 0 (and its scale tiles) goes
    // out before the epilogue, whose bounce scratch is confined to ring buffer 1.  (A grid of one block per tile degenerates to
    // the one-tile kernel: SAMRS_MX_PERSIST=0, kept for A/B runs.)
    constexpr int GROUP = 8;
    const int tiles_n = N / XBN, tiles_m = M / QBM, ntiles = tiles_n * tiles_m;
    const int per_group = GROUP * tiles_n;
#define MX_TILE(L_, tm_, tn_)                                                                    \
    do {                                                                                         \
        const int bid_ = xcd_remap((L_), ntiles);                                                \
        const int group_ = bid_ / per_group, first_m_ = group_ * GROUP;                           \
        const int gsz_ = (tiles_m - first_m_) < GROUP ? (tiles_m - first_m_) : GROUP;            \
        const int in_g_ = bid_ - group_ * per_group;                                             \
        (tm_) = first_m_ + in_g_ % gsz_;                                                         \
        (tn_) = in_g_ / gsz_;                                                                    \
    } while (0)
    int L = blockIdx.x, tile_m, tile_n;                    // the tile being FED (the DMA macros below read these)
    MX_TILE(L, tile_m, tile_n);
    int m0 = tile_m * QBM, n0 = tile_n * XBN;              // the tile being COMPUTED (epilogue)

    const int nst1 = K / XBK;                              // f16 stages
    const int nst4 = mx.Kp / MXK;                          // MX stages per correction segment
    int nmx = n0 >= mx.split_from_n ? 2 * nst4 : 0;        // MX stages of the tile being fed (wave-uniform): A_lo B_hi, then A_hi B_lo
    int nf16 = nst1 + ((ext && n0 < mx.split_from_n) ? 1 : 0);   // ... and its f16 stages (+ the outlier extension where no lo terms run)

    // DMA map as in gemm_et_x64_kernel; the per-lane byte offset depends on the row stride of the source (2 K vs Kp / 2)
    const int prow = 8 * wave + ((lane >> 2) & 7);
    const uint32_t lane_off = (uint32_t)(lane >> 5) * 64u + (uint32_t)qswz(prow, lane & 3) * 16u;
    const uint32_t voff16 = (uint32_t)prow * (uint32_t)LD * 2u + lane_off;
    const uint32_t voff4 = (uint32_t)prow * (uint32_t)(mx.Kp >> 1) + lane_off;
    const unsigned char* A16 = reinterpret_cast<const unsigned char*>(A) + (size_t)m0 * LD * 2;     // of the tile being fed
    const unsigned char* B16 = reinterpret_cast<const unsigned char*>(B) + (size_t)n0 * LD * 2;
    int fm0 = m0, fn0 = n0;                                // rows of the tile being fed (the fp4 sources)
    const size_t row4 = (size_t)(mx.Kp >> 1);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + (uint32_t)wave * 1024u);
    const uint32_t lds_sc = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 2u * XSB);
    // piece q_ (literal) of stage st_ into the ring buffer at byte offset wr_ (scales: buffer (wr_ != 0))
#define MX_PIECE(st_, wr_, q_)                                                                                       \
    do {                                                                                                             \
        const int st__ = (st_);                                                                                      \
        /* readfirstlane: inside the persistent loop LLVM moves some wave-uniform address chains to VGPRs (SGPR pressure) and */ \
        /* then hands the VGPR to the "s" operand of the inline asm ("s_mov_b32 m0, v0": does not assemble)                   */ \
        const uint32_t dst__ = __builtin_amdgcn_readfirstlane(                                                       \
            lds0 + (wr_) + ((q_) < 4 ? (q_) * 8192u : (uint32_t)(QBM * XBK * 2) + ((q_) - 4) * 8192u));              \
        if (st__ < nmx) {                                                                                            \
            const int seg__ = st__ >= nst4, s__ = st__ - seg__ * nst4;                                               \
            const unsigned char* src__ = (q_) < 4 ? (seg__ ? mx.a4_hi : mx.a4_lo) + ((size_t)fm0 + (q_) * 64) * row4         \
                                                   : (seg__ ? mx.b4_lo : mx.b4_hi) + ((size_t)fn0 + ((q_) - 4) * 64) * row4; \
            glds16_s(voff4, src__ + (size_t)s__ * 128, dst__);                                                       \
        } else {                                                                                                     \
            const unsigned char* src__ = (q_) < 4 ? A16 + (size_t)(q_) * 64 * LD * 2 : B16 + (size_t)((q_) - 4) * 64 * LD * 2; \
            glds16_s(voff16, src__ + (size_t)(st__ - nmx) * 128, dst__);                                             \
        }                                                                                                            \
    } while (0)
    // the stage's scale tiles: 2 KiB (A) + 4 KiB (B) as six 1-KiB pieces, one per wave 0 .. 5
#define MX_SCALES(st_, wr_)                                                                                          \
    do {                                                                                                             \
        const int st__ = (st_);                                                                                      \
        if (st__ < nmx && wave < 6) {                                                                                \
            const int seg__ = st__ >= nst4, s__ = st__ - seg__ * nst4;                                               \
            const unsigned char* src__ = wave < 2                                                                    \
                ? (seg__ ? mx.sa_hi : mx.sa_lo) + ((size_t)tile_m * nst4 + s__) * MX_SA_BYTES + wave * 1024          \
                : (seg__ ? mx.sb_lo : mx.sb_hi) + ((size_t)tile_n * nst4 + s__) * MX_SB_BYTES + (wave - 2) * 1024;   \
            glds16_s((uint32_t)lane * 16u, src__,                                                                    \
                     __builtin_amdgcn_readfirstlane(lds_sc + ((wr_) ? (uint32_t)MX_S_BYTES : 0u) + (uint32_t)wave * 1024u)); \
        }                                                                                                            \
    } while (0)
#define MX_ISSUE(st_, wr_)                                                                                           \
    do {                                                                                                             \
        MX_PIECE(st_, wr_, 0); MX_PIECE(st_, wr_, 1); MX_PIECE(st_, wr_, 2); MX_PIECE(st_, wr_, 3); MX_PIECE(st_, wr_, 4); \
        MX_PIECE(st_, wr_, 5); MX_PIECE(st_, wr_, 6); MX_PIECE(st_, wr_, 7); MX_PIECE(st_, wr_, 8); MX_SCALES(st_, wr_); \
    } while (0)

    f32x4_t acc[NI][8];

    const int fr = lane & 15, fq = lane >> 4;
    uint32_t offA[8], offB[NI];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = wm * 128 + j * 16 + fr; offA[j] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = QBM + wn * (16 * NI) + i * 16 + fr;
        offB[i] = ((r >> 3) * 512 + (r & 7) * 32 + qswz(r, fq) * 8) * 2;
    }
    const unsigned char* ldsb = reinterpret_cast<const unsigned char*>(lds);
    const uint32_t scA = 2u * XSB + (uint32_t)((wm * 64 + fq * 16 + fr) * 8);                    // + kh * 1024 + buffer * MX_S_BYTES
    const uint32_t scB = 2u * XSB + (uint32_t)MX_SA_BYTES + (uint32_t)((wn * 64 + fq * 16 + fr) * 8);   // + kh * 2048 + ...

    MX_ISSUE(0, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#define MX_READ(rd_, kh_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                         \
        fb[i] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offB[i]);                     \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                          \
        fa[j] = *reinterpret_cast<const uint4*>(ldsb + (rd_) + (kh_) * 512 + offA[j]);
#define MX_READ_SC(rd_, kh_)                                                                               \
    sa = *reinterpret_cast<const uint2*>(ldsb + scA + ((rd_) ? (uint32_t)MX_S_BYTES : 0u) + (kh_) * 1024); \
    sb = *reinterpret_cast<const uint2*>(ldsb + scB + ((rd_) ? (uint32_t)MX_S_BYTES : 0u) + (kh_) * 2048);
    // one row of MFMAs (m-tile j_) + the DMA piece that rides behind it
#define MX_DMA_AFTER(j_, dma_, st_, wr_)                                                                   \
    if (dma_) {                                                                                            \
        if ((j_) == 0) MX_PIECE(st_, wr_, 0); if ((j_) == 1) MX_PIECE(st_, wr_, 1);                        \
        if ((j_) == 2) MX_PIECE(st_, wr_, 2); if ((j_) == 3) MX_PIECE(st_, wr_, 3);                        \
        if ((j_) == 4) MX_PIECE(st_, wr_, 4); if ((j_) == 5) MX_PIECE(st_, wr_, 5);                        \
        if ((j_) == 6) { MX_PIECE(st_, wr_, 6); MX_SCALES(st_, wr_); }                                     \
        if ((j_) == 7) { MX_PIECE(st_, wr_, 7); MX_PIECE(st_, wr_, 8); }                                   \
    }
#define MX_MFMA_F16(dma_, st_, wr_)                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                        \
        _Pragma("unroll") for (int i = 0; i < NI; ++i) acc[i][j] = ET<PREC>::mfma16(fb[i], fa[j], acc[i][j]); \
        MX_DMA_AFTER(j, dma_, st_, wr_)                                                                    \
    }
    // first MFMA operand = the weight fragment (n rows): its scale is the B scale, byte i; second = activation fragment, byte j
#define MX_ROW(j_, OPB_, sav_)                                                                             \
    acc[0][j_] = mfma_mx4<0, OPB_>(fb[0], fa[j_], acc[0][j_], (int)sb.x, (int)(sav_));                     \
    acc[1][j_] = mfma_mx4<1, OPB_>(fb[1], fa[j_], acc[1][j_], (int)sb.x, (int)(sav_));                     \
    acc[2][j_] = mfma_mx4<2, OPB_>(fb[2], fa[j_], acc[2][j_], (int)sb.x, (int)(sav_));                     \
    acc[3][j_] = mfma_mx4<3, OPB_>(fb[3], fa[j_], acc[3][j_], (int)sb.x, (int)(sav_));                     \
    acc[4][j_] = mfma_mx4<0, OPB_>(fb[4], fa[j_], acc[4][j_], (int)sb.y, (int)(sav_));                     \
    /* pin the row here: unlike the plain MFMA intrinsics the scaled one is not convergent, and LLVM sinks all 80 of a stage */ \
    /* below the (branchy) DMA pieces that are meant to ride between the rows -- two k-halves of fragments live, 670 spills */  \
    asm volatile("" : "+v"(acc[0][j_]), "+v"(acc[1][j_]), "+v"(acc[2][j_]), "+v"(acc[3][j_]), "+v"(acc[4][j_]));
#define MX_MFMA_FP4(dma_, st_, wr_)                                                                        \
    MX_ROW(0, 0, sa.x) MX_DMA_AFTER(0, dma_, st_, wr_) MX_ROW(1, 1, sa.x) MX_DMA_AFTER(1, dma_, st_, wr_)  \
    MX_ROW(2, 2, sa.x) MX_DMA_AFTER(2, dma_, st_, wr_) MX_ROW(3, 3, sa.x) MX_DMA_AFTER(3, dma_, st_, wr_)  \
    MX_ROW(4, 0, sa.y) MX_DMA_AFTER(4, dma_, st_, wr_) MX_ROW(5, 1, sa.y) MX_DMA_AFTER(5, dma_, st_, wr_)  \
    MX_ROW(6, 2, sa.y) MX_DMA_AFTER(6, dma_, st_, wr_) MX_ROW(7, 3, sa.y) MX_DMA_AFTER(7, dma_, st_, wr_)

    // One stage (the schedule of gemm_et_x64_kernel<MODE = 3>: one block barrier per stage, DMA pieces spread between the MFMA
    // rows, the two wave groups half a stage apart).  The MX stages and the f16 stages run in two consecutive loops -- one MFMA
    // flavour per loop body; a single loop with a per-stage branch made the register allocator spill 670 registers -- and the
    // hand-over between them needs nothing: MX_PIECE decides per TARGET stage what it fetches.
#define MX_STAGE(READ_SC0_, READ_SC1_, MFMA_)                                                              \
    {                                                                                                      \
        uint4 fa[8], fb[NI];                                                                               \
        const uint32_t wr = XSB - rd;                                                                      \
        const bool dma0 = (grp == 0) && (t + 1 < nst);                                                     \
        const bool dma1 = (grp == 1) && (t + 2 < nst);                                                     \
        if (grp == 0) __builtin_amdgcn_s_barrier();                           /* R_t */                    \
        MX_READ(rd, 0)                                                                                     \
        READ_SC0_                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                     \
        MFMA_(dma0, t + 1, wr)                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        MX_READ(rd, 1)                                                                                     \
        READ_SC1_                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (grp == 1) __builtin_amdgcn_s_barrier();       /* R_{t+1} of group 1: buffer rd is free from here on */ \
        __builtin_amdgcn_s_setprio(1);                                                                     \
        MFMA_(dma1, t + 2, rd)                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                     \
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        rd = wr;                                                                                           \
    }
    for (;;) {
        const int nst = nmx + nf16;                        // stages of THIS tile (the feed variables still are its own here)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (grp == 1) {                                    // rendezvous R_0 of group 1; its share of stage 1 goes out first
            if (nst > 1) MX_ISSUE(1, XSB);
            __builtin_amdgcn_s_barrier();
        }
        uint32_t rd = 0;
        int t = 0;
        for (; t < nmx; ++t) {
            uint2 sa, sb;
            MX_STAGE(MX_READ_SC(rd, 0), MX_READ_SC(rd, 1), MX_MFMA_FP4)
        }
        for (; t < nst; ++t) MX_STAGE(, , MX_MFMA_F16)
        if (grp == 0) __builtin_amdgcn_s_barrier();        // both groups: same barrier count; every ring read of this tile is done

        // next tile: its stage 0 (fp4 or f16 pieces + the scale tiles) goes into ring buffer 0 / scale buffer 0 now and lands
        // under the epilogue
        const int Ln = L + (int)gridDim.x;
        const bool more = Ln < ntiles;
        const int em0 = m0, en0 = n0;
        if (more) {
            MX_TILE(Ln, tile_m, tile_n);
            fm0 = tile_m * QBM; fn0 = tile_n * XBN;
            A16 = reinterpret_cast<const unsigned char*>(A) + (size_t)fm0 * LD * 2;
            B16 = reinterpret_cast<const unsigned char*>(B) + (size_t)fn0 * LD * 2;
            nmx = fn0 >= mx.split_from_n ? 2 * nst4 : 0;
            nf16 = nst1 + ((ext && fn0 < mx.split_from_n) ? 1 : 0);
            MX_ISSUE(0, 0u);
        }
        {   // epilogue of tile (em0, en0); bounce scratch = ring buffer 1 only (72 KiB: 9 KiB per wave / 18 KiB per pair)
            unsigned char* upper = reinterpret_cast<unsigned char*>(lds) + XSB;
            if constexpr (!OUT_F32) {
                epilogue_pair_et<PREC, GELU, 2, true, false, MXO>(acc, upper, Cv, bias, nullptr, 1, N, em0 + wm * 128, en0 + (wn >> 1) * 160,
                                                                  wm, wn, lane, nullptr, nullptr, mxo);
            } else {
                epilogue_coalesced<PREC, true, false, 8, 1, NI, false, true>(acc, upper + wave * (XSB / 8), Cv, bias, nullptr, 1, N,
                                                                            em0 + wm * 128, en0 + wn * (16 * NI), accumulate, lane);
            }
        }
        if (!more) break;
        __builtin_amdgcn_s_barrier();          // scratch free again; stage 0 visible (each wave drained its pieces before its first store)
        L = Ln; m0 = fm0; n0 = fn0;
    }
#undef MX_STAGE
#undef MX_TILE
#undef MX_PIECE
#undef MX_SCALES
#undef MX_ISSUE
#undef MX_READ
#undef MX_READ_SC
#undef MX_DMA_AFTER
#undef MX_MFMA_F16
#undef MX_ROW
#undef MX_MFMA_FP4
}

// One (row, 32-k' block) per 8 lanes, 4 elements per lane.  Source: fp32 x (SRC_F32: hi = ET(x), lo = x - hi; also writes the ET
// copy when out_hi is given) or the ET pair (hi, lo) an existing producer wrote.  k' = padded index: group g = k' / GP holds the
// G source elements g G .. g G + G - 1 followed by GP - G zeros.
template <int PREC, bool SRC_F32>
__global__ __launch_bounds__(256) void mx4_pack_kernel(const float* __restrict__ x, const uint16_t* __restrict__ hi_in,
                                                       const uint16_t* __restrict__ lo_in, uint16_t* __restrict__ out_hi,
                                                       unsigned char* __restrict__ q_hi, unsigned char* __restrict__ q_lo,
                                                       unsigned char* __restrict__ s_hi, unsigned char* __restrict__ s_lo,
                                                       int rows, int K, int G, int GP, int is_b, int perm) {
    const int Kp = (K / G) * GP, nblk = Kp / 32, nst4 = Kp / MXK;
    const long item = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;           // (row, block)
    const int sub = threadIdx.x & 7;
    if (item >= (long)rows * nblk) return;
    const int r = (int)(item / nblk), b = (int)(item % nblk);
    // position 4 sub .. 4 sub + 3 of the block holds source elements at block offset 4 sub (natural order) or 8 (sub & 3) + 4 (sub >> 2)
    // (perm: the order store_attention_row_mx writes its rows in)
    // perm 2: the order of epilogue_pair_et<MXO> (position 8 fq + 4 ib + e holds column 16 ib + 4 fq + e; sub = 2 fq + ib)
    const int in_blk = perm == 1 ? 8 * (sub & 3) + 4 * (sub >> 2) : perm == 2 ? 16 * (sub & 1) + 4 * (sub >> 1) : sub * 4;
    const int kp = b * 32 + in_blk, g = kp / GP, off = kp % GP;             // GP % 4 == 0 and G % 4 == 0: a lane's 4 elements share a fate
    float h[4] = {0.f, 0.f, 0.f, 0.f}, l[4] = {0.f, 0.f, 0.f, 0.f};
    if (off < G) {
        const size_t src = (size_t)r * K + (size_t)g * G + off;
        if constexpr (SRC_F32) {
            const float4 v = *reinterpret_cast<const float4*>(x + src);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            uint2 hb;
            hb.x = pack2<PREC>(vv[0], vv[1]); hb.y = pack2<PREC>(vv[2], vv[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = ET<PREC>::to_float((uint16_t)(((e & 2) ? hb.y : hb.x) >> (16 * (e & 1))));
                l[e] = vv[e] - h[e];
            }
            if (out_hi) *reinterpret_cast<uint2*>(out_hi + src) = hb;
        } else {
            const uint2 hb = *reinterpret_cast<const uint2*>(hi_in + src), lb = *reinterpret_cast<const uint2*>(lo_in + src);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = ET<PREC>::to_float((uint16_t)(((e & 2) ? hb.y : hb.x) >> (16 * (e & 1))));
                l[e] = ET<PREC>::to_float((uint16_t)(((e & 2) ? lb.y : lb.x) >> (16 * (e & 1))));
            }
        }
    }
    float ah = fmaxf(fmaxf(fabsf(h[0]), fabsf(h[1])), fmaxf(fabsf(h[2]), fabsf(h[3])));
    float al = fmaxf(fmaxf(fabsf(l[0]), fabsf(l[1])), fmaxf(fabsf(l[2]), fabsf(l[3])));
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { ah = fmaxf(ah, __shfl_xor(ah, o, 64)); al = fmaxf(al, __shfl_xor(al, o, 64)); }
    const int bh = mx_scale_byte(ah), bl = mx_scale_byte(al);
    const uint32_t ch = fp4_pack4(h[0], h[1], h[2], h[3], bh), cl = fp4_pack4(l[0], l[1], l[2], l[3], bl);
    const size_t dst = (size_t)r * (Kp / 2) + (size_t)b * 16 + sub * 2;
    *reinterpret_cast<uint16_t*>(q_hi + dst) = (uint16_t)ch;
    *reinterpret_cast<uint16_t*>(q_lo + dst) = (uint16_t)cl;
    if (sub == 0) {
        const size_t si = mx_scale_index(is_b != 0, r, b, nst4);
        s_hi[si] = (unsigned char)bh;
        s_lo[si] = (unsigned char)bl;
    }
}

}  // namespace

// true when a plain ET launch of this shape (no 2-D addend, automatic variant) runs on gemm_et_x64p_kernel / gemm_et_w4x_kernel, the two
// kernels that take an operand row stride (tl_gemm_ld): whole rounds of 256 x 320 tiles, K in 64-k pair stages
static void gemm_env_once() {          // tuning knob for A/B runs: SAMRS_GEMM_VARIANT=<n>
    static const bool once = [] {
        if (const char* v = getenv("SAMRS_GEMM_VARIANT")) g_gemm_variant = atoi(v);
        return true;
    }();
    (void)once;
}
bool gemm_ld_ok(int M, int N, int K, bool gelu) {
    (void)gelu;
    gemm_env_once();
    const int gv = tl_gemm_variant >= 0 ? tl_gemm_variant : g_gemm_variant;
    if (gv != 8 || M % QBM || N % WBN || K % XBK || K % QBK) return false;
    const long t320 = (long)(M / QBM) * (N / WBN);
    return t320 >= 256 && t320 % 256 == 0;
}
int swap_gemm_ld(int ld) { const int old = tl_gemm_ld; tl_gemm_ld = ld; return old; }

// fp32 residual outputs with the outlier-column stage: the shapes the 256 x 320 pair-stage kernel takes by the automatic rule
bool gemm_ext_ok(int M, int N, int K) {
    gemm_env_once();
    const int gv = tl_gemm_variant >= 0 ? tl_gemm_variant : g_gemm_variant;
    return gv == 8 && M > 0 && M % QBM == 0 && N % WBN == 0 && K % XBK == 0 && K >= 2 * XBK && (long)(M / QBM) * (N / WBN) >= 256;
}
hipError_t launch_gemm_et_ext(int prec, const void* A, const void* B, const void* Ax, const void* Bx, float* C, const float* bias,
                              int M, int N, int K, hipStream_t s) {
    if (!A || !B || !Ax || !Bx || !C || !gemm_ext_ok(M, N, K)) return hipErrorInvalidValue;
    if (prec == PREC_BF16) return launch_gemm_x64_ext<PREC_BF16>(A, B, Ax, Bx, C, bias, M, N, K, s);
    if (prec == PREC_F16) return launch_gemm_x64_ext<PREC_F16>(A, B, Ax, Bx, C, bias, M, N, K, s);
    return hipErrorInvalidValue;
}

hipError_t launch_gemm_et(int prec, const void* A, const void* B, void* C, const float* bias,
                          const float* add2d, int add2d_period, int M, int N, int K, bool out_f32,
                          bool gelu, bool accumulate, hipStream_t s) {
    if (M % BM || N % BN || K % BK || M <= 0 || N <= 0 || K <= 0) return hipErrorInvalidValue;
    if (add2d && add2d_period <= 0) return hipErrorInvalidValue;
    if (add2d && gelu) return hipErrorInvalidValue;   // not needed by the path; the coalesced epilogue orders them differently
    gemm_env_once();
    // an engine handle's own choice (samrs_set_option "gemm_variant") overrides the process-wide test hook for its launches
    const int gv = tl_gemm_variant >= 0 ? tl_gemm_variant : g_gemm_variant;
    // a padded operand stride is understood by the persistent ET kernels only: refuse anything else instead of reading garbage
    if (tl_gemm_ld != 0 && (tl_gemm_ld < K || !gemm_ld_ok(M, N, K, gelu) || out_f32 || add2d)) return hipErrorInvalidValue;
#define GEMM_DISPATCH(P)                                                                                         \
    switch (gv) {                                                                                    \
        case 0: return launch_gemm_prec<P, false, 1>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s); \
        case 1: return launch_gemm_prec<P, true, 1>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);  \
        case 3: return launch_gemm_prec<P, false, 8>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s); \
        default: return launch_gemm_prec<P, true, 8>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s); \
    }
#ifdef SAMRS_EXPERIMENTS
    if (gv >= 60 && gv < 92 && prec == PREC_F16 && !out_f32 && M % QBM == 0 && N % QBN == 0) {
        dim3 grid((M / QBM) * (N / QBN)), block(QTHREADS);
        const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
        const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
#define ABL_CASE(x) case 60 + x: gemm_et_big_kernel<PREC_F16, false, false, x><<<grid, block, 0, s>>>(a, b, C, bias, add2d, add2d_period, M, N, K, 0); break;
        switch (gv) {
            ABL_CASE(0) ABL_CASE(1) ABL_CASE(4) ABL_CASE(6) ABL_CASE(8) ABL_CASE(7) ABL_CASE(16) ABL_CASE(22) ABL_CASE(23) ABL_CASE(24)
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
#endif
    // 40: K = 256 streaming kernel (decoder image side); automatic for its shapes once there are >= 2 tiles per CU
    if ((gv == 40 || (gv == 8 && M / K2_ROWS >= 512)) && k256_ok(M, N, K, out_f32, gelu, accumulate)) {
        if (prec == PREC_BF16) return launch_gemm_k256<PREC_BF16>(A, B, C, bias, add2d, add2d_period, M, N, s);
        if (prec == PREC_F16) return launch_gemm_k256<PREC_F16>(A, B, C, bias, add2d, add2d_period, M, N, s);
        return hipErrorInvalidValue;
    }
    // variant 8 ("auto", default): per-shape pick measured on MI355X (tools/gemm_bench.py) -- the
    // 2-blocks-per-CU kernel wins where the epilogue dominates (GELU output, or short K with a
    // narrow N), the 64-wide-K single-block kernel wins on long K / wide N.
    int variant = gv;
    // 20 / 21: pair-stage (64-deep, whole-cache-line DMA) 256x256 / 256x320 kernel; 22 / 23: the same with the DMA pieces
    // spread between the MFMAs.  Shapes they do not cover fall through to the automatic choice.
    if (variant == 28 && !(M % QBM == 0 && N % WBN == 0 && K % XBK == 0 && !add2d)) variant = 8;
    if (variant >= 20 && variant <= 27 || variant == 35) {     // 20 + mode * 2 + (NI == 5): mode bit 0 = spread DMA, bit 1 = one barrier per stage
        const int ni = (variant & 1) ? 5 : 4, mode = (variant - 20) >> 1;
        if (M % QBM == 0 && N % (64 * ni) == 0 && K % XBK == 0) {
#define X64_CASE(P, NI_, MD_) return launch_gemm_x64<P, NI_, MD_>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s)
#define X64_MODES(P)                                                             \
            switch (variant) {                                                   \
                case 20: X64_CASE(P, 4, 0); case 21: X64_CASE(P, 5, 0);          \
                case 22: X64_CASE(P, 4, 1); case 23: X64_CASE(P, 5, 1);          \
                case 24: X64_CASE(P, 4, 2); case 25: X64_CASE(P, 5, 2);          \
                case 26: X64_CASE(P, 4, 3); case 35: X64_CASE(P, 5, 7); default: X64_CASE(P, 5, 3); \
            }
            (void)mode;
            if (prec == PREC_F16) { X64_MODES(PREC_F16) }
            if (prec == PREC_BF16) { X64_MODES(PREC_BF16) }
#undef X64_MODES
#undef X64_CASE
            return hipErrorInvalidValue;
        }
        variant = 8;
    }
#ifdef SAMRS_EXPERIMENTS
    // 100 + abl: ablations of the barrier-light 256x320 kernel (f16, ET output, no GELU): timing experiments only
    if (variant >= 100 && variant < 196 && prec == PREC_F16 && !out_f32 && M % QBM == 0 && N % WBN == 0 && K % XBK == 0) {
        dim3 grid((M / QBM) * (N / WBN)), block(QTHREADS);
        const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
        const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
#define XABL_CASE(x) case 100 + x: gemm_et_x64_kernel<PREC_F16, false, false, 5, 3, x><<<grid, block, 0, s>>>(a, b, C, bias, add2d, add2d_period, M, N, K, 0, g_x64_skew); break;
        switch (variant) {
            XABL_CASE(0) XABL_CASE(1) XABL_CASE(2) XABL_CASE(4) XABL_CASE(16) XABL_CASE(6) XABL_CASE(7) XABL_CASE(22) XABL_CASE(23) XABL_CASE(17)
            XABL_CASE(32) XABL_CASE(33) XABL_CASE(64)
            default: return hipErrorInvalidValue;
        }
#undef XABL_CASE
        return hipGetLastError();
    }
#endif
    if (variant == 8) {
        const bool big_ok = M % QBM == 0 && K % QBK == 0;
        const long t256 = big_ok && N % QBN == 0 ? (long)(M / QBM) * (N / QBN) : 0;
        const long t320 = big_ok && N % WBN == 0 ? (long)(M / QBM) * (N / WBN) : 0;
        // the 256x320 tile: pair-stage kernel (whole-line DMA, one barrier per 64 k, DMA pieces spread between the MFMAs;
        // measured +1.5 ... +4 % over the 32-deep ring on all four encoder shapes, bit-identical) when K allows, else
        // the 32-deep staggered kernel
        const int wide = (K % XBK == 0) ? 27 : 10;
        // fp32 residual outputs (proj, lin2: N = 1280): 256x320 tiles -> an exact number of rounds over the 256 CUs
        if (out_f32 && t320 >= 256) variant = wide;
        // f16 outputs (qkv N = 3840, lin1+GELU N = 5120): the wide tile whenever it fills whole rounds (6 / 8 rounds at
        // batch 8, exactly one round for lin1 of a single image), else 256x256 as long as there are >= 4 rounds of tiles
        // (ET outputs without a 2-D addend: the persistent flavour, +3 % on qkv / lin1; its one-m-tile fp32 epilogue loses on proj)
        else if (!out_f32 && t320 >= 256 && t320 % 256 == 0) variant = (wide == 27 && !add2d) ? 28 : wide;
        else if (!out_f32 && N >= 2048 && t256 >= 1024) variant = 6;
        else variant = (gelu || (K <= 1536 && N <= 1536)) ? 7 : 5;
    }
    // ET + GELU outputs whose 256 x 256 tiles fill whole rounds (lin1 of ViT-H: 2560 tiles = 10 rounds at batch 8; 5 at batch 4) go to
    // the four-wave 128 x 128-wave-tile kernel (variant 38; bit-identical with the kernel it replaces, so no mode's arithmetic moves).
    // Measured on MI355X: as a 2.5 s loop of its own 412 us against 429 and 1.03 against 1.08 pJ / FLOP (profiles/r05_gemm_energy.txt);
    // INSIDE the tile loop, beside the decoder's kernels, the launch itself is no faster (0.4342 against 0.4330 ms) and the step gains
    // 0.2 % (142.68 / 142.68 / 142.88 against 142.21 / 142.57 / 142.52 images/s, alternated on one box: profiles/r05_ab_loop.txt).
    // Plain ET outputs (qkv: 7.5 rounds) and the fp32 outputs (2.5 rounds) are slower on it in both settings and stay where they were.
    // SAMRS_GEMM_W4X=0 switches the rule off (A/B runs).
    static const bool w4x_auto = [] { const char* v = getenv("SAMRS_GEMM_W4X"); return v ? atoi(v) != 0 : true; }();
    if (variant == 28 && w4x_auto && gelu && w4x_ok(M, N, K, add2d, out_f32)) {
        const long t256 = (long)(M / QBM) * (N / W4X_BN);
        if (t256 % 256 == 0 && t256 >= 1024) variant = 38;
    }
    // 30 / 31: 32x32x16 symmetric-schedule kernel, persistent / one tile per block.  SAMRS_GEMM_M32=<mask> lets the automatic
    // rule pick it for A/B runs of the whole loop: bit 0 = ET outputs (qkv, lin1), bit 1 = fp32 outputs (proj, lin2),
    // bit 2 = one tile per block instead of persistent, bit 3 = spread pieces
#ifdef SAMRS_EXPERIMENTS
    static const int m32_mask = [] { const char* v = getenv("SAMRS_GEMM_M32"); return v ? atoi(v) : 0; }();
    if ((variant == 27 || variant == 28) && tl_gemm_ld == 0 /* the m32 / w4 kernels take no operand stride */ && m32_ok(M, N, K, add2d) &&
        (m32_mask & (out_f32 ? 2 : 1)))
        variant = (m32_mask & 16) ? 34 : 30 + ((m32_mask & 4) ? 1 : 0) + ((m32_mask & 8) ? 2 : 0);
    if (variant >= 30 && variant <= 33 && !m32_ok(M, N, K, add2d)) variant = add2d ? 27 : 28;
    if (variant >= 30 && variant <= 33) {     // 30 / 31: all pieces in step 3; 32 / 33: pieces spread over two steps
        const bool pers = !(variant & 1);
        if (variant < 32) {
            if (prec == PREC_BF16) return launch_gemm_m32<PREC_BF16, 0>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, pers, s);
            if (prec == PREC_F16) return launch_gemm_m32<PREC_F16, 0>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, pers, s);
        } else {
            if (prec == PREC_BF16) return launch_gemm_m32<PREC_BF16, 1>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, pers, s);
            if (prec == PREC_F16) return launch_gemm_m32<PREC_F16, 1>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, pers, s);
        }
        return hipErrorInvalidValue;
    }
    // 34 / 36: the four-wave, 512-register flavour of the symmetric schedule (persistent / one tile per block)
    if ((variant == 34 || variant == 36) && !m32_ok(M, N, K, add2d)) variant = add2d ? 27 : 28;
    if (variant == 34 || variant == 36) {
        if (prec == PREC_BF16) return launch_gemm_w4<PREC_BF16>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, variant == 34, s);
        if (prec == PREC_F16) return launch_gemm_w4<PREC_F16>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, variant == 34, s);
        return hipErrorInvalidValue;
    }
#else
    if (variant >= 30 && variant <= 36) variant = add2d ? 27 : 28;     // the 32x32x16 kernels are not in this build (make EXPERIMENTS=1)
#endif
    // 38: the four-wave 256 x 256 kernel on 16x16x32 MFMAs with 128 x 128 wave tiles (round 5); shapes it does not cover fall back
    if (variant == 38 && !w4x_ok(M, N, K, add2d, out_f32)) variant = add2d ? 27 : 28;
    if (variant == 38) {
        if (prec == PREC_BF16) return launch_gemm_w4x<PREC_BF16>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_w4x<PREC_F16>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (variant == 28 && M % QBM == 0 && N % WBN == 0 && K % XBK == 0 && !add2d) {   // persistent pair-stage kernel
        if (prec == PREC_BF16) return launch_gemm_x64p<PREC_BF16>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_x64p<PREC_F16>(A, B, C, bias, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (variant == 28) variant = 27;
    if (variant == 27 && M % QBM == 0 && N % WBN == 0 && K % XBK == 0) {   // chosen by the automatic rule above
        if (prec == PREC_BF16) return launch_gemm_x64<PREC_BF16, 5, 3>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_x64<PREC_F16, 5, 3>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (variant == 9 && M % DBM == 0 && K % DBK == 0) {   // 2 blocks / CU, lock-step (one barrier per K step)
        if (prec == PREC_BF16) return launch_gemm_dual<PREC_BF16, false>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_dual<PREC_F16, false>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (variant == 7 && M % DBM == 0 && K % DBK == 0) {   // 2 blocks / CU
        if (prec == PREC_BF16) return launch_gemm_dual<PREC_BF16>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_dual<PREC_F16>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (variant == 11) variant = 10;      // (the round-1 persistent 32-deep kernel was removed: 28 is its successor)
    if (variant == 10 && M % QBM == 0 && N % WBN == 0 && K % QBK == 0) {   // 256x320 staggered kernel
        if (prec == PREC_BF16) return launch_gemm_big<PREC_BF16, 5>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_big<PREC_F16, 5>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (variant == 10) variant = 5;
    if (variant == 6 && M % QBM == 0 && N % QBN == 0 && K % QBK == 0) {   // 256x256 staggered kernel
        if (prec == PREC_BF16) return launch_gemm_big<PREC_BF16>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_big<PREC_F16>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if ((variant == 5 || variant == 6 || variant == 7 || variant == 9) && M % PBM == 0) {   // staggered-group pipelined kernel
        if (prec == PREC_BF16) return launch_gemm_stag<PREC_BF16>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        if (prec == PREC_F16) return launch_gemm_stag<PREC_F16>(A, B, C, bias, add2d, add2d_period, M, N, K, out_f32, gelu, accumulate, s);
        return hipErrorInvalidValue;
    }
    if (prec == PREC_BF16) { GEMM_DISPATCH(PREC_BF16) }
    if (prec == PREC_F16) { GEMM_DISPATCH(PREC_F16) }
    return hipErrorInvalidValue;
}

// LayerNorm folded into the neighbouring GEMMs (ViT-H: the residual stream has N = 1280 = LN_NS x 160 columns).
bool gemm_has_experiments() {
#ifdef SAMRS_EXPERIMENTS
    return true;
#else
    return false;
#endif
}
#ifndef SAMRS_EXPERIMENTS
hipError_t launch_gemm_et_stats(int, const void*, const void*, float*, const float*, void*, float*, int, int, int, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_gemm_et_fold(int, const void*, const void*, void*, const float*, const float*, const float*, int, int, int, bool, hipStream_t) { return hipErrorInvalidValue; }
#else
hipError_t launch_gemm_et_stats(int prec, const void* A, const void* B, float* C, const float* bias, void* Xh, float* stats,
                                int M, int N, int K, hipStream_t s) {
    if (!m32_ok(M, N, K, nullptr) || N != 160 * LN_NS || !Xh || !stats) return hipErrorInvalidValue;
    static const int spread = [] { const char* v = getenv("SAMRS_GEMM_M32"); return v ? ((atoi(v) & 8) ? 1 : 0) : 1; }();
    if (prec == PREC_BF16) return launch_gemm_m32_stats<PREC_BF16>(A, B, C, bias, Xh, stats, M, N, K, spread, s);
    if (prec == PREC_F16) return launch_gemm_m32_stats<PREC_F16>(A, B, C, bias, Xh, stats, M, N, K, spread, s);
    return hipErrorInvalidValue;
}
hipError_t launch_gemm_et_fold(int prec, const void* Xh, const void* Wf, void* C, const float* bias_f, const float* cvec,
                               const float* rowstat, int M, int N, int K, bool gelu, hipStream_t s) {
    if (M % QBM || N % WBN || K != 160 * LN_NS || !bias_f || !cvec || !rowstat) return hipErrorInvalidValue;
    if (prec == PREC_BF16) return launch_gemm_x64p_fold<PREC_BF16>(Xh, Wf, C, bias_f, cvec, rowstat, M, N, K, gelu, s);
    if (prec == PREC_F16) return launch_gemm_x64p_fold<PREC_F16>(Xh, Wf, C, bias_f, cvec, rowstat, M, N, K, gelu, s);
    return hipErrorInvalidValue;
}
#endif  // SAMRS_EXPERIMENTS

// One-launch split product (see seg_src_a): C = (A + A_lo)(B + B_lo)^T minus the lo x lo term, + bias; ET output (rounded once
// from the fp32 accumulators) or fp32 output, optionally accumulated into C.  Shapes of the pair-stage 256x320 tile only; the
// caller falls back to three accumulating launch_gemm_et passes when this returns hipErrorInvalidValue.
template <int PREC>
static hipError_t launch_gemm_split3_prec(const uint16_t* a, const uint16_t* al, const uint16_t* b, const uint16_t* bl, void* C,
                                          const float* bias, int M, int N, int K, bool out_f32, bool accumulate, int split_from_n,
                                          const float* add2d, int add2d_period, hipStream_t s) {
    if (out_f32) {
        // fp32 outputs: one tile per block; 256 x 320 tiles where N allows, else 256 x 256 (the neck: N = 256)
        if (N % WBN == 0)
            gemm_et_x64_kernel<PREC, true, false, 5, 3, 0, true><<<dim3((M / QBM) * (N / WBN)), dim3(QTHREADS), 0, s>>>(
                a, b, C, bias, add2d, add2d_period, M, N, K, accumulate ? 1 : 0, 0, al, bl);
        else
            gemm_et_x64_kernel<PREC, true, false, 4, 3, 0, true><<<dim3((M / QBM) * (N / QBN)), dim3(QTHREADS), 0, s>>>(
                a, b, C, bias, add2d, add2d_period, M, N, K, accumulate ? 1 : 0, 0, al, bl);
    } else {
        const int ntiles = (M / QBM) * (N / WBN);
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (n_cu <= 0) n_cu = 256;
        gemm_et_x64p_kernel<PREC, false, false, false, true><<<dim3(ntiles < n_cu ? ntiles : n_cu), dim3(QTHREADS), 0, s>>>(
            a, b, C, bias, M, N, K, 0, nullptr, nullptr, al, bl, split_from_n);
    }
    return hipGetLastError();
}

bool gemm_split3_ok(int M, int N, int K, bool out_f32) {
    return M > 0 && N > 0 && K > 0 && M % QBM == 0 && K % XBK == 0 && (N % WBN == 0 || (out_f32 && N % QBN == 0));
}

hipError_t launch_gemm_et_split3(int prec, const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, const float* bias,
                                 int M, int N, int K, bool out_f32, bool accumulate, hipStream_t s, int split_from_n,
                                 const float* add2d, int add2d_period) {
    if (!gemm_split3_ok(M, N, K, out_f32) || !A || !A_lo || !B || !B_lo || (accumulate && !out_f32)) return hipErrorInvalidValue;
    if (split_from_n < 0 || split_from_n % WBN || (split_from_n && out_f32)) return hipErrorInvalidValue;   // whole tiles; ET outputs only
    if (add2d && (!out_f32 || add2d_period <= 0)) return hipErrorInvalidValue;                              // the fp32 epilogue adds it
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* al = reinterpret_cast<const uint16_t*>(A_lo);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const uint16_t* bl = reinterpret_cast<const uint16_t*>(B_lo);
    if (prec == PREC_BF16) return launch_gemm_split3_prec<PREC_BF16>(a, al, b, bl, C, bias, M, N, K, out_f32, accumulate, split_from_n, add2d, add2d_period, s);
    if (prec == PREC_F16) return launch_gemm_split3_prec<PREC_F16>(a, al, b, bl, C, bias, M, N, K, out_f32, accumulate, split_from_n, add2d, add2d_period, s);
    return hipErrorInvalidValue;
}

// C (fp32) += A B^T + bias, then out_et = LayerNorm(C rows) * gamma + beta in the operand type, by the LayerNorm tail of the 256 x 320
// pair-stage kernel (gemm_et_x64_kernel<LNT>).  Only where that kernel is what launch_gemm_et would pick for the shape anyway (fp32
// output with at least one full round of 256 x 320 tiles) and N is one LayerNorm row; otherwise hipErrorInvalidValue and the caller
// launches GEMM and LayerNorm separately.  counters: M / 256 zeroed uint32 (left zero by every launch).
bool gemm_lntail_ok(int M, int N, int K) {
    return M > 0 && M % QBM == 0 && N % WBN == 0 && N <= 1280 && K % XBK == 0 && K >= 2 * XBK && (long)(M / QBM) * (N / WBN) >= 256;
}
hipError_t launch_gemm_et_lntail(int prec, const void* A, const void* B, float* C, const float* bias, int M, int N, int K,
                                 const float* gamma, const float* beta, float eps, void* out_et, unsigned int* counters, hipStream_t s) {
    if (!gemm_lntail_ok(M, N, K) || !A || !B || !C || !gamma || !beta || !out_et || !counters) return hipErrorInvalidValue;
    LnTail ln;
    ln.gamma = gamma; ln.beta = beta; ln.out = reinterpret_cast<uint16_t*>(out_et); ln.counters = counters; ln.eps = eps;
    dim3 grid((M / QBM) * (N / WBN)), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    if (prec == PREC_F16)
        gemm_et_x64_kernel<PREC_F16, true, false, 5, 3, 0, false, true><<<grid, block, 0, s>>>(a, b, C, bias, nullptr, 1, M, N, K, 1, g_x64_skew, nullptr, nullptr, ln);
    else if (prec == PREC_BF16)
        gemm_et_x64_kernel<PREC_BF16, true, false, 5, 3, 0, false, true><<<grid, block, 0, s>>>(a, b, C, bias, nullptr, 1, M, N, K, 1, g_x64_skew, nullptr, nullptr, ln);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

int swap_gelu_form(int v) { const int old = tl_gelu_form; tl_gelu_form = v; return old; }
void set_gemm_variant(int v) { g_gemm_variant = v; }
int swap_gemm_variant_override(int v) { const int old = tl_gemm_variant; tl_gemm_variant = v; return old; }
void set_gemm_skew(int xcd_units, int cu_units) { g_x64_skew = (xcd_units & 0xffff) | (cu_units << 16); }

hipError_t launch_gemm_f32_batch(const F32Batch& bt, int count, int lda, int ldc, int M, int N, int K, bool relu,
                                 bool accumulate, hipStream_t s) {
    if (K % FKS || M <= 0 || N <= 0 || (lda % 4) || count < 1 || count > F32_BATCH_MAX) return hipErrorInvalidValue;
    // waves per tile = K split: a function of K ALONE, so that the summation order (hence every bit of the
    // result) does not depend on how many rows -- prompts -- a call carries: predict(32 boxes) must equal
    // predict(20) + predict(12).  M is a few hundred rows on this path, so the deepest split always pays.
    int S = 1;
    while (S < 16 && K % (2 * FKS * S) == 0) S *= 2;
    dim3 grid((N + FN - 1) / FN, (M + FM - 1) / FM, count);
    const int r = relu ? 1 : 0, a = accumulate ? 1 : 0;
    switch (S) {
        case 1: gemm_f32_kernel<1><<<grid, 64, 0, s>>>(bt, lda, ldc, M, N, K, r, a); break;
        case 2: gemm_f32_kernel<2><<<grid, 128, 0, s>>>(bt, lda, ldc, M, N, K, r, a); break;
        case 4: gemm_f32_kernel<4><<<grid, 256, 0, s>>>(bt, lda, ldc, M, N, K, r, a); break;
        case 8: gemm_f32_kernel<8><<<grid, 512, 0, s>>>(bt, lda, ldc, M, N, K, r, a); break;
        default: gemm_f32_kernel<16><<<grid, 1024, 0, s>>>(bt, lda, ldc, M, N, K, r, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C,
                           int ldc, int M, int N, int K, bool relu, bool accumulate, hipStream_t s) {
    F32Batch bt{};
    bt.A[0] = A; bt.W[0] = W; bt.bias[0] = bias; bt.C[0] = C;
    return launch_gemm_f32_batch(bt, 1, lda, ldc, M, N, K, relu, accumulate, s);
}

hipError_t launch_gemm_et_gln(int prec, const void* A, const void* B, void* C, const float* bias, const float* gamma_beta,
                              int M, int N, int K, hipStream_t s, const void* A_lo, const void* B_lo) {
    if (M % DBM || N % DBN || K % DBK || M <= 0 || !gamma_beta || ((A_lo == nullptr) != (B_lo == nullptr))) return hipErrorInvalidValue;
    if (prec == PREC_BF16) return launch_gemm_dual_gln<PREC_BF16>(A, B, C, bias, gamma_beta, M, N, K, s, A_lo, B_lo);
    if (prec == PREC_F16) return launch_gemm_dual_gln<PREC_F16>(A, B, C, bias, gamma_beta, M, N, K, s, A_lo, B_lo);
    return hipErrorInvalidValue;
}


// ---- MXFP4 lo-term GEMM (gemm_et_mx_kernel) ----------------------------------------------------------------------------------
bool gemm_mx_ok(int M, int N, int K, int Kp) {
    return M > 0 && N > 0 && K > 0 && M % QBM == 0 && N % WBN == 0 && K % XBK == 0 && Kp > 0 && Kp % MXK == 0;
}
size_t mx_scale_bytes(int rows, int Kp, bool is_b) {
    return is_b ? (size_t)((rows + 319) / 320) * (Kp / MXK) * MX_SB_BYTES : (size_t)((rows + 255) / 256) * (Kp / MXK) * MX_SA_BYTES;
}
hipError_t launch_gemm_et_mx(int prec, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int Kp,
                             const void* a4_lo, const void* a4_hi, const void* sa_lo, const void* sa_hi,
                             const void* b4_hi, const void* b4_lo, const void* sb_hi, const void* sb_lo,
                             bool out_f32, bool accumulate, int split_from_n, hipStream_t s, bool gelu, void* o4_hi, void* o4_lo,
                             void* so_hi, void* so_lo, int ld, bool ext) {
    if ((gelu || o4_hi) && out_f32) return hipErrorInvalidValue;
    if (ld < 0 || (ld && ld < K + (ext ? XBK : 0)) || (ext && (!ld || o4_hi || out_f32 || split_from_n <= 0))) return hipErrorInvalidValue;
    // the MX-row epilogue goes with lo terms on every tile or (split_from_n == N) on none
    if (o4_hi && (!o4_lo || !so_hi || !so_lo || N % 80 || ((N / 80) * 96) % MXK || (split_from_n && split_from_n != N))) return hipErrorInvalidValue;
    if (!gemm_mx_ok(M, N, K, Kp) || !A || !B || !C || !a4_lo || !a4_hi || !sa_lo || !sa_hi || !b4_hi || !b4_lo || !sb_hi || !sb_lo)
        return hipErrorInvalidValue;
    if ((accumulate && !out_f32) || split_from_n < 0 || split_from_n % WBN || (split_from_n && out_f32)) return hipErrorInvalidValue;
    // no tile takes lo terms and the output is wanted as MX rows (lin1 of split 207): the plain persistent kernel with the MX-row
    // epilogue -- the f16 main loop of gemm_et_x64p_kernel is ~5 % faster than this kernel's (SAMRS_MX_LIN1=0: A/B switch)
    static const bool x64p_lin1 = [] { const char* v = getenv("SAMRS_MX_LIN1"); return !(v && atoi(v) == 0); }();
    if (x64p_lin1 && o4_hi && split_from_n == N && N % XBK == 0 && K % XBK == 0) {
        if (prec == PREC_F16) return launch_gemm_x64p_mxo<PREC_F16>(A, B, C, bias, M, N, K, gelu, o4_hi, o4_lo, so_hi, so_lo, s);
        if (prec == PREC_BF16) return launch_gemm_x64p_mxo<PREC_BF16>(A, B, C, bias, M, N, K, gelu, o4_hi, o4_lo, so_hi, so_lo, s);
        return hipErrorInvalidValue;
    }
    MxOperands mx;
    mx.a4_lo = (const unsigned char*)a4_lo; mx.a4_hi = (const unsigned char*)a4_hi;
    mx.b4_hi = (const unsigned char*)b4_hi; mx.b4_lo = (const unsigned char*)b4_lo;
    mx.sa_lo = (const unsigned char*)sa_lo; mx.sa_hi = (const unsigned char*)sa_hi;
    mx.sb_hi = (const unsigned char*)sb_hi; mx.sb_lo = (const unsigned char*)sb_lo;
    mx.Kp = Kp; mx.split_from_n = split_from_n;
    const int ntiles = (M / QBM) * (N / WBN);
    static const int n_cu = [] {
        const char* v = getenv("SAMRS_MX_PERSIST");          // 0: one block per tile (A/B runs)
        if (v && atoi(v) == 0) return 1 << 30;
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const dim3 grid(ntiles < n_cu ? ntiles : n_cu), block(QTHREADS);
    const uint16_t* a = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b = reinterpret_cast<const uint16_t*>(B);
    const int acc = accumulate ? 1 : 0;
    MxOut mxo;
    mxo.q_hi = (unsigned char*)o4_hi; mxo.q_lo = (unsigned char*)o4_lo; mxo.s_hi = (unsigned char*)so_hi; mxo.s_lo = (unsigned char*)so_lo;
#define MXL(P_)                                                                                                          \
    do {                                                                                                                 \
        if (out_f32) gemm_et_mx_kernel<P_, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, mx, MxOut(), ld, ext ? 1 : 0);                \
        else if (o4_hi && gelu) gemm_et_mx_kernel<P_, false, true, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, mx, mxo, ld, ext ? 1 : 0);  \
        else if (o4_hi) gemm_et_mx_kernel<P_, false, false, true><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, mx, mxo, ld, ext ? 1 : 0);         \
        else if (gelu) gemm_et_mx_kernel<P_, false, true, false><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, mx, MxOut(), ld, ext ? 1 : 0);               \
        else gemm_et_mx_kernel<P_, false><<<grid, block, 0, s>>>(a, b, C, bias, M, N, K, acc, mx, MxOut(), ld, ext ? 1 : 0);                       \
    } while (0)
    if (prec == PREC_F16) MXL(PREC_F16);
    else if (prec == PREC_BF16) MXL(PREC_BF16);
    else return hipErrorInvalidValue;
#undef MXL
    return hipGetLastError();
}

// fp32 [rows][K] (x) or the ET pair (hi_in, lo_in) -> fp4 hi / lo [rows][Kp / 2] + their scale tiles; Kp = K / G * GP
hipError_t launch_mx4_pack(int prec, const float* x, const void* hi_in, const void* lo_in, void* out_hi, void* q_hi, void* q_lo,
                           void* s_hi, void* s_lo, int rows, int K, int G, int GP, bool is_b, hipStream_t s, int perm) {
    if (rows <= 0 || K <= 0 || G <= 0 || K % G || G % 4 || GP % 32 || GP < G || ((K / G) * GP) % MXK) return hipErrorInvalidValue;
    if (!x && !(hi_in && lo_in)) return hipErrorInvalidValue;
    const long items = (long)rows * ((K / G) * GP / 32);
    const dim3 grid((unsigned)((items * 8 + 255) / 256)), block(256);
    const uint16_t* hi = reinterpret_cast<const uint16_t*>(hi_in);
    const uint16_t* lo = reinterpret_cast<const uint16_t*>(lo_in);
#define MXP(P_, F_) mx4_pack_kernel<P_, F_><<<grid, block, 0, s>>>(x, hi, lo, (uint16_t*)out_hi, (unsigned char*)q_hi, (unsigned char*)q_lo, \
                                                                  (unsigned char*)s_hi, (unsigned char*)s_lo, rows, K, G, GP, is_b ? 1 : 0, perm)
    if (prec == PREC_F16) { if (x) MXP(PREC_F16, true); else MXP(PREC_F16, false); }
    else if (prec == PREC_BF16) { if (x) MXP(PREC_BF16, true); else MXP(PREC_BF16, false); }
    else return hipErrorInvalidValue;
#undef MXP
    return hipGetLastError();
}
