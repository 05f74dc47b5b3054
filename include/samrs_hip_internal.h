/*
 * samrs_hip_internal.h -- NOT the drop-in boundary.  Kernel-level entry points (samrs_k_*: one kernel each, used by the parity
 * tests to check every kernel alone against the oracle) and test / measurement hooks (samrs_debug_*) of libsamrs_hip.so.
 * A caller that binds the engine (include/samrs_hip.h) needs nothing from this file; nothing here carries a
 * compatibility promise, and several entries keep process-wide or per-device state that product code must not share
 * (noted at each).  VERDICT r05 "what's weak" 12: a caller linking the public header could not tell product from probe.
 */
#ifndef SAMRS_HIP_INTERNAL_H
#define SAMRS_HIP_INTERNAL_H

#include "samrs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* -- test hook: run preprocess + patch embed + the first n_blocks encoder blocks for n_images
 * tiles and copy the fp32 residual stream [n_images*4096, embed_dim] (channels-last,
 * image_encoder.py:107-112) to x_out.  Invalidates the embedding slots. */
int samrs_debug_encoder_prefix(samrs_engine_t* e, const uint8_t* images, int n_images, int in_h,
                               int in_w, int n_blocks, float* x_out, void* stream);

/* -- test hook: the outlier K-columns (option "outlier_cols") picked at samrs_finalize_weights for block GEMM `gemm` (0 qkv, 1 lin1,
 * 2 lin2, 3 proj) of encoder block `block`: returns their number (<= 32) and writes the ascending indices to the HOST array out32. */
int samrs_debug_outlier_columns(samrs_engine_t* e, int block, int gemm, int32_t* out32);

/* -- process-wide test / tuning hooks of the KERNEL-LEVEL entry points below (samrs_k_gemm has no handle): GEMM tile variant,
 * and the start skew of the first round of GEMM blocks, per XCD / per CU group, in 1024-cycle units (0, 0 = off). */
void samrs_debug_set_gemm_variant(int variant);
void samrs_debug_set_gemm_skew(int xcd_units, int cu_units);
/* 1 when the library was built with `make EXPERIMENTS=1`: the kernels that were measured and not adopted (GEMM variants 30 - 36 on
 * v_mfma_f32_32x32x16, the LayerNorm fold "ln_fold", the timing ablations 60 - 92 / 100 - 196) exist; 0 in the product build, where those
 * variants fall back to the default kernels and "ln_fold" = 1 is refused. */
int samrs_debug_has_experiments(void);

/* -- test hook: copy a prefix of a named internal decoder buffer (Q, KF, KE, KVQ, OI, U1raw, U1, U2,
 * HYPER, ...) to a device buffer; used to localise run-to-run differences. */
int samrs_debug_copy_buffer(samrs_engine_t* e, const char* name, void* dst, size_t bytes, void* stream);

/* -- measurement hook: when enabled, samrs_set_images brackets every launch of the dominant kernel
 * (MLP lin1 + GELU GEMM, [n*4096, D] x [4D, D]^T) with hipEvents on the launch stream.
 * samrs_debug_dominant_kernel_time() synchronises those events, returns the average duration (ms),
 * the number of launches since the last call, and the GEMM's N / K. */
int samrs_debug_time_dominant_kernel(samrs_engine_t* e, int enable);
int samrs_debug_dominant_kernel_time(samrs_engine_t* e, float* avg_ms, int* launches, int* M, int* N, int* K);

/* -- kernel-level entry points (used by the parity tests to check each kernel alone) --------
 * All pointers are device pointers.  `prec` is enum samrs_precision; "et" = MFMA operand type
 * (bf16 or f16 bit patterns in uint16). */
int samrs_k_gemm(int prec, const void* A_et, const void* B_et, void* C, const float* bias,
                 const float* add2d, int add2d_period, int M, int N, int K,
                 int out_f32, int gelu, int accumulate, void* stream);
int samrs_k_gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C, int ldc,
                     int M, int N, int K, int relu, int accumulate, void* stream);
/* LayerNorm folded into the neighbouring GEMMs of an encoder block (image_encoder.py:168,177; residual stream of 1280 columns):
 *   samrs_k_gemm_stats       C (fp32) += A_et B_et^T + bias;  xh_et = ET(C);  stats[m][8] = (mean, sum of squared deviations)
 *                            of eight 160-element groups of row m.  M % 256 == 0, N == 1280, K % 128 == 0.
 *   samrs_k_ln_rowstat       rowstat[m] = (rstd_m, -rstd_m mean_m) merged from stats[m][.] with eps
 *   samrs_k_gemm_fold        C_et = [GELU](rstd_m (xh_et Wf_et^T - mean_m cvec) + bias_f) with rowstat from above.
 *                            M % 256 == 0, N % 320 == 0, K == 1280.
 *   samrs_k_ln_fold_weight   Wf_et = ET(W diag(gamma)), cvec[n] = sum_k Wf_et[n][k], bias_f = bias + W beta  (W fp32 [N][K])
 *   samrs_k_rowstats_convert xh_et = ET(X), stats as above, for X fp32 [rows][1280] */
int samrs_k_gemm_stats(int prec, const void* A_et, const void* B_et, float* C, const float* bias, void* xh_et, float* stats,
                       int M, int N, int K, void* stream);
int samrs_k_ln_rowstat(const float* stats, float* rowstat, int rows, float eps, void* stream);
int samrs_k_gemm_fold(int prec, const void* xh_et, const void* Wf_et, void* C_et, const float* bias_f, const float* cvec,
                      const float* rowstat, int M, int N, int K, int gelu, void* stream);
int samrs_k_ln_fold_weight(int prec, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf_et,
                           float* cvec, float* bias_f, int N, int K, void* stream);
int samrs_k_rowstats_convert(int prec, const float* X, void* xh_et, float* stats, int rows, int D, void* stream);
int samrs_k_convert(int prec, const float* in, void* out_et, int64_t n, void* stream);
int samrs_k_layernorm(int prec, const float* X, const float* gamma, const float* beta, float eps,
                      void* out_et, float* out_f32, int rows_out, int D, int window_mode,
                      int n_images, int grid, int window, void* stream);
/* qkv_et: [n_images*grid*grid, 3D] in token order; qkv_bias fp32 [3D] supplies k / v of the window
 * padding positions (zero tokens after norm1 in the reference). */
int samrs_k_window_attention(int prec, const void* qkv_et, const float* qkv_bias, const float* rel_h,
                             const float* rel_w, void* out_et, int n_images, int grid, int window,
                             int heads, int head_dim, void* stream);
/* TEST / BENCH hook (as is samrs_k_attention_mx with global = 1): the V^T workspace the engine owns is, here, one grow-only
 * buffer per device kept by the library; growing it synchronizes the device first.  Not for product code: an engine's own
 * encoder pass (samrs_set_images) uses the engine's workspace and never this one. */
int samrs_k_global_attention(int prec, const void* qkv_et, const float* rel_h, const float* rel_w,
                             void* out_et, int n_images, int grid, int heads, int head_dim,
                             void* stream);
/* neck (image_encoder.py:88-104): im2col of the 3x3 / pad 1 convolution on a channels-last ET tensor
 * in [n_images][grid][grid][C] -> A [n_images*grid*grid][9*C], k = (ky*3 + kx)*C + c, zero outside the image. */
int samrs_k_neck_im2col(const void* in_et, void* A_et, int n_images, int grid, int C, void* stream);
int samrs_k_postprocess(const float* lowres, int n_masks, int in_h, int in_w, int orig_h,
                        int orig_w, int img_size, int return_logits, void* out, void* stream);
/* First transposed conv of the mask upscaler as a GEMM with LayerNorm2d(64) + GELU in its epilogue
 * (segment_anything/modeling/mask_decoder.py:53-56): C_et[M,N] = GELU(LN64(A_et[M,K] B_et[N,K]^T + bias)),
 * every 64-column group of N normalised on its own, eps 1e-6; gamma_beta = gamma[64] | beta[64].
 * M % 256 == 0, N % 128 == 0, K % 32 == 0.  A_lo_et / B_lo_et (both or neither; samrs_k_convert_split): the split
 * remainders of the operands -- the product then runs as A_lo B + A B_lo + A B and C is written in FP32 [M][N]. */
int samrs_k_gemm_gln(int prec, const void* A_et, const void* B_et, void* C_et, const float* bias,
                     const float* gamma_beta, int M, int N, int K, const void* A_lo_et, const void* B_lo_et, void* stream);
/* Split-precision GEMM in one launch (the block GEMMs of the reference-grade mode, option "split" bits 16 / 32):
 * C = A B^T + A_lo B^T + A B_lo^T + bias over a three-segment K axis; A*, [M,K], B*, [N,K] in the operand type (hi / lo from
 * samrs_k_convert_split).  out_f32 = 0: C_et [M,N] rounded once from the fp32 accumulators; out_f32 = 1: C fp32 [M,N],
 * accumulate != 0 adds to what C holds (the residual stream).  M % 256 == 0, N % 320 == 0, K % 64 == 0, else SAMRS_ERR_BAD_SHAPE.
 * split_from_n (out_f32 = 0 only, a multiple of 320, 0 = everywhere): only output columns >= split_from_n take the lo terms. */
int samrs_k_gemm_split3(int prec, const void* A_et, const void* A_lo_et, const void* B_et, const void* B_lo_et, void* C,
                        const float* bias, int M, int N, int K, int out_f32, int accumulate, int split_from_n, void* stream);
/* Second transposed conv + GELU + hypernetwork product (mask_decoder.py:57-59,154-167) in one pass:
 * u1_et [n*grid*grid*4, 64] (rows = prompt, token, sub-pixel 1), w_et [128, 64] (rows = sub-pixel 2 x 32
 * channels), bias [128], hyper [n, n_mask_tokens, 32] -> low [n, n_sel, 4*grid, 4*grid] fp32 for mask
 * tokens sel0 .. sel0+n_sel-1 (n_sel 1 or 3).  grid*grid*4 % 1024 == 0.  w_lo_et != NULL: split precision -- u1 is then
 * FP32 [rows][64] (split into hi + lo in registers) and w_lo_et the remainder of the weight split. */
int samrs_k_upscale2_masks(int prec, const void* u1, const void* w_et, const void* w_lo_et, const float* bias,
                           const float* hyper, float* low, int n, int grid, int n_mask_tokens,
                           int sel0, int n_sel, void* stream);
/* The whole mask upscaler in one kernel (mask_decoder.py:53-59,154-167): keys_et [n*grid*grid, 256] -> ConvT #1 (w1_et [256][256]
 * rows = sub-pixel 1 x 64 channels, b1 [256]) -> LayerNorm2d(64) (ln = gamma[64] | beta[64], eps 1e-6) -> GELU -> ConvT #2
 * (w2_et [128][64] rows = sub-pixel 2 x 32 channels, b2 [128]) -> GELU -> dot with hyper [n, n_mask_tokens, 32] ->
 * low [n, n_sel, 4*grid, 4*grid] fp32.  keys_lo / w1_lo / w2_lo (all three or none): split precision.  grid % 16 == 0. */
int samrs_k_upscaler_fused(int prec, const void* keys_et, const void* keys_lo_et, const void* w1_et, const void* w1_lo_et,
                           const float* b1, const float* ln, const void* w2_et, const void* w2_lo_et, const float* b2,
                           const float* hyper, float* low, int n, int grid, int n_mask_tokens, int sel0, int n_sel, void* stream);
/* Operand split with the two correction terms on MXFP4 operands (gfx950 v_mfma_scale_f32_16x16x128_f8f6f4, e2m1 codes + one E8M0
 * scale per 32 k; option "lo_format" = 4):
 *   samrs_k_mx4_pack   x (fp32 [rows][K]) or the ET pair (hi_in, lo_in) -> fp4 codes of hi / lo, q_* [rows][Kp / 2] bytes, and their
 *                      scale tiles s_* (samrs_k_mx_scale_bytes(rows, Kp, is_b) bytes each; A-operand or B-operand tile order);
 *                      out_hi (optional, with x): ET(x).  Every group of G source elements becomes GP on the padded axis (zeros
 *                      behind it): Kp = K / G * GP, a multiple of 256; plain: G = GP = K.  is_b: bit 0 = B-operand scale tiles, bit 2 = the block order of a GEMM epilogue's MX rows (position 8 f + 4 i + e holds column 16 i + 4 f + e), bit 1 = the
 *                      block-internal element order in which the attention kernels emit their own MX rows (position 16 h + 4 g + e
 *                      of a block holds element 8 g + 4 h + e): what the proj weights are packed with.
 *   samrs_k_gemm_mx    C = A B^T + A4lo B4hi^T + A4hi B4lo^T + bias: the f16 product over K, the two fp4 products over Kp, fp32
 *                      accumulators throughout; out_f32 = 0: C_et rounded once, = 1: fp32 (accumulate != 0 adds to C).
 *                      M % 256 == 0, N % 320 == 0, K % 64 == 0; split_from_n as for samrs_k_gemm_split3. */
/* the attention kernels with their optional extra outputs: out_lo (the f16 split remainder) or, instead, hi / lo of the output as
 * MXFP4 on the per-head padded K axis (q_* [rows][heads * ceil32(head_dim) / 2], A-operand scale tiles); global = 0: windowed (14) */
int samrs_k_attention_mx(int prec, int global, const void* qkv_et, const float* qkv_bias, const float* rel_h, const float* rel_w,
                         void* out_et, void* out_lo_et, int n_images, int grid, int heads, int head_dim, void* q_hi, void* q_lo,
                         void* s_hi, void* s_lo, void* stream);
int64_t samrs_k_mx_scale_bytes(int rows, int Kp, int is_b);
/* LayerNorm that also emits its output as MXFP4 hi / lo (what the engine feeds the qkv GEMM in "lo_format" 4): D % 256 == 0 */
int samrs_k_layernorm_mx(int prec, const float* X, const float* gamma, const float* beta, float eps, void* out_et, int rows, int D,
                         void* q_hi, void* q_lo, void* s_hi, void* s_lo, void* stream);
int samrs_k_mx4_pack(int prec, const float* x, const void* hi_in, const void* lo_in, void* out_hi, void* q_hi, void* q_lo, void* s_hi,
                     void* s_lo, int rows, int K, int G, int GP, int is_b, void* stream);
int samrs_k_gemm_mx(int prec, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int Kp, const void* a4_lo,
                    const void* a4_hi, const void* sa_lo, const void* sa_hi, const void* b4_hi, const void* b4_lo, const void* sb_hi,
                    const void* sb_lo, int out_f32, int accumulate, int split_from_n, void* stream);
/* samrs_k_gemm_mx with an ET output, optionally the exact-erf GELU in the epilogue, and (o4_*: all four or none) the output ALSO written
 * as MXFP4 hi / lo on a K axis padded per 80-column wave tile to 96 -- [M][N / 80 * 48] bytes + A-operand scale tiles, block-internal
 * order = samrs_k_mx4_pack is_b bit 2: lin1 feeding lin2 in the all-split mode.  `gelu`: bit 0 = GELU, bit 1 = NO tile takes lo terms
 * (lin1 of split 207, whose lin2 alone is split: the plain persistent kernel with the MX-row epilogue; the a4 / b4 operands are ignored) */
int samrs_k_gemm_mx_gelu_mxout(int prec, const void* A, const void* B, void* C_et, const float* bias, int M, int N, int K, int Kp,
                               const void* a4_lo, const void* a4_hi, const void* sa_lo, const void* sa_hi, const void* b4_hi, const void* b4_lo,
                               const void* sb_hi, const void* sb_lo, int gelu, void* o4_hi, void* o4_lo, void* so_hi, void* so_lo, void* stream);
/* fp32 -> hi (= samrs_k_convert) and lo = ET(x - hi): the two-term operand split */
int samrs_k_convert_split(int prec, const float* in, void* out_hi_et, void* out_lo_et, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMRS_HIP_INTERNAL_H */
