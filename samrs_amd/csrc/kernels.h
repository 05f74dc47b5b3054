// kernels.h -- host-side launchers of every HIP kernel in libsamrs_hip (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- gemm.hip -------------------------------------------------------------------------------
hipError_t launch_gemm_et(int prec, const void* A, const void* B, void* C, const float* bias,
                          const float* add2d, int add2d_period, int M, int N, int K, bool out_f32,
                          bool gelu, bool accumulate, hipStream_t s);
// C (ET) = GELU(LayerNorm2d over every 64-column group of (A B^T + bias)), eps 1e-6; gamma_beta = gamma[64] | beta[64]
// A_lo / B_lo (both or neither): split-precision product of hi + lo operands; C is then FP32 [M][N] instead of ET
hipError_t launch_gemm_et_gln(int prec, const void* A, const void* B, void* C, const float* bias, const float* gamma_beta,
                              int M, int N, int K, hipStream_t s, const void* A_lo = nullptr, const void* B_lo = nullptr);
// LayerNorm folded into the neighbouring GEMMs (residual stream of N = 1280 columns):
//   producer  C (fp32) += A B^T + bias;  Xh = ET(C);  stats[m][N / 160] = (mean, sum of squared deviations) per 160 columns
//   (launch_ln_rowstat, encoder_kernels.hip: stats -> rowstat[m] = (rstd, -rstd mean))
//   consumer  C (ET) = [GELU](rstd_m (Xh Wf^T - mean_m cvec) + bias_f)
hipError_t launch_gemm_et_stats(int prec, const void* A, const void* B, float* C, const float* bias, void* Xh, float* stats,
                                int M, int N, int K, hipStream_t s);
hipError_t launch_gemm_et_fold(int prec, const void* Xh, const void* Wf, void* C, const float* bias_f, const float* cvec,
                               const float* rowstat, int M, int N, int K, bool gelu, hipStream_t s);
// Split-precision product in ONE launch: C = A B^T + A_lo B^T + A B_lo^T + bias over a three-segment K axis (fp32 accumulators stay
// in registers).  out_f32: C is fp32 [M][N] (optionally accumulated into), else ET rounded once.  gemm_split3_ok: M % 256,
// N % 320, K % 64 (the ViT-H block GEMMs); other shapes take three accumulating launch_gemm_et passes.
bool gemm_split3_ok(int M, int N, int K, bool out_f32 = false);      // fp32 outputs also take N % 256 == 0 (the neck)
hipError_t launch_gemm_et_split3(int prec, const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, const float* bias,
                                 int M, int N, int K, bool out_f32, bool accumulate, hipStream_t s, int split_from_n = 0,
                                 const float* add2d = nullptr, int add2d_period = 0);   // add2d: fp32 outputs only (patch embed + pos_embed)
// split_from_n (ET output only, a multiple of 320): only output columns >= split_from_n take the lo terms; the tiles in front
// of it are the plain hi x hi product (qkv: the v third alone)
void set_gemm_variant(int v);   // process-wide test hook (kernel-level entry points): 0 = register-staged tiles, ..., 8 = automatic
// proj / lin2 with the LayerNorm that follows them as a tail of the GEMM (gemm.hip gemm_et_x64_kernel<LNT>): C (fp32) += A B^T + bias,
// out_et = LN(C) in the operand type; hipErrorInvalidValue where the shape does not take that kernel (caller: separate launches)
bool gemm_has_experiments();     // built with -DSAMRS_EXPERIMENTS (make EXPERIMENTS=1): the 32x32x16 GEMM kernels, the LayerNorm fold, the timing ablations
bool gemm_lntail_ok(int M, int N, int K);
hipError_t launch_gemm_et_lntail(int prec, const void* A, const void* B, float* C, const float* bias, int M, int N, int K,
                                 const float* gamma, const float* beta, float eps, void* out_et, unsigned int* counters, hipStream_t s);
int swap_gelu_form(int v);                // thread-local erf form of lin1's GELU epilogue (1 = fp32-epsilon class, 2 = cheaper); returns the previous value
// Operand row stride of the calling thread's next plain ET GEMM launches (elements; 0 = K): the persistent ET kernels read A and B with
// it (gemm.hip tl_gemm_ld).  gemm_ld_ok: a launch of this shape runs on one of them (anything else refuses a stride).
int swap_gemm_ld(int ld);
bool gemm_ld_ok(int M, int N, int K, bool gelu);
// C (fp32 [M][N], the residual stream) += A B^T + A_x B_x^T + bias: proj / lin2 with ONE more 64-k stage read from two dense side operands
// A_x [M][64], B_x [N][64] (hi + lo of the outlier columns, engine.hip EncBlock::oc_*).  gemm_ext_ok: the shapes of the 256 x 320
// pair-stage kernel (ViT-H at >= 2 tiles); other shapes add the side product with an accumulating launch of their own.
bool gemm_ext_ok(int M, int N, int K);
hipError_t launch_gemm_et_ext(int prec, const void* A, const void* B, const void* Ax, const void* Bx, float* C, const float* bias,
                              int M, int N, int K, hipStream_t s);
int swap_gemm_variant_override(int v);   // thread-local override (-1 = none) used by engine handles; returns the previous value
void set_gemm_skew(int xcd_units, int cu_units);   // first-round start skew of the pair-stage GEMM (1024-cycle units)
hipError_t launch_gemm_f32(const float* A, int lda, const float* W, const float* bias, float* C,
                           int ldc, int M, int N, int K, bool relu, bool accumulate, hipStream_t s);
// up to F32_BATCH_MAX same-shape fp32 GEMMs in one launch: C[z] = (A[z] (+ A2[z])) * W[z]^T + bias[z]
constexpr int F32_BATCH_MAX = 5;
struct F32Batch {
    const float* A[F32_BATCH_MAX];
    const float* A2[F32_BATCH_MAX];     // optional elementwise addend of A (same lda), or null
    const float* W[F32_BATCH_MAX];
    const float* bias[F32_BATCH_MAX];   // or null
    float* C[F32_BATCH_MAX];
};
hipError_t launch_gemm_f32_batch(const F32Batch& bt, int count, int lda, int ldc, int M, int N, int K, bool relu,
                                 bool accumulate, hipStream_t s);

// ---- encoder_kernels.hip --------------------------------------------------------------------
// A_lo / out_lo (optional): the remainder of the two-term operand split (common.h split2_pack), same layout as the main output
hipError_t launch_patch_im2col(int prec, const uint8_t* img, void* A, int n_images, int in_h, int in_w,
                               int grid, int patch, hipStream_t s, void* A_lo = nullptr);
hipError_t launch_convert(int prec, const float* in, void* out, long n, hipStream_t s, void* out_lo = nullptr);
// folded LayerNorm (ViT-H): weight preparation (once) and the entry of the folded path (Xh = ET(X) + per-row partial statistics)
hipError_t launch_ln_fold_weight(int prec, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf,
                                 float* cvec, float* bias_f, int N, int K, hipStream_t s);
hipError_t launch_rowstats_convert(int prec, const float* X, void* Xh, float* stats, int rows, int D, hipStream_t s);
// stats [rows][8] (mean, M2) of 160-element groups -> rowstat [rows] = (rstd, -rstd mean), Chan merge in a fixed order
hipError_t launch_ln_rowstat(const float* stats, float* rowstat, int rows, float eps, hipStream_t s);
hipError_t launch_layernorm(int prec, const float* X, const float* gamma, const float* beta, float eps,
                            void* out_et, float* out_f32, int rows_out, int D, int window_mode, int grid,
                            int window, hipStream_t s, void* out_lo = nullptr,
                            // optional (plain row order, D % 256 == 0): the output's hi / lo as MXFP4 codes [rows][D / 2] + scale tiles
                            // (the A operands of launch_gemm_et_mx: no ET lo copy, no separate pack pass)
                            void* mx_q_hi = nullptr, void* mx_q_lo = nullptr, void* mx_s_hi = nullptr, void* mx_s_lo = nullptr,
                            int ld_out = 0 /* row stride of out_et in elements, 0 = D (plain ET output only) */,
                            // optional (plain row order, ld_out >= D + 64): hi + lo of n_oc <= 32 outlier columns as 64 more K columns
                            // of the row: [D + j] = lo of column oc_idx[j], [D + 32 + j] = its hi, zeros in unused slots
                            const int* oc_idx = nullptr, int n_oc = 0);
// weight side of that extension: out[r][col0 + j] = ET(W[r][idx[j]]), out[r][col0 + 32 + j] = ET(W - hi), W fp32 [N][K], out rows of stride ld
hipError_t launch_outlier_weight_ext(int prec, const float* W, int N, int K, const int* idx, int n_oc, void* out, int ld, int col0, hipStream_t s);
// the producers of the side operands of proj / lin2 (launch_gemm_et_ext): see the kernels in encoder_kernels.hip
hipError_t launch_outlier_side_weight(int prec, const float* W1, const float* b1, int D, const int* idx2, int n2, const int* idx1, int n1,
                                      void* out, int Ks, float* bias_out, hipStream_t s);
hipError_t launch_outlier_gather(const void* hi, const void* lo, int D, const int* idx, int n_oc, void* out, int rows, hipStream_t s);
// lin2's A_x: out [M][64] = lo | hi of GELU(Y Ws^T + bias), Y [M][K] with row stride lda, Ws [32][K], bias [32]; M % 64 == 0, K % 32 == 0
hipError_t launch_outlier_side_gemm(int prec, const void* Y, int lda, const void* Ws, const float* bias, int M, int K, void* out, hipStream_t s);
// squared L2 norms of the columns / rows of an fp32 matrix [N][K] (either output may be null); load-time scoring of outlier columns
hipError_t launch_weight_norms(const float* W, int N, int K, float* col_sq, float* row_sq, hipStream_t s);
// out_lo (optional, both attention kernels): the split remainder of `out` (reference-grade mode: proj on hi + lo operands)
hipError_t launch_window_attention(int prec, const void* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, void* out,
                                   int n_images, int grid, int window, int heads, int head_dim, hipStream_t s, void* out_lo = nullptr,
                                   // optional (instead of out_lo): hi / lo of the output as MXFP4 on the per-head padded K axis
                                   // [rows][heads * ceil32(head_dim) / 2], block-internal order of store_attention_row_mx
                                   void* mx_q_hi = nullptr, void* mx_q_lo = nullptr, void* mx_s_hi = nullptr, void* mx_s_lo = nullptr,
                                   // out_lo only: bit h set = head h writes its remainder (default: every head)
                                   uint32_t lo_heads = 0xffffffffu);
hipError_t launch_gelu_split(int prec, const float* in, void* hi, void* lo, long n, hipStream_t s);
// operand-range check: adds to *counter the elements of an ET tensor that sit at the operand type's saturation value or beyond (n % 8 == 0)
hipError_t launch_range_scan(int prec, const void* x, long n, unsigned long long* counter, hipStream_t s,
                             int cols = 0, int ld = 0 /* optional: rows of `cols` live elements at a stride of `ld` elements; n = rows * cols */);
// vt_ws: ET workspace of n_images * heads * head_dim * grid^2 elements (receives V transposed per head)
hipError_t launch_global_attention(int prec, const void* qkv, const float* rel_h, const float* rel_w, void* out,
                                   int n_images, int grid, int heads, int head_dim, void* vt_ws, hipStream_t s, void* out_lo = nullptr,
                                   void* mx_q_hi = nullptr, void* mx_q_lo = nullptr, void* mx_s_hi = nullptr, void* mx_s_lo = nullptr);
hipError_t launch_neck_im2col(const void* in, void* A, int n_images, int grid, int C, hipStream_t s);
hipError_t launch_transpose_f32(const float* in, float* out, int rows, int cols, hipStream_t s);

// ---- decoder_kernels.hip --------------------------------------------------------------------
struct PromptParams {
    const float* boxes;          // [n,4] or null
    const float* point_coords;   // [n,np,2] or null
    const int32_t* point_labels; // [n,np] or null
    int n_prompts, n_points;
    float img_size;
    const float* gauss;          // [2,128]
    const float* point_emb[4];   // 4 x [256]
    const float* not_a_point;    // [256]
    const float* iou_token;      // [256]
    const float* mask_tokens;    // [4,256]
};
// tokens [n, T, 256], T = 5 + n_points (+1 pad point when there is no box) + 2*(boxes != null)
hipError_t launch_prompt_tokens(const PromptParams& p, float* tokens, float* tokens2 /* optional second copy */, int T, hipStream_t s);
hipError_t launch_dense_pe(const float* gauss, float* pe /*[g*g,256]*/, int grid, hipStream_t s);
// mask prompt -> dense embedding [n, 4096, 256] (prompt_encoder.py:51-59)
struct MaskEmbedParams {
    const float *w0, *b0, *ln1w, *ln1b, *w3, *b3, *ln4w, *ln4b, *w6, *b6;
};
hipError_t launch_mask_embed(const MaskEmbedParams& p, const float* mask_in, float* dense, int n, int grid, hipStream_t s);
// out_f32[b, t, :] = emb[t, :] + (dense ? dense[b, t, :] : vec[:]) and its ET copy
hipError_t launch_make_keys(int prec, const float* emb, const float* dense, const float* vec, float* out_f32,
                            void* out_et, int n_batches, int tokens, int C, hipStream_t s);
hipError_t launch_add_f32(const float* a, const float* b, float* out, long n, hipStream_t s);
// token self attention: q,k,v [n*T, C] -> o [n*T, C]; heads of dim C/heads
hipError_t launch_token_self_attn(const float* q, const float* k, const float* v, float* o, int n, int T, int C,
                                  int heads, hipStream_t s);
// tokens -> image attention. qp [n*T, Ci] fp32; kp/vp ET rows of `ld` elements, batch stride in rows
// (0 = shared by all prompts); o [n*T, Ci] fp32.
constexpr int T2I_MAX_SPLITS = 16;
size_t t2i_workspace_floats(int n_prompts, int T);     // scratch for the per-split partial softmax states
hipError_t launch_t2i_attention(int prec, const float* qp, const void* kp, const void* vp, int ld, long batch_stride_rows,
                                float* out, float* workspace, int n, int T, int tokens, int Ci, int heads, hipStream_t s);
// image -> tokens attention. qi ET rows of `ld` elements (batch stride in rows, 0 = shared);
// kt, vt [n*T, Ci] fp32; out ET [n*tokens, Ci].
hipError_t launch_i2t_attention(int prec, const void* qi, int ld, long batch_stride_rows, const float* kt,
                                const float* vt, void* out, int n, int T, int tokens, int Ci, int heads, hipStream_t s);
// per row of `in` [rows, groups*gsize]: +0, LN over each group (eps), GELU -> ET
hipError_t launch_group_ln_gelu(int prec, const float* in, const float* gamma, const float* beta, float eps,
                                void* out, long rows, int groups, int gsize, hipStream_t s);
// low[b, c, Y, X] = sum_ch hyper[b, sel0 + c, ch] * up2[b, y, x, dy, dx, dy2, dx2, ch]
// fused ConvT #2 (K = 64 GEMM) + GELU + hypernetwork product: u1 [n*grid*grid*4][64] ET -> low [n][n_sel][4 grid][4 grid]
// w_lo != null: split precision -- u1 is then the FP32 output of the first transposed conv [rows][64] (split in registers)
hipError_t launch_upscale2_masks(int prec, const void* u1, const void* w, const void* w_lo, const float* bias, const float* hyper,
                                 float* low, int n, int grid, int n_mask_tokens, int sel0, int n_sel, hipStream_t s);
hipError_t launch_mask_product(int prec, const void* up2, const float* hyper, float* low, int n, int grid,
                               int n_mask_tokens, int sel0, int n_sel, hipStream_t s);
hipError_t launch_postprocess(const float* low, int n_masks, int in_h, int in_w, int orig_h, int orig_w,
                              int img_size, int return_logits, void* out, hipStream_t s);
hipError_t launch_paint(const uint8_t* masks, const int32_t* labels, int n, int h, int w, uint8_t* seg,
                        unsigned long long* areas, unsigned long long* class_pixels,
                        unsigned long long* class_instances, int n_classes, hipStream_t s);
// best-of-nsel by predicted IoU (first maximum): out [n][h][w] = masks[j][argmax_c iou[j][c]], quality[j], areas[j] = pixels set
hipError_t launch_select_best(const uint8_t* masks, const float* iou, int n, int nsel, int h, int w, uint8_t* out, float* quality,
                              unsigned long long* areas, hipStream_t s);
hipError_t launch_resample_pass(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* coef, int ksize,
                                int in_len, int out_len, int other, int horizontal, hipStream_t s);
// rotated-box polygons (int32 [n][nv][2]) -> mask prompts [n][out][out] fp32 (main_sam_rbox_mask_instance.py:125-141)
hipError_t launch_rbox_prompt(const int32_t* pts, int n, int nv, int h, int w, int th, int tw, int img_size, int out_size,
                              float* out, hipStream_t s,
                              int fill_rule = 0 /* 0: OpenCV <= 4.5.1 spans (ceil .. floor), 1: OpenCV >= 4.5.2 (round .. round) */);
// fused transformer.py:176-181: keys = LayerNorm(resid + out_proj(attention(q_i2t, k_tokens, v_tokens))) -> outF (fp32) and outE (ET)
hipError_t launch_i2t_fused(int prec, const void* qi, int ld, long q_bstride, const float* kt, const float* vt, const void* w,
                            const void* w_lo /* null: un-split out-projection */, const float* bias, const float* resid,
                            long r_bstride, const float* gamma, const float* beta, float eps, float* outF, void* outE,
                            void* outE_lo /* optional split remainder of outE */, int n, int T, int tokens, int Ci, int C, hipStream_t s);

// ---- gemm.hip: operand split with the two correction terms on MXFP4 operands (gemm_et_mx_kernel) ------------------------------
// C = A B^T (f16 segment, K) + A4lo B4hi^T + A4hi B4lo^T (block-scaled fp4 segments over the padded K axis Kp) + bias.
// a4_* [M][Kp / 2], b4_* [N][Kp / 2] bytes and their scale tiles come from launch_mx4_pack (sizes: mx_scale_bytes).  ET output
// rounded once from the fp32 accumulators, or fp32 output (optionally accumulated into C).  M % 256 == 0, N % 320 == 0,
// K % 64 == 0, Kp % 256 == 0.  split_from_n (ET outputs, a multiple of 320): columns below it take no lo terms.
bool gemm_mx_ok(int M, int N, int K, int Kp);
size_t mx_scale_bytes(int rows, int Kp, bool is_b);
hipError_t launch_gemm_et_mx(int prec, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int Kp,
                             const void* a4_lo, const void* a4_hi, const void* sa_lo, const void* sa_hi,
                             const void* b4_hi, const void* b4_lo, const void* sb_hi, const void* sb_lo,
                             bool out_f32, bool accumulate, int split_from_n, hipStream_t s,
                             // ET outputs only: GELU in the epilogue, and (o4_*: all four or none) the output ALSO as MXFP4 hi / lo on a K
                             // axis padded per 80-column wave tile to 96 ([M][N / 80 * 48] bytes, A-operand scale tiles, block-internal
                             // order = launch_mx4_pack perm 2): lin1 -> lin2 of the all-split mode
                             bool gelu = false, void* o4_hi = nullptr, void* o4_lo = nullptr, void* so_hi = nullptr, void* so_lo = nullptr,
                             // round 6: operand rows of stride ld (0 = K); ext: they carry the 64-column outlier extension behind their K
                             // live columns, which the tiles WITHOUT lo terms (n0 < split_from_n) read as one more f16 stage
                             int ld = 0, bool ext = false);
// x (fp32 [rows][K]) or the ET pair (hi_in, lo_in) -> fp4 codes of hi and lo [rows][Kp / 2] + E8M0 scale tiles (A layout, or the
// B layout when is_b); optional out_hi = ET(x).  Padded axis: every group of G source elements becomes GP (zeros behind it);
// Kp = K / G * GP must be a multiple of 256.  Plain: G = GP = K.
// perm: the block-internal element order of the attention kernels' own MX outputs (store_attention_row_mx: position 16 hh + 4 g + e
// holds element 8 g + 4 hh + e) -- for the proj weights, whose A operand those kernels write.
hipError_t launch_mx4_pack(int prec, const float* x, const void* hi_in, const void* lo_in, void* out_hi, void* q_hi, void* q_lo,
                           void* s_hi, void* s_lo, int rows, int K, int G, int GP, bool is_b, hipStream_t s, int perm = 0 /* 1: attention, 2: GEMM epilogue */);

// ---- upscaler_fused.hip ---------------------------------------------------------------------
// mask_decoder.py:53-59,154-167 in one kernel: keys [n * grid^2][256] ET -> ConvT #1 + LayerNorm2d + GELU -> ConvT #2 + GELU ->
// hypernetwork dot -> low [n][n_sel][4 grid][4 grid].  w1 [256][256], w2 [128][64] in the GEMM-B layouts of engine.hip, b1 [256] /
// b2 [128] tiled over the sub-pixels, ln = gamma[64] | beta[64].  *_lo (all three or none): split precision.  grid % 16 == 0.
hipError_t launch_upscaler_fused(int prec, const void* keys, const void* keys_lo, const void* w1, const void* w1_lo, const float* b1,
                                 const float* ln, const void* w2, const void* w2_lo, const float* b2, const float* hyper, float* low,
                                 int n, int grid, int n_mask_tokens, int sel0, int n_sel, hipStream_t s);

// ---- rle_kernels.hip ------------------------------------------------------------------------
// COCO RLE strings of n binary masks (uint8 [n][h][w], non-zero = set), packed behind *cursor (device int64, in / out)
// into `out` (device bytes, capacity out_cap) at 16-byte aligned offsets; table [n][3] (device int64) = (offset,
// length, n_counts), length < 0: did not fit (-length - 1 bytes were needed).  scratch: rle_scratch_bytes(n, h, w).
size_t rle_scratch_bytes(int n, int h, int w);
hipError_t launch_rle_encode(const uint8_t* masks, int n, int h, int w, void* scratch, unsigned char* out, long long out_cap,
                             long long* cursor, long long* table, hipStream_t s);
