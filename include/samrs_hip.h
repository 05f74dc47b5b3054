/*
 * samrs_hip.h -- C ABI of libsamrs_hip.so, the MI355X (gfx950) SAM box->mask engine.
 *
 * This is the drop-in boundary for the ONE hot path of ViTAE-Transformer/SAMRS: what
 * `SamPredictor.set_image()` / `predict_torch()` compute
 * (reference: Generate Dataset/segment_anything/predictor.py:34-90,168-245).
 * Plain pointers + sizes only; no torch types.  All data pointers are DEVICE pointers unless a
 * parameter says otherwise; every entry point that launches work takes the HIP stream to launch
 * on (`hipStream_t` passed as `void*`; NULL = the legacy default stream) and does NOT
 * synchronise -- exactly like the reference, where the sync happens at the caller's `.cpu()`
 * (Generate Dataset/main_sam_hbox_semantic.py:189).
 *
 * One engine handle per GPU per process; a handle is not thread-safe (neither is the
 * reference's stateful SamPredictor, predictor.py:84-90).
 *
 * Return convention: 0 = OK, negative = error code below; `samrs_last_error()` returns the
 * message.  The Python wrapper re-raises the reference's exception types / messages
 * (predictor.py:133-134,213-214: RuntimeError "An image must be set ...").
 */
#ifndef SAMRS_HIP_H
#define SAMRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMRS_ABI_VERSION 5

enum samrs_status {
    SAMRS_OK = 0,
    SAMRS_ERR_NOT_SET = -1,      /* predict before set_images (predictor.py:213-214)            */
    SAMRS_ERR_BAD_SHAPE = -2,    /* e.g. long side != 1024 (predictor.py:79-83)                 */
    SAMRS_ERR_BAD_ARG = -3,      /* points without labels (predictor.py:139-141), null ptrs ... */
    SAMRS_ERR_HIP = -4,          /* a HIP runtime call failed                                   */
    SAMRS_ERR_BAD_WEIGHTS = -5,  /* unknown / missing / mis-shaped tensor (strict load,
                                    build_sam.py:103-106)                                       */
    SAMRS_ERR_CAPACITY = -6,     /* more images / prompts than the handle was created for       */
    SAMRS_ERR_PRECISION = -7,    /* multimask predict on an embedding encoded below the mode the
                                    multimask outputs need (see samrs_get_slot_info)            */
    SAMRS_ERR_RANGE = -8         /* option "range_check" = 2: an encoder pass saturated values of
                                    an MFMA operand tensor (f16: |x| >= 65504)                  */
};

/* MFMA operand type.  Accumulation, residual stream, LayerNorm / softmax statistics and the
 * whole token side of the decoder are always fp32. */
enum samrs_precision {
    SAMRS_PREC_BF16 = 0,
    SAMRS_PREC_F16 = 1
};

/* Model hyper-parameters: Generate Dataset/segment_anything/build_sam.py:14-21,37-44,55-98. */
typedef struct samrs_config {
    int32_t embed_dim;               /* 1280 (vit_h) / 1024 (vit_l) / 768 (vit_b)            */
    int32_t depth;                   /* 32 / 24 / 12                                          */
    int32_t num_heads;               /* 16 / 16 / 12                                          */
    int32_t n_global;                /* number of valid entries in global_attn_indexes        */
    int32_t global_attn_indexes[8];  /* blocks that use global attention                      */
    int32_t img_size;                /* 1024                                                  */
    int32_t patch_size;              /* 16                                                    */
    int32_t window_size;             /* 14                                                    */
    int32_t out_chans;               /* 256 (== prompt / decoder width)                       */
    int32_t max_images;              /* encoder batch == number of embedding slots            */
    int32_t max_prompts;             /* prompts (boxes) decoded per pass: workspace size      */
    int32_t max_points;              /* max points per prompt                                 */
    int32_t precision;               /* enum samrs_precision                                  */
} samrs_config;

typedef struct samrs_engine samrs_engine_t;

/* -- lifetime ------------------------------------------------------------------------------
 * replaces: sam_model_registry[...](checkpoint) + sam.to(device) + SamPredictor(sam)
 * (main_sam_hbox_semantic.py:87-89).  `device` is the HIP device ordinal.  Returns NULL on
 * failure and writes a message into err (if non-NULL). */
samrs_engine_t* samrs_create(const samrs_config* cfg, int device, char* err, int err_len);
void samrs_destroy(samrs_engine_t* e);
const char* samrs_last_error(const samrs_engine_t* e);
int samrs_abi_version(void);

/* -- weights -------------------------------------------------------------------------------
 * replaces: sam.load_state_dict(state_dict) (build_sam.py:103-106).  One call per tensor with
 * the reference's key name, fp32, contiguous, HOST memory.  The engine
 * repacks into its MFMA operand layouts.  samrs_finalize_weights() fails with
 * SAMRS_ERR_BAD_WEIGHTS if any tensor is missing (strict). */
int samrs_load_weight(samrs_engine_t* e, const char* name, const float* host_data,
                      const int64_t* shape, int ndim);
int samrs_finalize_weights(samrs_engine_t* e, void* stream);

/* -- image side ----------------------------------------------------------------------------
 * replaces: SamPredictor.set_image / set_torch_image (predictor.py:34-90) incl.
 * Sam.preprocess (modeling/sam.py:164-174) and ImageEncoderViT.forward
 * (modeling/image_encoder.py:106-116).
 * `images`: n_images contiguous uint8 HWC RGB tiles of identical size in_h x in_w, ALREADY
 * resized so that max(in_h, in_w) == img_size (ResizeLongestSide.apply_image,
 * utils/transforms.py:26-31, stays on the host).  Embeddings land in slots
 * slot0 .. slot0+n_images-1. */
int samrs_set_images(samrs_engine_t* e, const uint8_t* images, int n_images, int in_h, int in_w,
                     int slot0, void* stream);
/* Ragged batch (the per-image H[] / W[] form): images[i] is the DEVICE pointer of tile i, in_h[i] x in_w[i]
 * with max(in_h[i], in_w[i]) == img_size; the three arrays themselves are HOST arrays of n_images
 * entries.  One encoder pass for tiles of different sizes (HRSC2016 / DIOR images after
 * ResizeLongestSide.apply_image, main_sam_rbox_mask_instance.py:99-101,157); each tile's embedding is
 * bit-identical to encoding it alone.  Remember each tile's (in_h, in_w) for samrs_predict. */
int samrs_set_images_ragged(samrs_engine_t* e, const uint8_t* const* images, const int* in_h,
                            const int* in_w, int n_images, int slot0, void* stream);
/* SamPredictor.get_image_embedding (predictor.py:247-258): fp32 [out_chans, 64, 64] (NCHW). */
int samrs_get_embedding(samrs_engine_t* e, int slot, float* out_chw, void* stream);
/* Install a precomputed embedding (fp32 NCHW) into a slot; marks it set. */
int samrs_set_embedding(samrs_engine_t* e, int slot, const float* emb_chw, void* stream);
/* SamPredictor.reset_image (predictor.py:264-271). */
int samrs_reset_image(samrs_engine_t* e, int slot);
/* The operand-split mode travels with the embedding (ABI 3): *split = the "split" option in force when the slot's image was
 * encoded (-1: installed by samrs_set_embedding), *split_depth = the number of leading blocks its block-GEMM bits covered.
 * samrs_predict(multimask = 1) on a slot whose mask lacks the block-GEMM bits this model's multimask outputs need (ViT-H: 64 or
 * 16; read-only option "grade_multimask") returns SAMRS_ERR_PRECISION instead of answering in whatever mode a single-mask
 * pipeline left behind -- unless option "allow_reduced" is 1 (set by an explicit SAMRS_SPLIT, or by the caller).  The
 * reference's SamPredictor.predict defaults to multimask_output=True (predictor.py:92-100, mask_decoder.py:101-106). */
int samrs_get_slot_info(const samrs_engine_t* e, int slot, int32_t* is_set, int32_t* split, int32_t* split_depth);

/* -- prompt side ---------------------------------------------------------------------------
 * replaces: SamPredictor.predict_torch (predictor.py:168-245) = PromptEncoder.forward
 * (modeling/prompt_encoder.py:128-173) + MaskDecoder.forward (modeling/mask_decoder.py:71-174)
 * + Sam.postprocess_masks (modeling/sam.py:133-162) + `> mask_threshold` (predictor.py:242-243).
 *
 *  boxes        [n_prompts,4] fp32 xyxy in the INPUT frame (after apply_boxes_torch) or NULL
 *  point_coords [n_prompts,n_points,2] fp32 input frame, or NULL
 *  point_labels [n_prompts,n_points] int32 (1 fg, 0 bg, -1 pad), required iff point_coords
 *  mask_input   [n_prompts,1,256,256] fp32 or NULL
 *  multimask    0 -> C = 1 (mask token 0), 1 -> C = 3 (tokens 1..3) (mask_decoder.py:102-107)
 *  return_logits 0 -> masks_out is uint8 {0,1} [n_prompts,C,orig_h,orig_w];
 *                1 -> masks_out is fp32 logits of the same shape
 *  in_h,in_w    size of the image handed to samrs_set_images (predictor.input_size)
 *  orig_h,orig_w size of the original image (predictor.original_size)
 *  masks_out / iou_out [n_prompts,C] / lowres_out [n_prompts,C,256,256]: caller-owned device
 *  buffers; any of them may be NULL to skip that output.
 *  n_prompts is unbounded like the reference's batch dimension: calls larger than the handle's
 *  max_prompts run as consecutive chunks on `stream` (results do not depend on the chunking).
 *  n_points is bounded by max_points (<= 8: the decoder keeps at most 16 tokens per prompt). */
int samrs_predict(samrs_engine_t* e, int slot, int n_prompts,
                  const float* boxes, const float* point_coords, const int32_t* point_labels,
                  int n_points, const float* mask_input, int multimask, int return_logits,
                  int in_h, int in_w, int orig_h, int orig_w,
                  void* masks_out, float* iou_out, float* lowres_out, void* stream);

/* -- "next row" N1: ordered painting + areas + class statistics on device --------------------
 * replaces the host loop of main_sam_hbox_semantic.py:195-206 and the sums of
 * statistic.py:15-21.  masks: uint8 [n,orig_h,orig_w] (the C = 1 output of samrs_predict),
 * labels: int32 [n] on device.  seg_mask (uint8 [orig_h,orig_w], caller initialises to 255,
 * :162) is overwritten in box order (later box wins).  areas_out int64 [n] = pixels per mask.
 * class_pixels / class_instances int64 [n_classes] are ACCUMULATED (area > 0 only); either may
 * be NULL. */
int samrs_paint(samrs_engine_t* e, const uint8_t* masks, const int32_t* labels, int n,
                int orig_h, int orig_w, uint8_t* seg_mask, int64_t* areas_out,
                int64_t* class_pixels, int64_t* class_instances, int n_classes, void* stream);

/* -- BASELINE.json configs[3] (the rbox / rhbox instance path WITH multimask_output=True; the reference's own scripts,
 * main_sam_rhbox_mask_instance.py:168 / main_sam_rbox_mask_instance.py:164 / main_sam_hbox_mask_instance.py:165, all pass
 * multimask_output=False and need none of this): of the n_sel masks per object keep the one with the highest predicted IoU (first maximum),
 * replacing the host-side `argmax` / gather / `.sum()`.  masks uint8 [n][n_sel][h][w], iou fp32 [n][n_sel] (both outputs of
 * samrs_predict) -> best_out uint8 [n][h][w], quality_out fp32 [n], areas_out int64 [n] (pixels set). */
int samrs_select_best(samrs_engine_t* e, const uint8_t* masks, const float* iou, int n, int n_sel, int h, int w,
                      uint8_t* best_out, float* quality_out, int64_t* areas_out, void* stream);

/* -- the reference's per-instance output: COCO RLE of every full-resolution mask (main_sam_hbox_semantic.py:201-202:
 * `maskUtils.encode(np.asfortranarray(mask))` + `.decode('ascii')`; counts half stated in-tree at utils/amg.py:107-135).
 * masks: uint8 [n][h][w] on device (non-zero = set; the C = 1 output of samrs_predict).  The n ASCII strings are packed into
 * `out` (device bytes, capacity out_capacity) at 16-byte aligned offsets behind *cursor (device int64, in / out: the first
 * free byte; several calls append to one buffer), and table [n][3] (device int64) receives (offset, length, number of counts)
 * per mask; length < 0 means the string did not fit (-length - 1 bytes were needed) and was not written.  Column-major runs
 * starting with zeros, delta-coded 5-bit groups + 48, exactly as cocoapi's rleToString.  h * w < 2^30, w <= 8192.
 * The run table this entry point builds lives in ONE scratch buffer per handle: all samrs_rle_encode calls on a handle must be
 * stream-ordered with each other (same stream, or an event between them); two handles never share it. */
int samrs_rle_encode(samrs_engine_t* e, const uint8_t* masks, int n, int h, int w, uint8_t* out, int64_t out_capacity,
                     int64_t* cursor, int64_t* table, void* stream);

/* -- "next row" N3: one separable pass of Pillow's 8-bit resample (ResizeLongestSide.apply_image,
 * utils/transforms.py:26-31 -> PIL Image.resize BILINEAR).  `bounds` int32 [out_len,2] = (first input
 * index, tap count), `coef` int32 [out_len,ksize] 22-bit fixed point, both computed on the host exactly
 * as Pillow does (samrs_amd/transforms.py).  horizontal: in [other,in_len,3] -> out [other,out_len,3];
 * vertical: in [in_len,other,3] -> out [out_len,other,3].  Bit-exact with PIL. */
int samrs_resample_pass_u8(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* coef,
                           int ksize, int in_len, int out_len, int other, int horizontal, void* stream);

/* -- per-engine options (each handle has its own; the environment variable named in brackets only sets the value a NEW handle
 * starts with).  Returns SAMRS_ERR_BAD_ARG for an unknown name.
 *   "split"          [SAMRS_SPLIT; default 79 where the one-launch split GEMM covers the block shapes (ViT-H), else 15] bit mask of
 *                    the rounding points that run as a two-term operand split (hi + lo, three MFMAs, ~2^-22 operand error
 *                    instead of 2^-11): 1 = patch embed, 2 = neck, 4 = decoder image->token out-projection, 8 = decoder upscaler
 *                    (both transposed convs).  15 = every block GEMM at the 1x f16 rate: enough for IoU >= 0.9995 on the single
 *                    mask of token 0 (what the hbox-semantic path asks for); 0 = the round-2 engine's arithmetic.
 *                    Block-GEMM bits: 64 = the v third of the qkv product + proj (q and k pass through the softmax), 16 = all of
 *                    qkv + proj, 32 = the MLP GEMMs (63 / 127 = every MFMA operand of the path), 128 = lin2 alone of the MLP GEMMs
 *                    ("lo_format" 4 only; 207 = 79 | 128 is the error budget's cheapest mode with more margin on the multimask
 *                    outputs: oracle/error_budget.py plans10).  With "lo_format" 4 a split GEMM costs 1.25 - 1.8x a plain one (f16 lo
 *                    terms: 3x).  79 = 15 | 64 holds IoU >= 0.999 on the three multimask tokens too (0.9992 on 192 masks) at 0.92x
 *                    the throughput of 15 and is what a ViT-H handle starts with.  The block-GEMM bits need lo copies of the
 *                    block weights: set them BEFORE samrs_finalize_weights; they can be cleared and set again afterwards.
 *                    What each bit buys in mask pixels: oracle/error_budget.py, DESIGN.md 2.
 *   "split_depth"    [SAMRS_SPLIT_DEPTH, default 0 = automatic] the block-GEMM bits apply to the first N encoder blocks only: an
 *                    operand error made in an early block is carried through every later one, one made in the last blocks is
 *                    not.  Automatic: every block for 16 / 32, the leading three quarters (24 of 32) for 64.
 *   "decoder_fusion" [SAMRS_DECODER_FUSION, default 1] 0 = run the decoder with its un-fused kernels (separate GEMM / LayerNorm /
 *                    product launches; never split): the fused-vs-unfused parity test and timing experiments.
 *   "ln_fold"        [SAMRS_LN_FOLD, default 0] (EXPERIMENTS builds only; embed_dim 1280; only while no block-GEMM bit of "split" is set) 1 = fold the encoder blocks' LayerNorms into the qkv / lin1
 *                    GEMMs.  Measured slower on MI355X.  Must be on before samrs_finalize_weights for the folded weights to
 *                    exist; can be flipped afterwards.
 *   "gemm_variant"   [default -1 = automatic] GEMM tile variant for this handle's launches (tools/gemm_bench.py lists them).
 *   "upscaler_fused" [SAMRS_UPSCALER_FUSED, default 1] 1 = the mask upscaler as ONE kernel (samrs_k_upscaler_fused); 0 = ConvT #1 as a GEMM with a
 *                    LayerNorm2d + GELU epilogue, then the ConvT #2 + GELU + product kernel (A/B runs, tests).
 *   "split_passes"   [SAMRS_SPLIT_PASSES, default 0] reference-grade bits of "split" only: 1 = the three terms of a split block GEMM as
 *                    three accumulating launches through an fp32 scratch (the generic route, every shape); 0 = as ONE launch over a
 *                    three-segment K axis where the shape fits the 256 x 320 tile (ViT-H; samrs_k_gemm_split3), else the generic route.
 *   "lo_format"      [SAMRS_LO_FORMAT; default 4 where the MX kernel covers the block shapes (ViT-H), else 0] operand format of the two
 *                    correction terms of the block-GEMM split (bits 64 / 16 and 32): 0 = f16 (a split GEMM is three f16
 *                    passes), 4 = MXFP4 (e2m1 + E8M0 scale per 32 k on gfx950's block-scaled MFMA: the corrections run at 4x the f16
 *                    rate on a quarter of the stages, a split GEMM costs 1.5 passes; oracle/error_budget.py plans10 for what the
 *                    format costs in mask pixels: nothing measurable).  The fp4 weight copies are made at samrs_finalize_weights
 *                    when a block-GEMM bit is set by then; afterwards the option can be flipped between 0 and 4 (A/B runs).
 *   "allow_reduced"  [SAMRS_ALLOW_REDUCED; default 0, 1 when SAMRS_SPLIT is set] 1 = samrs_predict(multimask = 1) accepts embeddings
 *                    encoded below the multimask grade (see samrs_get_slot_info) instead of returning SAMRS_ERR_PRECISION.
 *   "gelu_fast"      [SAMRS_GELU_FAST, default -1 = automatic] erf of the GELU in lin1's epilogue (image_encoder.py:177-180, common.py:18-26;
 *                    168 M elements per launch at ViT-H): 0 = Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: fp32-epsilon class, two
 *                    transcendentals per element), 1 = 7.1.28 (|error| <= 3e-7, one v_rcp_f32 and no exp2: -3.3 % on the dominant
 *                    kernel; both are three orders of magnitude under the f16 rounding the value receives next).  Automatic: 1 in the
 *                    1x-rate modes ("split" without a block-GEMM bit: what the single-mask pipelines run), 0 in every reference-grade
 *                    mode, whose arithmetic stays bit for bit what its committed parity statistics were measured on.  Applies where
 *                    lin1 runs on the persistent 256 x 320 kernel (ViT-H shapes); elsewhere the option has no effect.
 *   "ln_tail"        [SAMRS_LN_TAIL, default 0] 1 = the LayerNorm that follows proj (norm2, image_encoder.py:177) and lin2 (norm1 of the
 *                    next block, :168) runs as a TAIL of those GEMM launches instead of a launch of its own: the block that stores the
 *                    last of the four 256 x 320 tiles of a 256-row panel normalises the panel's rows (release / counter / acquire between
 *                    the four blocks, nobody waits: gemm.hip LnTail).  Bit-identical with the stand-alone LayerNorm, and measured SLOWER
 *                    (61.8 against 57.2 ms per 8-tile step): one CU draws ~20 GB/s from HBM, so a panel takes it 62 us where a LayerNorm
 *                    launch on all 256 CUs takes 53 for the whole tensor (profiles/r05_ln_tail.txt).  Kept as an option for the record
 *                    and for its test; 1x-rate modes only, batches of 4+ tiles at ViT-H.
 *   "operand_pad"    [SAMRS_OPERAND_PAD, default 1; read at samrs_finalize_weights for the copies, switchable afterwards] ViT-H: the K = 1280
 *                    operands of the plain qkv / lin1 launches -- the LayerNorm output and the weights -- are stored with a row stride of 1408
 *                    elements (2816 B = eleven 256-byte units) instead of 1280 (ten units: the 256 rows a tile fetches per k-slice then fall
 *                    on half of the memory channels); the persistent ET kernels read A and B with that stride.  Costs 0.8 GB for the padded
 *                    weight copies.  Only where the bytes live changes: bit-identical output in every mode.
 *   "range_check"    [SAMRS_RANGE_CHECK, default 0] the f16 operand type has 11 mantissa bits (what the IoU >= 0.999 bar needs) but
 *                    tops out at 65504, and every conversion on the path SATURATES there instead of overflowing to inf -- silently.
 *                    1 = after each producer of an MFMA-operand tensor in the encoder (both LayerNorm outputs, q | k | v, the
 *                    attention output, GELU(lin1), the residual stream in front of the neck, the neck's LayerNorm2d output, the
 *                    decoder's layer-0 keys) a scan adds the number of elements at the saturation value (or inf / nan) to a
 *                    per-engine counter, read -- and, by writing any value, reset -- through the option "saturated"; ~7 extra
 *                    passes over those tensors per block (+ 25 % encoder time): a validation mode for a NEW CHECKPOINT, not the
 *                    production setting.  2 = the same, and samrs_set_images* returns SAMRS_ERR_RANGE when its pass saturated
 *                    anything (one stream synchronisation per pass).  The remedy is the bf16 operand type (fp32 exponent range;
 *                    misses the IoU bar by its 8 mantissa bits: DESIGN.md 2) -- there is no silent fallback.
 *   "saturated"      read: the counter of "range_check" (clamped to INT_MAX; synchronizes the device); write: reset.
 *   "outlier_cols"   [SAMRS_OUTLIER_COLS, default 7 = bit 0 (qkv / lin1) | bit 1 (lin2) | bit 2 (proj); non-zero at samrs_finalize_weights
 *                    for the picking, switchable afterwards] checkpoint
 *                    weights have outlier channels (a few LayerNorm gammas 10 - 100x the rest, hidden units / v channels that run
 *                    thousands of times hotter), and an f16 operand's rounding error is relative to ITS magnitude: those few K-columns
 *                    carry most of the operand error of a block GEMM.  At load time every block GEMM's columns are scored from the
 *                    fp32 weights alone (operand magnitude proxy x weight column norm); those above "outlier_ratio_pct" of the
 *                    median score (at most 32 per GEMM) are its outlier columns.  For the plain qkv / lin1 launches their hi + lo
 *                    split rides as 64 more K columns of the SAME launch (the LayerNorm writes the operand side, the weights carry
 *                    the matching columns): + 1 / 20 of those two GEMMs in the blocks that have such columns, nothing elsewhere.
 *                    lin2 / proj read operands that other kernels write, so their 64 columns travel as a dense side operand
 *                    ([M][64]: the attention kernel's lo output gathered for proj; for lin2 the outlier hidden units recomputed
 *                    in fp32 by a 128-column side GEMM, exact GELU, split) and enter as one more K stage of the same launch.
 *                    Weights without outliers (seeded-normal test models) pick nothing: bit-identical output, zero cost.
 *   "outlier_ratio_pct" [SAMRS_OUTLIER_RATIO_PCT, default 400; before samrs_finalize_weights only]
 *   "outlier_blocks" / "outlier_columns" / "outlier_dominant_blocks"  read-only: encoder blocks with at least one outlier column in
 *                    qkv / lin1; the number of columns picked over all four block GEMMs; blocks in which the picked columns carry more
 *                    than half of the qkv or proj operand-error mass -- in the v-third modes (79 / 207) those blocks run the plain
 *                    launches with the exact lo terms of their outlier columns instead of the MXFP4 lo terms of all columns. */
int samrs_set_option(samrs_engine_t* e, const char* name, int value);
int samrs_get_option(const samrs_engine_t* e, const char* name, int* value);

/* Rotated-box MASK prompts, replacing the cv2 pre-step of `Generate Dataset/main_sam_rbox_mask_instance.py:125-141`
 * (fillPoly -> +-1000 -> resize to the ResizeLongestSide shape -> pad with -1000 -> resize to 256x256).
 * pts: device int32 [n][n_vertices][2] (x, y) in ORIGINAL image pixels (the reference's `.astype(np.int32)`),
 * 3 <= n_vertices <= 8; (h, w) original size; (th, tw) = ResizeLongestSide.get_preprocess_shape(h, w, img_size);
 * out: device fp32 [n][out_size][out_size] = the `mask_input` of samrs_predict (out_size = 256). */
int samrs_rbox_mask_prompt(const int32_t* pts, int n, int n_vertices, int h, int w, int th, int tw,
                           int img_size, int out_size, float* out, void* stream);
/* The same with the polygon fill rule named.  cv2.fillPoly's scanline spans changed in OpenCV 4.5.2 (modules/imgproc/src/drawing.cpp
 * FillEdgeCollection): up to 4.5.1 a span covers ceil(x_left) .. floor(x_right) of the 16.16 fixed-point edge crossings, since 4.5.2
 * both ends are rounded half up; the 8-connected boundary lines drawn on top are the same.  The reference pins no OpenCV version
 * (`Generate Dataset/main_sam_rbox_mask_instance.py:126-129`), and on FAIR1M-shaped boxes ~18 % of the polygons differ by 1 - 8
 * boundary pixels between the two, so the caller says which cv2 it replaces (samrs_amd.transforms picks by cv2.__version__ when
 * cv2 is importable).  samrs_rbox_mask_prompt == fill_rule SAMRS_FILL_CV2_LE_451. */
enum samrs_fill_rule { SAMRS_FILL_CV2_LE_451 = 0, SAMRS_FILL_CV2_GE_452 = 1 };
int samrs_rbox_mask_prompt_rule(const int32_t* pts, int n, int n_vertices, int h, int w, int th, int tw,
                                int img_size, int out_size, int fill_rule, float* out, void* stream);

/* The kernel-level entry points (samrs_k_*) and the test / measurement hooks (samrs_debug_*) that the parity tests, bench.py's
 * in-situ kernel timer and the tools/ scripts use are declared in samrs_hip_internal.h: exported by the same library, NOT part of
 * the drop-in boundary, and without any compatibility promise. */

#ifdef __cplusplus
}
#endif
#endif /* SAMRS_HIP_H */
