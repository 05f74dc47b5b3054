// engine.hip -- the C ABI of libsamrs_hip.so (include/samrs_hip.h): engine handle, strict weight
// loading + repacking, the encoder / decoder launch sequences.
//
// Launch sequences follow the reference graph, restated for this kernel set:
//   set_images : modeling/sam.py:164-174 -> modeling/image_encoder.py:106-116,166-182,88-104
//   predict    : modeling/prompt_encoder.py:128-173 -> modeling/mask_decoder.py:71-174 with
//                modeling/transformer.py:62-106,151-182 -> modeling/sam.py:133-162
// (paths under Generate Dataset/segment_anything/).  Image-side work shared by all prompts of a
// call (layer-0 key/value/query projections when there is no mask prompt) is computed once.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/samrs_hip.h"
#include "../../include/samrs_hip_internal.h"
#include "common.h"
#include "kernels.h"

namespace {

struct DevTensor {
    float* p = nullptr;           // fp32 on device
    std::vector<int64_t> shape;
    size_t numel = 0;
};

// Per-engine options (samrs_set_option); the environment only supplies the DEFAULTS a new handle starts with:
//   decoder_fusion (SAMRS_DECODER_FUSION, default 1): 0 = run the decoder with its un-fused kernels (separate GEMM / LayerNorm /
//                  product launches) -- the fused-vs-unfused parity test and timing experiments;
//   ln_fold        (SAMRS_LN_FOLD, default 0): fold the encoder blocks' LayerNorms into the qkv / lin1 GEMMs (embed_dim 1280);
//   split          (SAMRS_SPLIT, default 15): bit mask of the rounding points that run as a two-term operand split (3 MFMAs,
//                  ~2^-22 operand error): 1 patch embed, 2 neck, 4 decoder i2t out-projection, 8 decoder upscaler (both
//                  transposed convs).  oracle/error_budget.py measures what each bit buys; DESIGN.md 2 has the table.
//                  16 = the blocks' qkv + proj GEMMs, 32 = the blocks' MLP GEMMs: the REFERENCE-GRADE bits -- three times the MFMA
//                  work of the GEMMs they cover, not part of the default; they need their lo weights, i.e. must be set before
//                  samrs_finalize_weights (SAMRS_SPLIT=63 or the option), and can be cleared / set again afterwards.
// SPLIT_ATTN_V: the attention-side split restricted to the v third of qkv (+ proj): q and k pass through the softmax and buy
// next to nothing (error_budget.py plans4 / plans6); SPLIT_ATTN set as well = all of qkv.  One-launch route only (ViT-H shapes).
// SPLIT_LIN2 (round 4; needs lo_format 4): lin2 alone of the MLP GEMMs takes the lo terms -- the error budget's cheapest way to more margin
// on the multimask outputs (error_budget.py plans10); lin1 then runs on the MX kernel only to have its epilogue emit H's fp4 rows.
enum { SPLIT_PATCH = 1, SPLIT_NECK = 2, SPLIT_OI = 4, SPLIT_UP = 8, SPLIT_DEFAULT = 15, SPLIT_ATTN = 16, SPLIT_MLP = 32, SPLIT_ATTN_V = 64,
       SPLIT_LIN2 = 128, SPLIT_ATTN_ANY = SPLIT_ATTN | SPLIT_ATTN_V, SPLIT_ALL = 255 };
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

struct DecAttn {
    const float *qw, *qb, *kw, *kb, *vw, *vb, *ow, *ob;
};

struct DecLayer {
    DecAttn self, t2i, i2t;
    const float *n1w, *n1b, *n2w, *n2b, *n3w, *n3b, *n4w, *n4b;
    const float *m1w, *m1b, *m2w, *m2b;
    uint16_t* kvq_w = nullptr;    // ET [384][256] = [Wk_t2i; Wv_t2i; Wq_i2t]
    float* kvq_b = nullptr;       // [384]
    float* kvq_pe = nullptr;      // [tokens][384] = [PE Wk^T | 0 | PE Wq^T]
    uint16_t* i2t_ow = nullptr;   // ET [256][128]
    uint16_t* i2t_ow_lo = nullptr;   // its split remainder
};

struct EncBlock {
    bool global = false;
    const float *ln1w, *ln1b, *ln2w, *ln2b, *qkv_b, *proj_b, *lin1_b, *lin2_b, *rel_h, *rel_w;
    uint16_t *qkv_w = nullptr, *proj_w = nullptr, *lin1_w = nullptr, *lin2_w = nullptr;
    uint16_t *qkv_w_lo = nullptr, *proj_w_lo = nullptr, *lin1_w_lo = nullptr, *lin2_w_lo = nullptr;   // reference-grade bits only
    // copies of qkv_w / lin1_w with a row stride of ldk elements instead of K = D (engine field ldk: operands off the 2560-byte stride)
    uint16_t *qkv_wp = nullptr, *lin1_wp = nullptr;
    // option "lo_format" = 4: the lo terms of the attention-side split on MXFP4 operands (gemm.hip gemm_et_mx_kernel): fp4 codes of
    // hi and lo of the weights + their scale tiles (B layout); proj's K axis padded per head (80 -> 96) so that no MX block
    // straddles two heads
    unsigned char *qkv_w4[2] = {nullptr, nullptr}, *qkv_s4[2] = {nullptr, nullptr};       // [0] = hi, [1] = lo
    unsigned char *proj_w4[2] = {nullptr, nullptr}, *proj_s4[2] = {nullptr, nullptr};
    // ... and of the MLP weights (bit 32): lin1 plain, lin2 on the K axis that lin1's epilogue writes its MX rows on (80 -> 96 per wave tile)
    unsigned char *lin1_w4[2] = {nullptr, nullptr}, *lin1_s4[2] = {nullptr, nullptr};
    unsigned char *lin2_w4[2] = {nullptr, nullptr}, *lin2_s4[2] = {nullptr, nullptr};
    // Outlier columns (option "outlier_cols"; oracle/outlier_budget.py): per block GEMM ([0] qkv, [1] lin1, [2] lin2, [3] proj) the K-columns
    // whose operand magnitude x weight column norm stands out (> ratio x the median), at most 32, picked from the fp32 weights at load
    // time.  For qkv / lin1 their hi + lo split rides as 64 extra K columns of the same launch: the LayerNorm writes the operand side
    // (encoder_kernels.hip), qkv_wx / lin1_wx are dense [N][D + 64] copies of the weights with the weight side appended (on the
    // padded-stride route the same 64 columns sit in the pad region of qkv_wp / lin1_wp instead).
    int oc_n[4] = {0, 0, 0, 0};
    int* oc_idx[4] = {nullptr, nullptr, nullptr, nullptr};
    // share of the picked columns in the GEMM's squared-score mass (sum over S of score^2 / sum over all columns): > 1/2 = the operand error
    // of this GEMM is dominated by its outlier columns.  Decides between the exact f16 lo terms of those columns and the MXFP4 lo terms of
    // ALL columns in the v-third modes (run_encoder: oc_dominant)
    float oc_share[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t oc_heads = 0;         // bit h: attention head h holds an outlier column of proj (the only heads whose output remainder is needed)
    uint16_t *qkv_wx = nullptr, *lin1_wx = nullptr;
    // lin2 / proj: their A operands (GELU(lin1), the attention output) are written by other kernels, so the 64 columns travel as a
    // dense side operand A_x [M][64] (engine OCX) against oc_bx [D][64] = W_hi[:, S] | W_lo[:, S], one more K stage of the same
    // launch (gemm.hip EXT).  lin2's A_x needs GELU(lin1) of the outlier hidden units BEFORE its rounding: a side GEMM of the
    // LayerNorm output against those <= 32 rows of lin1's weight (lin2_ws, lin2_sb the matching bias) with the exact GELU and the split in
    // its epilogue (encoder_kernels.hip outlier_side_gemm_kernel).
    uint16_t* oc_bx[4] = {nullptr, nullptr, nullptr, nullptr};      // [2] lin2, [3] proj
    uint16_t* lin2_ws = nullptr;       // [32][D (+ 64 when lin1 carries outlier columns of its own)]
    float* lin2_sb = nullptr;          // [32]
    // LayerNorm folded into qkv / lin1 (ViT-H): W diag(gamma) in ET, its row sums, b + W beta
    uint16_t *qkv_wf = nullptr, *lin1_wf = nullptr;
    float *qkv_c = nullptr, *qkv_bf = nullptr, *lin1_c = nullptr, *lin1_bf = nullptr;
};

// LayerNorm folding (encoder blocks, embed_dim 1280 only).  OFF by default: measured slower than the stand-alone LayerNorm
// kernel on MI355X (anything added to a GEMM epilogue runs while the matrix pipe idles, the stand-alone kernel
// streams at ~6 TB/s).  SAMRS_LN_FOLD=1 at load time or samrs_set_option(e, "ln_fold", 1) BEFORE samrs_finalize_weights prepares
// the folded weights; the switch can then be flipped at run time (A/B runs, the folded-vs-unfolded parity test).

}  // namespace

struct samrs_engine {
    samrs_config cfg{};
    int device = 0;
    int prec = 0;
    bool finalized = false;
    std::string err;
    std::map<std::string, DevTensor> w;        // fp32 device copies keyed by reference name
    std::vector<void*> owned;                   // everything hipMalloc'ed by the engine

    // derived sizes
    int grid = 64, tokens = 4096, D = 0, C = 256, hd = 0, nwin = 5;
    int T_max = 0;
    bool decoder_fusion = true, ln_fold = false;   // per-engine options, see the top of this file
    int split = SPLIT_DEFAULT;
    int split_ready = SPLIT_DEFAULT;                // bits whose lo weights / workspaces exist (fixed at samrs_finalize_weights)
    int gemm_variant = -1;                          // -1 = the library default (launch_gemm_et's automatic choice)
    int split_depth = 0;                            // reference-grade bits apply to the first N blocks (0 = all)
    bool split_passes = false;                      // reference-grade block GEMMs as three accumulating launches instead of one (A/B)
    int lo_format = 0;                              // 0: lo terms on f16 operands (three-segment f16 GEMM); 4: on MXFP4 operands
    bool mx_ready = false;                          // the fp4 weight copies + activation workspaces exist (fixed at samrs_finalize_weights)
    int mx_gp = 0, mx_kp_proj = 0;                  // proj's padded K axis: heads x mx_gp (head_dim rounded up to 32)
    bool mx_mlp_ready = false;                      // the same for the MLP GEMMs (bit 32 set at samrs_finalize_weights)
    int mx_kp_lin2 = 0;                             // lin2's padded K axis: 4 D / 80 x 96
    unsigned char *H4[2] = {nullptr, nullptr}, *SH4[2] = {nullptr, nullptr};       // GELU(lin1) as fp4 hi / lo, written by lin1's epilogue
    unsigned char *Y4[2] = {nullptr, nullptr}, *SY4[2] = {nullptr, nullptr};       // LN output as fp4 hi / lo + scale tiles (A layout)
    unsigned char *AO4[2] = {nullptr, nullptr}, *SAO4[2] = {nullptr, nullptr};     // attention output likewise (padded K axis)
    bool upscaler_fused = true;                     // one-kernel upscaler (upscaler_fused.hip) instead of ConvT1 GEMM + ConvT2 kernel

    // encoder weights / workspaces
    std::vector<EncBlock> blocks;
    uint16_t *patch_w = nullptr, *neck0_w = nullptr, *neck2_w = nullptr;
    uint16_t *patch_w_lo = nullptr, *neck0_w_lo = nullptr, *neck2_w_lo = nullptr;   // split remainders (common.h split2_pack)
    float* X = nullptr;            // residual stream fp32 [Bi*tokens, D]
    uint16_t* Y = nullptr;         // LN out (ET) [Bi*tokens, D]; folded path: the residual stream itself rounded to ET
    float* STATS = nullptr;        // folded path: per-row (mean, M2) of eight 160-column groups [Bi*tokens][8][2]
    float* ROWSTAT = nullptr;      // folded path: per-row (rstd, -rstd mean) [Bi*tokens][2]
    bool can_fold = false;         // embed_dim == 1280 and the folded weights exist
    uint16_t* QKV = nullptr;       // [Bi*tokens, 3D], token order
    uint16_t* AO = nullptr;        // attention out [Bi*tokens, D]
    uint16_t *Ylo = nullptr, *AOlo = nullptr, *Hlo = nullptr;   // reference-grade split: remainders of Y, AO, H
    float* F32T = nullptr;         // reference-grade split: fp32 result of a three-pass qkv / lin1 product [Bi*tokens, 4D]
    uint16_t* VTG = nullptr;       // V of a global-attention block transposed per head: [Bi][heads][hd][tokens]
    uint16_t* H = nullptr;         // MLP hidden [Bi*tokens, 4D]  (also patch im2col / neck im2col)
    float* N1 = nullptr;           // neck fp32 [Bi*tokens, C]
    uint16_t* N1e = nullptr;       // [Bi*tokens, C]
    float* EMB = nullptr;          // [slots][tokens][C] fp32 (token-major)
    std::vector<char> slot_set;
    // Precision is a property of the EMBEDDING: the "split" mask (and the depth its block-GEMM bits reached) a slot's image was
    // encoded with; -1 = installed by samrs_set_embedding (the caller's numbers, nothing to say about them).  samrs_predict
    // checks it against what the requested outputs need (grade_multimask) instead of trusting whoever touched "split" last.
    std::vector<int> slot_split, slot_depth;
    int grade_multimask = 0;       // block-GEMM bits (any of them) the three multimask tokens need on this model; 0 = none
    bool allow_reduced = false;    // option "allow_reduced": multimask predicts on a slot encoded below that grade are the caller's choice
    // option "range_check" (0 off, 1 count, 2 count and fail): after every producer of an MFMA-operand tensor in the encoder a
    // scan counts the elements sitting at the operand type's saturation value (f16: +-65504, what common.h's saturating
    // conversions write) or beyond into *range_counter (device); read through option "saturated"
    int ln_tail = 0;               // option "ln_tail": 1 = the LayerNorm behind proj / lin2 as a tail of those launches (measured slower: off)
    unsigned int* ln_counters = nullptr;   // per 256-row panel: tiles of the running proj / lin2 launch that have stored (gemm.hip LnTail)
    // option "operand_pad" (default 1): the K = D operands of the plain qkv / lin1 launches -- the LayerNorm output and the weights -- are
    // stored with a row stride of ldk = D + 128 elements where D rows are an even number of 256-byte units (ViT-H: 2560 B -> 2816 B), so
    // that the rows a tile fetches per k-slice spread over all memory channels instead of half of them (gemm.hip tl_gemm_ld)
    int operand_pad_on = 1;
    int ldk = 0;                   // 0: no padded copies exist (other widths)
    // option "outlier_cols" (default 7; SAMRS_OUTLIER_COLS; bit 0: qkv / lin1, bit 1: lin2, bit 2: proj): hi + lo terms for the outlier
    // K-columns of the plain block-GEMM launches (EncBlock::oc_*).  Columns are picked in samrs_finalize_weights (the option must be on by then); later it switches their use.
    // "outlier_ratio_pct" (default 400): a column is an outlier when its score exceeds this percentage of its GEMM's median score.
    // Read-only: "outlier_blocks" (blocks with at least one such column in qkv / lin1), "outlier_columns" (their total over the
    // four block GEMMs).  Weights without outliers (every seeded-normal test model) pick nothing: bit-identical, zero cost.
    int outlier_on = 7 /* bit 0: qkv / lin1, bit 1: lin2, bit 2: proj */, outlier_ratio_pct = 400, outlier_blocks = 0, outlier_columns = 0;
    int outlier_dominant_blocks = 0;   // blocks whose qkv or proj operand error is dominated by outlier columns (EncBlock::oc_share > 1/2)
    float* oc_scratch = nullptr;   // load-time scratch: column / row norms
    uint16_t* OCX = nullptr;       // [M][64]: side operand A_x of the running proj / lin2 launch
    bool oc_resid = false;         // some block has outlier columns in lin2 / proj
    int gelu_fast = -1;            // option "gelu_fast": -1 automatic (on in the 1x-rate modes: no block-GEMM bit in "split"), 0 off, 1 on
    int range_check = 0;
    unsigned long long* range_counter = nullptr;
    unsigned long long range_seen = 0;             // counter value at the end of the last checked encoder pass (mode 2)

    // decoder weights
    std::vector<DecLayer> layers;
    DecAttn fin{};
    uint16_t* fin_kv_w = nullptr;  // ET [256][256] = [Wk; Wv]
    float *fin_kv_b = nullptr, *fin_pe = nullptr;
    uint16_t *up1_w = nullptr, *up2_w = nullptr, *up1_w_lo = nullptr, *up2_w_lo = nullptr;
    float *up1_b = nullptr, *up2_b = nullptr, *up_ln = nullptr;   // up_ln = LayerNorm2d gamma[64] | beta[64]
    float* PE = nullptr;           // dense PE [tokens][C]

    // decoder workspaces
    float *TOK0 = nullptr, *Q = nullptr, *TA = nullptr, *TQ = nullptr, *TK = nullptr, *TV = nullptr, *TO = nullptr;
    float *MH = nullptr, *QP = nullptr, *KT = nullptr, *VT = nullptr, *O128 = nullptr, *T2IW = nullptr;
    // Per embedding slot, written when the slot's image is set (prepare_slot_keys) and read-only for every predict on it: the
    // layer-0 image side of the two-way transformer without a mask prompt is the same for every box of an image (keys =
    // embedding + no_mask_embed, their k / v / q projections), so a second predict call / box chunk on the image costs nothing here
    float *K0F = nullptr;          // layer-0 keys fp32 [max_images][tokens][C]
    uint16_t* K0E = nullptr;       // ... in the operand type
    uint16_t* KVQ0 = nullptr;      // their K_t2i | V_t2i | Q_i2t projections [max_images][tokens][3 C / 2]
    float* KF = nullptr;           // per-prompt keys fp32 [Bb*tokens][C]
    uint16_t* KE = nullptr;
    uint16_t* KE_lo = nullptr;     // split remainder of the final keys (operand of the first transposed conv)
    float* DENSE = nullptr;        // mask-prompt dense embedding (allocated on first use)
    uint16_t* KVQ = nullptr;       // [Bb*tokens][384]
    uint16_t* OI = nullptr;        // [Bb*tokens][128]
    float* U1raw = nullptr;        // [Bb*tokens][256]
    uint16_t* U1 = nullptr;        // [Bb*tokens][256]
    uint16_t* U2 = nullptr;        // [Bb*tokens*4][128]
    float *HY1 = nullptr, *HY2 = nullptr, *HYPER = nullptr, *IOU = nullptr, *LOW = nullptr;

    // COCO RLE scratch (samrs_rle_encode), grown on demand
    void* rle_scratch = nullptr;
    size_t rle_scratch_bytes = 0;

    // optional in-situ timing of the dominant kernel (MLP lin1 + GELU GEMM) with HIP events
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;   // recorded pairs
    std::vector<hipEvent_t> tpool;                         // recycled events
};

namespace {

int fail(samrs_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}

#define CK(e, expr)                                                                                    \
    do {                                                                                               \
        hipError_t _err = (expr);                                                                      \
        if (_err != hipSuccess)                                                                        \
            return fail((e), SAMRS_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_err)); \
    } while (0)

template <typename T>
hipError_t dalloc(samrs_engine* e, T** p, size_t count) {
    void* q = nullptr;
    hipError_t r = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
    if (r != hipSuccess) return r;
    e->owned.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return hipSuccess;
}

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Every entry point that touches the device runs on the engine's device and then puts the CALLER's current device
// back (a process that drives several GPUs, or a handle collected at an arbitrary time, must not find its thread's
// device changed under it).
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) status = hipSetDevice(dev);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
// an engine's own GEMM tile choice (samrs_set_option "gemm_variant") applies to the launches of its entry points only
struct GemmVariantScope {
    int prev;
    explicit GemmVariantScope(int v) : prev(swap_gemm_variant_override(v)) {}
    ~GemmVariantScope() { (void)swap_gemm_variant_override(prev); }
};
#define ON_DEVICE(e) DeviceGuard _dg((e)->device); CK((e), _dg.status); GemmVariantScope _gvs((e)->gemm_variant)

// X (the fp32 residual stream) += A W^T + A_x B_x^T + bias -- proj / lin2 with the hi + lo terms of their outlier columns (EncBlock::oc_bx):
// one launch with one more K stage where the 256 x 320 pair-stage kernel takes the shape (gemm.hip EXT), else the plain launch followed
// by an accumulating launch of the 64-column side product on the 128 x 128 kernel (any shape; small models only: it re-reads X)
hipError_t resid_gemm_ext(samrs_engine* e, int prec, const void* A, const void* Wt, const void* Ax, const void* Bx, const float* bias,
                          int M, int N, int K, hipStream_t s);

bool is_global(const samrs_config& c, int i) {
    for (int k = 0; k < c.n_global; ++k)
        if (c.global_attn_indexes[k] == i) return true;
    return false;
}

// ---- the strict name/shape contract -----------------------------------
void required_tensors(const samrs_engine* e, std::vector<std::pair<std::string, std::vector<int64_t>>>& out) {
    const samrs_config& c = e->cfg;
    const int64_t D = c.embed_dim, g = e->grid, hd = e->hd, C = c.out_chans, P = c.patch_size;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { out.emplace_back(n, std::move(s)); };
    add("image_encoder.pos_embed", {1, g, g, D});
    add("image_encoder.patch_embed.proj.weight", {D, 3, P, P});
    add("image_encoder.patch_embed.proj.bias", {D});
    for (int i = 0; i < c.depth; ++i) {
        const std::string p = "image_encoder.blocks." + std::to_string(i);
        const int64_t s = is_global(c, i) ? g : c.window_size;
        add(p + ".norm1.weight", {D}); add(p + ".norm1.bias", {D});
        add(p + ".attn.rel_pos_h", {2 * s - 1, hd}); add(p + ".attn.rel_pos_w", {2 * s - 1, hd});
        add(p + ".attn.qkv.weight", {3 * D, D}); add(p + ".attn.qkv.bias", {3 * D});
        add(p + ".attn.proj.weight", {D, D}); add(p + ".attn.proj.bias", {D});
        add(p + ".norm2.weight", {D}); add(p + ".norm2.bias", {D});
        add(p + ".mlp.lin1.weight", {4 * D, D}); add(p + ".mlp.lin1.bias", {4 * D});
        add(p + ".mlp.lin2.weight", {D, 4 * D}); add(p + ".mlp.lin2.bias", {D});
    }
    add("image_encoder.neck.0.weight", {C, D, 1, 1});
    add("image_encoder.neck.1.weight", {C}); add("image_encoder.neck.1.bias", {C});
    add("image_encoder.neck.2.weight", {C, C, 3, 3});
    add("image_encoder.neck.3.weight", {C}); add("image_encoder.neck.3.bias", {C});
    add("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", {2, C / 2});
    for (int i = 0; i < 4; ++i) add("prompt_encoder.point_embeddings." + std::to_string(i) + ".weight", {1, C});
    add("prompt_encoder.not_a_point_embed.weight", {1, C});
    add("prompt_encoder.no_mask_embed.weight", {1, C});
    add("prompt_encoder.mask_downscaling.0.weight", {4, 1, 2, 2}); add("prompt_encoder.mask_downscaling.0.bias", {4});
    add("prompt_encoder.mask_downscaling.1.weight", {4}); add("prompt_encoder.mask_downscaling.1.bias", {4});
    add("prompt_encoder.mask_downscaling.3.weight", {16, 4, 2, 2}); add("prompt_encoder.mask_downscaling.3.bias", {16});
    add("prompt_encoder.mask_downscaling.4.weight", {16}); add("prompt_encoder.mask_downscaling.4.bias", {16});
    add("prompt_encoder.mask_downscaling.6.weight", {C, 16, 1, 1}); add("prompt_encoder.mask_downscaling.6.bias", {C});
    auto attn = [&](const std::string& p, int64_t internal) {
        add(p + ".q_proj.weight", {internal, C}); add(p + ".q_proj.bias", {internal});
        add(p + ".k_proj.weight", {internal, C}); add(p + ".k_proj.bias", {internal});
        add(p + ".v_proj.weight", {internal, C}); add(p + ".v_proj.bias", {internal});
        add(p + ".out_proj.weight", {C, internal}); add(p + ".out_proj.bias", {C});
    };
    for (int i = 0; i < 2; ++i) {
        const std::string p = "mask_decoder.transformer.layers." + std::to_string(i);
        attn(p + ".self_attn", C);
        attn(p + ".cross_attn_token_to_image", C / 2);
        attn(p + ".cross_attn_image_to_token", C / 2);
        for (int k = 1; k <= 4; ++k) {
            add(p + ".norm" + std::to_string(k) + ".weight", {C});
            add(p + ".norm" + std::to_string(k) + ".bias", {C});
        }
        add(p + ".mlp.lin1.weight", {2048, C}); add(p + ".mlp.lin1.bias", {2048});
        add(p + ".mlp.lin2.weight", {C, 2048}); add(p + ".mlp.lin2.bias", {C});
    }
    attn("mask_decoder.transformer.final_attn_token_to_image", C / 2);
    add("mask_decoder.transformer.norm_final_attn.weight", {C}); add("mask_decoder.transformer.norm_final_attn.bias", {C});
    add("mask_decoder.iou_token.weight", {1, C});
    add("mask_decoder.mask_tokens.weight", {4, C});
    add("mask_decoder.output_upscaling.0.weight", {C, C / 4, 2, 2}); add("mask_decoder.output_upscaling.0.bias", {C / 4});
    add("mask_decoder.output_upscaling.1.weight", {C / 4}); add("mask_decoder.output_upscaling.1.bias", {C / 4});
    add("mask_decoder.output_upscaling.3.weight", {C / 4, C / 8, 2, 2}); add("mask_decoder.output_upscaling.3.bias", {C / 8});
    for (int i = 0; i < 4; ++i) {
        const std::string p = "mask_decoder.output_hypernetworks_mlps." + std::to_string(i) + ".layers";
        add(p + ".0.weight", {C, C}); add(p + ".0.bias", {C});
        add(p + ".1.weight", {C, C}); add(p + ".1.bias", {C});
        add(p + ".2.weight", {C / 8, C}); add(p + ".2.bias", {C / 8});
    }
    const std::string p = "mask_decoder.iou_prediction_head.layers";
    add(p + ".0.weight", {C, C}); add(p + ".0.bias", {C});
    add(p + ".1.weight", {C, C}); add(p + ".1.bias", {C});
    add(p + ".2.weight", {4, C}); add(p + ".2.bias", {4});
}

const float* W(samrs_engine* e, const std::string& n) { return e->w.at(n).p; }

DecAttn dec_attn(samrs_engine* e, const std::string& p) {
    return DecAttn{W(e, p + ".q_proj.weight"), W(e, p + ".q_proj.bias"), W(e, p + ".k_proj.weight"),
                   W(e, p + ".k_proj.bias"),   W(e, p + ".v_proj.weight"), W(e, p + ".v_proj.bias"),
                   W(e, p + ".out_proj.weight"), W(e, p + ".out_proj.bias")};
}

// fp32 device tensor -> new ET device tensor (and, if asked for, the remainder of its two-term split); optionally frees the fp32 copy
int to_et(samrs_engine* e, const std::string& name, uint16_t** out, bool free_f32, hipStream_t s, uint16_t** out_lo = nullptr) {
    DevTensor& t = e->w.at(name);
    CK(e, dalloc(e, out, t.numel));
    if (out_lo) CK(e, dalloc(e, out_lo, t.numel));
    CK(e, launch_convert(e->prec, t.p, *out, (long)t.numel, s, out_lo ? *out_lo : nullptr));
    if (free_f32) {
        CK(e, hipStreamSynchronize(s));
        for (auto it = e->owned.begin(); it != e->owned.end(); ++it)
            if (*it == t.p) { e->owned.erase(it); break; }
        CK(e, hipFree(t.p));
        t.p = nullptr;
    }
    return SAMRS_OK;
}

hipError_t resid_gemm_ext(samrs_engine* e, int prec, const void* A, const void* Wt, const void* Ax, const void* Bx, const float* bias,
                          int M, int N, int K, hipStream_t s) {
    if (gemm_ext_ok(M, N, K)) return launch_gemm_et_ext(prec, A, Wt, Ax, Bx, e->X, bias, M, N, K, s);
    const hipError_t r = launch_gemm_et(prec, A, Wt, e->X, bias, nullptr, 0, M, N, K, true, false, true, s);
    if (r != hipSuccess) return r;
    GemmVariantScope base_kernel(1);
    return launch_gemm_et(prec, Ax, Bx, e->X, nullptr, nullptr, 0, M, N, 64, true, false, true, s);
}

// Outlier columns of the four block GEMMs of encoder block `i`, from the fp32 weights alone (they must still be resident: call before
// to_et frees them).  score_c = (magnitude proxy of operand column c) x || W[:, c] ||; the proxies: qkv / lin1 (A = a LayerNorm
// output): |gamma_c| + |beta_c|; lin2 (A = GELU(lin1)): || W1[c, :] || rms(gamma2) + |b1_c|; proj (A = the attention output, a
// convex combination of v rows): || Wv[c, :] || rms(gamma1) + |bv_c|.  A column is picked when its score exceeds ratio x the median
// score of its GEMM; at most 32 per GEMM (the largest), ascending.  oracle/outlier_budget.py restates the rule on the CPU and prices it.
int pick_outlier_columns(samrs_engine* e, int i, hipStream_t s) {
    const int D = e->D, H = 4 * D;
    EncBlock& b = e->blocks[i];
    const std::string p = "image_encoder.blocks." + std::to_string(i);
    if (!e->oc_scratch) CK(e, dalloc(e, &e->oc_scratch, (size_t)2 * 4 * D));
    float *dcol = e->oc_scratch, *drow = e->oc_scratch + 4 * D;
    std::vector<float> qkv_col(D), qkv_row(3 * D), l1_col(D), l1_row(H), l2_col(H), pj_col(D);
    auto norms = [&](const std::string& name, int N, int K, float* col, float* row) -> int {
        CK(e, launch_weight_norms(W(e, name), N, K, dcol, row ? drow : nullptr, s));
        CK(e, hipMemcpyAsync(col, dcol, sizeof(float) * K, hipMemcpyDeviceToHost, s));
        if (row) CK(e, hipMemcpyAsync(row, drow, sizeof(float) * N, hipMemcpyDeviceToHost, s));
        CK(e, hipStreamSynchronize(s));
        return SAMRS_OK;
    };
    int rc;
    if ((rc = norms(p + ".attn.qkv.weight", 3 * D, D, qkv_col.data(), qkv_row.data()))) return rc;
    if ((rc = norms(p + ".mlp.lin1.weight", H, D, l1_col.data(), l1_row.data()))) return rc;
    if ((rc = norms(p + ".mlp.lin2.weight", D, H, l2_col.data(), nullptr))) return rc;
    if ((rc = norms(p + ".attn.proj.weight", D, D, pj_col.data(), nullptr))) return rc;
    hipError_t copy_err = hipSuccess;
    auto host = [&](const std::string& name, size_t n) {
        std::vector<float> v(n);
        const hipError_t r = hipMemcpy(v.data(), W(e, name), sizeof(float) * n, hipMemcpyDeviceToHost);
        if (r != hipSuccess) copy_err = r;
        return v;
    };
    const std::vector<float> g1 = host(p + ".norm1.weight", D), b1 = host(p + ".norm1.bias", D), g2 = host(p + ".norm2.weight", D),
                             b2 = host(p + ".norm2.bias", D), bq = host(p + ".attn.qkv.bias", 3 * D), bl = host(p + ".mlp.lin1.bias", H);
    CK(e, copy_err);
    double r1 = 0, r2 = 0;
    for (int c = 0; c < D; ++c) { r1 += (double)g1[c] * g1[c]; r2 += (double)g2[c] * g2[c]; }
    const float rms1 = (float)std::sqrt(r1 / D), rms2 = (float)std::sqrt(r2 / D);
    std::vector<float> sc[4];
    sc[0].resize(D); sc[1].resize(D); sc[2].resize(H); sc[3].resize(D);
    for (int c = 0; c < D; ++c) {
        sc[0][c] = (std::fabs(g1[c]) + std::fabs(b1[c])) * std::sqrt(qkv_col[c]);
        sc[1][c] = (std::fabs(g2[c]) + std::fabs(b2[c])) * std::sqrt(l1_col[c]);
        sc[3][c] = (std::sqrt(qkv_row[2 * D + c]) * rms1 + std::fabs(bq[2 * D + c])) * std::sqrt(pj_col[c]);
    }
    for (int c = 0; c < H; ++c) sc[2][c] = (std::sqrt(l1_row[c]) * rms2 + std::fabs(bl[c])) * std::sqrt(l2_col[c]);
    const float ratio = e->outlier_ratio_pct * 0.01f;
    for (int g = 0; g < 4; ++g) {
        std::vector<float> tmp(sc[g]);
        std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
        const float med = tmp[tmp.size() / 2];
        std::vector<int> idx;
        for (int c = 0; c < (int)sc[g].size(); ++c) if (sc[g][c] > ratio * med) idx.push_back(c);
        if (idx.size() > 32) {
            std::partial_sort(idx.begin(), idx.begin() + 32, idx.end(), [&](int a, int c) { return sc[g][a] > sc[g][c]; });
            idx.resize(32);
            std::sort(idx.begin(), idx.end());
        }
        b.oc_n[g] = (int)idx.size();
        e->outlier_columns += b.oc_n[g];
        if (g == 3) {
            b.oc_heads = 0;
            for (int c : idx) b.oc_heads |= (e->hd > 0 && c / e->hd < 32) ? (1u << (c / e->hd)) : 0xffffffffu;
        }
        double all2 = 0.0, sel2 = 0.0;
        for (float v : sc[g]) all2 += (double)v * v;
        for (int c : idx) sel2 += (double)sc[g][c] * sc[g][c];
        b.oc_share[g] = all2 > 0.0 ? (float)(sel2 / all2) : 0.f;
        if (b.oc_n[g]) {
            CK(e, dalloc(e, &b.oc_idx[g], 32));
            idx.resize(32, 0);
            CK(e, hipMemcpy(b.oc_idx[g], idx.data(), sizeof(int) * 32, hipMemcpyHostToDevice));
        }
    }
    if (b.oc_n[0] || b.oc_n[1]) e->outlier_blocks += 1;
    if (b.oc_share[0] > 0.5f || b.oc_share[3] > 0.5f) e->outlier_dominant_blocks += 1;
    return SAMRS_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int samrs_abi_version(void) { return SAMRS_ABI_VERSION; }

const char* samrs_last_error(const samrs_engine_t* e) { return e ? e->err.c_str() : "null engine"; }

samrs_engine_t* samrs_create(const samrs_config* cfg, int device, char* err, int err_len) {
    auto bad = [&](const char* m) -> samrs_engine_t* {
        if (err && err_len > 0) snprintf(err, err_len, "%s", m);
        return nullptr;
    };
    if (!cfg) return bad("null config");
    if (cfg->img_size != 1024 || cfg->patch_size != 16 || cfg->window_size != 14 || cfg->out_chans != 256)
        return bad("only img_size 1024 / patch 16 / window 14 / out_chans 256 are supported (build_sam.py:62-80)");
    if (cfg->embed_dim % 128 || cfg->embed_dim % cfg->num_heads) return bad("embed_dim must be a multiple of 128 and of num_heads");
    const int hd = cfg->embed_dim / cfg->num_heads;
    if (hd != 64 && hd != 80) return bad("head_dim must be 64 or 80");
    if (cfg->precision != SAMRS_PREC_BF16 && cfg->precision != SAMRS_PREC_F16) return bad("bad precision");
    if (cfg->max_images < 1 || cfg->max_prompts < 1 || cfg->max_points < 0 || cfg->n_global < 0 || cfg->n_global > 8)
        return bad("bad capacity / n_global");
    if (5 + cfg->max_points + 1 + 2 > 16) return bad("max_points too large (token count must stay <= 16)");
    {
        DeviceGuard dg(device);
        if (dg.status != hipSuccess) return bad("hipSetDevice failed (no HIP device? this library has no CPU fallback)");
    }
    samrs_engine* e = new samrs_engine();
    e->cfg = *cfg;
    e->device = device;
    e->prec = cfg->precision;
    e->grid = cfg->img_size / cfg->patch_size;
    e->tokens = e->grid * e->grid;
    e->D = cfg->embed_dim;
    e->C = cfg->out_chans;
    e->hd = hd;
    e->nwin = (e->grid + cfg->window_size - 1) / cfg->window_size;
    e->T_max = 5 + cfg->max_points + 1 + 2;
    e->slot_set.assign(cfg->max_images, 0);
    e->slot_split.assign(cfg->max_images, 0);
    e->slot_depth.assign(cfg->max_images, 0);
    e->decoder_fusion = env_int("SAMRS_DECODER_FUSION", 1) != 0;
    e->ln_fold = env_int("SAMRS_LN_FOLD", 0) != 0 && gemm_has_experiments();
    // default: the cheap rounding points everywhere; where the one-launch split GEMM covers the block shapes (ViT-H), also the
    // v third of qkv + proj in the leading blocks -- that is what it takes to hold IoU >= 0.999 on the multimask (C4)
    // fixtures at ViT-H, for 0.90x the throughput.  SAMRS_SPLIT=15 / option "split" = 15: the 1x-rate arithmetic.
    const bool h_like = gemm_split3_ok(e->tokens, 3 * e->D, e->D) && gemm_split3_ok(e->tokens, e->D, e->D) && (2 * e->D) % 320 == 0;
    e->split = env_int("SAMRS_SPLIT", SPLIT_DEFAULT | (h_like ? SPLIT_ATTN_V : 0)) & SPLIT_ALL;
    // ViT-H: the multimask tokens hold IoU >= 0.999 only from the v-third / proj split on; below ViT-H the 1x rate does
    e->grade_multimask = h_like ? SPLIT_ATTN_ANY : 0;
    // an explicit SAMRS_SPLIT is the operator's decision about the whole process, multimask outputs included
    e->allow_reduced = env_int("SAMRS_ALLOW_REDUCED", getenv("SAMRS_SPLIT") ? 1 : 0) != 0;
    e->upscaler_fused = env_int("SAMRS_UPSCALER_FUSED", 1) != 0;
    e->split_passes = env_int("SAMRS_SPLIT_PASSES", 0) != 0;
    e->split_depth = env_int("SAMRS_SPLIT_DEPTH", 0);
    e->lo_format = (h_like && env_int("SAMRS_LO_FORMAT", 4) == 4) ? 4 : 0;
    e->gelu_fast = env_int("SAMRS_GELU_FAST", -1);
    e->ln_tail = env_int("SAMRS_LN_TAIL", 0) != 0;
    e->operand_pad_on = env_int("SAMRS_OPERAND_PAD", 1) != 0;
    e->outlier_on = env_int("SAMRS_OUTLIER_COLS", 7) & 7;
    e->outlier_ratio_pct = env_int("SAMRS_OUTLIER_RATIO_PCT", 400);
    if (e->outlier_ratio_pct < 101) e->outlier_ratio_pct = 101;
    if (e->gelu_fast > 1) e->gelu_fast = 1;
    if (const int rc0 = env_int("SAMRS_RANGE_CHECK", 0)) {
        if (samrs_set_option(e, "range_check", rc0) != SAMRS_OK) {
            if (err && err_len > 0) snprintf(err, err_len, "SAMRS_RANGE_CHECK=%d: %s", rc0, e->err.c_str());
            samrs_destroy(e);
            return nullptr;
        }
    }
    return e;
}

void samrs_destroy(samrs_engine_t* e) {
    if (!e) return;
    DeviceGuard dg(e->device);
    for (void* p : e->owned) (void)hipFree(p);
    if (e->rle_scratch) (void)hipFree(e->rle_scratch);
    delete e;
}

int samrs_load_weight(samrs_engine_t* e, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!e || !name || !host || !shape || ndim < 1 || ndim > 4) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_load_weight: bad argument");
    if (e->finalized) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "weights already finalized");
    ON_DEVICE(e);
    DevTensor t;
    t.shape.assign(shape, shape + ndim);
    t.numel = 1;
    for (int i = 0; i < ndim; ++i) t.numel *= (size_t)shape[i];
    const std::string n(name);
    std::vector<float> tmp;
    const float* src = host;
    if (n == "image_encoder.neck.2.weight" && ndim == 4) {
        // [co][ci][ky][kx] -> [co][(ky*3+kx)*Ci + ci]  (k order of neck_im2col_kernel)
        const int64_t Co = shape[0], Ci = shape[1], KH = shape[2], KW = shape[3];
        tmp.resize(t.numel);
        for (int64_t co = 0; co < Co; ++co)
            for (int64_t ci = 0; ci < Ci; ++ci)
                for (int64_t ky = 0; ky < KH; ++ky)
                    for (int64_t kx = 0; kx < KW; ++kx)
                        tmp[(co * KH * KW + ky * KW + kx) * Ci + ci] = host[((co * Ci + ci) * KH + ky) * KW + kx];
        src = tmp.data();
    } else if ((n == "mask_decoder.output_upscaling.0.weight" || n == "mask_decoder.output_upscaling.3.weight") && ndim == 4) {
        // ConvTranspose2d [ci][co][dy][dx] -> GEMM B [(dy*2+dx)*Co + co][ci]   (mask_decoder.py:53-59)
        const int64_t Ci = shape[0], Co = shape[1];
        tmp.resize(t.numel);
        for (int64_t ci = 0; ci < Ci; ++ci)
            for (int64_t co = 0; co < Co; ++co)
                for (int64_t s = 0; s < 4; ++s) tmp[(s * Co + co) * Ci + ci] = host[(ci * Co + co) * 4 + s];
        src = tmp.data();
    }
    auto it = e->w.find(n);
    if (it != e->w.end()) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "duplicate tensor %s", name);
    CK(e, dalloc(e, &t.p, t.numel));
    CK(e, hipMemcpy(t.p, src, t.numel * sizeof(float), hipMemcpyHostToDevice));
    e->w.emplace(n, std::move(t));
    return SAMRS_OK;
}

int samrs_finalize_weights(samrs_engine_t* e, void* stream) {
    if (!e) return SAMRS_ERR_BAD_ARG;
    if (e->finalized) return SAMRS_OK;
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(e);
    // ---- strict check (build_sam.py:106 load_state_dict raises on any mismatch) ----
    std::vector<std::pair<std::string, std::vector<int64_t>>> req;
    required_tensors(e, req);
    for (auto& r : req) {
        auto it = e->w.find(r.first);
        if (it == e->w.end()) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "missing tensor %s", r.first.c_str());
        if (it->second.shape != r.second) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "shape mismatch for %s", r.first.c_str());
    }
    if (e->w.size() != req.size()) {
        for (auto& kv : e->w) {
            bool found = false;
            for (auto& r : req) if (r.first == kv.first) { found = true; break; }
            if (!found) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "unexpected tensor %s", kv.first.c_str());
        }
    }
    const samrs_config& c = e->cfg;
    const int D = e->D, C = e->C, tokens = e->tokens;
    int rc;
    // ---- encoder weights -> ET ----
    if ((rc = to_et(e, "image_encoder.patch_embed.proj.weight", &e->patch_w, true, s, &e->patch_w_lo))) return rc;
    if ((rc = to_et(e, "image_encoder.neck.0.weight", &e->neck0_w, true, s, &e->neck0_w_lo))) return rc;
    if ((rc = to_et(e, "image_encoder.neck.2.weight", &e->neck2_w, true, s, &e->neck2_w_lo))) return rc;
    e->blocks.resize(c.depth);
    for (int i = 0; i < c.depth; ++i) {
        const std::string p = "image_encoder.blocks." + std::to_string(i);
        EncBlock& b = e->blocks[i];
        b.global = is_global(c, i);
        // padded operand rows (field ldk) exist where a D-element row is an even number of 256-byte units AND the qkv / lin1 shapes can
        // run on the kernels that take a stride (N a multiple of 320): ViT-H
        if (i == 0) e->ldk = (e->operand_pad_on && D % 256 == 0 && (3 * D) % 320 == 0 && (4 * D) % 320 == 0) ? D + 128 : 0;
        b.ln1w = W(e, p + ".norm1.weight"); b.ln1b = W(e, p + ".norm1.bias");
        b.ln2w = W(e, p + ".norm2.weight"); b.ln2b = W(e, p + ".norm2.bias");
        if (D == 1280 && e->ln_fold && gemm_has_experiments()) {       // folded LayerNorm needs the fp32 weights: before to_et() frees them
            CK(e, dalloc(e, &b.qkv_wf, (size_t)3 * D * D)); CK(e, dalloc(e, &b.qkv_c, (size_t)3 * D)); CK(e, dalloc(e, &b.qkv_bf, (size_t)3 * D));
            CK(e, dalloc(e, &b.lin1_wf, (size_t)4 * D * D)); CK(e, dalloc(e, &b.lin1_c, (size_t)4 * D)); CK(e, dalloc(e, &b.lin1_bf, (size_t)4 * D));
            CK(e, launch_ln_fold_weight(e->prec, W(e, p + ".attn.qkv.weight"), b.ln1w, b.ln1b, W(e, p + ".attn.qkv.bias"), b.qkv_wf,
                                        b.qkv_c, b.qkv_bf, 3 * D, D, s));
            CK(e, launch_ln_fold_weight(e->prec, W(e, p + ".mlp.lin1.weight"), b.ln2w, b.ln2b, W(e, p + ".mlp.lin1.bias"), b.lin1_wf,
                                        b.lin1_c, b.lin1_bf, 4 * D, D, s));
            e->can_fold = true;
        }
        const bool lo_a = (e->split & SPLIT_ATTN_ANY) != 0, lo_m = (e->split & SPLIT_MLP) != 0, lo_l2 = (e->split & SPLIT_LIN2) != 0;
        if (lo_a && e->lo_format == 4 && gemm_mx_ok(c.max_images * e->tokens, 3 * D, D, D)) {
            // fp4 copies of hi / lo of the attention-side weights (from the fp32 tensors, before to_et frees them)
            const int gp = (e->hd + 31) / 32 * 32, kp = c.num_heads * gp;
            if (gemm_mx_ok(c.max_images * e->tokens, D, D, kp)) {
                e->mx_gp = gp; e->mx_kp_proj = kp;
                for (int h = 0; h < 2; ++h) {
                    CK(e, dalloc(e, &b.qkv_w4[h], (size_t)3 * D * D / 2)); CK(e, dalloc(e, &b.qkv_s4[h], mx_scale_bytes(3 * D, D, true)));
                    CK(e, dalloc(e, &b.proj_w4[h], (size_t)D * kp / 2)); CK(e, dalloc(e, &b.proj_s4[h], mx_scale_bytes(D, kp, true)));
                }
                CK(e, launch_mx4_pack(e->prec, W(e, p + ".attn.qkv.weight"), nullptr, nullptr, nullptr, b.qkv_w4[0], b.qkv_w4[1], b.qkv_s4[0],
                                      b.qkv_s4[1], 3 * D, D, D, D, true, s));
                CK(e, launch_mx4_pack(e->prec, W(e, p + ".attn.proj.weight"), nullptr, nullptr, nullptr, b.proj_w4[0], b.proj_w4[1], b.proj_s4[0],
                                      b.proj_s4[1], D, D, e->hd, gp, true, s, /* perm: the attention kernels' block order */ true));
                e->mx_ready = true;
            }
        }
        if ((lo_m || lo_l2) && e->lo_format == 4 && gemm_mx_ok(c.max_images * e->tokens, 4 * D, D, D) && (4 * D) % 80 == 0) {
            const int kp2 = 4 * D / 80 * 96;
            if (gemm_mx_ok(c.max_images * e->tokens, D, 4 * D, kp2)) {
                e->mx_kp_lin2 = kp2;
                for (int h = 0; h < 2; ++h) {
                    CK(e, dalloc(e, &b.lin1_w4[h], (size_t)4 * D * D / 2)); CK(e, dalloc(e, &b.lin1_s4[h], mx_scale_bytes(4 * D, D, true)));
                    CK(e, dalloc(e, &b.lin2_w4[h], (size_t)D * kp2 / 2)); CK(e, dalloc(e, &b.lin2_s4[h], mx_scale_bytes(D, kp2, true)));
                }
                CK(e, launch_mx4_pack(e->prec, W(e, p + ".mlp.lin1.weight"), nullptr, nullptr, nullptr, b.lin1_w4[0], b.lin1_w4[1], b.lin1_s4[0],
                                      b.lin1_s4[1], 4 * D, D, D, D, true, s));
                CK(e, launch_mx4_pack(e->prec, W(e, p + ".mlp.lin2.weight"), nullptr, nullptr, nullptr, b.lin2_w4[0], b.lin2_w4[1], b.lin2_s4[0],
                                      b.lin2_s4[1], D, 4 * D, 80, 96, true, s, /* perm: the order of lin1's epilogue */ 2));
                e->mx_mlp_ready = true;
            }
        }
        // outlier columns: picked and their weight-side extension written while the fp32 weights are still resident
        if (e->outlier_on && (rc = pick_outlier_columns(e, i, s))) return rc;
        if (e->ldk) {
            const size_t ldb = (size_t)e->ldk * 2;
            CK(e, dalloc(e, &b.qkv_wp, (size_t)3 * D * e->ldk)); CK(e, dalloc(e, &b.lin1_wp, (size_t)4 * D * e->ldk));
            CK(e, hipMemsetAsync(b.qkv_wp, 0, (size_t)3 * D * ldb, s)); CK(e, hipMemsetAsync(b.lin1_wp, 0, (size_t)4 * D * ldb, s));
            if (b.oc_n[0]) CK(e, launch_outlier_weight_ext(e->prec, W(e, p + ".attn.qkv.weight"), 3 * D, D, b.oc_idx[0], b.oc_n[0], b.qkv_wp, e->ldk, D, s));
            if (b.oc_n[1]) CK(e, launch_outlier_weight_ext(e->prec, W(e, p + ".mlp.lin1.weight"), 4 * D, D, b.oc_idx[1], b.oc_n[1], b.lin1_wp, e->ldk, D, s));
        }
        if (b.oc_n[0]) {
            CK(e, dalloc(e, &b.qkv_wx, (size_t)3 * D * (D + 64)));
            CK(e, launch_outlier_weight_ext(e->prec, W(e, p + ".attn.qkv.weight"), 3 * D, D, b.oc_idx[0], b.oc_n[0], b.qkv_wx, D + 64, D, s));
        }
        if (b.oc_n[1]) {
            CK(e, dalloc(e, &b.lin1_wx, (size_t)4 * D * (D + 64)));
            CK(e, launch_outlier_weight_ext(e->prec, W(e, p + ".mlp.lin1.weight"), 4 * D, D, b.oc_idx[1], b.oc_n[1], b.lin1_wx, D + 64, D, s));
        }
        if (b.oc_n[2]) {          // lin2: weight side [D][64] + the side weights / bias of the hidden units' recomputation
            CK(e, dalloc(e, &b.oc_bx[2], (size_t)D * 64));
            CK(e, launch_outlier_weight_ext(e->prec, W(e, p + ".mlp.lin2.weight"), D, 4 * D, b.oc_idx[2], b.oc_n[2], b.oc_bx[2], 64, 0, s));
            CK(e, dalloc(e, &b.lin2_sb, (size_t)32));
            const int k0 = D + (b.oc_n[1] ? 64 : 0);
            CK(e, dalloc(e, &b.lin2_ws, (size_t)32 * k0));
            CK(e, launch_outlier_side_weight(e->prec, W(e, p + ".mlp.lin1.weight"), W(e, p + ".mlp.lin1.bias"), D, b.oc_idx[2], b.oc_n[2],
                                             b.oc_idx[1], b.oc_n[1], b.lin2_ws, k0, b.lin2_sb, s));
            e->oc_resid = true;
        }
        if (b.oc_n[3]) {          // proj
            CK(e, dalloc(e, &b.oc_bx[3], (size_t)D * 64));
            CK(e, launch_outlier_weight_ext(e->prec, W(e, p + ".attn.proj.weight"), D, D, b.oc_idx[3], b.oc_n[3], b.oc_bx[3], 64, 0, s));
            e->oc_resid = true;
        }
        if ((rc = to_et(e, p + ".attn.qkv.weight", &b.qkv_w, true, s, lo_a ? &b.qkv_w_lo : nullptr))) return rc;
        if ((rc = to_et(e, p + ".attn.proj.weight", &b.proj_w, true, s, lo_a ? &b.proj_w_lo : nullptr))) return rc;
        if ((rc = to_et(e, p + ".mlp.lin1.weight", &b.lin1_w, true, s, lo_m ? &b.lin1_w_lo : nullptr))) return rc;
        if ((rc = to_et(e, p + ".mlp.lin2.weight", &b.lin2_w, true, s, lo_m ? &b.lin2_w_lo : nullptr))) return rc;
        {
            const size_t wb = (size_t)D * 2;
            if (e->ldk) {
                const size_t ldb = (size_t)e->ldk * 2;
                CK(e, hipMemcpy2DAsync(b.qkv_wp, ldb, b.qkv_w, wb, wb, (size_t)3 * D, hipMemcpyDeviceToDevice, s));
                CK(e, hipMemcpy2DAsync(b.lin1_wp, ldb, b.lin1_w, wb, wb, (size_t)4 * D, hipMemcpyDeviceToDevice, s));
            }
            const size_t ldx = (size_t)(D + 64) * 2;
            if (b.qkv_wx) CK(e, hipMemcpy2DAsync(b.qkv_wx, ldx, b.qkv_w, wb, wb, (size_t)3 * D, hipMemcpyDeviceToDevice, s));
            if (b.lin1_wx) CK(e, hipMemcpy2DAsync(b.lin1_wx, ldx, b.lin1_w, wb, wb, (size_t)4 * D, hipMemcpyDeviceToDevice, s));
        }
        b.ln1w = W(e, p + ".norm1.weight"); b.ln1b = W(e, p + ".norm1.bias");
        b.ln2w = W(e, p + ".norm2.weight"); b.ln2b = W(e, p + ".norm2.bias");
        b.qkv_b = W(e, p + ".attn.qkv.bias"); b.proj_b = W(e, p + ".attn.proj.bias");
        b.lin1_b = W(e, p + ".mlp.lin1.bias"); b.lin2_b = W(e, p + ".mlp.lin2.bias");
        b.rel_h = W(e, p + ".attn.rel_pos_h"); b.rel_w = W(e, p + ".attn.rel_pos_w");
    }
    // ---- decoder: dense PE, fused image-side projection weights, PE projections ----
    CK(e, dalloc(e, &e->PE, (size_t)tokens * C));
    CK(e, launch_dense_pe(W(e, "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"), e->PE, e->grid, s));
    const int Ci = C / 2;  // 128
    float* tmpw = nullptr;
    CK(e, dalloc(e, &tmpw, (size_t)3 * Ci * C));
    e->layers.resize(2);
    for (int i = 0; i < 2; ++i) {
        const std::string p = "mask_decoder.transformer.layers." + std::to_string(i);
        DecLayer& L = e->layers[i];
        L.self = dec_attn(e, p + ".self_attn");
        L.t2i = dec_attn(e, p + ".cross_attn_token_to_image");
        L.i2t = dec_attn(e, p + ".cross_attn_image_to_token");
        L.n1w = W(e, p + ".norm1.weight"); L.n1b = W(e, p + ".norm1.bias");
        L.n2w = W(e, p + ".norm2.weight"); L.n2b = W(e, p + ".norm2.bias");
        L.n3w = W(e, p + ".norm3.weight"); L.n3b = W(e, p + ".norm3.bias");
        L.n4w = W(e, p + ".norm4.weight"); L.n4b = W(e, p + ".norm4.bias");
        L.m1w = W(e, p + ".mlp.lin1.weight"); L.m1b = W(e, p + ".mlp.lin1.bias");
        L.m2w = W(e, p + ".mlp.lin2.weight"); L.m2b = W(e, p + ".mlp.lin2.bias");
        // [Wk_t2i; Wv_t2i; Wq_i2t]
        CK(e, hipMemcpyAsync(tmpw, L.t2i.kw, sizeof(float) * Ci * C, hipMemcpyDeviceToDevice, s));
        CK(e, hipMemcpyAsync(tmpw + Ci * C, L.t2i.vw, sizeof(float) * Ci * C, hipMemcpyDeviceToDevice, s));
        CK(e, hipMemcpyAsync(tmpw + 2 * Ci * C, L.i2t.qw, sizeof(float) * Ci * C, hipMemcpyDeviceToDevice, s));
        CK(e, dalloc(e, &L.kvq_w, (size_t)3 * Ci * C));
        CK(e, launch_convert(e->prec, tmpw, L.kvq_w, (long)3 * Ci * C, s));
        CK(e, dalloc(e, &L.kvq_b, (size_t)3 * Ci));
        CK(e, hipMemcpyAsync(L.kvq_b, L.t2i.kb, sizeof(float) * Ci, hipMemcpyDeviceToDevice, s));
        CK(e, hipMemcpyAsync(L.kvq_b + Ci, L.t2i.vb, sizeof(float) * Ci, hipMemcpyDeviceToDevice, s));
        CK(e, hipMemcpyAsync(L.kvq_b + 2 * Ci, L.i2t.qb, sizeof(float) * Ci, hipMemcpyDeviceToDevice, s));
        CK(e, dalloc(e, &L.kvq_pe, (size_t)tokens * 3 * Ci));
        CK(e, hipMemsetAsync(L.kvq_pe, 0, sizeof(float) * tokens * 3 * Ci, s));
        CK(e, launch_gemm_f32(e->PE, C, L.t2i.kw, nullptr, L.kvq_pe, 3 * Ci, tokens, Ci, C, false, false, s));
        CK(e, launch_gemm_f32(e->PE, C, L.i2t.qw, nullptr, L.kvq_pe + 2 * Ci, 3 * Ci, tokens, Ci, C, false, false, s));
        if ((rc = to_et(e, p + ".cross_attn_image_to_token.out_proj.weight", &L.i2t_ow, false, s, &L.i2t_ow_lo))) return rc;
    }
    e->fin = dec_attn(e, "mask_decoder.transformer.final_attn_token_to_image");
    CK(e, hipMemcpyAsync(tmpw, e->fin.kw, sizeof(float) * Ci * C, hipMemcpyDeviceToDevice, s));
    CK(e, hipMemcpyAsync(tmpw + Ci * C, e->fin.vw, sizeof(float) * Ci * C, hipMemcpyDeviceToDevice, s));
    CK(e, dalloc(e, &e->fin_kv_w, (size_t)2 * Ci * C));
    CK(e, launch_convert(e->prec, tmpw, e->fin_kv_w, (long)2 * Ci * C, s));
    CK(e, dalloc(e, &e->fin_kv_b, (size_t)2 * Ci));
    CK(e, hipMemcpyAsync(e->fin_kv_b, e->fin.kb, sizeof(float) * Ci, hipMemcpyDeviceToDevice, s));
    CK(e, hipMemcpyAsync(e->fin_kv_b + Ci, e->fin.vb, sizeof(float) * Ci, hipMemcpyDeviceToDevice, s));
    CK(e, dalloc(e, &e->fin_pe, (size_t)tokens * 2 * Ci));
    CK(e, hipMemsetAsync(e->fin_pe, 0, sizeof(float) * tokens * 2 * Ci, s));
    CK(e, launch_gemm_f32(e->PE, C, e->fin.kw, nullptr, e->fin_pe, 2 * Ci, tokens, Ci, C, false, false, s));
    // upscaler: weights were reordered at load to GEMM-B layout; biases are tiled over the 4 sub-pixels
    if ((rc = to_et(e, "mask_decoder.output_upscaling.0.weight", &e->up1_w, false, s, &e->up1_w_lo))) return rc;
    if ((rc = to_et(e, "mask_decoder.output_upscaling.3.weight", &e->up2_w, false, s, &e->up2_w_lo))) return rc;
    CK(e, dalloc(e, &e->up1_b, (size_t)C));
    CK(e, dalloc(e, &e->up2_b, (size_t)C / 2));
    CK(e, dalloc(e, &e->up_ln, (size_t)C / 2));
    CK(e, hipMemcpyAsync(e->up_ln, W(e, "mask_decoder.output_upscaling.1.weight"), sizeof(float) * C / 4, hipMemcpyDeviceToDevice, s));
    CK(e, hipMemcpyAsync(e->up_ln + C / 4, W(e, "mask_decoder.output_upscaling.1.bias"), sizeof(float) * C / 4, hipMemcpyDeviceToDevice, s));
    for (int k = 0; k < 4; ++k) {
        CK(e, hipMemcpyAsync(e->up1_b + k * (C / 4), W(e, "mask_decoder.output_upscaling.0.bias"), sizeof(float) * C / 4, hipMemcpyDeviceToDevice, s));
        CK(e, hipMemcpyAsync(e->up2_b + k * (C / 8), W(e, "mask_decoder.output_upscaling.3.bias"), sizeof(float) * C / 8, hipMemcpyDeviceToDevice, s));
    }

    // ---- workspaces ----
    const size_t Bi = c.max_images, Bb = c.max_prompts;
    const size_t M = Bi * tokens;
    const size_t Mmax = M;
    CK(e, dalloc(e, &e->X, M * D));
    CK(e, dalloc(e, &e->ln_counters, M / 256 + 1));
    CK(e, hipMemsetAsync(e->ln_counters, 0, (M / 256 + 1) * sizeof(unsigned int), s));
    const size_t y_ld_max = e->ldk ? (size_t)e->ldk : (e->outlier_blocks ? (size_t)D + 64 : (size_t)D);    // ldk = D + 128 holds the 64 extension columns too
    CK(e, dalloc(e, &e->Y, Mmax * y_ld_max));
    CK(e, hipMemsetAsync(e->Y, 0, Mmax * y_ld_max * 2, s));
    if (e->can_fold) { CK(e, dalloc(e, &e->STATS, Mmax * 16)); CK(e, dalloc(e, &e->ROWSTAT, Mmax * 2)); }
    CK(e, dalloc(e, &e->QKV, Mmax * 3 * D));
    CK(e, dalloc(e, &e->AO, M * D));
    CK(e, dalloc(e, &e->VTG, M * D));
    if (e->oc_resid) {
        CK(e, dalloc(e, &e->OCX, M * 64));
        if (!(e->split & SPLIT_ATTN_ANY)) CK(e, dalloc(e, &e->AOlo, M * D));       // proj's outlier columns take the attention output's lo half
    }
    e->split_ready = SPLIT_DEFAULT | (e->split & SPLIT_MLP) | ((e->split & SPLIT_ATTN_ANY) ? SPLIT_ATTN_ANY : 0) | (e->mx_mlp_ready ? SPLIT_LIN2 : 0);
    if ((e->split & SPLIT_LIN2) && !e->mx_mlp_ready) e->split &= ~SPLIT_LIN2;          // no MX kernel for these shapes (or lo_format 0): bit ignored
    if (e->split & (SPLIT_ATTN_ANY | SPLIT_MLP | SPLIT_LIN2)) {
        CK(e, dalloc(e, &e->Ylo, M * D));
        if (e->split & SPLIT_MLP) CK(e, dalloc(e, &e->F32T, M * 4 * D));      // the generic attention-side route allocates it on first use
        if (e->split & SPLIT_ATTN_ANY) CK(e, dalloc(e, &e->AOlo, M * D));
        if (e->mx_mlp_ready) {
            for (int h = 0; h < 2; ++h) {
                CK(e, dalloc(e, &e->H4[h], M * e->mx_kp_lin2 / 2)); CK(e, dalloc(e, &e->SH4[h], mx_scale_bytes((int)M, e->mx_kp_lin2, false)));
                if (!e->Y4[h]) { CK(e, dalloc(e, &e->Y4[h], M * D / 2)); CK(e, dalloc(e, &e->SY4[h], mx_scale_bytes((int)M, D, false))); }
            }
        }
        if (e->mx_ready) {
            for (int h = 0; h < 2; ++h) {
                if (!e->Y4[h]) {
                CK(e, dalloc(e, &e->Y4[h], M * D / 2)); CK(e, dalloc(e, &e->SY4[h], mx_scale_bytes((int)M, D, false)));
                }
                CK(e, dalloc(e, &e->AO4[h], M * e->mx_kp_proj / 2)); CK(e, dalloc(e, &e->SAO4[h], mx_scale_bytes((int)M, e->mx_kp_proj, false)));
            }
        }
        if (e->split & SPLIT_MLP) CK(e, dalloc(e, &e->Hlo, M * 4 * D));
    }
    size_t hsz = M * 4 * D;
    if (M * 9 * C * 2 > hsz) hsz = M * 9 * C * 2;      // neck im2col, hi + lo
    if (M * 768 > hsz) hsz = M * 768;
    CK(e, dalloc(e, &e->H, hsz));
    CK(e, dalloc(e, &e->N1, M * C));
    CK(e, dalloc(e, &e->N1e, M * C));
    CK(e, dalloc(e, &e->EMB, M * C));
    const size_t BT = Bb * e->T_max;
    CK(e, dalloc(e, &e->TOK0, BT * C)); CK(e, dalloc(e, &e->Q, BT * C)); CK(e, dalloc(e, &e->TA, BT * C));
    CK(e, dalloc(e, &e->TQ, BT * C)); CK(e, dalloc(e, &e->TK, BT * C)); CK(e, dalloc(e, &e->TV, BT * C));
    CK(e, dalloc(e, &e->TO, BT * C)); CK(e, dalloc(e, &e->MH, BT * 2048));
    CK(e, dalloc(e, &e->QP, BT * Ci)); CK(e, dalloc(e, &e->KT, BT * Ci)); CK(e, dalloc(e, &e->VT, BT * Ci));
    CK(e, dalloc(e, &e->O128, BT * Ci));
    CK(e, dalloc(e, &e->T2IW, t2i_workspace_floats(c.max_prompts, e->T_max)));
    CK(e, dalloc(e, &e->K0F, (size_t)c.max_images * tokens * C)); CK(e, dalloc(e, &e->K0E, (size_t)c.max_images * tokens * C));
    CK(e, dalloc(e, &e->KVQ0, (size_t)c.max_images * tokens * 3 * (C / 2)));
    CK(e, dalloc(e, &e->KF, Bb * tokens * C)); CK(e, dalloc(e, &e->KE, Bb * tokens * C));
    CK(e, dalloc(e, &e->KE_lo, Bb * tokens * C));
    CK(e, dalloc(e, &e->KVQ, Bb * tokens * 3 * Ci)); CK(e, dalloc(e, &e->OI, Bb * tokens * Ci));
    CK(e, dalloc(e, &e->U1raw, Bb * tokens * C)); CK(e, dalloc(e, &e->U1, Bb * tokens * C));
    CK(e, dalloc(e, &e->U2, Bb * tokens * 4 * (C / 2)));
    CK(e, dalloc(e, &e->HY1, 5 * Bb * C)); CK(e, dalloc(e, &e->HY2, 5 * Bb * C));
    CK(e, dalloc(e, &e->HYPER, Bb * 4 * (C / 8))); CK(e, dalloc(e, &e->IOU, Bb * 4));
    CK(e, dalloc(e, &e->LOW, Bb * 3 * 256 * 256));
    CK(e, hipStreamSynchronize(s));
    e->finalized = true;
    return SAMRS_OK;
}

// -------------------------------------------------------------------------------------------------
// images[i]: device pointer of tile i (uint8 HWC, in_h[i] x in_w[i], long side == img_size).  Tiles of one call may
// differ in size (HRSC / DIOR images after ResizeLongestSide): only the im2col reads pixels, everything downstream
// works on the zero-padded 64 x 64 token grid (sam.py:170-173).
static int prepare_slot_keys(samrs_engine_t* e, int slot0, int n, hipStream_t s);

static int encode(samrs_engine_t* e, const uint8_t* const* images, const int* in_h, const int* in_w, int n, int slot0,
                  void* stream, int n_blocks, bool do_neck) {
    if (!e || !images || !in_h || !in_w) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_set_images: null argument");
    if (!e->finalized) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "weights not finalized");
    const samrs_config& c = e->cfg;
    if (n < 1 || slot0 < 0 || slot0 + n > c.max_images) return fail(e, SAMRS_ERR_CAPACITY, "n_images/slot out of range (max_images=%d)", c.max_images);
    for (int i = 0; i < n; ++i) {
        if (!images[i]) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_set_images: null image pointer (tile %d)", i);
        if (in_h[i] < 1 || in_w[i] < 1 || in_h[i] > c.img_size || in_w[i] > c.img_size || (in_h[i] != c.img_size && in_w[i] != c.img_size))
            return fail(e, SAMRS_ERR_BAD_SHAPE, "set_torch_image input must be BCHW with long side %d.", c.img_size);
    }
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(e);
    const int D = e->D, C = e->C, g = e->grid, tokens = e->tokens, prec = e->prec;
    const int M = n * tokens;
    for (int i = 0; i < n; ++i) e->slot_set[slot0 + i] = 0;

    // patch embed: im2col (normalise + zero pad) -> GEMM (+bias +pos_embed) -> X.  One im2col launch per run of
    // same-size tiles that sit back to back in memory (the whole batch for a contiguous tile stack).
    // SPLIT_PATCH: pixels and weights as hi + lo (three GEMM passes into X) -- the normalised pixel values span +-2.6 and
    // their f16 rounding alone cost 266 of the 899 class-map pixels the round-2 engine lost at ViT-H (oracle/error_budget.py)
    const size_t KP = (size_t)3 * c.patch_size * c.patch_size;
    const bool sp_patch = (e->split & SPLIT_PATCH) != 0;
    uint16_t* Hlo = e->H + (size_t)M * KP;
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && in_h[j] == in_h[i] && in_w[j] == in_w[i] &&
               images[j] == images[i] + (size_t)(j - i) * in_h[i] * in_w[i] * 3) ++j;
        CK(e, launch_patch_im2col(prec, images[i], e->H + (size_t)i * tokens * KP, j - i, in_h[i], in_w[i], g, c.patch_size, s,
                                  sp_patch ? Hlo + (size_t)i * tokens * KP : nullptr));
        i = j;
    }
    // split products of patch embed and neck: ONE launch over a three-segment K axis where the shape fits the pair-stage tile
    // (gemm.hip seg_src_a), else three accumulating passes
    const bool one3p = !e->split_passes;
    if (sp_patch && one3p && gemm_split3_ok((int)M, D, (int)KP, true)) {
        CK(e, launch_gemm_et_split3(prec, e->H, Hlo, e->patch_w, e->patch_w_lo, e->X, W(e, "image_encoder.patch_embed.proj.bias"),
                                    (int)M, D, (int)KP, true, false, s, 0, W(e, "image_encoder.pos_embed"), tokens));
    } else {
        CK(e, launch_gemm_et(prec, e->H, e->patch_w, e->X, W(e, "image_encoder.patch_embed.proj.bias"),
                             W(e, "image_encoder.pos_embed"), tokens, M, D, (int)KP, true, false, false, s));
        if (sp_patch) {
            CK(e, launch_gemm_et(prec, Hlo, e->patch_w, e->X, nullptr, nullptr, 0, M, D, (int)KP, true, false, true, s));
            CK(e, launch_gemm_et(prec, e->H, e->patch_w_lo, e->X, nullptr, nullptr, 0, M, D, (int)KP, true, false, true, s));
        }
    }
    // Folded LayerNorm (embed_dim 1280): no LayerNorm launches inside the blocks.  Y holds the residual stream rounded to ET and
    // STATS its per-row partial statistics, both written by the epilogue of the GEMM that produced X (proj, lin2; here, once,
    // by rowstats_convert); qkv / lin1 run on the gamma-folded weights and normalise in their epilogue (gemm.hip).
    // reference-grade bits: the block GEMMs on hi + lo operands (qkv / lin1: three passes into an fp32 scratch, then one
    // rounding to the operand type; proj / lin2: two more accumulating passes into the residual stream)
    const bool any_attn = (e->split & SPLIT_ATTN_ANY) != 0, any_mlp = (e->split & SPLIT_MLP) != 0;
    const bool fold = e->can_fold && e->ln_fold && !any_attn && !any_mlp;
    // "split_depth" > 0: the block-GEMM bits apply to the first split_depth blocks only -- an operand error made early is carried
    // (and amplified) through every later block, one made in the last blocks is not (error_budget.py plans5 / plans6)
    // 0 = automatic: every block for the full bits (16 / 32), the leading three quarters for the v-third form (64)
    const int depth_full = e->split_depth > 0 ? e->split_depth : c.depth;
    const int depth_v = e->split_depth > 0 ? e->split_depth : (3 * c.depth + 3) / 4;
    // the three split terms of a block GEMM as ONE launch over a three-segment K axis (gemm.hip seg_src_a) where the shape
    // fits the 256 x 320 tile (ViT-H); SAMRS_SPLIT_PASSES=1 / option "split_passes" keeps the three accumulating launches (A/B)
    const bool one3 = !e->split_passes;
    const bool fast_gelu = e->gelu_fast >= 0 ? e->gelu_fast != 0 : !(e->split & (SPLIT_ATTN_ANY | SPLIT_MLP | SPLIT_LIN2));
    // The LayerNorm behind proj (norm2) and behind lin2 (the next block's norm1) as a tail of those GEMMs (gemm.hip LnTail): OPT-IN
    // (option "ln_tail" = 1).  Built, bit-identical with the stand-alone kernel, and measured slower (profiles/r05_ln_tail.txt): the
    // panel is normalised by ONE CU, and one CU draws ~20 GB/s from HBM -- 62 us for the 1.3 MB of a panel, against 53 us for a
    // LayerNorm launch that uses all 256.  The 1x-rate modes only (the reference-grade modes want the LayerNorm's lo / MXFP4 outputs),
    // and only where the shapes take the 256 x 320 kernel (at least one full round of tiles: batches of 4 tiles and more at ViT-H).
    const bool ln_tail = !fold && !(e->split & (SPLIT_ATTN_ANY | SPLIT_MLP | SPLIT_LIN2)) && e->ln_tail > 0 &&
                         e->ln_counters && gemm_lntail_ok(M, D, D) && gemm_lntail_ok(M, D, 4 * D);
    bool y_ready = false;          // Y already holds norm1 of the block about to start (written by the previous block's lin2 launch)
    if (fold && n_blocks > 0) {
        CK(e, launch_rowstats_convert(prec, e->X, e->Y, e->STATS, M, D, s));
        CK(e, launch_ln_rowstat(e->STATS, e->ROWSTAT, M, 1e-6f, s));
    }
    // option "range_check": scan an operand tensor right after its producer (same stream)
#define RANGE_SCAN(ptr_, count_) do { if (e->range_check) CK(e, launch_range_scan(prec, (ptr_), (long)(count_), e->range_counter, s)); } while (0)
    // ... of a tensor whose rows carry pad columns (the LayerNorm output on the padded-stride route): the live columns only
#define RANGE_SCAN_ROWS(ptr_, rows_, cols_, ld_) do { if (e->range_check) CK(e, launch_range_scan(prec, (ptr_), (long)(rows_) * (cols_), e->range_counter, s, (cols_), (ld_))); } while (0)
    // padded operand rows for the plain qkv / lin1 launches (they run on the persistent ET kernels at these shapes: gemm_ld_ok)
    const bool pad_ok = e->ldk && e->operand_pad_on && !fold && !ln_tail;
    // outlier-column extension of the plain qkv / lin1 launches (EncBlock::oc_*): not with the folded / tail LayerNorm forms (other producers of Y)
    const bool oc_any = e->outlier_on && e->outlier_columns > 0 && !fold && !ln_tail;
    const bool oc_ok = oc_any && (e->outlier_on & 1);
    int y_ld = D, y_live = D;      // row stride Y currently holds, and how many columns of a row are operand values
    for (int i = 0; i < c.depth && i < n_blocks; ++i) {
        const EncBlock& b = e->blocks[i];
        y_ld = D; y_live = D;
        const bool attn_full = (e->split & SPLIT_ATTN) && i < depth_full;
        // v-third modes (79 / 207): where the outlier columns carry more than half of this block's qkv or proj operand-error mass, the block
        // runs the plain launches with the EXACT f16 lo terms of those columns instead of the MXFP4 lo terms of all columns -- an outlier
        // column shares its fp4 block scale with 31 neighbours (their lo terms quantise to zero, its own keeps ~2 bits).  Measured on
        // heavy-tailed weights (tests/test_outlier_gpu.py): multimask IoU min 0.99848 -> the 1x-rate mode's 0.99906 with the columns treated.
        const bool oc_dominant = oc_any && (e->outlier_on & 5) == 5 && !attn_full && (b.oc_share[0] > 0.5f || b.oc_share[3] > 0.5f);
        const bool sp_attn = attn_full || ((e->split & SPLIT_ATTN_V) && i < depth_v && !oc_dominant), sp_mlp = any_mlp && i < depth_full;
        // v third only: needs the tile mask of the one-launch kernel; other shapes split all of qkv
        const int v_from = (sp_attn && !attn_full && one3 && gemm_split3_ok(M, 3 * D, D) && (2 * D) % 320 == 0) ? 2 * D : 0;
        const bool mx_mlp = e->lo_format == 4 && e->mx_mlp_ready && !e->split_passes && gemm_mx_ok(M, 4 * D, D, D) && gemm_mx_ok(M, D, 4 * D, e->mx_kp_lin2);
        const bool sp_lin2 = (e->split & SPLIT_LIN2) && !sp_mlp && mx_mlp && i < depth_full;     // lin2 alone (lin1 only emits H's MX rows)
        const bool mx_attn = e->lo_format == 4 && e->mx_ready && !e->split_passes && gemm_mx_ok(M, 3 * D, D, D) && (2 * D) % 320 == 0;
        // norm1 + qkv in plain token order for both block kinds; the windowed kernel partitions
        // on the fly and takes k / v of padding positions from the qkv bias
        if (fold) {
            CK(e, launch_gemm_et_fold(prec, e->Y, b.qkv_wf, e->QKV, b.qkv_bf, b.qkv_c, e->ROWSTAT, M, 3 * D, D, false, s));
        } else if (sp_attn && mx_attn) {
            // lo terms on MXFP4 operands: the LayerNorm emits the fp4 codes + scales of its output's hi and lo (no ET lo copy).
            // Outlier columns of norm1's output: the v third is covered by its fp4 correction segments (they span every column); the q and k
            // tiles of the v-third form run plain f16 and read the 64-column extension of the padded rows as one more stage
            const int nocx = (oc_ok && !attn_full && b.oc_n[0] && e->ldk && b.qkv_wp) ? b.oc_n[0] : 0;
            CK(e, launch_layernorm(prec, e->X, b.ln1w, b.ln1b, 1e-6f, e->Y, nullptr, M, D, 0, g, 0, s, nullptr,
                                   e->Y4[0], e->Y4[1], e->SY4[0], e->SY4[1], nocx ? e->ldk : 0, nocx ? b.oc_idx[0] : nullptr, nocx));
            if (nocx) { y_ld = e->ldk; y_live = D + 64; }
            CK(e, launch_gemm_et_mx(prec, e->Y, nocx ? b.qkv_wp : b.qkv_w, e->QKV, b.qkv_b, M, 3 * D, D, D, e->Y4[1], e->Y4[0], e->SY4[1], e->SY4[0],
                                    b.qkv_w4[0], b.qkv_w4[1], b.qkv_s4[0], b.qkv_s4[1], false, false, attn_full ? 0 : 2 * D, s,
                                    false, nullptr, nullptr, nullptr, nullptr, nocx ? e->ldk : 0, nocx != 0));
        } else if (sp_attn) {
            CK(e, launch_layernorm(prec, e->X, b.ln1w, b.ln1b, 1e-6f, e->Y, nullptr, M, D, 0, g, 0, s, e->Ylo));
            if (one3 && gemm_split3_ok(M, 3 * D, D)) {     // one launch, ET output rounded once from the register accumulators
                CK(e, launch_gemm_et_split3(prec, e->Y, e->Ylo, b.qkv_w, b.qkv_w_lo, e->QKV, b.qkv_b, M, 3 * D, D, false, false, s, v_from));
            } else {
                if (!e->F32T) CK(e, dalloc(e, &e->F32T, (size_t)c.max_images * tokens * 4 * D));
                CK(e, launch_gemm_et(prec, e->Ylo, b.qkv_w, e->F32T, nullptr, nullptr, 0, M, 3 * D, D, true, false, false, s));
                CK(e, launch_gemm_et(prec, e->Y, b.qkv_w_lo, e->F32T, nullptr, nullptr, 0, M, 3 * D, D, true, false, true, s));
                CK(e, launch_gemm_et(prec, e->Y, b.qkv_w, e->F32T, b.qkv_b, nullptr, 0, M, 3 * D, D, true, false, true, s));
                CK(e, launch_convert(prec, e->F32T, e->QKV, (long)M * 3 * D, s));
            }
        } else {
            // outlier columns of norm1's output: 64 more K columns (lo | hi of up to 32 columns) in the same launch -- on the padded-stride
            // route in the pad region of the rows (K = D + 64 of the ldk-element rows), elsewhere on dense rows of D + 64 elements
            const int noc = (oc_ok && b.oc_n[0] && b.qkv_wx) ? b.oc_n[0] : 0;
            const int Kq = noc ? D + 64 : D;
            const int ldq = (pad_ok && b.qkv_wp && gemm_ld_ok(M, 3 * D, Kq, false)) ? e->ldk : 0;
            const int ldy = ldq ? ldq : Kq;                                 // row stride of Y for this launch
            if (!y_ready) CK(e, launch_layernorm(prec, e->X, b.ln1w, b.ln1b, 1e-6f, e->Y, nullptr, M, D, 0, g, 0, s, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                 ldy == D ? 0 : ldy, noc ? b.oc_idx[0] : nullptr, noc));
            y_ld = ldy; y_live = Kq;
            const int prev_ld = swap_gemm_ld(ldq);
            const hipError_t qe = launch_gemm_et(prec, e->Y, ldq ? b.qkv_wp : (noc ? b.qkv_wx : b.qkv_w), e->QKV, b.qkv_b, nullptr, 0, M, 3 * D, Kq, false, false, false, s);
            (void)swap_gemm_ld(prev_ld);
            CK(e, qe);
        }
        y_ready = false;
        const bool mx_ao = sp_attn && mx_attn;      // the attention kernels write the proj GEMM's MX operands themselves
        // outlier columns of proj (plain launch only): the attention kernel also writes the lo half of its output, a gather makes A_x
        const int nop = (oc_any && (e->outlier_on & 4) && !sp_attn && b.oc_n[3] && b.oc_bx[3] && e->AOlo && e->OCX) ? b.oc_n[3] : 0;
        const bool ao_lo = (sp_attn && !mx_ao) || nop;
        if (!b.global)
            CK(e, launch_window_attention(prec, e->QKV, b.qkv_b, b.rel_h, b.rel_w, e->AO, n, g, c.window_size, c.num_heads, e->hd, s,
                                          ao_lo ? e->AOlo : nullptr, mx_ao ? e->AO4[0] : nullptr, mx_ao ? e->AO4[1] : nullptr,
                                          mx_ao ? e->SAO4[0] : nullptr, mx_ao ? e->SAO4[1] : nullptr,
                                          // the remainder only feeds the gather of proj's outlier columns: the heads that hold them
                                          (nop && !sp_attn) ? b.oc_heads : 0xffffffffu));
        else
            CK(e, launch_global_attention(prec, e->QKV, b.rel_h, b.rel_w, e->AO, n, g, c.num_heads, e->hd, e->VTG, s,
                                          ao_lo ? e->AOlo : nullptr, mx_ao ? e->AO4[0] : nullptr, mx_ao ? e->AO4[1] : nullptr,
                                          mx_ao ? e->SAO4[0] : nullptr, mx_ao ? e->SAO4[1] : nullptr));
        if (e->range_check) {       // norm1 output, q | k | v, attention output
            RANGE_SCAN_ROWS(e->Y, M, y_live, y_ld);  // the live columns only: pad columns may hold an earlier pass's values
            RANGE_SCAN(e->QKV, (size_t)M * 3 * D);
            RANGE_SCAN(e->AO, (size_t)M * D);
        }
        y_ld = D; y_live = D;
        // lin1 takes the plain launch below exactly when none of these holds; then norm2 writes the padded layout for it
        const bool lin1_plain = !fold && !sp_lin2 && !sp_mlp;
        const int nol = (oc_ok && lin1_plain && b.oc_n[1] && b.lin1_wx) ? b.oc_n[1] : 0;    // outlier columns of norm2's output (see qkv above)
        const int Kl = nol ? D + 64 : D;
        const int ldl = (pad_ok && b.lin1_wp && lin1_plain && gemm_ld_ok(M, 4 * D, Kl, true)) ? e->ldk : 0;
        const int ldy2 = ldl ? ldl : Kl;
        if (fold) {
            CK(e, launch_gemm_et_stats(prec, e->AO, b.proj_w, e->X, b.proj_b, e->Y, e->STATS, M, D, D, s));
            CK(e, launch_ln_rowstat(e->STATS, e->ROWSTAT, M, 1e-6f, s));
        } else {
            if (sp_attn && mx_attn) {
                // the attention kernel wrote hi / lo of its output as fp4 on the per-head padded K axis; the GEMM adds into the residual stream
                CK(e, launch_gemm_et_mx(prec, e->AO, b.proj_w, e->X, b.proj_b, M, D, D, e->mx_kp_proj, e->AO4[1], e->AO4[0], e->SAO4[1],
                                        e->SAO4[0], b.proj_w4[0], b.proj_w4[1], b.proj_s4[0], b.proj_s4[1], true, true, 0, s));
            } else if (sp_attn && one3 && gemm_split3_ok(M, D, D)) {
                CK(e, launch_gemm_et_split3(prec, e->AO, e->AOlo, b.proj_w, b.proj_w_lo, e->X, b.proj_b, M, D, D, true, true, s));
            } else {
                if (sp_attn) {
                    CK(e, launch_gemm_et(prec, e->AOlo, b.proj_w, e->X, nullptr, nullptr, 0, M, D, D, true, false, true, s));
                    CK(e, launch_gemm_et(prec, e->AO, b.proj_w_lo, e->X, nullptr, nullptr, 0, M, D, D, true, false, true, s));
                }
                if (ln_tail && !sp_attn)
                    CK(e, launch_gemm_et_lntail(prec, e->AO, b.proj_w, e->X, b.proj_b, M, D, D, b.ln2w, b.ln2b, 1e-6f, e->Y, e->ln_counters, s));
                else if (nop) {
                    CK(e, launch_outlier_gather(e->AO, e->AOlo, D, b.oc_idx[3], nop, e->OCX, M, s));
                    CK(e, resid_gemm_ext(e, prec, e->AO, b.proj_w, e->OCX, b.oc_bx[3], b.proj_b, M, D, D, s));
                } else
                    CK(e, launch_gemm_et(prec, e->AO, b.proj_w, e->X, b.proj_b, nullptr, 0, M, D, D, true, false, true, s));
            }
            if (ln_tail && !sp_attn) {
                // norm2 came out of the proj launch
            } else if (sp_mlp && mx_mlp)
                CK(e, launch_layernorm(prec, e->X, b.ln2w, b.ln2b, 1e-6f, e->Y, nullptr, M, D, 0, g, 0, s, nullptr, e->Y4[0], e->Y4[1], e->SY4[0], e->SY4[1]));
            else {
                const int ldn = (sp_mlp || ldy2 == D) ? 0 : ldy2;
                CK(e, launch_layernorm(prec, e->X, b.ln2w, b.ln2b, 1e-6f, e->Y, nullptr, M, D, 0, g, 0, s, sp_mlp ? e->Ylo : nullptr, nullptr, nullptr, nullptr,
                                       nullptr, ldn, nol ? b.oc_idx[1] : nullptr, nol));
                y_ld = ldn ? ldn : D; y_live = ldn ? Kl : D;
            }
        }
        hipEvent_t t0 = nullptr, t1 = nullptr;
        if (e->timing) {
            auto get = [&](hipEvent_t* ev) -> hipError_t {
                if (!e->tpool.empty()) { *ev = e->tpool.back(); e->tpool.pop_back(); return hipSuccess; }
                return hipEventCreate(ev);
            };
            CK(e, get(&t0)); CK(e, get(&t1));
            CK(e, hipEventRecord(t0, s));
        }
        if (fold) CK(e, launch_gemm_et_fold(prec, e->Y, b.lin1_wf, e->H, b.lin1_bf, b.lin1_c, e->ROWSTAT, M, 4 * D, D, true, s));
        else if (sp_lin2) {
            // lin1 without lo terms (split_from_n = N: no MX stages; the a4 / b4 operands are not touched), its epilogue emits H's fp4 rows
            CK(e, launch_gemm_et_mx(prec, e->Y, b.lin1_w, e->H, b.lin1_b, M, 4 * D, D, D, e->Y4[1], e->Y4[0], e->SY4[1], e->SY4[0], b.lin1_w4[0],
                                    b.lin1_w4[1], b.lin1_s4[0], b.lin1_s4[1], false, false, 4 * D, s, true, e->H4[0], e->H4[1], e->SH4[0], e->SH4[1]));
        } else if (sp_mlp && mx_mlp) {
            // lo terms on MXFP4: ET output with the exact-erf GELU in the epilogue, which also emits H as fp4 hi / lo for lin2
            CK(e, launch_gemm_et_mx(prec, e->Y, b.lin1_w, e->H, b.lin1_b, M, 4 * D, D, D, e->Y4[1], e->Y4[0], e->SY4[1], e->SY4[0], b.lin1_w4[0],
                                    b.lin1_w4[1], b.lin1_s4[0], b.lin1_s4[1], false, false, 0, s, true, e->H4[0], e->H4[1], e->SH4[0], e->SH4[1]));
        } else if (sp_mlp) {
            if (one3 && gemm_split3_ok(M, 4 * D, D)) {
                CK(e, launch_gemm_et_split3(prec, e->Y, e->Ylo, b.lin1_w, b.lin1_w_lo, e->F32T, b.lin1_b, M, 4 * D, D, true, false, s));
            } else {
                CK(e, launch_gemm_et(prec, e->Ylo, b.lin1_w, e->F32T, nullptr, nullptr, 0, M, 4 * D, D, true, false, false, s));
                CK(e, launch_gemm_et(prec, e->Y, b.lin1_w_lo, e->F32T, nullptr, nullptr, 0, M, 4 * D, D, true, false, true, s));
                CK(e, launch_gemm_et(prec, e->Y, b.lin1_w, e->F32T, b.lin1_b, nullptr, 0, M, 4 * D, D, true, false, true, s));
            }
            CK(e, launch_gelu_split(prec, e->F32T, e->H, e->Hlo, (long)M * 4 * D, s));
        } else {
            // the 1x-rate modes take the cheaper erf in lin1's GELU epilogue (common.h gelu_erf2_et: -3.3 % on the dominant kernel);
            // every mode with a block-GEMM split bit keeps the arithmetic its parity statistics were measured on, bit for bit
            const int prev_form = swap_gelu_form(fast_gelu ? 2 : 1);
            const int prev_ld = swap_gemm_ld(ldl);
            const hipError_t le = launch_gemm_et(prec, e->Y, ldl ? b.lin1_wp : (nol ? b.lin1_wx : b.lin1_w), e->H, b.lin1_b, nullptr, 0, M, 4 * D, Kl, false, true, false, s);
            (void)swap_gemm_ld(prev_ld);
            (void)swap_gelu_form(prev_form);
            CK(e, le);
        }
        // outlier columns of lin2 (plain launches only): the pre-activations of those <= 32 hidden units once more, in fp32, from the
        // LayerNorm output that still sits in Y (rows of stride ldy2) -> exact GELU -> lo | hi = A_x
        const int nol2 = (oc_any && (e->outlier_on & 2) && lin1_plain && b.oc_n[2] && b.oc_bx[2] && e->OCX && b.lin2_ws &&
                          // the side weights carry lin1's own extension columns: Y must hold them in this launch (else they are stale)
                          (b.oc_n[1] == 0 || nol > 0)) ? b.oc_n[2] : 0;
        if (e->timing) {
            CK(e, hipEventRecord(t1, s));
            e->tev.emplace_back(t0, t1);
        }
        if (e->range_check) {       // norm2 output (still in Y: lin1 has read it, nothing has overwritten it) and GELU(lin1)
            RANGE_SCAN_ROWS(e->Y, M, y_live, y_ld);
            RANGE_SCAN(e->H, (size_t)M * 4 * D);
        }
        if (fold) {
            CK(e, launch_gemm_et_stats(prec, e->H, b.lin2_w, e->X, b.lin2_b, e->Y, e->STATS, M, D, 4 * D, s));
            if (i + 1 < c.depth) CK(e, launch_ln_rowstat(e->STATS, e->ROWSTAT, M, 1e-6f, s));
        } else {
            if ((sp_mlp && mx_mlp) || sp_lin2) {
                CK(e, launch_gemm_et_mx(prec, e->H, b.lin2_w, e->X, b.lin2_b, M, D, 4 * D, e->mx_kp_lin2, e->H4[1], e->H4[0], e->SH4[1], e->SH4[0],
                                        b.lin2_w4[0], b.lin2_w4[1], b.lin2_s4[0], b.lin2_s4[1], true, true, 0, s));
            } else if (sp_mlp && one3 && gemm_split3_ok(M, D, 4 * D)) {
                CK(e, launch_gemm_et_split3(prec, e->H, e->Hlo, b.lin2_w, b.lin2_w_lo, e->X, b.lin2_b, M, D, 4 * D, true, true, s));
            } else {
                if (sp_mlp) {
                    CK(e, launch_gemm_et(prec, e->Hlo, b.lin2_w, e->X, nullptr, nullptr, 0, M, D, 4 * D, true, false, true, s));
                    CK(e, launch_gemm_et(prec, e->H, b.lin2_w_lo, e->X, nullptr, nullptr, 0, M, D, 4 * D, true, false, true, s));
                }
                if (ln_tail && !sp_mlp && i + 1 < c.depth && i + 1 < n_blocks) {
                    const EncBlock& nb = e->blocks[i + 1];
                    CK(e, launch_gemm_et_lntail(prec, e->H, b.lin2_w, e->X, b.lin2_b, M, D, 4 * D, nb.ln1w, nb.ln1b, 1e-6f, e->Y, e->ln_counters, s));
                    y_ready = true;
                } else if (nol2) {
                    CK(e, launch_outlier_side_gemm(prec, e->Y, ldy2, b.lin2_ws, b.lin2_sb, M, Kl, e->OCX, s));
                    CK(e, resid_gemm_ext(e, prec, e->H, b.lin2_w, e->OCX, b.oc_bx[2], b.lin2_b, M, D, 4 * D, s));
                } else
                    CK(e, launch_gemm_et(prec, e->H, b.lin2_w, e->X, b.lin2_b, nullptr, 0, M, D, 4 * D, true, false, true, s));
            }
        }
    }
    if (!do_neck) return SAMRS_OK;
    // neck: 1x1 conv -> LN2d -> 3x3 conv -> LN2d   (all channels-last).  Folded path: Y already is ET(X).
    // SPLIT_NECK: both neck convolutions on hi + lo operands (three GEMM passes each; 0.13 % of the encoder FLOPs).  The neck is
    // the last thing in front of the embedding: nothing downstream averages its operand rounding away (error_budget.py:
    // 258 + 259 of 899 class-map pixels at ViT-H).  Scratch: QKV (free after the last block) holds the lo halves.
    const bool sp_neck = (e->split & SPLIT_NECK) != 0;
    uint16_t* lo_buf = e->QKV;
    if (sp_neck) CK(e, launch_convert(prec, e->X, e->Y, (long)M * D, s, lo_buf));
    else if (!(fold && c.depth > 0 && n_blocks >= c.depth)) CK(e, launch_convert(prec, e->X, e->Y, (long)M * D, s));
    RANGE_SCAN(e->Y, (size_t)M * D);       // the RAW residual stream rounded to the operand type: the one operand without a LayerNorm in front
    if (sp_neck && one3p && gemm_split3_ok((int)M, C, D, true)) {
        CK(e, launch_gemm_et_split3(prec, e->Y, lo_buf, e->neck0_w, e->neck0_w_lo, e->N1, nullptr, (int)M, C, D, true, false, s));
    } else {
        CK(e, launch_gemm_et(prec, e->Y, e->neck0_w, e->N1, nullptr, nullptr, 0, M, C, D, true, false, false, s));
        if (sp_neck) {
            CK(e, launch_gemm_et(prec, lo_buf, e->neck0_w, e->N1, nullptr, nullptr, 0, M, C, D, true, false, true, s));
            CK(e, launch_gemm_et(prec, e->Y, e->neck0_w_lo, e->N1, nullptr, nullptr, 0, M, C, D, true, false, true, s));
        }
    }
    CK(e, launch_layernorm(prec, e->N1, W(e, "image_encoder.neck.1.weight"), W(e, "image_encoder.neck.1.bias"), 1e-6f,
                           e->N1e, nullptr, M, C, 0, g, 0, s, sp_neck ? lo_buf : nullptr));
    RANGE_SCAN(e->N1e, (size_t)M * C);
    CK(e, launch_neck_im2col(e->N1e, e->H, n, g, C, s));
    uint16_t* H2lo = e->H + (size_t)M * 9 * C;
    if (sp_neck) CK(e, launch_neck_im2col(lo_buf, H2lo, n, g, C, s));
    if (sp_neck && one3p && gemm_split3_ok((int)M, C, 9 * C, true)) {
        CK(e, launch_gemm_et_split3(prec, e->H, H2lo, e->neck2_w, e->neck2_w_lo, e->N1, nullptr, (int)M, C, 9 * C, true, false, s));
    } else {
        CK(e, launch_gemm_et(prec, e->H, e->neck2_w, e->N1, nullptr, nullptr, 0, M, C, 9 * C, true, false, false, s));
        if (sp_neck) {
            CK(e, launch_gemm_et(prec, H2lo, e->neck2_w, e->N1, nullptr, nullptr, 0, M, C, 9 * C, true, false, true, s));
            CK(e, launch_gemm_et(prec, e->H, e->neck2_w_lo, e->N1, nullptr, nullptr, 0, M, C, 9 * C, true, false, true, s));
        }
    }
    CK(e, launch_layernorm(prec, e->N1, W(e, "image_encoder.neck.3.weight"), W(e, "image_encoder.neck.3.bias"), 1e-6f,
                           nullptr, e->EMB + (size_t)slot0 * tokens * C, M, C, 0, g, 0, s));
    { const int rc = prepare_slot_keys(e, slot0, n, s); if (rc != SAMRS_OK) return rc; }
    if (e->range_check) {
        RANGE_SCAN(e->K0E + (size_t)slot0 * tokens * C, (size_t)n * tokens * C);       // the decoder's layer-0 keys
        if (e->range_check == 2) {
            // fail loudly: this pass (and every earlier one since the last reset) must not have saturated an operand.  Costs a
            // stream synchronisation per encoder pass -- a validation mode for new checkpoints, not the production setting.
            unsigned long long now = 0;
            CK(e, hipStreamSynchronize(s));
            CK(e, hipMemcpy(&now, e->range_counter, sizeof(now), hipMemcpyDeviceToHost));
            const unsigned long long before = e->range_seen;
            e->range_seen = now;
            if (now > before)
                return fail(e, SAMRS_ERR_RANGE, "%llu operand values of this encoder pass saturated the %s range (|x| >= %s): the masks of these "
                            "images are not the reference's.  Use precision bf16 (fp32 exponent range, 8 mantissa bits) for this "
                            "checkpoint, or option \"range_check\" = 1 to count without failing", now - before,
                            prec == PREC_F16 ? "f16" : "bf16", prec == PREC_F16 ? "65504" : "inf");
        }
    }
#undef RANGE_SCAN
    for (int i = 0; i < n; ++i) {
        e->slot_set[slot0 + i] = 1;
        e->slot_split[slot0 + i] = e->split;
        e->slot_depth[slot0 + i] = (e->split & (SPLIT_ATTN | SPLIT_MLP | SPLIT_LIN2)) ? depth_full : (e->split & SPLIT_ATTN_V) ? depth_v : 0;
    }
    return SAMRS_OK;
}

static int encode_stack(samrs_engine_t* e, const uint8_t* images, int n, int in_h, int in_w, int slot0, void* stream,
                        int n_blocks, bool do_neck) {
    if (!e || !images) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_set_images: null argument");
    if (n < 1 || n > e->cfg.max_images) return fail(e, SAMRS_ERR_CAPACITY, "n_images/slot out of range (max_images=%d)", e->cfg.max_images);
    std::vector<const uint8_t*> ptr(n);
    std::vector<int> hs(n, in_h), ws(n, in_w);
    for (int i = 0; i < n; ++i) ptr[i] = images + (size_t)i * (in_h > 0 ? in_h : 0) * (in_w > 0 ? in_w : 0) * 3;
    return encode(e, ptr.data(), hs.data(), ws.data(), n, slot0, stream, n_blocks, do_neck);
}

int samrs_set_images(samrs_engine_t* e, const uint8_t* images, int n, int in_h, int in_w, int slot0, void* stream) {
    return encode_stack(e, images, n, in_h, in_w, slot0, stream, 1 << 30, true);
}

int samrs_set_images_ragged(samrs_engine_t* e, const uint8_t* const* images, const int* in_h, const int* in_w, int n,
                            int slot0, void* stream) {
    return encode(e, images, in_h, in_w, n, slot0, stream, 1 << 30, true);
}

int samrs_debug_encoder_prefix(samrs_engine_t* e, const uint8_t* images, int n, int in_h, int in_w, int n_blocks,
                               float* x_out, void* stream) {
    if (!x_out) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_debug_encoder_prefix: null output");
    const int rc = encode_stack(e, images, n, in_h, in_w, 0, stream, n_blocks, false);
    if (rc) return rc;
    CK(e, hipMemcpyAsync(x_out, e->X, sizeof(float) * (size_t)n * e->tokens * e->D, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SAMRS_OK;
}

int samrs_get_embedding(samrs_engine_t* e, int slot, float* out_chw, void* stream) {
    if (!e || !out_chw) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_get_embedding: null argument");
    if (slot < 0 || slot >= e->cfg.max_images) return fail(e, SAMRS_ERR_CAPACITY, "slot out of range");
    if (!e->slot_set[slot]) return fail(e, SAMRS_ERR_NOT_SET, "An image must be set with .set_image(...) to generate an embedding.");
    ON_DEVICE(e);
    CK(e, launch_transpose_f32(e->EMB + (size_t)slot * e->tokens * e->C, out_chw, e->tokens, e->C, (hipStream_t)stream));
    return SAMRS_OK;
}

// layer-0 image side of slots [slot0, slot0 + n): see the K0F / K0E / KVQ0 members
static int prepare_slot_keys(samrs_engine_t* e, int slot0, int n, hipStream_t s) {
    const int C = e->C, Ci = C / 2, tokens = e->tokens, prec = e->prec;
    const DecLayer& L = e->layers[0];
    for (int i = 0; i < n; ++i) {
        const size_t o = (size_t)(slot0 + i) * tokens * C;
        CK(e, launch_make_keys(prec, e->EMB + o, nullptr, W(e, "prompt_encoder.no_mask_embed.weight"), e->K0F + o, e->K0E + o, 1, tokens, C, s));
    }
    CK(e, launch_gemm_et(prec, e->K0E + (size_t)slot0 * tokens * C, L.kvq_w, e->KVQ0 + (size_t)slot0 * tokens * 3 * Ci, L.kvq_b, L.kvq_pe,
                         tokens, n * tokens, 3 * Ci, C, false, false, false, s));
    return SAMRS_OK;
}

int samrs_set_embedding(samrs_engine_t* e, int slot, const float* emb_chw, void* stream) {
    if (!e || !emb_chw) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_set_embedding: null argument");
    if (!e->finalized) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "weights not finalized");
    if (slot < 0 || slot >= e->cfg.max_images) return fail(e, SAMRS_ERR_CAPACITY, "slot out of range");
    ON_DEVICE(e);
    CK(e, launch_transpose_f32(emb_chw, e->EMB + (size_t)slot * e->tokens * e->C, e->C, e->tokens, (hipStream_t)stream));
    { const int rc = prepare_slot_keys(e, slot, 1, (hipStream_t)stream); if (rc != SAMRS_OK) return rc; }
    e->slot_set[slot] = 1;
    e->slot_split[slot] = -1;
    e->slot_depth[slot] = 0;
    return SAMRS_OK;
}

int samrs_get_slot_info(const samrs_engine_t* e, int slot, int32_t* is_set, int32_t* split, int32_t* split_depth) {
    if (!e) return SAMRS_ERR_BAD_ARG;
    if (slot < 0 || slot >= e->cfg.max_images) return SAMRS_ERR_CAPACITY;
    if (is_set) *is_set = e->slot_set[slot];
    if (split) *split = e->slot_set[slot] ? e->slot_split[slot] : 0;
    if (split_depth) *split_depth = e->slot_set[slot] ? e->slot_depth[slot] : 0;
    return SAMRS_OK;
}

int samrs_reset_image(samrs_engine_t* e, int slot) {
    if (!e) return SAMRS_ERR_BAD_ARG;
    if (slot < 0 || slot >= e->cfg.max_images) return fail(e, SAMRS_ERR_CAPACITY, "slot out of range");
    e->slot_set[slot] = 0;
    return SAMRS_OK;
}

// -------------------------------------------------------------------------------------------------
static int predict_chunk(samrs_engine_t* e, int slot, int n, const float* boxes, const float* point_coords,
                         const int32_t* point_labels, int n_points, const float* mask_input, int multimask,
                         int return_logits, int in_h, int in_w, int orig_h, int orig_w, void* masks_out, float* iou_out,
                         float* lowres_out, void* stream);

// The reference takes any number of prompts per call (its instance drivers pass every object of an image at once,
// main_sam_rbox_mask_instance.py:159-164).  The engine's workspaces hold max_prompts prompts, so a larger call is
// run as consecutive chunks of max_prompts on the same stream, each writing its slice of the caller's buffers;
// results do not depend on the chunking (no cross-prompt arithmetic, no atomics anywhere on the path).
int samrs_predict(samrs_engine_t* e, int slot, int n, const float* boxes, const float* point_coords,
                  const int32_t* point_labels, int n_points, const float* mask_input, int multimask,
                  int return_logits, int in_h, int in_w, int orig_h, int orig_w, void* masks_out, float* iou_out,
                  float* lowres_out, void* stream) {
    if (!e) return SAMRS_ERR_BAD_ARG;
    if (n < 1) return fail(e, SAMRS_ERR_BAD_ARG, "n_prompts must be >= 1");
    // The three multimask tokens need more operand precision than token 0 (C4 fixtures at ViT-H: IoU 0.9983 - 0.9992 at the 1x
    // rate, >= 0.999 from the v-third split on).  The mode an image was encoded in travels with its slot, so a multimask
    // predict on an embedding some single-mask pipeline produced is refused instead of silently answering in that mode.
    if (multimask && !e->allow_reduced && e->grade_multimask && slot >= 0 && slot < e->cfg.max_images && e->slot_set[slot]) {
        const int sm = e->slot_split[slot];
        // the depth the IoU >= 0.999 claim was measured at: every block for the full bits, the leading three quarters for the
        // v-third form (the automatic depths of run_encoder); an embedding whose split reached fewer blocks ("split_depth" set by
        // hand) is not multimask-grade either (round-4 advisor finding: the recorded depth was never consulted)
        const int need_depth = sm < 0 ? 0 : (sm & (SPLIT_ATTN | SPLIT_MLP | SPLIT_LIN2)) ? e->cfg.depth : (3 * e->cfg.depth + 3) / 4;
        if (sm >= 0 && (!(sm & e->grade_multimask) || e->slot_depth[slot] < need_depth ||
                        (e->split & (SPLIT_OI | SPLIT_UP)) != (SPLIT_OI | SPLIT_UP)))
            return fail(e, SAMRS_ERR_PRECISION, "multimask_output=True on an embedding encoded with split=%d over %d of %d blocks (decoder split=%d): "
                        "this model's multimask outputs need a block-GEMM split bit (64 or 16) at its automatic depth and the decoder "
                        "bits 4 | 8 to hold IoU >= 0.999; re-encode the image in the engine's default mode, or set option "
                        "\"allow_reduced\" = 1", sm, e->slot_depth[slot], need_depth, e->split);
    }
    const int cap = e->cfg.max_prompts;
    const size_t nsel = multimask ? 3 : 1;
    const size_t mask_stride = nsel * (size_t)(orig_h > 0 ? orig_h : 0) * (size_t)(orig_w > 0 ? orig_w : 0) * (return_logits ? 4 : 1);
    const int np = point_coords ? n_points : 0;
    for (int off = 0; off < n; off += cap) {
        const int m = (n - off) < cap ? (n - off) : cap;
        const int rc = predict_chunk(
            e, slot, m, boxes ? boxes + (size_t)off * 4 : nullptr, point_coords ? point_coords + (size_t)off * np * 2 : nullptr,
            point_labels ? point_labels + (size_t)off * np : nullptr, n_points,
            mask_input ? mask_input + (size_t)off * 256 * 256 : nullptr, multimask, return_logits, in_h, in_w, orig_h, orig_w,
            masks_out ? (void*)((unsigned char*)masks_out + (size_t)off * mask_stride) : nullptr,
            iou_out ? iou_out + (size_t)off * nsel : nullptr, lowres_out ? lowres_out + (size_t)off * nsel * 256 * 256 : nullptr, stream);
        if (rc) return rc;
    }
    return SAMRS_OK;
}

static int predict_chunk(samrs_engine_t* e, int slot, int n, const float* boxes, const float* point_coords,
                         const int32_t* point_labels, int n_points, const float* mask_input, int multimask,
                         int return_logits, int in_h, int in_w, int orig_h, int orig_w, void* masks_out, float* iou_out,
                         float* lowres_out, void* stream) {
    if (!e->finalized) return fail(e, SAMRS_ERR_BAD_WEIGHTS, "weights not finalized");
    const samrs_config& c = e->cfg;
    if (slot < 0 || slot >= c.max_images) return fail(e, SAMRS_ERR_CAPACITY, "slot out of range");
    if (!e->slot_set[slot]) return fail(e, SAMRS_ERR_NOT_SET, "An image must be set with .set_image(...) before mask prediction.");
    if (n < 1 || n > c.max_prompts) return fail(e, SAMRS_ERR_CAPACITY, "n_prompts=%d exceeds max_prompts=%d", n, c.max_prompts);
    if (!boxes && !point_coords && !mask_input) return fail(e, SAMRS_ERR_BAD_ARG, "at least one prompt (points, boxes or mask_input) is required");
    if (point_coords && !point_labels) return fail(e, SAMRS_ERR_BAD_ARG, "point_labels must be supplied if point_coords is supplied.");
    if (point_coords && (n_points < 1 || n_points > c.max_points)) return fail(e, SAMRS_ERR_CAPACITY, "n_points=%d exceeds max_points=%d", n_points, c.max_points);
    if (in_h < 1 || in_w < 1 || in_h > c.img_size || in_w > c.img_size || orig_h < 1 || orig_w < 1)
        return fail(e, SAMRS_ERR_BAD_SHAPE, "bad input/original size");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(e);
    const int C = e->C, Ci = C / 2, tokens = e->tokens, prec = e->prec, g = e->grid;
    const int npt = point_coords ? n_points + (boxes ? 0 : 1) : 0;
    const int T = 5 + npt + (boxes ? 2 : 0);
    const int BT = n * T;
    const int Mi = n * tokens;
    auto lin = [&](const float* A, int lda, const float* Wt, const float* b, float* Cout, int ldc, int M, int N, int K,
                   bool relu, bool acc) { return launch_gemm_f32(A, lda, Wt, b, Cout, ldc, M, N, K, relu, acc, s); };
    auto lin2 = [&](const float* A, const float* A2, int lda, const float* Wt, const float* b, float* Cout, int ldc, int M,
                    int N, int K) {     // (A + A2) W^T + b
        F32Batch bt{};
        bt.A[0] = A; bt.A2[0] = A2; bt.W[0] = Wt; bt.bias[0] = b; bt.C[0] = Cout;
        return launch_gemm_f32_batch(bt, 1, lda, ldc, M, N, K, false, false, s);
    };
    auto ln_tok = [&](const float* gw, const float* gb) {
        return launch_layernorm(prec, e->Q, gw, gb, 1e-5f, nullptr, e->Q, BT, C, 0, g, 0, s);
    };

    // ---- prompt encoder (prompt_encoder.py:128-173) ----
    PromptParams pp{};
    pp.boxes = boxes; pp.point_coords = point_coords; pp.point_labels = point_labels;
    pp.n_prompts = n; pp.n_points = point_coords ? n_points : 0;
    pp.img_size = (float)c.img_size;
    pp.gauss = W(e, "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix");
    for (int i = 0; i < 4; ++i) pp.point_emb[i] = W(e, "prompt_encoder.point_embeddings." + std::to_string(i) + ".weight");
    pp.not_a_point = W(e, "prompt_encoder.not_a_point_embed.weight");
    pp.iou_token = W(e, "mask_decoder.iou_token.weight");
    pp.mask_tokens = W(e, "mask_decoder.mask_tokens.weight");
    CK(e, launch_prompt_tokens(pp, e->TOK0, e->Q, T, s));       // tokens and (a copy) the initial queries

    const float* emb = e->EMB + (size_t)slot * tokens * C;
    const bool shared0 = (mask_input == nullptr);
    // layer-0 image side of this slot (prepare_slot_keys, at set_image time)
    const float* k0f = e->K0F + (size_t)slot * tokens * C;
    const uint16_t* kvq0 = e->KVQ0 + (size_t)slot * tokens * 3 * Ci;
    if (!shared0) {
        if (!e->DENSE) CK(e, dalloc(e, &e->DENSE, (size_t)c.max_prompts * tokens * C));
        MaskEmbedParams mp{W(e, "prompt_encoder.mask_downscaling.0.weight"), W(e, "prompt_encoder.mask_downscaling.0.bias"),
                           W(e, "prompt_encoder.mask_downscaling.1.weight"), W(e, "prompt_encoder.mask_downscaling.1.bias"),
                           W(e, "prompt_encoder.mask_downscaling.3.weight"), W(e, "prompt_encoder.mask_downscaling.3.bias"),
                           W(e, "prompt_encoder.mask_downscaling.4.weight"), W(e, "prompt_encoder.mask_downscaling.4.bias"),
                           W(e, "prompt_encoder.mask_downscaling.6.weight"), W(e, "prompt_encoder.mask_downscaling.6.bias")};
        CK(e, launch_mask_embed(mp, mask_input, e->DENSE, n, g, s));
        CK(e, launch_make_keys(prec, emb, e->DENSE, nullptr, e->KF, e->KE, n, tokens, C, s));
    }

    // ---- two-way transformer (transformer.py:62-106,151-182) ----
    for (int li = 0; li < 2; ++li) {
        const DecLayer& L = e->layers[li];
        const bool sh = shared0 && li == 0;     // image side still identical for every prompt
        // (1) token self attention: q/k from queries (+ prompt PE after layer 0), v from queries -- one launch
        {
            F32Batch bt{};
            const float* pe = li > 0 ? e->TOK0 : nullptr;
            bt.A[0] = e->Q; bt.A2[0] = pe; bt.W[0] = L.self.qw; bt.bias[0] = L.self.qb; bt.C[0] = e->TQ;
            bt.A[1] = e->Q; bt.A2[1] = pe; bt.W[1] = L.self.kw; bt.bias[1] = L.self.kb; bt.C[1] = e->TK;
            bt.A[2] = e->Q; bt.A2[2] = nullptr; bt.W[2] = L.self.vw; bt.bias[2] = L.self.vb; bt.C[2] = e->TV;
            CK(e, launch_gemm_f32_batch(bt, 3, C, C, BT, C, C, false, false, s));
        }
        CK(e, launch_token_self_attn(e->TQ, e->TK, e->TV, e->TO, n, T, C, 8, s));
        CK(e, lin(e->TO, C, L.self.ow, L.self.ob, e->Q, C, BT, C, C, false, li > 0));
        CK(e, ln_tok(L.n1w, L.n1b));
        // image-side projections for this layer: K_t2i | V_t2i | Q_i2t  (PE folded in as add2d)
        const long bstride = sh ? 0 : tokens;
        const uint16_t* kvq = sh ? kvq0 : e->KVQ;
        if (!sh) CK(e, launch_gemm_et(prec, e->KE, L.kvq_w, e->KVQ, L.kvq_b, L.kvq_pe, tokens, Mi, 3 * Ci, C, false, false, false, s));
        // (2) tokens -> image
        CK(e, lin2(e->Q, e->TOK0, C, L.t2i.qw, L.t2i.qb, e->QP, Ci, BT, Ci, C));
        CK(e, launch_t2i_attention(prec, e->QP, kvq, kvq + Ci, 3 * Ci, bstride, e->O128, e->T2IW, n, T, tokens, Ci, 8, s));
        CK(e, lin(e->O128, Ci, L.t2i.ow, L.t2i.ob, e->Q, C, BT, C, Ci, false, true));
        CK(e, ln_tok(L.n2w, L.n2b));
        // (3) MLP (ReLU)
        CK(e, lin(e->Q, C, L.m1w, L.m1b, e->MH, 2048, BT, 2048, C, true, false));
        CK(e, lin(e->MH, 2048, L.m2w, L.m2b, e->Q, C, BT, C, 2048, false, true));
        CK(e, ln_tok(L.n3w, L.n3b));
        // (4) image -> tokens
        {
            F32Batch bt{};
            bt.A[0] = e->Q; bt.A2[0] = e->TOK0; bt.W[0] = L.i2t.kw; bt.bias[0] = L.i2t.kb; bt.C[0] = e->KT;
            bt.A[1] = e->Q; bt.A2[1] = nullptr; bt.W[1] = L.i2t.vw; bt.bias[1] = L.i2t.vb; bt.C[1] = e->VT;
            CK(e, launch_gemm_f32_batch(bt, 2, C, Ci, BT, Ci, C, false, false, s));
        }
        if (e->decoder_fusion && tokens % 32 == 0) {
            // attention + out_proj + residual + norm4 in one pass over the keys (layer 0 without a mask prompt: the residual
            // is the shared image embedding, batch stride 0)
            // the fp32 copy of the keys is the NEXT layer's residual; after the last layer only the ET copy is read
            // (final t2i projections, upscaler), so its 4 bytes per element are not written
            // SPLIT_OI: attention output and out-projection weights as hi + lo; SPLIT_UP: the last layer also writes the split
            // remainder of the final keys for the first transposed conv
            CK(e, launch_i2t_fused(prec, kvq + 2 * Ci, 3 * Ci, bstride, e->KT, e->VT, L.i2t_ow,
                                   (e->split & SPLIT_OI) ? L.i2t_ow_lo : nullptr, L.i2t.ob, sh ? k0f : e->KF,
                                   sh ? 0 : tokens, L.n4w, L.n4b, 1e-5f, li == 1 ? nullptr : e->KF, e->KE,
                                   (li == 1 && (e->split & SPLIT_UP)) ? e->KE_lo : nullptr, n, T, tokens, Ci, C, s));
        } else {
            CK(e, launch_i2t_attention(prec, kvq + 2 * Ci, 3 * Ci, bstride, e->KT, e->VT, e->OI, n, T, tokens, Ci, 8, s));
            if (sh)
                CK(e, launch_gemm_et(prec, e->OI, L.i2t_ow, e->KF, L.i2t.ob, k0f, tokens, Mi, C, Ci, true, false, false, s));
            else
                CK(e, launch_gemm_et(prec, e->OI, L.i2t_ow, e->KF, L.i2t.ob, nullptr, 0, Mi, C, Ci, true, false, true, s));
            CK(e, launch_layernorm(prec, e->KF, L.n4w, L.n4b, 1e-5f, e->KE, e->KF, Mi, C, 0, g, 0, s));
        }
    }
    // final tokens -> image attention (transformer.py:98-104)
    CK(e, lin2(e->Q, e->TOK0, C, e->fin.qw, e->fin.qb, e->QP, Ci, BT, Ci, C));
    CK(e, launch_gemm_et(prec, e->KE, e->fin_kv_w, e->KVQ, e->fin_kv_b, e->fin_pe, tokens, Mi, 2 * Ci, C, false, false, false, s));
    CK(e, launch_t2i_attention(prec, e->QP, e->KVQ, e->KVQ + Ci, 2 * Ci, tokens, e->O128, e->T2IW, n, T, tokens, Ci, 8, s));
    CK(e, lin(e->O128, Ci, e->fin.ow, e->fin.ob, e->Q, C, BT, C, Ci, false, true));
    CK(e, ln_tok(W(e, "mask_decoder.transformer.norm_final_attn.weight"), W(e, "mask_decoder.transformer.norm_final_attn.bias")));

    // ---- heads (mask_decoder.py:156-172): 4 hypernetwork MLPs + the IoU MLP, layer by layer in one launch each ----
    {
        const std::string hp = "mask_decoder.output_hypernetworks_mlps.", ip = "mask_decoder.iou_prediction_head.layers";
        const size_t hs = (size_t)c.max_prompts * C;
        F32Batch l0{}, l1{}, l2{};
        for (int i = 0; i < 5; ++i) {
            const std::string p = i < 4 ? hp + std::to_string(i) + ".layers" : ip;
            l0.A[i] = i < 4 ? e->Q + (size_t)(1 + i) * C : e->Q;       // mask token i / IoU token of every prompt
            l0.W[i] = W(e, p + ".0.weight"); l0.bias[i] = W(e, p + ".0.bias"); l0.C[i] = e->HY1 + i * hs;
            l1.A[i] = e->HY1 + i * hs;
            l1.W[i] = W(e, p + ".1.weight"); l1.bias[i] = W(e, p + ".1.bias"); l1.C[i] = e->HY2 + i * hs;
            if (i < 4) {
                l2.A[i] = e->HY2 + i * hs;
                l2.W[i] = W(e, p + ".2.weight"); l2.bias[i] = W(e, p + ".2.bias"); l2.C[i] = e->HYPER + i * (C / 8);
            }
        }
        CK(e, launch_gemm_f32_batch(l0, 5, T * C, C, n, C, C, true, false, s));
        CK(e, launch_gemm_f32_batch(l1, 5, C, C, n, C, C, true, false, s));
        CK(e, launch_gemm_f32_batch(l2, 4, C, 4 * (C / 8), n, C / 8, C, false, false, s));
        // IoU head, last layer: only the columns the caller gets (mask_decoder.py:102-107), written straight into its buffer
        if (iou_out) {
            const int s0 = multimask ? 1 : 0, ns = multimask ? 3 : 1;
            CK(e, lin(e->HY2 + 4 * hs, C, W(e, ip + ".2.weight") + (size_t)s0 * C, W(e, ip + ".2.bias") + s0, iou_out, ns, n, ns, C, false, false));
        }
    }
    // ---- upscaler (mask_decoder.py:53-59,154-155) as two GEMMs + fused tail ----
    // A/B knob (timing experiments): SAMRS_DECODER_FUSION=0 runs the un-fused upscaler kernels
    const bool fuse = e->decoder_fusion;
    // SPLIT_UP (fused path only): both transposed convs on hi + lo operands -- ConvT #1 writes its LayerNorm2d + GELU output in
    // fp32 (U1raw), ConvT #2 splits that in registers.  error_budget.py: 376 + 395 of the 899 class-map pixels at ViT-H.
    const bool sp_up = fuse && (e->split & SPLIT_UP) && tokens % 32 == 0;      // KE_lo exists (written by the fused i2t kernel)
    const int sel0 = multimask ? 1 : 0, nsel = multimask ? 3 : 1;     // mask_decoder.py:102-107
    float* low = lowres_out ? lowres_out : e->LOW;
    const bool one_kernel = fuse && e->upscaler_fused && g % 16 == 0;
    if (one_kernel) {
        // both transposed convs, LayerNorm2d, both GELUs and the hypernetwork product in one kernel (upscaler_fused.hip): the
        // [rows][256] intermediate never leaves the CU
        CK(e, launch_upscaler_fused(prec, e->KE, sp_up ? e->KE_lo : nullptr, e->up1_w, sp_up ? e->up1_w_lo : nullptr, e->up1_b, e->up_ln,
                                    e->up2_w, sp_up ? e->up2_w_lo : nullptr, e->up2_b, e->HYPER, low, n, g, 4, sel0, nsel, s));
    } else if (sp_up && Mi % 256 == 0 && (g * g * 4) % 1024 == 0) {
        CK(e, launch_gemm_et_gln(prec, e->KE, e->up1_w, e->U1raw, e->up1_b, e->up_ln, Mi, C, C, s, e->KE_lo, e->up1_w_lo));
    } else if (fuse && Mi % 256 == 0) {   // ConvT #1 as a GEMM with LayerNorm2d(64) + GELU fused into its epilogue
        CK(e, launch_gemm_et_gln(prec, e->KE, e->up1_w, e->U1, e->up1_b, e->up_ln, Mi, C, C, s));
    } else {
        CK(e, launch_gemm_et(prec, e->KE, e->up1_w, e->U1raw, e->up1_b, nullptr, 0, Mi, C, C, true, false, false, s));
        CK(e, launch_group_ln_gelu(prec, e->U1raw, W(e, "mask_decoder.output_upscaling.1.weight"),
                                   W(e, "mask_decoder.output_upscaling.1.bias"), 1e-6f, e->U1, (long)Mi, 4, C / 4, s));
    }
    if (one_kernel) {
        // done above
    } else if (sp_up && Mi % 256 == 0 && (g * g * 4) % 1024 == 0) {
        CK(e, launch_upscale2_masks(prec, e->U1raw, e->up2_w, e->up2_w_lo, e->up2_b, e->HYPER, low, n, g, 4, sel0, nsel, s));
    } else if (fuse && (g * g * 4) % 1024 == 0) {
        CK(e, launch_upscale2_masks(prec, e->U1, e->up2_w, nullptr, e->up2_b, e->HYPER, low, n, g, 4, sel0, nsel, s));
    } else {   // tiny grids (test configurations): GEMM + separate product
        CK(e, launch_gemm_et(prec, e->U1, e->up2_w, e->U2, e->up2_b, nullptr, 0, Mi * 4, C / 2, C / 4, false, true, false, s));
        CK(e, launch_mask_product(prec, e->U2, e->HYPER, low, n, g, 4, sel0, nsel, s));
    }
    // ---- postprocess (sam.py:133-162) + threshold (predictor.py:242-243) ----
    if (masks_out)
        CK(e, launch_postprocess(low, n * nsel, in_h, in_w, orig_h, orig_w, c.img_size, return_logits, masks_out, s));
    return SAMRS_OK;
}

int samrs_paint(samrs_engine_t* e, const uint8_t* masks, const int32_t* labels, int n, int h, int w, uint8_t* seg,
                int64_t* areas, int64_t* cpix, int64_t* cins, int n_classes, void* stream) {
    if (!e || !masks || !labels || n < 1 || h < 1 || w < 1) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_paint: bad argument");
    if ((cpix || cins) && !areas) return fail(e, SAMRS_ERR_BAD_ARG, "samrs_paint: class statistics need areas_out");
    ON_DEVICE(e);
    CK(e, launch_paint(masks, labels, n, h, w, seg, (unsigned long long*)areas, (unsigned long long*)cpix,
                       (unsigned long long*)cins, n_classes, (hipStream_t)stream));
    return SAMRS_OK;
}

void samrs_debug_set_gemm_variant(int v) { set_gemm_variant(v); }
void samrs_debug_set_gemm_skew(int xcd_units, int cu_units) { set_gemm_skew(xcd_units, cu_units); }
int samrs_debug_has_experiments(void) { return gemm_has_experiments() ? 1 : 0; }
int samrs_select_best(samrs_engine_t* e, const uint8_t* masks, const float* iou, int n, int n_sel, int h, int w, uint8_t* best_out,
                      float* quality_out, int64_t* areas_out, void* stream) {
    if (!e || !masks || !iou || !best_out || !quality_out || !areas_out || n < 1 || n_sel < 1 || h < 1 || w < 1)
        return fail(e, SAMRS_ERR_BAD_ARG, "samrs_select_best: bad argument");
    ON_DEVICE(e);
    CK(e, launch_select_best(masks, iou, n, n_sel, h, w, best_out, quality_out, (unsigned long long*)areas_out, (hipStream_t)stream));
    return SAMRS_OK;
}

// per-engine options (see the top of this file)
int samrs_set_option(samrs_engine_t* e, const char* name, int value) {
    if (!e || !name) return SAMRS_ERR_BAD_ARG;
    const std::string n(name);
    if (n == "decoder_fusion") e->decoder_fusion = value != 0;
    else if (n == "ln_fold") {
        if (value && !gemm_has_experiments())
            return fail(e, SAMRS_ERR_BAD_ARG, "ln_fold: the folded LayerNorm (measured slower: DESIGN.md 6) is built with make EXPERIMENTS=1 only");
        e->ln_fold = value != 0;
    }
    else if (n == "split") {
        if (e->finalized && (value & (SPLIT_ATTN_ANY | SPLIT_MLP | SPLIT_LIN2) & ~e->split_ready))
            return fail(e, SAMRS_ERR_BAD_ARG, "split bits 16 / 32 / 64 / 128 (block GEMMs) need their lo weights: set them before the weights are "
                                              "finalized (SAMRS_SPLIT or options={'split': ...})");
        e->split = value & SPLIT_ALL;
    }
    else if (n == "gemm_variant") e->gemm_variant = value;
    else if (n == "upscaler_fused") e->upscaler_fused = value != 0;
    else if (n == "split_passes") e->split_passes = value != 0;
    else if (n == "split_depth") e->split_depth = value > 0 ? value : 0;
    else if (n == "allow_reduced") e->allow_reduced = value != 0;
    else if (n == "gelu_fast") e->gelu_fast = value < 0 ? -1 : (value != 0);
    else if (n == "ln_tail") e->ln_tail = value > 0;
    else if (n == "operand_pad") e->operand_pad_on = value != 0;
    else if (n == "outlier_cols") e->outlier_on = value & 7;
    else if (n == "outlier_ratio_pct") {
        if (e->finalized) return fail(e, SAMRS_ERR_BAD_ARG, "outlier_ratio_pct: the columns are picked when the weights are finalized; set it before");
        if (value < 101) return fail(e, SAMRS_ERR_BAD_ARG, "outlier_ratio_pct must exceed 100 (a column's score as a percentage of the median score)");
        e->outlier_ratio_pct = value;
    }
    else if (n == "range_check") {
        if (value < 0 || value > 2) return fail(e, SAMRS_ERR_BAD_ARG, "range_check is 0 (off), 1 (count) or 2 (count, and samrs_set_images fails)");
        if (value && !e->range_counter) {
            ON_DEVICE(e);
            CK(e, dalloc(e, &e->range_counter, 1));
            CK(e, hipMemset(e->range_counter, 0, sizeof(unsigned long long)));
        }
        if (value == 2 && e->range_check != 2 && e->range_counter) {
            // mode 2 fails the pass that ADDS to the counter: start from what mode 1 has counted so far, or a clean image would be blamed
            ON_DEVICE(e);
            CK(e, hipDeviceSynchronize());
            CK(e, hipMemcpy(&e->range_seen, e->range_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost));
        }
        e->range_check = value;
    }
    else if (n == "saturated") {                    // write = reset (any value)
        if (e->range_counter) {
            ON_DEVICE(e);
            CK(e, hipDeviceSynchronize());
            CK(e, hipMemset(e->range_counter, 0, sizeof(unsigned long long)));
        }
        e->range_seen = 0;
    }
    else if (n == "lo_format") {
        if (value != 0 && value != 4) return fail(e, SAMRS_ERR_BAD_ARG, "lo_format is 0 (f16 lo terms) or 4 (MXFP4 lo terms)");
        if (value == 4 && e->finalized && !e->mx_ready && !e->mx_mlp_ready)
            return fail(e, SAMRS_ERR_BAD_ARG, "lo_format 4 needs the fp4 weight copies: set it (and a block-GEMM split bit) before the weights are "
                                              "finalized; it covers the attention-side split of models whose block GEMMs fit the 256 x 320 tile (ViT-H)");
        e->lo_format = value;
    }
    else return fail(e, SAMRS_ERR_BAD_ARG, "unknown option %s", name);
    return SAMRS_OK;
}
int samrs_get_option(const samrs_engine_t* e, const char* name, int* value) {
    if (!e || !name || !value) return SAMRS_ERR_BAD_ARG;
    const std::string n(name);
    if (n == "decoder_fusion") *value = e->decoder_fusion;
    else if (n == "ln_fold") *value = e->ln_fold;
    else if (n == "split") *value = e->split;
    else if (n == "gemm_variant") *value = e->gemm_variant;
    else if (n == "upscaler_fused") *value = e->upscaler_fused;
    else if (n == "split_passes") *value = e->split_passes;
    else if (n == "split_depth") *value = e->split_depth;
    else if (n == "allow_reduced") *value = e->allow_reduced;
    else if (n == "gelu_fast") *value = e->gelu_fast;
    else if (n == "ln_tail") *value = e->ln_tail;
    else if (n == "operand_pad") *value = e->operand_pad_on;
    else if (n == "outlier_cols") *value = e->outlier_on;
    else if (n == "outlier_ratio_pct") *value = e->outlier_ratio_pct;
    else if (n == "outlier_blocks") *value = e->outlier_blocks;       // read-only
    else if (n == "outlier_columns") *value = e->outlier_columns;     // read-only
    else if (n == "outlier_dominant_blocks") *value = e->outlier_dominant_blocks;   // read-only
    else if (n == "range_check") *value = e->range_check;
    else if (n == "saturated") {                    // synchronizes the device: a diagnostic, not a hot-path call
        unsigned long long c = 0;
        if (e->range_counter) {
            DeviceGuard dg(e->device);
            if (dg.status != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(&c, e->range_counter, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return SAMRS_ERR_HIP;
        }
        *value = c > 0x7fffffffull ? 0x7fffffff : (int)c;
    }
    else if (n == "lo_format") *value = (e->finalized && !e->mx_ready && !e->mx_mlp_ready) ? 0 : e->lo_format;
    else if (n == "grade_multimask") *value = e->grade_multimask;     // read-only
    else return SAMRS_ERR_BAD_ARG;
    return SAMRS_OK;
}

// COCO RLE strings of n masks, packed behind *cursor into `out` (see samrs_hip.h)
int samrs_rle_encode(samrs_engine_t* e, const uint8_t* masks, int n, int h, int w, uint8_t* out, int64_t out_capacity,
                     int64_t* cursor, int64_t* table, void* stream) {
    if (!e || !masks || !out || !cursor || !table || n < 1 || h < 1 || w < 1 || out_capacity < 16)
        return fail(e, SAMRS_ERR_BAD_ARG, "samrs_rle_encode: bad argument");
    if ((size_t)h * w >= (1ull << 30) || w > 8192)
        return fail(e, SAMRS_ERR_BAD_SHAPE, "samrs_rle_encode: mask too large (h * w must be < 2^30, w <= 8192)");
    ON_DEVICE(e);
    hipStream_t s = (hipStream_t)stream;
    const int chunk = 32;                               // masks per pass: bounds the scratch (5.3 MB per 1024^2 mask)
    const size_t need = rle_scratch_bytes(n < chunk ? n : chunk, h, w);
    if (need > e->rle_scratch_bytes) {
        if (e->rle_scratch) {
            CK(e, hipStreamSynchronize(s));
            CK(e, hipFree(e->rle_scratch));               // device-synchronising: nothing still reads the old scratch
            e->rle_scratch = nullptr; e->rle_scratch_bytes = 0;
        }
        CK(e, hipMalloc(&e->rle_scratch, need));
        e->rle_scratch_bytes = need;
    }
    for (int off = 0; off < n; off += chunk) {
        const int m = n - off < chunk ? n - off : chunk;
        CK(e, launch_rle_encode(masks + (size_t)off * h * w, m, h, w, e->rle_scratch, out, (long long)out_capacity,
                                (long long*)cursor, (long long*)table + (size_t)off * 3, s));
    }
    return SAMRS_OK;
}

// test hook: copy (a prefix of) a named internal decoder buffer to a caller device buffer
int samrs_debug_copy_buffer(samrs_engine_t* e, const char* name, void* dst, size_t bytes, void* stream) {
    if (!e || !name || !dst) return SAMRS_ERR_BAD_ARG;
    const std::string n(name);
    const void* src = nullptr;
    if (n == "Q") src = e->Q; else if (n == "TOK0") src = e->TOK0; else if (n == "KF") src = e->KF; else if (n == "KE") src = e->KE;
    else if (n == "KVQ") src = e->KVQ; else if (n == "OI") src = e->OI; else if (n == "U1raw") src = e->U1raw;
    else if (n == "U1") src = e->U1; else if (n == "U2") src = e->U2; else if (n == "HYPER") src = e->HYPER;
    else if (n == "K0F") src = e->K0F; else if (n == "K0E") src = e->K0E; else if (n == "O128") src = e->O128;
    else if (n == "MH") src = e->MH; else if (n == "KT") src = e->KT; else if (n == "VT") src = e->VT; else if (n == "QP") src = e->QP;
    if (!src) return fail(e, SAMRS_ERR_BAD_ARG, "unknown buffer %s", name);
    CK(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SAMRS_OK;
}

int samrs_debug_time_dominant_kernel(samrs_engine_t* e, int enable) {
    if (!e) return SAMRS_ERR_BAD_ARG;
    e->timing = enable != 0;
    return SAMRS_OK;
}

int samrs_debug_dominant_kernel_time(samrs_engine_t* e, float* avg_ms, int* launches, int* M, int* N, int* K) {
    if (!e || !avg_ms || !launches) return SAMRS_ERR_BAD_ARG;
    double tot = 0.0;
    int n = 0;
    for (auto& pr : e->tev) {
        CK(e, hipEventSynchronize(pr.second));
        float ms = 0.f;
        CK(e, hipEventElapsedTime(&ms, pr.first, pr.second));
        tot += ms;
        ++n;
        e->tpool.push_back(pr.first);
        e->tpool.push_back(pr.second);
    }
    e->tev.clear();
    *avg_ms = n ? (float)(tot / n) : 0.f;
    *launches = n;
    if (M) *M = 0;   // rows vary with the batch of each call; the caller knows its batch
    if (N) *N = 4 * e->D;
    if (K) *K = e->D;
    return SAMRS_OK;
}

// ---- kernel-level entry points -------------------------------------------------------------------
#define KRET(expr)                                                                                  \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) fprintf(stderr, "libsamrs_hip: %s: %s\n", __func__, hipGetErrorString(_e)); \
        return _e == hipSuccess ? SAMRS_OK : SAMRS_ERR_HIP;                                         \
    } while (0)

int samrs_k_gemm(int prec, const void* A, const void* B, void* C, const float* bias, const float* add2d, int period,
                 int M, int N, int K, int out_f32, int gelu, int accumulate, void* stream) {
    KRET(launch_gemm_et(prec, A, B, C, bias, add2d, period, M, N, K, out_f32 != 0, gelu != 0, accumulate != 0, (hipStream_t)stream));
}
int samrs_k_gemm_stats(int prec, const void* A, const void* B, float* C, const float* bias, void* xh, float* stats, int M, int N,
                       int K, void* stream) {
    KRET(launch_gemm_et_stats(prec, A, B, C, bias, xh, stats, M, N, K, (hipStream_t)stream));
}
int samrs_k_gemm_fold(int prec, const void* xh, const void* Wf, void* C, const float* bias_f, const float* cvec, const float* rowstat,
                      int M, int N, int K, int gelu, void* stream) {
    KRET(launch_gemm_et_fold(prec, xh, Wf, C, bias_f, cvec, rowstat, M, N, K, gelu != 0, (hipStream_t)stream));
}
int samrs_k_ln_rowstat(const float* stats, float* rowstat, int rows, float eps, void* stream) {
    KRET(launch_ln_rowstat(stats, rowstat, rows, eps, (hipStream_t)stream));
}
int samrs_k_ln_fold_weight(int prec, const float* W, const float* gamma, const float* beta, const float* bias, void* Wf, float* cvec,
                           float* bias_f, int N, int K, void* stream) {
    KRET(launch_ln_fold_weight(prec, W, gamma, beta, bias, Wf, cvec, bias_f, N, K, (hipStream_t)stream));
}
int samrs_k_rowstats_convert(int prec, const float* X, void* xh, float* stats, int rows, int D, void* stream) {
    KRET(launch_rowstats_convert(prec, X, xh, stats, rows, D, (hipStream_t)stream));
}
int samrs_k_gemm_f32(const float* A, int lda, const float* Wt, const float* bias, float* C, int ldc, int M, int N, int K,
                     int relu, int accumulate, void* stream) {
    KRET(launch_gemm_f32(A, lda, Wt, bias, C, ldc, M, N, K, relu != 0, accumulate != 0, (hipStream_t)stream));
}
int samrs_k_convert(int prec, const float* in, void* out, int64_t n, void* stream) {
    KRET(launch_convert(prec, in, out, (long)n, (hipStream_t)stream));
}
int samrs_k_layernorm(int prec, const float* X, const float* gamma, const float* beta, float eps, void* out_et,
                      float* out_f32, int rows_out, int D, int window_mode, int n_images, int grid, int window, void* stream) {
    (void)n_images;
    KRET(launch_layernorm(prec, X, gamma, beta, eps, out_et, out_f32, rows_out, D, window_mode, grid, window, (hipStream_t)stream));
}
int samrs_k_layernorm_mx(int prec, const float* X, const float* gamma, const float* beta, float eps, void* out_et, int rows, int D,
                         void* q_hi, void* q_lo, void* s_hi, void* s_lo, void* stream) {
    KRET(launch_layernorm(prec, X, gamma, beta, eps, out_et, nullptr, rows, D, 0, 0, 0, (hipStream_t)stream, nullptr, q_hi, q_lo, s_hi, s_lo));
}
int samrs_k_window_attention(int prec, const void* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, void* out,
                             int n_images, int grid, int window, int heads, int head_dim, void* stream) {
    KRET(launch_window_attention(prec, qkv, qkv_bias, rel_h, rel_w, out, n_images, grid, window, heads, head_dim, (hipStream_t)stream));
}
// V^T workspace of the two global-attention TEST / BENCH hooks below (the engine owns its own: e->VTG).  One grow-only buffer per
// DEVICE, looked up under a lock.  A buffer that is outgrown is RETIRED, never freed: another thread of the same device may hold the
// old pointer between this function's return and its own kernel launch (the pointer leaves the lock), and a device synchronisation
// only covers launches that are already enqueued.  Buffers at least double, so the retired ones sum to less than the live
// one; these entry points exist for tests/ and tools/ only (samrs_hip_internal.h says so).
static void* kernel_hook_workspace(size_t need) {
    static std::mutex mu;
    static std::map<int, std::pair<void*, size_t>> per_dev;
    static std::vector<void*> retired;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto& w = per_dev[dev];
    if (need > w.second) {
        const size_t want = need > 2 * w.second ? need : 2 * w.second;
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        if (w.first) retired.push_back(w.first);
        w = {p, want};
    }
    return w.first;
}
int samrs_k_global_attention(int prec, const void* qkv, const float* rel_h, const float* rel_w, void* out, int n_images,
                             int grid, int heads, int head_dim, void* stream) {
    // test / bench hook: the V^T workspace the engine owns is a grow-only static here
    const size_t need = (size_t)n_images * heads * head_dim * grid * grid * 2;
    void* ws = kernel_hook_workspace(need);
    if (!ws) return SAMRS_ERR_HIP;
    KRET(launch_global_attention(prec, qkv, rel_h, rel_w, out, n_images, grid, heads, head_dim, ws, (hipStream_t)stream));
}
int samrs_k_attention_mx(int prec, int global, const void* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, void* out,
                         void* out_lo, int n_images, int grid, int heads, int head_dim, void* q_hi, void* q_lo, void* s_hi, void* s_lo,
                         void* stream) {
    if (!global)
        KRET(launch_window_attention(prec, qkv, qkv_bias, rel_h, rel_w, out, n_images, grid, 14, heads, head_dim, (hipStream_t)stream, out_lo,
                                     q_hi, q_lo, s_hi, s_lo));
    const size_t need = (size_t)n_images * heads * head_dim * grid * grid * 2;
    void* ws = kernel_hook_workspace(need);
    if (!ws) return SAMRS_ERR_HIP;
    KRET(launch_global_attention(prec, qkv, rel_h, rel_w, out, n_images, grid, heads, head_dim, ws, (hipStream_t)stream, out_lo, q_hi, q_lo, s_hi, s_lo));
}
int samrs_resample_pass_u8(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* coef, int ksize,
                           int in_len, int out_len, int other, int horizontal, void* stream) {
    if (!in || !out || !bounds || !coef || ksize < 1 || in_len < 1 || out_len < 1 || other < 1) return SAMRS_ERR_BAD_ARG;
    KRET(launch_resample_pass(in, out, bounds, coef, ksize, in_len, out_len, other, horizontal, (hipStream_t)stream));
}
// test hook: the outlier K-columns picked for block GEMM `gemm` (0 qkv, 1 lin1, 2 lin2, 3 proj) of encoder block `block`; returns the count
// (<= 32) and copies the indices (ascending) to the HOST array `out32`, or a negative status
int samrs_debug_outlier_columns(samrs_engine_t* e, int block, int gemm, int32_t* out32) {
    if (!e || !out32 || gemm < 0 || gemm > 3 || block < 0 || block >= (int)e->blocks.size()) return SAMRS_ERR_BAD_ARG;
    const EncBlock& b = e->blocks[block];
    if (b.oc_n[gemm] > 0) {
        ON_DEVICE(e);
        CK(e, hipMemcpy(out32, b.oc_idx[gemm], sizeof(int32_t) * b.oc_n[gemm], hipMemcpyDeviceToHost));
    }
    return b.oc_n[gemm];
}
int samrs_rbox_mask_prompt(const int32_t* pts, int n, int n_vertices, int h, int w, int th, int tw, int img_size, int out_size,
                           float* out, void* stream) {
    if (!pts || !out) return SAMRS_ERR_BAD_ARG;
    KRET(launch_rbox_prompt(pts, n, n_vertices, h, w, th, tw, img_size, out_size, out, (hipStream_t)stream, SAMRS_FILL_CV2_LE_451));
}
int samrs_rbox_mask_prompt_rule(const int32_t* pts, int n, int n_vertices, int h, int w, int th, int tw, int img_size, int out_size,
                                int fill_rule, float* out, void* stream) {
    if (!pts || !out || (fill_rule != SAMRS_FILL_CV2_LE_451 && fill_rule != SAMRS_FILL_CV2_GE_452)) return SAMRS_ERR_BAD_ARG;
    KRET(launch_rbox_prompt(pts, n, n_vertices, h, w, th, tw, img_size, out_size, out, (hipStream_t)stream, fill_rule));
}
int samrs_k_neck_im2col(const void* in, void* A, int n_images, int grid, int C, void* stream) {
    if (!in || !A || n_images < 1 || grid < 1 || C < 8 || C % 8) return SAMRS_ERR_BAD_ARG;
    KRET(launch_neck_im2col(in, A, n_images, grid, C, (hipStream_t)stream));
}
int samrs_k_postprocess(const float* low, int n_masks, int in_h, int in_w, int orig_h, int orig_w, int img_size,
                        int return_logits, void* out, void* stream) {
    KRET(launch_postprocess(low, n_masks, in_h, in_w, orig_h, orig_w, img_size, return_logits, out, (hipStream_t)stream));
}
int samrs_k_gemm_gln(int prec, const void* A, const void* B, void* C, const float* bias, const float* gamma_beta, int M, int N,
                     int K, const void* A_lo, const void* B_lo, void* stream) {
    KRET(launch_gemm_et_gln(prec, A, B, C, bias, gamma_beta, M, N, K, (hipStream_t)stream, A_lo, B_lo));
}
int samrs_k_gemm_split3(int prec, const void* A, const void* A_lo, const void* B, const void* B_lo, void* C, const float* bias,
                        int M, int N, int K, int out_f32, int accumulate, int split_from_n, void* stream) {
    if (!gemm_split3_ok(M, N, K, out_f32 != 0)) return SAMRS_ERR_BAD_SHAPE;
    KRET(launch_gemm_et_split3(prec, A, A_lo, B, B_lo, C, bias, M, N, K, out_f32 != 0, accumulate != 0, (hipStream_t)stream, split_from_n));
}
int samrs_k_upscale2_masks(int prec, const void* u1, const void* w, const void* w_lo, const float* bias, const float* hyper,
                           float* low, int n, int grid, int n_mask_tokens, int sel0, int n_sel, void* stream) {
    KRET(launch_upscale2_masks(prec, u1, w, w_lo, bias, hyper, low, n, grid, n_mask_tokens, sel0, n_sel, (hipStream_t)stream));
}
int samrs_k_upscaler_fused(int prec, const void* keys, const void* keys_lo, const void* w1, const void* w1_lo, const float* b1,
                           const float* ln, const void* w2, const void* w2_lo, const float* b2, const float* hyper, float* low, int n,
                           int grid, int n_mask_tokens, int sel0, int n_sel, void* stream) {
    KRET(launch_upscaler_fused(prec, keys, keys_lo, w1, w1_lo, b1, ln, w2, w2_lo, b2, hyper, low, n, grid, n_mask_tokens, sel0, n_sel,
                               (hipStream_t)stream));
}
int64_t samrs_k_mx_scale_bytes(int rows, int Kp, int is_b) { return (int64_t)mx_scale_bytes(rows, Kp, is_b != 0); }
int samrs_k_mx4_pack(int prec, const float* x, const void* hi_in, const void* lo_in, void* out_hi, void* q_hi, void* q_lo, void* s_hi,
                     void* s_lo, int rows, int K, int G, int GP, int is_b, void* stream) {
    // is_b bit 1: the attention kernels' block-internal element order (launch_mx4_pack perm)
    return launch_mx4_pack(prec, x, hi_in, lo_in, out_hi, q_hi, q_lo, s_hi, s_lo, rows, K, G, GP, (is_b & 1) != 0, (hipStream_t)stream,
                           (is_b & 2) ? 1 : (is_b & 4) ? 2 : 0) == hipSuccess
               ? SAMRS_OK : SAMRS_ERR_BAD_SHAPE;
}
int samrs_k_gemm_mx(int prec, const void* A, const void* B, void* C, const float* bias, int M, int N, int K, int Kp, const void* a4_lo,
                    const void* a4_hi, const void* sa_lo, const void* sa_hi, const void* b4_hi, const void* b4_lo, const void* sb_hi,
                    const void* sb_lo, int out_f32, int accumulate, int split_from_n, void* stream) {
    return launch_gemm_et_mx(prec, A, B, C, bias, M, N, K, Kp, a4_lo, a4_hi, sa_lo, sa_hi, b4_hi, b4_lo, sb_hi, sb_lo, out_f32 != 0,
                             accumulate != 0, split_from_n, (hipStream_t)stream) == hipSuccess ? SAMRS_OK : SAMRS_ERR_BAD_SHAPE;
}
int samrs_k_gemm_mx_gelu_mxout(int prec, const void* A, const void* B, void* C_et, const float* bias, int M, int N, int K, int Kp,
                               const void* a4_lo, const void* a4_hi, const void* sa_lo, const void* sa_hi, const void* b4_hi, const void* b4_lo,
                               const void* sb_hi, const void* sb_lo, int gelu, void* o4_hi, void* o4_lo, void* so_hi, void* so_lo, void* stream) {
    // gelu: bit 0 = GELU in the epilogue, bit 1 = NO tile takes lo terms (split_from_n = N: lin1 of split 207, routed to the persistent
    // plain kernel with the MX-row epilogue)
    return launch_gemm_et_mx(prec, A, B, C_et, bias, M, N, K, Kp, a4_lo, a4_hi, sa_lo, sa_hi, b4_hi, b4_lo, sb_hi, sb_lo, false, false,
                             (gelu & 2) ? N : 0, (hipStream_t)stream, (gelu & 1) != 0, o4_hi, o4_lo, so_hi, so_lo) == hipSuccess
               ? SAMRS_OK : SAMRS_ERR_BAD_SHAPE;
}
int samrs_k_convert_split(int prec, const float* in, void* out_hi, void* out_lo, int64_t n, void* stream) {
    KRET(launch_convert(prec, in, out_hi, (long)n, (hipStream_t)stream, out_lo));
}

}  // extern "C"
