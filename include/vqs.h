/* libvqs_hip -- MI355X (gfx950) native VQAScore hot path for CLIP-FlanT5.  C ABI.
 *
 * The reference (linzhiqiu/t2v_metrics) is pure Python and has no FFI; the forward pass this library
 * replaces is the one HuggingFace `transformers` executes underneath the v3.0 CLIP-FlanT5 wrapper
 * (SURVEY.md §0, §8a).  Each entry point below names the reference-side call it stands in for.
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HIP); tensors are dense row-major;
 *   - bf16 tensors are raw IEEE bfloat16 (uint16_t bits), exactly `torch.bfloat16` storage;
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream);
 *   - functions enqueue work on `stream` and return without synchronising it; the caller syncs;
 *   - the library never allocates device memory: packed weights and workspaces are caller-provided
 *     (sized by the *_bytes queries);
 *   - return value 0 = success, negative = error (VQS_ERR_*); vqs_last_error() gives the message;
 *   - one handle per device/stream; a handle is not re-entrant.
 */
#ifndef VQS_H
#define VQS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQS_OK 0
#define VQS_ERR_INVALID (-1)      /* bad argument / unsupported shape */
#define VQS_ERR_MISSING_WEIGHT (-2)
#define VQS_ERR_WORKSPACE (-3)    /* caller buffer too small */
#define VQS_ERR_HIP (-4)          /* a HIP call or kernel launch failed */
#define VQS_ERR_STATE (-5)        /* e.g. weights not bound */

#define VQS_IMAGE_TOKEN_INDEX (-200) /* /root/reference/t2v_metrics/constants.py:7 */
#define VQS_IGNORE_INDEX (-100)      /* /root/reference/t2v_metrics/constants.py:6 */

typedef struct vqs_handle vqs_handle;

/* Architecture of one CLIP-FlanT5 checkpoint.  Replaces the HF config objects read by
 * `model_cls.from_pretrained` (/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:201,222-229):
 * CLIPVisionConfig (HF models/clip/configuration_clip.py:97-109) and T5Config
 * (HF models/t5/configuration_t5.py:44-62). */
typedef struct vqs_config {
    /* vision tower */
    int32_t vis_hidden;      /* 1024 */
    int32_t vis_layers_run;  /* layers executed = index of hidden_states[-2] = 23 for the 24-layer tower */
    int32_t vis_heads;       /* 16 (head dim must be 64) */
    int32_t vis_mlp;         /* 4096 */
    int32_t vis_patch;       /* 14 */
    int32_t vis_image;       /* 336 */
    float vis_ln_eps;        /* 1e-5 */
    /* T5 */
    int32_t d_model;         /* 2048 (xl) / 4096 (xxl) */
    int32_t n_heads;         /* 32 / 64 */
    int32_t d_kv;            /* 64 (required) */
    int32_t d_ff;            /* 5120 / 10240 */
    int32_t enc_layers;      /* 24 */
    int32_t dec_layers;      /* 24 */
    int32_t vocab;           /* 32128 */
    int32_t rel_buckets;     /* 32 */
    int32_t rel_max_distance;/* 128 */
    float t5_ln_eps;         /* 1e-6 */
} vqs_config;

/* One named bf16 device tensor.  Names are the HF state_dict keys listed by
 * t2v_metrics_amd/weights.py::weight_specs (vision.*, mm_projector.*, T5 keys, lm_head.weight). */
typedef struct vqs_weight_desc {
    const char* name;
    const void* d_data;   /* device, bf16, contiguous */
    int64_t numel;
} vqs_weight_desc;

/* Replaces: constructing the HF modules (mm_utils.py:201) -- no device work. */
int vqs_create(const vqs_config* cfg, vqs_handle** out);
void vqs_destroy(vqs_handle* h);
const char* vqs_last_error(const vqs_handle* h);

/* Replaces: `model.to(device, dtype=torch.bfloat16)` (mm_utils.py:228).  The caller keeps every bound
 * tensor alive.  Fused / re-laid-out copies (QKV concatenation, wi_0|wi_1 interleave, padded patch
 * kernel, bucket LUT) are written into `d_packed` (vqs_packed_bytes() bytes, caller-owned). */
size_t vqs_packed_bytes(const vqs_handle* h);
int vqs_bind_weights(vqs_handle* h, const vqs_weight_desc* weights, int32_t n_weights, void* d_packed,
                     size_t packed_bytes, void* stream);

/* Replaces: CLIPVisionModel.forward(..., output_hidden_states=True).hidden_states[-2][:, 1:]
 * (HF models/clip/modeling_clip.py:613-656) followed by the mlp2x_gelu projector.
 *   d_pixels  bf16 [n_img, 3, image, image]  (already CLIP-normalised)
 *   d_feats   bf16 [n_img, n_patches, d_model]  (output) */
size_t vqs_encode_workspace_bytes(const vqs_handle* h, int32_t n_img);
int vqs_encode_images(vqs_handle* h, const void* d_pixels, int32_t n_img, void* d_feats, void* d_ws, size_t ws_bytes,
                      void* stream);

/* Replaces: embed + splice, T5ForConditionalGeneration.forward(inputs_embeds, attention_mask, labels)
 * (HF models/t5/modeling_t5.py:939-1066) and the scoring tail exp(-CrossEntropyLoss(mean)) of the v3.0
 * wrapper.
 *   d_feats        bf16  [n_img, n_patches, d_model] from vqs_encode_images
 *   d_img_index    int32 [B]      row of d_feats used by pair b
 *   d_input_ids    int32 [B, L]   prompt ids, exactly one VQS_IMAGE_TOKEN_INDEX per row, 0 = right padding
 *   d_labels       int32 [B, T]   answer ids, VQS_IGNORE_INDEX = padding, T <= 16
 *   d_label_logprobs float [B, T] (output) log P(label_t | image, prompt, labels_<t); 0 at ignored positions
 *   d_scores       float [B]      (output) exp(mean of the valid label log-probs)
 * Encoder length S_e = L - 1 + n_patches must be <= 2048 + n_patches. */
size_t vqs_score_workspace_bytes(const vqs_handle* h, int32_t B, int32_t L, int32_t T);
int vqs_score(vqs_handle* h, const void* d_feats, const int32_t* d_img_index, const int32_t* d_input_ids,
              const int32_t* d_labels, int32_t B, int32_t L, int32_t T, float* d_label_logprobs, float* d_scores,
              void* d_ws, size_t ws_bytes, void* stream);

/* Greedy decoding (the reference's `model.generate(images=, texts=)`, V_3.0_README.md:316-325 -> HF GenerationMixin greedy
 * search of T5ForConditionalGeneration, decoder_start_token_id = pad = 0): encoder once, then max_new
 * (<= VQS_MAX_NEW_TOKENS) incremental decoder steps over a self-attention K/V cache held in the workspace, each appending
 * argmax(logits) to d_tokens (int32 [B, max_new], device).  Every step is executed (no host sync); the caller cuts each
 * row at its first EOS (id 1).  Workspace: vqs_generate_workspace_bytes(h, B, L, max_new). */
#define VQS_MAX_NEW_TOKENS 512
size_t vqs_generate_workspace_bytes(const vqs_handle* h, int32_t B, int32_t L, int32_t max_new);
int vqs_generate(vqs_handle* h, const void* d_feats, const int32_t* d_img_index, const int32_t* d_input_ids, int32_t B,
                 int32_t L, int32_t max_new, int32_t* d_tokens, void* d_ws, size_t ws_bytes, void* stream);

/* Byte offset of a named intermediate inside the vqs_score / vqs_encode_images workspace (parity tests read
 * stages through this): "enc_in" fp32 [B,S_e,D], "enc_out" bf16 [B,S_e,D], "dec_out" bf16 [B*T,D] (final decoder norm = lm_head operand), "logits" fp32 [B*T, ld],
 * "vit_hidden" fp32 [n_img, 1+P, hidden] (patch rows = hidden_states[-2]; CLS row lags one sub-layer), "enc_len" int32 [B], "flags" int32[1] (bit 0 = malformed prompt; bit 1 = a pair's label log-probs are not finite: with the fp16 options an activation left the fp16 range -- rerun with vit_fp16=0 / enc_fp16=0).
 * For encode-stage names pass B = n_img, L = T = 0.  Returns -1 for an unknown name.
 * *ld_out (optional) receives the row stride in elements. */
int64_t vqs_workspace_offset(const vqs_handle* h, const char* name, int32_t B, int32_t L, int32_t T, int64_t* ld_out);

/* Kernel timing for bench.py: when enabled, every GEMM launch is bracketed by HIP events on `stream`.
 * vqs_profile_read synchronises the recorded events, returns the number of GEMM launches since the last
 * reset, and fills total GEMM milliseconds and total algorithmic GEMM FLOPs (2*M*N*K). */
int vqs_profile_enable(vqs_handle* h, int32_t on);
int vqs_profile_read(vqs_handle* h, double* gemm_ms, double* gemm_flops, int32_t reset);
/* Algorithmic bytes (operands read once + results written once) of the GEMM launches since the last reset;
 * call before the resetting vqs_profile_read. */
int vqs_profile_bytes(vqs_handle* h, double* gemm_bytes);
/* Text table (one line per GEMM call site: launches, ms, TFLOP/s) of the launches covered by the last vqs_profile_read. */
const char* vqs_profile_report(vqs_handle* h);

/* ---- single-kernel entry points (parity tests and micro-benchmarks call the kernels through these) ---- */
/* epilogue: 0 bf16, 1 bf16+quick_gelu, 2 bf16+erf-gelu, 3 fp32, 4 fp32 + residual, 5 gated gelu_new (W rows
 * interleaved per 32: wi_0 block then wi_1 block; C is [M, N/2]), 6 head-major scatter (q/k/v = C, C+B*H*S*64, ...)
 * variant (bits 0-7): 0 one tile per workgroup, 2 / 5 ping-pong, 3 the library's choice by epilogue and weight shape (engine default: quad form for a 16-bit result with K >= 128, stream form for few rows per batch entry, else the 8-wave persistent kernel), 10 quad form, 11 the 8-wave rule of rounds 1-2 for every launch; 1 / 4 / 6-9 (the lab forms of rounds 1-3, among them the four-wave wide form) are refused; bits 8-15 = gm, bits 16-23 = ns:
 * the workgroup -> tile ORDER (groups of gm M-tiles x all N-tiles, N cut into ns column ranges walked one after the other;
 * 0 = the library's choice by shape).  The order only permutes which workgroup computes a tile when: results are bitwise
 * identical.  Bit 24: result rows leave with the non-temporal hint; bits 25-26: A-panel L2 prefetch of the lock-step kernel
 * (0 by shape, 1 on, 2 off) -- cache-policy hints, bitwise-neutral as well.  Bits 27-28: operand type of a 16-bit-result launch (quad
 * form only): 0 A, W and C bf16; 1 IEEE fp16 A, W and C (bias stays bf16; not the gated epilogue); 2 fp16 A and W, bf16 C (epilogues 0
 * and 5) -- the instantiations options vit_fp16 / enc_fp16 run. */
int vqs_gemm(const void* d_A, const void* d_W, void* d_C, const void* d_bias, const float* d_resid, int32_t M, int32_t N,
             int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t epilogue, int32_t S, int32_t H, int32_t variant,
             void* stream);
/* The fused residual + RMSNorm pieces of the T5 encoder, exposed for the parity tests (persistent variants 3 / 5 / 7):
 *   epilogue 7 (producer): d_hres[M,N] (fp32) += A.W^T in place; d_C bf16 [M,ldc] = d_hres * d_lnw[col] (the NEXT RMSNorm's
 *     operand without its per-row 1/rms); d_rowss_out[ceil(N/256)][M] = per-256-column-tile sums of squares of the rows;
 *   d_rowss_in != NULL (consumer, epilogues 0 / 3 / 5 / 6): accumulator row r is scaled by
 *     rsqrt(sum_p d_rowss_in[p][r] * rs_invd + rs_eps) before the epilogue -- HF modeling_t5.py:59-72 applied after the
 *     contraction instead of before it. */
int vqs_gemm_rms(const void* d_A, const void* d_W, void* d_C, float* d_hres, const void* d_lnw, float* d_rowss_out,
                 const float* d_rowss_in, int32_t rowss_parts, float rs_invd, float rs_eps, int32_t M, int32_t N, int32_t K,
                 int32_t lda, int32_t ldw, int32_t ldc, int32_t epilogue, int32_t S, int32_t H, int32_t variant, void* stream);
int vqs_attention(const void* d_q, const void* d_k, const void* d_v, void* d_out, const float* d_bias_table,
                  const int32_t* d_key_len, int32_t B, int32_t H, int32_t S, float scale, void* stream);
/* Input-pipeline tail: d_u8 uint8 [N,H,W,3] (host-decoded, padded, PIL-resized RGB) -> d_out bf16 [N,3,H,W] =
 * ((x * 1/255) - mean[c]) / std[c] in fp32, rounded to bf16 -- HF CLIPImageProcessor rescale + normalize
 * (models/clip/image_processing_clip.py:22-33) followed by the reference's .to(bfloat16) (mm_utils.py:228).
 * mean3 / std3 are HOST pointers to 3 floats. */
int vqs_normalize_u8(const void* d_u8, void* d_out, int32_t N, int32_t H, int32_t W, const float* mean3, const float* std3,
                     void* stream);

/* Rotary position embedding (rotate-half), in place on d_x bf16 [B,H,S,hd]; d_cos / d_sin fp32 [B*S, half] hold the per-token
 * angles' cos / sin for dims i < half (dim i pairs with i + half) -- HF models/qwen2_5_vl/modeling_qwen2_5_vl.py:153-172
 * (vision, 2-D) and :557-599 (multimodal sections).  Dims >= 2*half are left alone. */
int vqs_rope(void* d_x, const float* d_cos, const float* d_sin, int32_t B, int32_t H, int32_t S, int32_t hd, int32_t half,
             void* stream);
/* Flash attention with head_dim hd = 128, grouped-query heads (query head h reads key/value head h / (H / Hkv)) and an
 * optional causal mask -- the attention of HF Qwen2_5_VLAttention / Qwen2_5_VLVisionAttention
 * (models/qwen2_5_vl/modeling_qwen2_5_vl.py:187-208,602-689; 80-wide tower heads zero-padded to 128).
 * q [B,H,S,hd], k/v [B,Hkv,S,hd] bf16 head-major; out bf16 [B*S, H*hd]; key_len [B] or NULL. */
int vqs_attention_hd(const void* d_q, const void* d_k, const void* d_v, void* d_out, const int32_t* d_key_len, int32_t B,
                     int32_t H, int32_t Hkv, int32_t S, int32_t hd, float scale, int32_t causal, void* stream);
int vqs_decoder_attention(const void* d_q, const void* d_k, const void* d_v, void* d_out, const float* d_bias_table,
                          const int32_t* d_key_len, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ldq, int32_t ldk,
                          int32_t cross, void* stream);
/* d_delta != NULL (bf16 [M,D], the previous sub-layer's output): d_x += d_delta, written back to the fp32 stream,
 * before normalising -- the fused residual update */
int vqs_rmsnorm(float* d_x, const void* d_delta, const void* d_w, void* d_out, int32_t M, int32_t D, float eps,
                void* stream);
int vqs_layernorm(float* d_x, const void* d_delta, const void* d_w, const void* d_b, void* d_out, int32_t out_f32,
                  int32_t M, int32_t D, float eps, void* stream);
/* Deferred-store forms of the fused residual update (both norms of a transformer layer then move 22 instead of 24 bytes
 * per element; same fp32 additions in the same order as two stored updates -- HF modeling_t5.py:140,400,431,
 * modeling_clip.py:366,371):  d_delta2 == NULL, store_x == 0: normalise d_x + d_delta, d_x is NOT modified;
 * d_delta2 != NULL (store_x must be 1): d_x = (d_x + d_delta) + d_delta2, stored, then normalised.
 * kind 0 = RMSNorm (d_b ignored), 1 = LayerNorm; output bf16. */
int vqs_norm_deferred(int32_t kind, float* d_x, const void* d_delta, const void* d_delta2, int32_t store_x, const void* d_w,
                      const void* d_b, void* d_out, int32_t M, int32_t D, float eps, void* stream);
int vqs_score_head(const float* d_logits, int32_t ldl, int32_t V, const int32_t* d_labels, float* d_label_logprobs,
                   float* d_scores, int32_t B, int32_t T, void* stream);
/* host-side bucket function used to build the bias tables (HF models/t5/modeling_t5.py:216-262) */
/* Execution-form options of a handle; the defaults are what the engine ships with, the alternatives compute the same
 * function and exist so that the parity tests can hold each form against the default and the oracle
 * (tests/test_gpu_e2e.py).  The library reads NO environment variables.
 *   "cross_mode"   1 (default) reassociated decoder cross-attention, 0 per-layer K|V projection (what HF executes)
 *   "splitk"       1 (default) split-K for the decoder's skinny nn.Linear GEMMs, 0 single GEMMs
 *   "dec_precise"  1 (default) the scoring decoder holds the activations that matter as split-bf16 / fp32 (16 significant bits
 *                  into the bf16 MFMA as two stacked row planes, fp32 partial sums: round 4), 0 the bf16 decoder
 *                  of rounds 1-3 (vqs_generate always runs that one)
 *   "vit_fp16"     1 (default) the vision tower and the projector hold their 16-bit tensors in IEEE fp16: fp16 copies of their linear weights (made by vqs_bind_weights; bf16 -> fp16 is
 *                  exact for 2^-14 <= |w| < 65 520), fp16 activations through the same kernels on fp16 MFMAs (same rate, same bytes),
 *                  fp32 accumulation / residual stream / statistics unchanged, the feature tensor still bf16 (written by the projector's last GEMM: fp16 operands, bf16 result).  11
 *                  significant bits instead of 8 where the error attribution (profiles/r4_error_attribution.md) puts most of what is
 *                  left of the end-to-end |delta log P| (round 4's 16-pair sample of the benchmarked batch: max 1.45e-3 -> 7.0e-4, mean 7.0e-4 -> 3.3e-4,
 *                  throughput 201.0 -> 200.5 pairs/s; profiles/r4_call18_*).  CLIP was trained in fp16; the T5 stack is not fp16-safe and
 *                  is not touched.  0: bf16 there too, the reference's dtype (mm_utils.py:228) and rounds 1-3's tower.  1 needs
 *                  gemm_variant 3.  The tower's / projector's
 *                  linear weights must be finite and below 65 520 in magnitude (any CLIP checkpoint is: the model was trained in fp16);
 *                  the library does not look -- the Python binding checks at bind time and names the tensors (engine.fp16_unsafe_weights).
 *   "proj_fp16"    1 (default) with vit_fp16 the selected features (hidden_states[-2]: the residual STREAM cast to 16 bits) and the projector's hidden
 *                  tensor are fp16 as well; 0 = those two stay bf16 while the tower's blocks run fp16.  The stream is a sum over every block's
 *                  output: it is the tower site whose bind-time range proof fails first (see "Range safety" below).
 *   Range safety of the three fp16 options (round 6).  The library does not look at weights; the Python binding does, at bind time
 *                  (t2v_metrics_amd/engine.py fp16_range_proof): for every fp16 site of an option it computes a bound of |T| from the weights
 *                  alone -- norm outputs sqrt(D) |g| (+ |b|), norm-fed linears by Cauchy-Schwarz, attention outputs by the value bound, sum-fed
 *                  linears by the row's l1 norm, the tower's stream by the sum of its blocks' output bounds -- and switches an option that is on BY
 *                  DEFAULT off (one warning) where a bound exceeds half of the fp16 maximum: fp16 runs only where no input can overflow it.  An
 *                  option the caller set explicitly is honoured.  Backstop for that case: flags bit 1 (a non-finite label log-prob); the model
 *                  wrapper then re-scores the batch with the fp16 options off instead of raising (clip_t5_model.py score_pairs).
 *   "enc_fp16"     1 (default, round 5) the ATTENTION SIDE of the T5 encoder holds its 16-bit tensors in IEEE fp16: both RMSNorm outputs, q / k / v, the
 *                  softmax probabilities and the attention output; q|k|v, o and the gated wi read fp16 copies of their weights (made by
 *                  vqs_bind_weights: 7.2 GB more packed buffer at XXL) -- what HF's own fp16 T5 path holds in fp16.  The sub-layer outputs (the
 *                  o / wo results added into the fp32 stream), the gated FFN product, the wo GEMM and the encoder's final output stay bf16:
 *                  Flan-T5 leaves the fp16 range in its FFN (HF keeps `wo` in fp32 for that, modeling_t5.py _keep_in_fp32_modules).  Same
 *                  MFMA rate and bytes; measured effect and the attribution behind it: profiles/r5_error_attribution_xxl.md.  0: bf16 there
 *                  too (the reference's dtype, rounds 1-4's encoder).  1 needs gemm_variant 3 and fused_norm 0.  Weight range as for
 *                  vit_fp16 (the Python binding checks at bind time); an activation that leaves the fp16 range turns into a non-finite
 *                  score, which vqs_score reports through its status word (flags bit 1) -- the binding raises and names this option.
 *   "dec_fp16"     1 (default, round 5; effective with dec_precise = 1 and cross_mode = 1) the precise decoder's cross-attention SCORE path and
 *                  what it attends over are IEEE fp16: the encoder's output (its final norm writes fp16), the cross q (from both planes of the
 *                  split norm output, rounded once), q.Wk (fp16 copy of Wk^T, 0.8 GB at XXL), the probabilities; q.Wk, scores and P.E run the
 *                  batched 8-wave / stream kernels' fp16 instantiations, P.E still leaves as a split-bf16 tensor.  The reassociated
 *                  cross-attention makes the roundings of q.Wk and P coherent over all keys: these three bf16 roundings were most of the
 *                  precise decoder's floor (profiles/r5_error_attribution_xxl.md).  0: bf16 there (round 4's precise decoder).  1 needs
 *                  gemm_variant 3.  vqs_generate and the bf16 decoder always read a bf16 encoder output.
 *   "stream_gemm"  1 (default) skinny batched GEMMs (<= 128 rows per entry: the reassociated cross-attention's two products over
 *                  the encoder output) run the HBM-streaming form (csrc/gemm_stream.inc), 0 the persistent 256-row kernel;
 *                  bitwise equal
 *   "norm_defer"   1 (default) deferred store of the fp32 stream in the norm kernels (bitwise equal), 0 store in every norm
 *   "fused_norm"   0 (default) separate add+norm kernels, 1 residual update + RMSNorm operand in the o / wo GEMM epilogues
 *   "gemm_variant" 3 (default) the quad form (four waves of 128x128, 16x16x32 MFMAs; csrc/gemm_quad.inc) for every bf16-result
 *                  launch with one batch entry and K >= 128, the persistent 8-wave kernels for the rest (fp32 results, batched,
 *                  tiny K); 11 the persistent 8-wave kernels everywhere (rounds 1-2's product path); 0 one tile per workgroup;
 *                  2 / 5 the ping-pong schedule.  All of them -- and the stream form (csrc/gemm_stream.inc) -- are bitwise equal:
 *                  a 16x16x32 MFMA adds a k-step's 32 products in the order two 32x32x16 MFMAs do, and every form walks k
 *                  ascending in one accumulator chain (tests/test_gpu_kernels.py holds each against variant 0 with torch.equal;
 *                  a call site nevertheless keeps ONE form for every M, csrc/gemm_quad.inc).  Other values are rejected (the A/B
 *                  forms of rounds 1-3 left the tree in round 4)
 *   per-shape cache-policy names ("tile_order:", "l2_touch:", "nt_store:"): lab switches, see include/vqs_debug.h
 * Returns VQS_ERR_INVALID for an unknown name or value. */
int vqs_set_option(vqs_handle* h, const char* name, int32_t value);
/* the current value of one of the scalar options above (not the per-shape "tile_order:" / "nt_store:" / "l2_touch:" entries); what a
 * checker asks to mirror the arithmetic the handle will run (the rounding-matched oracle, tests/test_gpu_stage_locked.py) */
int vqs_get_option(const vqs_handle* h, const char* name, int32_t* value);

/* Test hooks (stage taps, host-side restatements of the kernels' index arithmetic) and the cache-policy / execution-form lab
 * switches of vqs_set_option are declared and documented in include/vqs_debug.h -- not part of the drop-in boundary. */
/* Host-side arithmetic, no device access: dynamic LDS bytes vqs_attention / vqs_attention_hd request per workgroup for
 * sequence length S (hd 0 / 64 / 128).  The kernels' occupancy hangs on it (160 KiB of LDS per CU in 1 280-B granules);
 * -1 on bad arguments. */
int64_t vqs_attention_lds_bytes(int32_t S, int32_t has_bias, int32_t hd);
int32_t vqs_relpos_bucket(int32_t relative_position, int32_t bidirectional, int32_t num_buckets, int32_t max_distance);

#ifdef __cplusplus
}
#endif
#endif /* VQS_H */
