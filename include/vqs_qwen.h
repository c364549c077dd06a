/* libvqs_hip -- C ABI of the Qwen2.5-VL VQAScore row (BASELINE.json configs[4]).
 *
 * Replaces, for the reference's Qwen wrapper
 * (/root/reference/t2v_metrics/models/vqascore_models/qwen2vl_model.py:110-133 load, :190-230 one prefill per sample via
 * model.generate(max_new_tokens=1, output_scores=True)), the arithmetic HF executes in
 * Qwen2_5_VLForConditionalGeneration (models/qwen2_5_vl/modeling_qwen2_5_vl.py):
 *   vqs_qwen_encode_vision  <- Qwen2_5_VisionTransformerPretrainedModel.forward :408-471 (patch embed, 32 window/full
 *                              attention blocks with 2-D RoPE, patch merger)
 *   vqs_qwen_score          <- get_placeholder_mask/masked_scatter :1094-1232, Qwen2_5_VLTextModel.forward :790-873,
 *                              lm_head on the last position (= scores[0] of generate)
 *   vqs_qwen_prefill/decode <- the same forward with past_key_values (Qwen2_5_VLAttention.forward :653-717), one position per call
 * Same conventions as include/vqs.h: plain device pointers, caller-owned buffers, no allocation, no stream sync, 0 or a
 * negative VQS_ERR_* code, message via vqs_qwen_last_error.  Integer layout work (window permutation, rotary tables from
 * the 3-D positions, placeholder slots) is the caller's: t2v_metrics_amd/qwen/layout.py builds those arrays. */
#ifndef VQS_QWEN_H
#define VQS_QWEN_H
#include <stddef.h>
#include <stdint.h>
#include "vqs.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vqs_qwen_handle vqs_qwen_handle;

typedef struct vqs_qwen_config {
    /* vision tower (Qwen2_5_VLVisionConfig) */
    int32_t v_depth, v_hidden, v_heads, v_mlp, v_patch_dim, v_merge_unit, v_out_hidden;
    int32_t v_fullatt_mask;          /* bit i set: block i attends within a whole frame, else within its window */
    float v_eps;
    /* language model (Qwen2_5_VLTextConfig) */
    int32_t t_vocab, t_hidden, t_layers, t_heads, t_kv_heads, t_mlp;
    float t_eps;
} vqs_qwen_config;

int vqs_qwen_create(const vqs_qwen_config* cfg, vqs_qwen_handle** out);
void vqs_qwen_destroy(vqs_qwen_handle* h);
const char* vqs_qwen_last_error(const vqs_qwen_handle* h);

/* Weights: HF state_dict names ("model.visual.blocks.0.attn.qkv.weight", ...), bf16, device, [out, in] row-major.  The
 * library packs fused / padded copies (language-model heads padded to 128 lanes, gate|up interleaved, K padded to 64) into d_packed
 * and reads the other tensors where they lie: the descs' device pointers must stay valid for the handle's lifetime. */
/* GEMM launch timing for the bench's roofline object (same contract as vqs_profile_* in vqs.h): when enabled, every GEMM
 * launch of the following calls is bracketed by HIP events on the launch stream.  vqs_qwen_profile_read synchronises on
 * them and returns the number of launches, their summed duration (ms), algorithmic FLOPs (2*M*N*K) and operand + result
 * bytes.  The only entry point of this header that synchronises. */
int vqs_qwen_profile_enable(vqs_qwen_handle* h, int32_t on);
int vqs_qwen_profile_read(vqs_qwen_handle* h, double* gemm_ms, double* gemm_flops, double* gemm_bytes, int32_t reset);

/* ---- The range-safe fp16 forms (round 6; option "fp16", default 1 where the model is eligible).
 * The reference runs this model in bf16 (qwen2vl_model.py:110-133, torch_dtype bfloat16); BASELINE.json asks for |delta log P| <= 1e-3 against fp32,
 * which a bf16 prefill misses by 4-6 x (profiles/r5_qwen_error_attribution.md, r6_call1_*: every 16-bit activation class has to carry >= 11 bits).
 * With "fp16" = 1 every 16-bit activation T of the tower, the merger and the prefill is held as IEEE fp16(T * sigma_T) and every GEMM reads fp16
 * copies of the packed weights.  sigma_T = 2^-s is fixed at BIND time from a bound of |T| PROVEN from the weights alone (RMSNorm output <=
 * sqrt(D) |g|; a norm-fed linear by Cauchy-Schwarz; attention output = a convex combination of value rows; |SiLU(g) u| <= |g| |u|; a sum-fed
 * linear by its row's l1 norm: csrc/qwen_decode.hip) so that no input can drive a stored fp16 value beyond half of the fp16 maximum --
 * range-safe by construction, no calibration data.  GEMM epilogues and norm kernels carry the scales (powers of two: exact); accumulation,
 * statistics and softmax stay fp32; the precise tail and the decode step are unchanged (the KV cache stays bf16 in true units).
 * vqs_qwen_bind_weights computes the bounds on the device and SYNCHRONISES THE STREAM ONCE to read them (only with the option on).
 * If a weight does not fit fp16 or a bound is not finite the handle falls back to the bf16 forms (vqs_qwen_get_option "fp16" reads 0, the reason
 * is in vqs_qwen_last_error).  d_merged of vqs_qwen_encode_vision is an OPAQUE 16-bit tensor in the handle's operand format (bf16, or fp16 behind
 * the merger's scale): hand it to vqs_qwen_score / prefill of the same handle under the same option value.
 * "fp16" may be set to 0 / back to 1 on a bound handle (the bf16 forms are always resident); 1 needs the option on when the weights were bound. */
int vqs_qwen_get_option(const vqs_qwen_handle* h, const char* name, int64_t* value);   /* "fp16" (what runs), "fp16_requested", "fp16_eligible", "tail_precise", "rope_fused" (what runs), "x_pitch" */
/* The proof's result: per site the bound of |T| and sigma_T, in the order vision blocks (x0, qkv, dattn, x1, act, dmlp each), merger
 * (norm output, mlp.0 output, merged tokens), language-model layers (six each).  Returns the number of sites (0: no proof made); fills at most cap. */
int vqs_qwen_range_report(const vqs_qwen_handle* h, float* bounds, float* sigmas, int32_t cap);

size_t vqs_qwen_packed_bytes(const vqs_qwen_handle* h);
int vqs_qwen_bind_weights(vqs_qwen_handle* h, const vqs_weight_desc* descs, int32_t n, void* d_packed, size_t packed_bytes,
                          void* stream);

/* Vision tower over N patches of videos/images that share one (t, h, w) grid (HF processor order).  Window blocks run on
 * a WINDOWED layout of Np >= N rows in which every attention window owns win_len slots (partial windows at the right /
 * bottom edge are padded); the full-attention blocks run on the original (frame-contiguous) order.
 *   d_patches    bf16 [N, v_patch_dim]           flattened 2 x 14 x 14 x 3 receptive fields
 *   d_row_map    int32 [Np]                      windowed slot -> source patch row, -1 = padding slot
 *   d_inv_row    int32 [N]                       patch row -> its windowed slot
 *   d_win_valid  int32 [Np / win_len]            real patches per window (they fill the window's first slots)
 *   d_cell_inv   int32 [N / merge_unit]          original merged cell -> windowed cell slot
 *   d_cos_w/d_sin_w  fp32 [Np, head_dim/2]       2-D rotary tables in windowed order;  d_cos_f/d_sin_f [N, ...] original order
 *   frame_len    patches per frame (h * w)
 *   d_merged     16-bit [N / merge_unit, v_out_hidden] out, ORIGINAL cell order (bf16; with option fp16: fp16 behind the merger's scale -- opaque, see above) */
size_t vqs_qwen_vision_workspace_bytes(const vqs_qwen_handle* h, int32_t N, int32_t Np);
int vqs_qwen_encode_vision(vqs_qwen_handle* h, const void* d_patches, int32_t N, const int32_t* d_row_map,
                           const int32_t* d_inv_row, const int32_t* d_win_valid, const int32_t* d_cell_inv, const float* d_cos_w,
                           const float* d_sin_w, const float* d_cos_f, const float* d_sin_f, int32_t Np, int32_t win_len,
                           int32_t frame_len, void* d_merged, void* d_ws, size_t ws_bytes, void* stream);

/* Language-model prefill + last-position logits.
 *   d_input_ids  int32 [B, L] right-padded          d_vis_slot int32 [B, L]: row of d_merged for placeholder tokens, else -1
 *   d_seq_len    int32 [B]                           d_last_row int32 [B] = b*L + seq_len[b] - 1
 *   d_cos/d_sin  fp32 [B*L, head_dim/2]              M-RoPE tables already section-selected per token
 *   d_logits     fp32 [B, t_vocab]                   out */
size_t vqs_qwen_score_workspace_bytes(const vqs_qwen_handle* h, int32_t B, int32_t L);
int vqs_qwen_score(vqs_qwen_handle* h, const void* d_merged, const int32_t* d_input_ids, const int32_t* d_vis_slot,
                   const int32_t* d_seq_len, const int32_t* d_last_row, const float* d_cos, const float* d_sin, int32_t B,
                   int32_t L, float* d_logits, void* d_ws, size_t ws_bytes, void* stream);

/* Generation beyond the first token (the reference's forward with max_new_tokens > 1, qwen2vl_model.py:222-230, and generate(),
 * :495-563, run HF generate with its KV cache): vqs_qwen_prefill is vqs_qwen_score that also keeps every layer's K (after the rotary
 * embedding) and V in a caller-owned cache; vqs_qwen_decode runs ONE further position per sample against it.
 *   d_kv         bf16, vqs_qwen_kv_bytes(B, Lmax): per layer K then V, each [B, t_kv_heads, Lmax, 128]; Lmax >= L + steps to come
 *   decode: d_ids int32 [B] the token entering each sample, d_len int32 [B] its index in the sample's sequence (= tokens cached so
 *   far; the caller advances it), d_cos / d_sin fp32 [B, head_dim/2] rotary table of that position (each M-RoPE axis at the prompt's
 *   last position + 1 + step, as HF generate advances position_ids); d_logits fp32 [B, t_vocab] of the new position.
 * Samples that have stopped may be fed any token: rows are independent. */
/* Largest Lmax the cached decode step accepts (its one-row attention keeps Lmax fp32 scores in LDS); prefill and decode refuse a larger
 * cache with VQS_ERR_INVALID before any launch.  The decode kernels also ignore / clamp a device-side length outside [0, Lmax). */
#define VQS_QWEN_MAX_CACHE_POSITIONS 36864
size_t vqs_qwen_kv_bytes(const vqs_qwen_handle* h, int32_t B, int32_t Lmax);
int vqs_qwen_prefill(vqs_qwen_handle* h, const void* d_merged, const int32_t* d_input_ids, const int32_t* d_vis_slot,
                     const int32_t* d_seq_len, const int32_t* d_last_row, const float* d_cos, const float* d_sin, int32_t B,
                     int32_t L, float* d_logits, void* d_ws, size_t ws_bytes, void* d_kv, size_t kv_bytes, int32_t Lmax, void* stream);
size_t vqs_qwen_decode_workspace_bytes(const vqs_qwen_handle* h, int32_t B);
int vqs_qwen_decode(vqs_qwen_handle* h, const int32_t* d_ids, const int32_t* d_len, const float* d_cos, const float* d_sin, int32_t B,
                    int32_t Lmax, void* d_kv, size_t kv_bytes, float* d_logits, void* d_ws, size_t ws_bytes, void* stream);

/* Test hook (not part of the drop-in boundary): register a caller-owned device buffer for a named intermediate of the NEXT passes;
 * when a pass produces it, it is copied there on the pass's stream.  Names: vis.pre, vis.<i>.{h,xn0,q0,k0,q,k,v,attn,d_attn,xn1,ff,d_mlp},
 * vis.{h_out,xnm,mid,merged_w}; txt.emb, txt.<i>.{h,xn0,q0,k0,q,k,v,attn,d_attn,xn1,ff,d_mlp}, txt.{h_out,xnf}; of vqs_qwen_decode:
 * dec.emb, dec.<i>.{h,xn0,qkv,q,attn,d_attn,xn1,ff,d_mlp}, dec.{h_out,xnf} -- tensors in the ENGINE's
 * layouts (padded windowed rows, 128-lane heads, q0 / k0 before and q / k after the rotary embedding, padded ff width).  name == NULL clears every tap.
 * tests/test_gpu_qwen.py checks every launch of a pass against the rounding-matched oracle through it. */
int vqs_qwen_debug_tap(vqs_qwen_handle* h, const char* name, void* d_dst, size_t bytes);

/* Test hook: "x_pitch" = row pitch (elements, a multiple of 64, >= hidden) of the language model's normalised activations and packed
 * gate|up weight rows; the library's choice is hidden for hidden <= 2048 and the next power of two >= 4096 above that.
 * Must be called before vqs_qwen_bind_weights.  Results do not depend on it.
 * Execution-form switch: "tail_precise" 1 (default, round 5) = vqs_qwen_score / vqs_qwen_prefill take their logits from a PRECISE re-evaluation
 * of every sample's last prompt position -- the one row the reference reads (qwen2vl_model.py:222-301: scores[0] of generate) -- carried
 * layer by layer beside the bf16 prefill with 16 significant bits (split-bf16 operands as stacked rows of the same GEMMs, fp32 partial sums,
 * fp32 q / softmax / sub-layer outputs; attention over the layer's bf16 K / V); 0 = the logits of the bf16 prefill's last row (rounds 2-4).
 * Same function either way; the attribution behind it and the measured effect: profiles/r5_qwen_error_attribution.md.  Any time after create.
 * "fp16": see "The range-safe fp16 forms" above.
 * "rope_fused" 1 (default): under the fp16 forms, with 128-lane language-model heads, the rotary embedding of q / k is applied in
 * the q|k|v GEMM's epilogue -- on the fp32 projection, before its ONE rounding -- and rope_qk_kernel is not launched for the language model (the taps
 * txt.<i>.q0 / k0 then do not exist).  0: the separate kernel on the rounded projection (two roundings; rounds 2-5).  vqs_qwen_get_option reads what runs.
 * The tower's 80-lane heads are not aligned to the epilogue's 128-column blocks and keep the separate kernel. */
int vqs_qwen_debug_option(vqs_qwen_handle* h, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
