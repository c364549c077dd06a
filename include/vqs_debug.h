/* vqs_debug.h -- test hooks and lab switches of libvqs_hip.so.  NOT part of the drop-in boundary (include/vqs.h): nothing the
 * reference-side binding of INTEGRATION.md needs is declared here.  The parity tests (tests/test_gpu_stage_locked.py,
 * tests/test_gpu_bench_config.py, tests/test_properties.py) and the lab tools (tools/lab_call.py) are the only callers. */
#ifndef VQS_DEBUG_H
#define VQS_DEBUG_H
#include "vqs.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Stage tap (parity tests): register a caller-owned device buffer for a named intermediate of the NEXT passes; when a
 * pass produces it, it is copied there on the pass's stream (device to device, no synchronisation).  The workspace
 * buffers are reused layer after layer, so this is how a test reads EVERY layer's tensors and checks each launch
 * against the oracle on the engine's own inputs (tests/test_gpu_stage_locked.py).  Names: "<stack>.<layer>.<what>" or
 * "<stack>.<what>": vit.{patch_out,h0,feat_in,pmid}; vit.<i>.{xn0,q,k,v,attn,d_attn,xn1,mid,d_mlp};
 * enc.emb; enc.<i>.{xn0,q,k,v,attn,d_attn,xn1,ff,d_ff}; dec.emb; dec.<i>.{xn0,qkv,sattn,d_self,xn1,cq,cqk,cscores,cprobs,cctx,
 * cattn,d_cross,xn2,ff,d_ff}.  bytes = capacity of d_dst (a pass fails with VQS_ERR_WORKSPACE if it is too small);
 * name == NULL clears every tap, d_dst == NULL removes one.  With no tap registered a pass pays one empty() test. */
int vqs_debug_tap(vqs_handle* h, const char* name, void* d_dst, size_t bytes);
/* Host-side test hook, no device access: the element offsets into a head-major [B, hx, S, hdim] tensor that the GEMM's
 * head-major epilogue (EPI_HEADS, the QKV projections' scatter -- HF modeling_t5.py:311-323 view/transpose) uses for the
 * rows row0 + 8k, k = 0..n-1, computed by the SAME inline functions as the kernel (one division, then steps).  S >= 8. */
int vqs_debug_heads_rows(int32_t row0, int32_t S, int32_t hx, int32_t hdim, int32_t n, int64_t* off_out);
/* Host-side test hook, no device access: the tile every workgroup slot of a persistent GEMM launch (grid workgroups, a multiple
 * of 8) computes under tile order (gm, ns), by the SAME inline functions as the launcher and the kernels: out[4*i .. 4*i+3] =
 * (slot, m0, n0, batch entry) for the M x N x K x batch problem, i < number of tiles.  gm = ns = 0 asks for the library's choice
 * by shape (the Infinity-Cache working-set rule, vqs_kernels.h resolve_tile_order).  Returns the resolved gm | ns << 8 (an
 * illegal ns falls back to 1) or a negative error. */
int vqs_debug_tile_order(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t gm, int32_t ns, int32_t grid, int32_t* out);
/* Host-side test hook, no device access: the kernel family a vqs_gemm launch of this shape resolves to, by the launcher's own
 * function (gemm.hip gemm_form): 10 = the quad form (four waves, 16x16x32 MFMAs; operand offsets relative to the output tile, no
 * size limit), 3 = an 8-wave persistent kernel (32-bit byte offsets into a batch entry's operands), 12 = the stream form (<= 128 rows
 * per batch entry, W streamed from HBM; bitwise the results of 3: one family), 13 = a quad call site's few-row launch in the slim form
 * (gemm_slim.inc, round 6: 128 x 128 tiles for launches whose 256 x 256 tiles leave most CUs idle; bitwise the results of 10: one family),
 * 0 = one tile per workgroup (64-bit pointers), -1 = not launchable.  Two properties the tests pin: the form of a bf16-result launch is a function of the
 * epilogue and the WEIGHT's shape only, never of M (a pair's bits must not depend on its batch), and an operand of 4 GiB or more
 * per batch entry never reaches a 32-bit kernel (it falls back to family 0 where that computes the same function, else -1). */
int vqs_debug_gemm_form(int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t epilogue, int32_t batch, int32_t variant,
                        int32_t S, int32_t inner, int32_t inner_kv);
/* Device-side test hook: the batched form of vqs_gemm that the engine's decoder launches (entry z reads A + z*sA, W + z*sW and
 * writes C + z*sC, strides in elements; EPI_F32 or EPI_BF16 only), with the split-bf16 result store (split_off != 0: the lo plane
 * bf16(acc - hi) goes to C + split_off) and no_stream = 1 keeping the launch off the stream form (csrc/gemm_stream.inc).  The
 * per-kernel tests hold the stream form against the persistent kernel and against per-entry one-tile launches, bitwise. */
int vqs_debug_gemm_batched(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc,
                           int32_t epilogue, int32_t batch, int64_t sA, int64_t sW, int64_t sC, int64_t split_off, int32_t no_stream,
                           int32_t variant, void* stream);
/* Device-side test hooks for the fp16 instantiations (options vit_fp16 / enc_fp16): the single-kernel entry points of vqs.h name bf16
 * tensors; these run the same launches on IEEE fp16 ones so that the per-kernel tests cover the fp16 kernels at shapes of their own
 * (edge tiles, ragged key lengths), not only at the tower's / encoder's shapes through whole passes.
 *   vqs_debug_attention_f16: vqs_attention with q / k / v / out in fp16 (d_bias_table NULL: the vision tower's kernel; non-NULL: the T5
 *     encoder's, with the per-head position-bias table and the key-length mask).
 *   vqs_debug_norm16: vqs_norm_deferred's forms with 16-bit types by `types`: 1 = RMSNorm, bf16 deltas in, fp16 operand out (T5 encoder),
 *     3 = LayerNorm, fp16 deltas in, fp16 operand out (vision tower); d_delta may be NULL (plain norm).  Other combinations: VQS_ERR_INVALID.
 * The fp16 GEMMs are reached through vqs_gemm's variant word (bits 27-28, vqs.h). */
int vqs_debug_attention_f16(const void* d_q, const void* d_k, const void* d_v, void* d_out, const float* d_bias_table, const int32_t* d_key_len,
                            int32_t B, int32_t H, int32_t S, float scale, void* stream);
int vqs_debug_norm16(int32_t kind, float* d_x, const void* d_delta, const void* d_delta2, int32_t store_x, const void* d_w, const void* d_b,
                     void* d_out, int32_t M, int32_t D, float eps, int32_t types, void* stream);
/* Tap window: taps copy only the rows of `count` consecutive outer entries starting at `first` -- pairs for the T5 stacks
 * (rows [first*S, (first+count)*S) of an [B*S, W] tensor, the same fraction of a head-major [B, H, S, 64] or a [B*T, W] one),
 * images for the vision tower (so the window's pairs must reference images first .. first+count-1 in order, as the bench
 * batch does).  Every intermediate is laid out outer-entry-major, so the window is ONE contiguous byte range of each
 * tensor and the tap buffers shrink by B / count: this is how a stage-locked check runs on the BENCHMARKED configuration
 * (XXL, 256 pairs: full taps would need ~290 GB) for a few sampled pairs.  count == 0 restores whole-tensor taps. */
int vqs_debug_tap_window(vqs_handle* h, int32_t first, int32_t count);

/* Lab switches of vqs_set_option (cache-policy hints, bitwise-neutral; the library's rule by shape is the product setting):
 *   "tile_order:<N>x<K>"  value = gm | ns << 8: tile order (see vqs_gemm) of every GEMM of a pass whose weight is [N, K];
 *                  0 removes the entry (back to the library's choice for that shape).  Bitwise-neutral.
 *   "l2_touch:<N>x<K>"    1: the lock-step GEMM prefetches the A panel two K-tiles ahead into L2 for the big launches whose weight is
 *                  [N, K], 2: it does not, 0: the library's rule by shape.  A hint: bitwise-neutral.
 *   "nt_store:<N>x<K>"    1: the results of every GEMM of a pass whose weight is [N, K] are stored with the non-temporal hint,
 *                  2: plain stores, 0: the library's choice for the call site.  A cache-policy hint: bitwise-neutral.
 */

#ifdef __cplusplus
}
#endif
#endif /* VQS_DEBUG_H */
