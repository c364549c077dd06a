// Internal launcher interface between the C-ABI host code (vqs_api.cpp) and the gfx950 kernels.
// Everything here is device-pointer plumbing; no torch types, no allocation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vqs {

typedef uint16_t bf16_t;   // raw bf16 bits

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C = epilogue(A[M,K] . W[N,K]^T), bf16 operands (both K-contiguous, HF nn.Linear layout), fp32 accumulate.
enum GemmEpilogue : int {
    EPI_BF16 = 0,        // C bf16 [M,ldc] = acc (+bias)
    EPI_BF16_QGELU = 1,  // quick_gelu(acc + bias)           HF activations.py:117-123
    EPI_BF16_GELU = 2,   // erf-GELU(acc + bias)             mlp2x_gelu projector
    EPI_F32 = 3,         // C fp32 [M,ldc] = acc (+bias)
    EPI_F32_RESID = 4,   // C fp32 [M,ldc] = resid + acc (+bias); resid may alias C
    EPI_GATED = 5,       // C bf16 [M,N/2] = gelu_new(acc[wi_0 col]) * acc[wi_1 col]; W rows interleaved in blocks of 32
    EPI_HEADS = 6,       // scatter to head-major [B,H,S,64] tensors: which = col / inner selects heads_out[which]
    EPI_RESID_RMS = 7,   // fused residual update + un-normalised RMSNorm operand (persistent kernels only):
                         //   hres[M,N] (fp32) += acc;  C bf16 [M,ldc] = hres * lnw[col];  rowss_out[n0/256][row] = sum_cols hres^2
    EPI_COUNT = 8
};

struct GemmParams {
    const bf16_t* A;
    const bf16_t* W;
    void* C;
    const bf16_t* bias;      // [N] or nullptr
    const float* resid;      // EPI_F32_RESID
    int M, N, K;
    int lda, ldw, ldc;       // in elements
    // EPI_HEADS
    int S, H, inner;
    int f16 = 0;             // any epilogue; host-side only (selects the kernel): 4 = as 2 and 3 = as 1 with the scaled epilogue (acc_scale / out_scale below); 1 = A, W and the 16-bit result are IEEE fp16 instead of bf16
                             // (quad form only; bias stays bf16) -- the fp16 vision tower; 2 = A and W fp16, the 16-bit result bf16 (plain and
                             // gated epilogue: the T5 encoder's o / wi of option enc_fp16).  Sits in what was alignment padding: no other
                             // field moved, the kernels' argument block is byte for byte what it was.
    bf16_t* heads_out[3];
    int hd = 0;              // head width (0 = 64); 128 for Qwen2.5-VL
    int hd_src = 0;          // columns per head in the GEMM's N when narrower than hd (0 = hd): the product's heads are written into
                             // the first hd_src lanes of hd-lane slots (the caller keeps the other lanes zero); inner / inner_kv count
                             // GEMM columns.  Quad form only.
    int inner_kv = 0;        // width of the k and of the v column ranges (0 = inner); grouped-query models
    int Hkv = 0;             // heads of the k / v tensors (0 = H)
    // EPI_GATED
    int gate_act = 0;        // 0 gelu_new (T5), 1 SiLU (Qwen SwiGLU); bias (if any) is in packed column order
    // batched GEMM (persistent variant only): entry z uses A + z*sA, W + z*sW, C + z*sC (strides in elements)
    int batch = 1;
    long long sA = 0, sW = 0, sC = 0;
    // workgroup -> tile order (gemm.hip tile_of_slot): M-tiles per group and number of N column ranges; 0 = default (8, 1).
    // A permutation of the tile list only -- results do not depend on it.
    int tile_gm = 0, tile_ns = 0;
    // EPI_BF16, 8-wave staged epilogue only: the result leaves as a SPLIT-bf16 tensor -- hi = bf16(acc) at C (as always) and
    // lo = bf16(acc - hi) at C + split_off (elements; 0 = off).  hi + lo carries 16 significant bits of the fp32 accumulator;
    // the consumer stacks the two planes as rows of one GEMM and adds the two fp32 results (the precise decoder, vqs_api.cpp).
    long long split_off = 0;
    int no_stream = 0;       // 1: never the stream form (gemm_stream.inc) nor the slim form (gemm_slim.inc) -- A/B switch; results are bitwise the same either way
    int l2_touch = 0;        // lock-step persistent kernel: L2 prefetch of the A panel two K-tiles ahead; 0 = by shape (gemm.hip), 1 on, 2 off (a hint)
    int nt_store = 0;        // persistent kernels: result rows leave with the non-temporal hint (same bytes; a cache-policy hint)
    // EPI_RESID_RMS (producer side of the fused residual + RMSNorm)
    float* hres = nullptr;           // [M, ldh] fp32 residual stream, read-modified-written
    int ldh = 0;
    const bf16_t* lnw = nullptr;     // [N] weight of the NEXT RMSNorm
    float* rowss_out = nullptr;      // [ceil(N/256)][M] per-tile partial sums of squares of the updated rows
    // consumer side (any epilogue): the A operand is x*lnw un-normalised; scale accumulator row r by
    // rsqrt(sum_p rowss_in[p][r] * rs_invd + rs_eps) before the epilogue proper
    const float* rowss_in = nullptr;
    int rowss_parts = 0;
    float rs_invd = 0.0f, rs_eps = 0.0f;
    // f16 == 3 (quad form only, round 6): fp16 operands and an fp16 result behind power-of-two scales -- the range-safe fp16 forms of the
    // Qwen2.5-VL row (vqs_qwen.cpp: every 16-bit activation T is held as fp16(T * sigma_T), sigma_T = 2^-s chosen at bind time from a proven
    // bound of |T|).  The epilogue computes t = acc * acc_scale + bias (acc_scale = 1 / sigma of the A operand: t is in true units), applies
    // the activation / gate to t and stores fp16(result * out_scale).  Both 1: bitwise gemm_f16_quad.  Appended: no other field moved.
    float acc_scale = 1.0f, out_scale = 1.0f;
    // f16 == 3 with EPI_HEADS, 128-lane heads (round 6): the rotary embedding of q and k inside the epilogue -- per GEMM row
    // (token) a table row of rope_half = 64 cos / sin values (fp32 [M, 64]); lane d < 64 of a q / k head pairs with lane d + 64 (both sit in the same
    // MFMA lane, four blocks apart), rotated in fp32 on the UNROUNDED projection (one rounding instead of two), v heads untouched.  nullptr: no rotation.
    const float* rope_cos = nullptr;
    const float* rope_sin = nullptr;
};

// EPI_HEADS row split, shared by the kernel and its host-side test hook (vqs_debug_heads_rows): GEMM row -> (sample,
// position).  A lane stores 16 rows per tile, 8 apart: one division for the first, then a step -- an integer division by the
// run-time S costs ~28 VALU instructions, 16 of them per lane and tile were 460 of the epilogue's 1 278 (ISA).
#if defined(__HIPCC__)
#define VQS_HD __host__ __device__
#else
#define VQS_HD
#endif
// Row `row` of the GEMM is position hs = row % S of sample hb = row / S; its head-major destination is element
// (hb * hx * S + hs) * hdim of the target tensor ([B, hx, S, hdim], this head's base already added).
VQS_HD inline void heads_off_first(int row, int S, int hx, int hdim, int& hs, long long& off) {
    const int hb = row / S;
    hs = row - hb * S;
    off = ((long long)hb * hx * S + hs) * hdim;
}
// row += 8 (requires S >= 8: at most one sample boundary per step); wrap = (hx - 1) * S * hdim
VQS_HD inline void heads_off_step8(int S, int hdim, long long wrap, int& hs, long long& off) {
    hs += 8;
    off += 8 * hdim;
    if (hs >= S) {
        hs -= S;
        off += wrap;
    }
}

static constexpr int GEMM_BM = 256, GEMM_BN = 256;   // output tile of every GEMM kernel

// ----------------------------------------------------------------------------------------------------
// Workgroup slot -> output tile.  Slot `pid` runs on XCD pid % 8 (hardware round-robin), so each XCD is given a CONTIGUOUS run
// of the tile order below and its 32 concurrent workgroups work on neighbouring tiles that share panels in that XCD's L2.
// Order of the tiles of one batch entry: N is cut into `ns` column ranges walked one after the other IN TIME (every XCD does
// its share of range 0, then of range 1, ...); inside a range, groups of `gm` M-tiles x all N-tiles of the range, M fastest.
// gm = 8, ns = 1 is the map of rounds 1-2 (an 8 x 4 window of concurrent tiles per XCD).  (gm, ns) only permute WHICH
// workgroup computes a tile WHEN: every tile's arithmetic is unchanged, results are bitwise identical for any choice
// (test_gemm_tile_order_is_a_permutation, test_gemm_persistent_many_tiles).  What they steer is the working set that has to
// stay in the 256 MB Infinity Cache while the chip sweeps W: 8 XCDs x (gm M-tiles x K) of A plus (N / ns x K) of W.
// ns > 1 requires tiles_m * (tiles_n / ns) % 8 == 0 and tiles_n % ns == 0 (the launcher checks; no remainders to balance).
// ----------------------------------------------------------------------------------------------------
VQS_HD inline void tile_of_slot(int pid, int nwg, int tiles_m, int tiles_n, int gm, int ns, int& m0, int& n0, int& bz) {
    const int xcd = pid & 7, local = pid >> 3;
    const int tiles_pb = tiles_m * tiles_n;
    int t_lin, n_lo = 0, cols = tiles_n;
    if (ns > 1) {                                          // batch == 1, no remainders (launcher)
        cols = tiles_n / ns;
        const int per_xcd = (tiles_m * cols) >> 3;          // tiles of one range per XCD
        const int part = local / per_xcd;
        t_lin = xcd * per_xcd + (local - part * per_xcd);
        n_lo = part * cols;
        bz = 0;
    } else {
        const int q = nwg >> 3, r = nwg & 7;
        t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        bz = t_lin / tiles_pb;                              // batched GEMM: consecutive tiles stay in one batch entry
        t_lin -= bz * tiles_pb;
    }
    const int width = gm * cols;
    const int group = t_lin / width;
    const int first_m = group * gm;
    const int gsz = tiles_m - first_m < gm ? tiles_m - first_m : gm;
    const int x = t_lin - group * width;
    m0 = (first_m + x % gsz) * GEMM_BM;
    n0 = (n_lo + x / gsz) * GEMM_BN;
}


// Tile order of a launch (see tile_of_slot): an explicit (tile_gm, tile_ns) of the caller is honoured where it is legal;
// 0 selects the library's choice by shape.  That choice follows the working set the chip holds while it sweeps W --
// 8 XCDs x (gm M-tiles x K) of A plus (N / ns x K) of W, in bf16 -- against the 256 MB Infinity Cache.  Measured on two
// MI355X boxes (profiles/r2_call25_*.jsonl, r2_call26_*.jsonl: isolated sweeps of gm x ns per shape, interleaved in-situ A/B):
//   * (8, 1), the map of rounds 1-2, while that working set is <= 180 MB: every ViT and T5-XL shape but wo, T5-XXL o;
//   * otherwise groups of 4 M-tiles AND two column ranges -- T5-XXL wi (302 -> 151 MB) +1.1 / +1.8 %, qkv (235 -> 117 MB)
//     +1.5 / +2.1 %, wo (K = 10 240: 420 -> 210 MB) +3.6 / +2.9 % in isolation; +0.4 ... +1.0 % per XXL step in situ;
//   * one column range when N has fewer than 16 tiles (T5-XL wo, N = 2 048: (4, 1) +0.6 % in situ, (4, 2) -2 %).
//   16-M-tile groups lose 10 % at every shape (a 16 x 2 window of concurrent tiles per XCD re-fetches more), 3 / 5 / 6-tile
//   groups and four column ranges are never better than (4, 2).
// ns > 1 needs equal column ranges and equal per-XCD shares (no remainders), one batch entry and enough tiles for the
// persistent grid; otherwise 1.
inline double tile_order_working_set_mb(int N, int K, int gm, int ns) {
    return (8.0 * gm * GEMM_BM * (double)K * 2.0 + ((double)N / ns) * (double)K * 2.0) / 1.0e6;
}
inline void resolve_tile_order(GemmParams& p, int persistent_wgs) {
    const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM, tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;
    int gm = p.tile_gm, ns = p.tile_ns;
    if (gm <= 0 && ns <= 0 && p.batch <= 1 && tiles_m * tiles_n >= 8 * persistent_wgs) {      // the library's choice (big launches only)
        gm = tile_order_working_set_mb(p.N, p.K, 8, 1) <= 180.0 ? 8 : 4;
        ns = (gm == 4 && tiles_n / 2 >= 8) ? 2 : 1;                 // a column range holds at least one 4 x 8 window's width
    }
    if (gm <= 0) gm = 8;
    if (ns <= 0) ns = 1;
    if (gm > 64) gm = 64;
    if (ns > 1 && (p.batch > 1 || (tiles_n % ns) != 0 || ((tiles_m * (tiles_n / ns)) & 7) != 0 || tiles_m * tiles_n < 8 * persistent_wgs)) ns = 1;
    p.tile_gm = gm;
    p.tile_ns = ns;
}


// variant 0 = direct-to-LDS (global_load_lds) staging; variant 1 = register-staged (debug / A-B)
int gemm_form(const GemmParams& p, int epilogue, int variant);   // kernel family a launch resolves to (gemm.hip), host arithmetic
bool gemm_takes_slim(const GemmParams& p, int epilogue, int variant);   // a quad call site (form 10) whose launch runs the few-row slim form (gemm_slim.inc: same bits)
hipError_t launch_gemm(const GemmParams& p, int epilogue, int variant, hipStream_t stream);

// ---------------------------------------------------------------- attention (attn.hip)
struct AttnParams {
    const bf16_t* q;          // [B,H,S,64]
    const bf16_t* k;          // [B,H,S,64]
    const bf16_t* v;          // [B,H,S,64]
    bf16_t* out;              // [B*S, H*64] token-major
    const float* bias_table;  // [H, 2*S-1] fp32 (index = key - query + S - 1) or nullptr
    const int* key_len;       // [B] valid keys per sample or nullptr (= S)
    int B, H, S;
    float scale;
    // generalised kernel (Qwen2.5-VL row): hd = 128 selects it; q [B,H,S,hd], k/v [B,Hkv,S,hd], out [B*S, H*hd]
    int hd = 0;               // 0 / 64: the kernels above
    int Hkv = 0;              // key/value heads (0 = H); query head h reads head h / (H / Hkv)
    int causal = 0;           // key <= query
    int out_hd = 0;           // hd = 128 only: lanes of a head written to `out`, which is then [B*S, H*out_hd] (0 = hd); the Qwen2.5-VL
                              // tower's 80-lane heads leave compact so that the proj GEMM contracts over 1280, not 2048
    int f16 = 0;              // hd = 64 only: q / k / v / out are IEEE fp16 tensors instead of bf16 (the fp16 vision tower; with a bias table: the T5 encoder of option enc_fp16)
};
hipError_t launch_attention(const AttnParams& p, hipStream_t stream);
size_t attention_lds_bytes(int S, bool has_bias, int hd);   // dynamic LDS request of that launch (host-side arithmetic)

// decoder attention (T <= 16 query rows per sample; fp32 VALU): self (causal + bucket bias) and cross.
struct DecAttnParams {
    const bf16_t* q;          // [B*T, ldq] token-major; head h at columns h*64
    const bf16_t* k;          // self: [B*T, ldk] token-major; cross: [B,H,S,64] head-major
    const bf16_t* v;
    bf16_t* out;              // [B*T, H*64]
    const float* bias_table;  // self: [H, T] fp32 indexed by (query - key); cross: nullptr
    const int* key_len;       // cross: [B]; self: nullptr
    int B, H, T, S;           // S = number of keys (T for self)
    int ldq, ldk;
    int cross;
    // incremental decoding over a K/V cache (self form only; 0 = the teacher-forced defaults)
    long long kv_stride_b = 0;   // elements between the K/V rows of consecutive samples (default T * ldk)
    int qpos0 = 0;               // decoder position of query row 0 (causal mask and bias use qpos0 + t - key)
    int bias_ld = 0;             // row length of bias_table (default T)
    // precise self form (teacher-forced only): q / k / v are FP32 [B*T, ldq] (the pointers above reinterpreted), the output is a
    // split-bf16 tensor: hi plane at out, lo plane at out + out_plane (elements)
    int precise = 0;
    long long out_plane = 0;
};
hipError_t launch_decoder_attention(const DecAttnParams& p, hipStream_t stream);

// ---------------------------------------------------------------- norms + glue (elementwise.hip)
// delta != nullptr: x += delta (fp32, written back) first -- the fused residual update of the previous sub-layer
// delta2 / store_x select the deferred-store forms (elementwise.hip): (delta, store_x=false) normalises x + delta without
// writing the stream; (delta, delta2) stores x = (x + delta) + delta2.
hipError_t launch_rmsnorm(float* x, const bf16_t* delta, const bf16_t* w, bf16_t* out, int M, int D, float eps,
                          hipStream_t s, const bf16_t* delta2 = nullptr, bool store_x = true, int out_ld = 0, bool out_f16 = false);
// out_ld: row pitch of out (0 = D); out_f16: the operand leaves as IEEE fp16 (deltas stay bf16): the T5 encoder of option enc_fp16
hipError_t launch_layernorm(float* x, const bf16_t* delta, const bf16_t* w, const bf16_t* b, void* out, int out_f32, int M,
                            int D, float eps, hipStream_t s, const bf16_t* delta2 = nullptr, bool store_x = true, bool f16 = false);
// f16: the deltas and the 16-bit output are IEEE fp16 tensors (the fp16 vision tower); w / b stay bf16
// pixels bf16 [N,3,IMG,IMG] -> rows [N*G*G, Kpad] in (c,ky,kx) order, zero padded
hipError_t launch_im2col(const bf16_t* pixels, bf16_t* out, int N, int img, int patch, int kpad, hipStream_t s);
// hidden[n, 0] = cls + pos[0]; hidden[n, 1+p] = patch_out[n*P+p] + pos[1+p]   (fp32 out)
hipError_t launch_vit_assemble(const float* patch_out, const bf16_t* cls, const bf16_t* pos, float* hidden, int N,
                               int P, int D, hipStream_t s);
// stream fp32 [N, 1+P, D] -> bf16 [N*P, D] dropping the CLS row
hipError_t launch_drop_cls_cast(float* hidden, const bf16_t* delta, bf16_t* out, int N, int P, int D,
                                hipStream_t s, bool f16 = false, int out_f16 = -1, float out_scale = 1.0f);   // f16: `delta` is an IEEE fp16 tensor; out_f16: is `out` (-1 = as delta); out = 16-bit(value * out_scale)
// n 16-bit elements (n % 8 == 0, 16-byte aligned): bf16 -> fp16 (to_f16) or fp16 -> bf16, round-to-nearest-even
hipError_t launch_cast16(const bf16_t* in, bf16_t* out, size_t n, bool to_f16, hipStream_t s);
// per sample: sentinel position and spliced length from int32 ids [B,L] (pad = 0 trailing, sentinel = -200)
hipError_t launch_prompt_scan(const int* ids, int B, int L, int P, int* sent_pos, int* enc_len, int* err_flag,
                              hipStream_t s);
// inputs_embeds fp32 [B, S_e, D]
hipError_t launch_embed_splice(const int* ids, const int* sent_pos, const int* enc_len, const int* img_index,
                               const bf16_t* shared, const bf16_t* proj, float* out, int B, int L, int P, int D,
                               int vocab, hipStream_t s);
// decoder_input_ids = shift_right(labels); h[b,t] = shared[id]  (fp32 out)
// rows t = 0..T-1 hold decoder positions pos0 + t: id = labels[b, pos0 + t - 1] (start token for position 0)
hipError_t launch_decoder_embed(const int* labels, int ld_labels, const bf16_t* shared, float* out, int B, int T, int D,
                                int vocab, hipStream_t s, int pos0 = 0);
hipError_t launch_rope(bf16_t* x, const float* cs, const float* sn, int B, int H, int S, int hd, int half, hipStream_t s);
// q [B, Hq, S, hd] and k [B, Hk, S, hd] in one launch (16-byte accesses when half % 8 == 0; otherwise two launch_rope calls)
// greedy-decoding step of the Qwen2.5-VL language model (qwen_decode.hip): rotate + append the new token's q / k / v, one-row attention
hipError_t launch_qwen_decode_rope_append(const bf16_t* qkv, const float* cs, const float* sn, const int* len, bf16_t* q_out, bf16_t* kc,
                                          bf16_t* vc, int B, int Hq, int Hkv, int hd, int half, int Lmax, hipStream_t s);
hipError_t launch_qwen_decode_attn(const bf16_t* q, const bf16_t* kc, const bf16_t* vc, const int* len, bf16_t* out, int B, int Hq,
                                   int Hkv, int Lmax, float scale, hipStream_t s);
hipError_t launch_qwen_tail_attn(const float* q, const bf16_t* k, const bf16_t* v, const int* count, bf16_t* out, long long out_plane, int B, int Hq,
                                 int Hkv, int Lmax, float scale, hipStream_t s, bool kv_f16 = false, float vscale = 1.0f);
// bind-time range proof of the Qwen2.5-VL row's fp16 forms (qwen_decode.hip): per-row bounds of a linear's output, their maximum into `slot`
hipError_t launch_rowbound(const bf16_t* W, long long ldw, int N, int K, int mode, const bf16_t* g, int g_len, float R, const float* u,
                           const float* uconst, const bf16_t* bias, float* out, float* slot, hipStream_t s);
hipError_t launch_absmax_bf16(const bf16_t* x, size_t n, float R, float* slot, hipStream_t s);
hipError_t launch_gate_pair_bound(const float* rb, int mlp_p, int ld, float* u_act, float* slot, hipStream_t s);
hipError_t launch_f16_to_bf16_rows(const bf16_t* src, bf16_t* dst, int rows, long long cols, long long src_ld, long long dst_ld, float unscale, hipStream_t s);
hipError_t launch_gather_rows_f32(const float* src, const int* map, float* dst, int rows, int cols, int src_ld, hipStream_t s);
hipError_t launch_qwen_tail_rope_q(const float* qkv, int ld, const float* cs, const float* sn, const int* row, float* q_out, int B, int Hq, int hd,
                                   int half, hipStream_t s);
hipError_t launch_rope_qk(bf16_t* q, bf16_t* k, const float* cs, const float* sn, int B, int Hq, int Hk, int S, int hd, int half,
                          hipStream_t s, bool f16 = false);   // f16: q / k are IEEE fp16 tensors (half % 8 == 0 only)
// RMSNorm of the Qwen2.5-VL row's scaled fp16 forms: deltas are fp16 tensors held behind power-of-two scales (x += delta * d1 (+ delta2 * d2)),
// the operand leaves as fp16(value * osc)
hipError_t launch_rmsnorm_f16s(float* x, const bf16_t* delta, const bf16_t* w, bf16_t* out, int M, int D, float eps, hipStream_t s,
                               const bf16_t* delta2, bool store_x, int out_ld, float d1, float d2, float osc, bool out_bf16 = false);
hipError_t launch_rowss_to_rs(const float* rowss, int parts, int M, float invd, float eps, float* rs, hipStream_t s);
hipError_t launch_reduce_slices(const float* part, int nslices, size_t n, bf16_t* out, hipStream_t s);
// ---- the precise decoder's glue (elementwise.hip).  A split-bf16 tensor is two bf16 planes [2][rows][cols]: hi = bf16(x),
// lo = bf16(x - hi); a GEMM consumes it as 2*rows stacked rows and the two fp32 results are added.
// RMSNorm of the decoder rows: x (fp32 stream) += delta (fp32, optional; written back), out = split(w * x * rsqrt(mean x^2 + eps))
hipError_t launch_rmsnorm_split(float* x, const float* delta, const bf16_t* w, bf16_t* out, long long out_plane, int M, int D,
                                float eps, hipStream_t s);
// Sum of GEMM partials: part is [nslices][2*rows][ldp] fp32 (split-K slices of a stacked hi / lo launch);
//   y[r][c] = (sum_k part[k][r][c]) + (sum_k part[k][rows + r][c])   -- hi rows first, slices in index order (deterministic)
// mode 0: out fp32 [rows, ld_out] = y;  mode 1: out = split(y), planes [rows, cols] at out and out + out_plane (ld_out = cols);
// mode 2: cols = 2F in the packed wi order (blocks of 32 gate | 32 linear columns): out = split(gelu_new(gate) * linear), [rows, F] planes
enum SumPlanesMode : int { SUM_F32 = 0, SUM_SPLIT = 1, SUM_GATED_SPLIT = 2, SUM_F16 = 3 };   // SUM_F16: out IEEE fp16 [rows, ld_out] = y (one rounding)
// out bf16 [rows, ld_out] = f(sum of fp32 partial slices (+ bias)); gated: packed gate|up columns -> SiLU(gate) * up  (elementwise.hip)
hipError_t launch_reduce_slices_act(const float* part, int nslices, long long slice_stride, int rows, int cols, int ldp, const bf16_t* bias,
                                    int gated, bf16_t* out, int ld_out, hipStream_t s);
hipError_t launch_sum_planes(const float* part, int nslices, long long slice_stride, int rows, int cols, int ldp, int mode, void* out,
                             int ld_out, long long out_plane, hipStream_t s, const bf16_t* bias = nullptr, int silu = 0);
// argmax of logits row (b*T + T-1) -> tokens[b, dst_col] (dst_col < 0: T-1)
hipError_t launch_argmax_append(const float* logits, int ldl, int V, int* tokens, int ld_tokens, int B, int T,
                                hipStream_t s, int dst_col = -1, int* flags = nullptr);   // flags: bit 1 is set when a logit of a scanned row is not finite
// bias tables from the [buckets,H] bf16 embedding and a host-computed bucket LUT
hipError_t launch_relpos_table(const bf16_t* rel_weight, const int* bucket_lut_bidir, const int* bucket_lut_causal,
                               int lut_len, int buckets, float* enc_table, int H, int S, float* dec_table, int T,
                               hipStream_t s);
// logits fp32 [B*T, ldl] -> label_logprobs [B,T] (0 where label == -100) and scores [B]
hipError_t launch_score_head(const float* logits, int ldl, int V, const int* labels, float* label_logprobs,
                             float* scores, int B, int T, hipStream_t s, int* flags = nullptr);
// reassociated decoder cross-attention helpers
hipError_t launch_transpose_pad(const bf16_t* in, bf16_t* out, int B, int S, int D, int S_pad, hipStream_t s);
hipError_t launch_masked_softmax(const float* scores, bf16_t* probs, const int* key_len, int B, int rows, int S_pad,
                                 hipStream_t s, bool out_f16 = false);
hipError_t launch_transpose(const bf16_t* in, bf16_t* out, int rows, int cols, hipStream_t s);
// weight packing helpers (bind time)
hipError_t launch_copy_rows(const bf16_t* src, bf16_t* dst, int rows, int cols, int src_ld, int dst_ld,
                            int dst_row_offset, hipStream_t s);   // dst[dst_row_offset + r, 0:cols] = src[r, 0:cols], zero pad to dst_ld
hipError_t launch_gather_rows_bf16(const bf16_t* src0, const bf16_t* src1, const int* map, bf16_t* dst, int rows, int cols,
                                   int src_ld, int dst_ld, hipStream_t s);
hipError_t launch_gather_cols_bf16(const bf16_t* src, const int* cmap, bf16_t* dst, int rows, int src_ld, int dst_cols,
                                   hipStream_t s);
hipError_t launch_gather_rows_f32(const float* src, const int* map, float* dst, int rows, int D, hipStream_t s);
hipError_t launch_qwen_embed(const int* ids, const int* vis_slot, const bf16_t* embed, const bf16_t* merged, float* out, int rows,
                             int D, int vocab, hipStream_t s, bool merged_f16 = false, float merged_unscale = 1.0f);
hipError_t launch_u8_to_norm_bf16(const unsigned char* in, bf16_t* out, int N, int H, int W, const float* mean3,
                                  const float* std3, hipStream_t s);
hipError_t launch_interleave_gate(const bf16_t* wi0, const bf16_t* wi1, bf16_t* dst, int F, int D, hipStream_t s);

}  // namespace vqs
