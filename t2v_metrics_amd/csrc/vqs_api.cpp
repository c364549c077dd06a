// Host side of libvqs_hip: the C ABI of include/vqs.h and the launch sequence of the CLIP-FlanT5
// scoring pass.  The sequence restates what HF executes for the reference:
//   CLIPVisionModel.forward   HF models/clip/modeling_clip.py:613-656  -> encode_images()
//   T5Stack encoder           HF models/t5/modeling_t5.py:663-750      -> score(): encoder loop
//   T5Stack decoder + lm_head HF models/t5/modeling_t5.py:1026-1047    -> score(): decoder loop
// No device allocation, no stream synchronisation, no CPU fallback: if a launch fails the call
// returns VQS_ERR_HIP and the message says which one.
#include "../../include/vqs.h"
#include "../../include/vqs_debug.h"
#include "vqs_kernels.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <map>
#include <array>

using vqs::bf16_t;

struct WEntry {
    const bf16_t* p;
    int64_t numel;
};

#if defined(VQS_ATTN_TIMING) && VQS_ATTN_TIMING
namespace vqs { hipError_t lab_set_attn_timing(unsigned long long* d_buf); }   // attn.hip, timing build only
#endif

struct vqs_handle {
    vqs_config c;
    std::string err;
    std::unordered_map<std::string, WEntry> w;
    bool bound = false;
    int gemm_variant = 0;
    // derived
    int P = 0, Sv = 0, kpatch = 0, kpad = 0, I = 0;
    // packed (device) weights
    const bf16_t* patch_w = nullptr;
    std::vector<const bf16_t*> vit_qkv_w, vit_qkv_b, enc_qkv, enc_wi, dec_qkv, dec_ckv, dec_wi, dec_ckT;
    int cross_mode = 1;   // 1 = reassociated cross-attention (default), 0 = per-layer K|V projection of the encoder output
    int stream_gemm = 1;  // 1 = skinny batched launches (M <= 128 rows per entry) take the HBM-streaming GEMM form (gemm_stream.inc; bitwise the 8-wave forms), 0 = never
    int vit_fp16 = 1;     // 1 (default) = the vision tower and the projector run on IEEE fp16 operands (11 significant bits instead of bf16's 8, same MFMA rate and
                          // bytes): fp16 copies of their weights (made at bind time), fp16 activations, fp32 accumulation / residual stream / statistics
                          // as before; the image features leave as bf16 (the C ABI's type).  CLIP was trained in fp16; the T5 stack is NOT fp16-safe
                          // and stays bf16.  0 = the bf16 tower of rounds 1-3 (what the reference's dtype would give)
    std::vector<const bf16_t*> vit_qkv_w16, vit_out_w16, vit_fc1_w16, vit_fc2_w16;   // the fp16 copies (packed buffer)
    const bf16_t* proj0_w16 = nullptr;
    const bf16_t* proj2_w16 = nullptr;
    int dec_precise = 1;  // 1 = the scoring decoder holds its activations as split-bf16 / fp32 (decoder_pass_precise), 0 = bf16 (rounds 1-3)
    int proj_fp16 = 1;    // 1 (default) = with vit_fp16 the selected features (hidden_states[-2], the residual STREAM itself cast to 16 bits) and the projector's
                          // hidden tensor are fp16 too; 0 = they stay bf16 while the tower's blocks run fp16.  Round 6: the stream is a sum over all
                          // blocks' outputs -- the one site of the tower whose bind-time range proof (engine.py fp16_range_proof) fails first
    int proj_fs_shift = 0, proj_mid_shift = 0;   // with proj_fp16: the selected features are held as fp16(x * 2^-fs), the projector's hidden tensor as
                          // fp16(x * 2^-mid) -- the scales the bind-time range proof asks for (engine.py; 0 = none: the unscaled kernels, bit for bit round 5)
    int enc_fp16 = 1;     // 1 (default, round 5) = the ATTENTION SIDE of the T5 encoder runs on IEEE fp16 tensors: both norm outputs, q / k / v, the
                          // softmax probabilities and the attention output are fp16, and q|k|v, o and the gated wi read fp16 copies of their weights
                          // (made at bind time) -- what HF's own fp16 T5 path holds in fp16.  The sub-layer outputs (o / wo results), the gated FFN
                          // product and the wo GEMM stay bf16: those are where Flan-T5 leaves the fp16 range (HF keeps `wo` in fp32 for it,
                          // modeling_t5.py _keep_in_fp32_modules); the residual stream is fp32 here anyway.  Same MFMA rate, same bytes, three more
                          // significant bits on five of the encoder's seven 16-bit tensor classes (profiles/r5_error_attribution_xxl.md).  0 = bf16
    std::vector<const bf16_t*> enc_qkv16, enc_o16, enc_wi16;   // the fp16 copies (packed buffer)
    int dec_fp16 = 1;     // 1 (default, round 5; effective with dec_precise = 1 and cross_mode = 1) = the precise decoder's cross-attention SCORE path
                          // and the tensor it attends over are IEEE fp16: the encoder's output E (its final norm writes fp16), the cross q (from BOTH
                          // planes of the split norm output, rounded once to fp16), q.Wk (fp16 copy of Wk^T), the probabilities; the three products
                          // q.Wk, (q.Wk).E^T and P.E run on fp16 MFMAs (the batched 8-wave and stream kernels' fp16 instantiations), P.E still leaves as
                          // a split-bf16 tensor.  The reassociated cross-attention makes the rounding of q.Wk and of P COHERENT over all ~600 keys,
                          // which is why these three bf16 roundings were most of the precise decoder's floor (profiles/r5_error_attribution_xxl.md).
                          // 0 = bf16 there (round 4's precise decoder).  vqs_generate and dec_precise = 0 / cross_mode = 0 always run bf16.
    std::vector<const bf16_t*> dec_ckT16;                      // fp16 copies of Wk^T (packed buffer)
    const int* lut_bidir = nullptr;
    const int* lut_causal = nullptr;
    int lut_len = 0;
    std::vector<int> h_lut_bidir, h_lut_causal;
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<std::pair<std::string, double>> ev_what;   // per profiled GEMM launch: call-site label, FLOPs
    std::string prof_report;
    int norm_defer = 1;        // VQS_NORM_DEFER=0: every norm stores the updated fp32 stream (24 instead of 22 B/elem per layer)
    int fused_norm = 0;        // VQS_FUSED_NORM=1: residual update + RMSNorm operand in the o / wo GEMM epilogues (lab; +1 %)
    int splitk = 1;            // VQS_SPLITK=0 disables split-K in the decoder's skinny GEMMs (lab A/B)
    double prof_flops = 0.0;
    double prof_bytes = 0.0;   // algorithmic operand + result bytes of the profiled GEMM launches
    // stage taps (vqs_debug_tap): point name -> (caller buffer, capacity); the pass copies the named intermediate there
    struct Tap { void* dst; size_t cap; };
    std::unordered_map<std::string, Tap> taps;
    // tap window (vqs_debug_tap_window): taps copy outer entries [tap_first, tap_first + tap_count) of tap_outer (pairs of the
    // running T5 pass / images of the running vision pass; set by the pass); tap_count == 0: whole tensors
    int tap_first = 0, tap_count = 0, tap_outer = 0;
    // per-shape tile order of the big GEMMs (option "tile_order:<N>x<K>"): a permutation of the tile list, results unchanged
    struct TileOrder { int N, K, gm, ns; };
    std::vector<TileOrder> tile_orders;
    // per-shape non-temporal result stores (option "nt_store:<N>x<K>" = 1 on, 2 off; absent = the library's choice)
    struct NtStore { int N, K, on; };
    std::vector<NtStore> nt_stores, l2_touches;      // l2_touches: option "l2_touch:<N>x<K>" = 1 on, 2 off
};

namespace {

int fail(vqs_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Carves named regions out of a caller buffer; with base == nullptr it only measures.
struct Carver {
    char* base;
    size_t off = 0;
    std::unordered_map<std::string, size_t>* names;
    template <typename T>
    T* take(size_t n, const char* name = nullptr) {
        off = align_up(off);
        if (names && name) (*names)[name] = off;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

struct EncodeWs {
    bf16_t* im2col;
    float* patch_out;
    float* pre;
    float* hidden;
    bf16_t *delta, *delta2;
    bf16_t *xn, *q, *k, *v, *attn, *mid, *feat_in, *pmid;
    size_t total;
};

EncodeWs carve_encode(const vqs_handle* h, char* base, int N, std::unordered_map<std::string, size_t>* names = nullptr) {
    const vqs_config& c = h->c;
    Carver cv{base, 0, names};
    EncodeWs w;
    const size_t NP = (size_t)N * h->P, NS = (size_t)N * h->Sv;
    w.im2col = cv.take<bf16_t>(NP * h->kpad);
    w.patch_out = cv.take<float>(NP * c.vis_hidden);
    w.pre = cv.take<float>(NS * c.vis_hidden);
    w.hidden = cv.take<float>(NS * c.vis_hidden, "vit_hidden");
    w.delta = cv.take<bf16_t>(NS * c.vis_hidden);
    w.delta2 = cv.take<bf16_t>(NS * c.vis_hidden);
    w.xn = cv.take<bf16_t>(NS * c.vis_hidden);
    w.q = cv.take<bf16_t>(NS * c.vis_hidden);
    w.k = cv.take<bf16_t>(NS * c.vis_hidden);
    w.v = cv.take<bf16_t>(NS * c.vis_hidden);
    w.attn = cv.take<bf16_t>(NS * c.vis_hidden);
    w.mid = cv.take<bf16_t>(NS * c.vis_mlp);
    w.feat_in = cv.take<bf16_t>(NP * c.vis_hidden);
    w.pmid = cv.take<bf16_t>(NP * c.d_model);
    w.total = align_up(cv.off);
    return w;
}

struct ScoreWs {
    int *sent_pos, *enc_len, *flags;
    float *enc_table, *dec_table, *hidden;
    bf16_t *delta, *delta2, *ddelta;
    float* rowss;             // [ceil(D/256)][B*S] partial row sums of squares (fused residual + RMSNorm)
    float* rs;                // [B*S] 1/rms per row, derived from rowss
    bf16_t *xn, *q, *k, *v, *attn, *ff, *enc_out, *ck, *cv;
    float* dhid;
    bf16_t *dxn, *dqkv, *dattn, *dq, *dff;     // dxn, dattn, dff, cctx hold TWO planes [2][rows][width] (split-bf16; plane 0 = the bf16 tensor of rounds 1-3)
    float *dqkv32, *ddelta32, *dscratch;       // precise decoder: fp32 q|k|v, fp32 sub-layer output, GEMM partials
    size_t dscratch_bytes;
    bf16_t *enc_outT, *cqk, *cprobs, *cctx;
    float* cscores;
    int S_pad;
    float* logits;
    int ldl;
    bf16_t* kvcache;          // incremental decoding: [dec_layers][B][Tc][2I] (k | v of every decoded position), else nullptr
    int Tc;
    size_t total;
};

// T = decoder rows per pair held at once; Tc > 0 additionally carves the self-attention K/V cache of vqs_generate for Tc
// positions (the decoder buffers are then sized for T rows per step -- 1 -- and the bias table for Tc distances).
ScoreWs carve_score(const vqs_handle* h, char* base, int B, int L, int T,
                    std::unordered_map<std::string, size_t>* names = nullptr, int Tc = 0) {
    const vqs_config& c = h->c;
    Carver cv{base, 0, names};
    ScoreWs w;
    const int S = L - 1 + h->P;
    const size_t M = (size_t)B * S, MT = (size_t)B * T;
    const int D = c.d_model, I = h->I, F = c.d_ff, H = c.n_heads;
    w.sent_pos = cv.take<int>(B);
    w.enc_len = cv.take<int>(B, "enc_len");
    w.flags = cv.take<int>(4, "flags");
    w.enc_table = cv.take<float>((size_t)H * (2 * S - 1));
    w.dec_table = cv.take<float>((size_t)H * (T > Tc ? T : Tc));
    w.hidden = cv.take<float>(M * D, "enc_in");   // the fp32 residual stream; holds enc_in until layer 0 runs
    w.delta = cv.take<bf16_t>(M * D);            // bf16 sub-layer output waiting to be added by the next norm
    w.delta2 = cv.take<bf16_t>(M * D);           // second pending delta (deferred stream store, VQS_NORM_DEFER)
    w.xn = cv.take<bf16_t>(M * D);
    w.rowss = cv.take<float>((size_t)((D + 255) / 256) * M);
    w.rs = cv.take<float>(M);
    w.q = cv.take<bf16_t>(M * I);
    w.k = cv.take<bf16_t>(M * I);
    w.v = cv.take<bf16_t>(M * I);
    w.attn = cv.take<bf16_t>(M * I);
    w.ff = cv.take<bf16_t>(M * F);
    w.enc_out = cv.take<bf16_t>(M * D, "enc_out");
    w.ck = cv.take<bf16_t>(M * I);
    w.cv = cv.take<bf16_t>(M * I);
    w.dhid = cv.take<float>(MT * D);
    w.ddelta = cv.take<bf16_t>(MT * D);
    w.dxn = cv.take<bf16_t>(2 * MT * D, "dec_out");   // after a pass: the final-norm output (lm_head operand); hi plane, then lo plane
    w.dqkv = cv.take<bf16_t>(MT * 3 * I);
    w.dattn = cv.take<bf16_t>(2 * MT * I);
    w.dq = cv.take<bf16_t>(MT * I);
    w.dff = cv.take<bf16_t>(2 * MT * F);
    w.dqkv32 = cv.take<float>(MT * 3 * I);
    w.ddelta32 = cv.take<float>(MT * D);
    {   // fp32 partials of one decoder GEMM: [slices][2 * MT stacked rows][N] with slices * N <= 16 384 when split (dec_slices) or
        // one slice of the widest N -- a function of the architecture, so no launch ever has to fall back for lack of scratch
        size_t widest = 16384;
        for (size_t n : {(size_t)3 * I, (size_t)2 * F, (size_t)c.vocab, (size_t)D}) widest = n > widest ? n : widest;
        w.dscratch_bytes = 2 * MT * widest * sizeof(float);
        w.dscratch = cv.take<float>(2 * MT * widest);
    }
    w.S_pad = (S + 63) / 64 * 64;
    w.enc_outT = cv.take<bf16_t>((size_t)B * D * w.S_pad);
    w.cqk = cv.take<bf16_t>(MT * H * D);
    w.cscores = cv.take<float>(MT * H * w.S_pad);
    w.cprobs = cv.take<bf16_t>(MT * H * w.S_pad);
    w.cctx = cv.take<bf16_t>(2 * MT * H * D);
    w.ldl = c.vocab;
    w.logits = cv.take<float>(MT * w.ldl, "logits");
    w.Tc = Tc;
    w.kvcache = Tc > 0 ? cv.take<bf16_t>((size_t)c.dec_layers * B * Tc * 2 * I) : nullptr;
    w.total = align_up(cv.off);
    return w;
}

struct PackedLayout {
    size_t patch_w;
    std::vector<size_t> vit_qkv_w, vit_qkv_b, enc_qkv, enc_wi, dec_qkv, dec_ckv, dec_wi, dec_ckT;
    std::vector<size_t> vit_qkv_w16, vit_out_w16, vit_fc1_w16, vit_fc2_w16;    // fp16 copies for option vit_fp16 (always laid out: 0.45 GB for ViT-L)
    size_t proj0_w16, proj2_w16;
    std::vector<size_t> enc_qkv16, enc_o16, enc_wi16;                          // fp16 copies for option enc_fp16 (always laid out: 7.2 GB at XXL, 1.9 GB at XL)
    std::vector<size_t> dec_ckT16;                                             // fp16 copies of Wk^T for option dec_fp16 (0.8 GB at XXL)
    size_t lut_bidir, lut_causal;
    size_t total;
};

PackedLayout packed_layout(const vqs_handle* h) {
    const vqs_config& c = h->c;
    Carver cv{nullptr, 0, nullptr};
    PackedLayout pl;
    auto take = [&](size_t elems) {
        cv.off = align_up(cv.off);
        size_t o = cv.off;
        cv.off += elems * sizeof(bf16_t);
        return o;
    };
    const size_t hid = c.vis_hidden, D = c.d_model, I = h->I, F = c.d_ff;
    pl.patch_w = take(hid * h->kpad);
    for (int i = 0; i < c.vis_layers_run; ++i) {
        pl.vit_qkv_w.push_back(take(3 * hid * hid));
        pl.vit_qkv_b.push_back(take(3 * hid));
    }
    for (int i = 0; i < c.enc_layers; ++i) {
        pl.enc_qkv.push_back(take(3 * I * D));
        pl.enc_wi.push_back(take(2 * F * D));
    }
    for (int i = 0; i < c.dec_layers; ++i) {
        pl.dec_qkv.push_back(take(3 * I * D));
        pl.dec_ckv.push_back(take(2 * I * D));
        pl.dec_ckT.push_back(take(I * D));
        pl.dec_wi.push_back(take(2 * F * D));
    }
    for (int i = 0; i < c.vis_layers_run; ++i) {
        pl.vit_qkv_w16.push_back(take(3 * hid * hid));
        pl.vit_out_w16.push_back(take(hid * hid));
        pl.vit_fc1_w16.push_back(take((size_t)c.vis_mlp * hid));
        pl.vit_fc2_w16.push_back(take((size_t)c.vis_mlp * hid));
    }
    pl.proj0_w16 = take(D * hid);
    pl.proj2_w16 = take(D * D);
    for (int i = 0; i < c.enc_layers; ++i) {
        pl.enc_qkv16.push_back(take(3 * I * D));
        pl.enc_o16.push_back(take(D * I));
        pl.enc_wi16.push_back(take(2 * F * D));
    }
    for (int i = 0; i < c.dec_layers; ++i) pl.dec_ckT16.push_back(take(I * D));
    pl.lut_bidir = take(2 * (size_t)(c.rel_max_distance + 1));   // int32 = 2 bf16 slots each
    pl.lut_causal = take(2 * (size_t)(c.rel_max_distance + 1));
    pl.total = align_up(cv.off);
    return pl;
}

#define HIPCHK(h, expr, what)                                                                          \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(h, VQS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(_e));             \
    } while (0)

int get_w(vqs_handle* h, const std::string& name, int64_t numel, const bf16_t** out) {
    auto it = h->w.find(name);
    if (it == h->w.end()) return fail(h, VQS_ERR_MISSING_WEIGHT, "missing weight: " + name);
    if (it->second.numel != numel)
        return fail(h, VQS_ERR_INVALID, "weight " + name + ": numel " + std::to_string(it->second.numel) +
                                            " != expected " + std::to_string(numel));
    *out = it->second.p;
    return VQS_OK;
}

#define GETW(var, name, numel)                                   \
    const bf16_t* var = nullptr;                                 \
    do {                                                         \
        int _r = get_w(h, (name), (int64_t)(numel), &var);       \
        if (_r != VQS_OK) return _r;                             \
    } while (0)

struct GemmCall {
    const bf16_t* A;
    const bf16_t* W;
    void* C;
    const bf16_t* bias = nullptr;
    const float* resid = nullptr;
    int M, N, K;
    int lda, ldw, ldc;
    int epi;
    int S = 0, H = 0, inner = 0;
    bf16_t* heads[3] = {nullptr, nullptr, nullptr};
    int batch = 1;
    long long sA = 0, sW = 0, sC = 0;
    // fused residual + RMSNorm (see vqs_kernels.h)
    float* hres = nullptr;
    int ldh = 0;
    const bf16_t* lnw = nullptr;
    float* rowss_out = nullptr;
    const float* rowss_in = nullptr;
    int rowss_parts = 0;
    float rs_invd = 0.0f, rs_eps = 0.0f;
    int nt_store = 0;          // result rows leave with the non-temporal hint (call sites whose multi-GB output is streamed once)
    long long split_off = 0;   // EPI_BF16: also store the lo plane of a split-bf16 result at C + split_off (vqs_kernels.h)
    float acc_scale = 1.0f, out_scale = 1.0f;   // f16 = 3 / 4: the scaled quad families (vqs_kernels.h)
    int no_stream = 0;         // 1: keep this launch off the stream form (same bits either way)
    int f16 = 0;               // A, W and the 16-bit result are IEEE fp16 (the fp16 vision tower; quad form only)
};

int run_gemm(vqs_handle* h, const GemmCall& g, hipStream_t st, const char* what) {
    vqs::GemmParams p;
    p.A = g.A; p.W = g.W; p.C = g.C; p.bias = g.bias; p.resid = g.resid;
    p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldw = g.ldw; p.ldc = g.ldc;
    p.S = g.S > 0 ? g.S : 1; p.H = g.H; p.inner = g.inner > 0 ? g.inner : 1;
    p.heads_out[0] = g.heads[0]; p.heads_out[1] = g.heads[1]; p.heads_out[2] = g.heads[2];
    p.batch = g.batch; p.sA = g.sA; p.sW = g.sW; p.sC = g.sC;
    p.hres = g.hres; p.ldh = g.ldh; p.lnw = g.lnw; p.rowss_out = g.rowss_out;
    p.rowss_in = g.rowss_in; p.rowss_parts = g.rowss_parts; p.rs_invd = g.rs_invd; p.rs_eps = g.rs_eps;
    for (const vqs_handle::TileOrder& t : h->tile_orders)
        if (t.N == g.N && t.K == g.K) { p.tile_gm = t.gm; p.tile_ns = t.ns; }
    p.nt_store = g.nt_store;
    p.split_off = g.split_off;
    p.f16 = g.f16; p.acc_scale = g.acc_scale; p.out_scale = g.out_scale;
    p.no_stream = (h->stream_gemm && !g.no_stream) ? 0 : 1;
    for (const vqs_handle::NtStore& t : h->l2_touches)
        if (t.N == g.N && t.K == g.K && g.M >= 4096) p.l2_touch = t.on;
    for (const vqs_handle::NtStore& t : h->nt_stores)
        if (t.N == g.N && t.K == g.K && g.M >= 4096) p.nt_store = t.on == 1;      // the big launches only (the decoder shares (N, K))
    if (h->prof) {
        while (h->ev.size() < h->ev_used + 2) {
            hipEvent_t e;
            HIPCHK(h, hipEventCreate(&e), "hipEventCreate");
            h->ev.push_back(e);
        }
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used], st), "hipEventRecord");
    }
    HIPCHK(h, vqs::launch_gemm(p, g.epi, g.batch > 1 ? 3 : h->gemm_variant, st), std::string("gemm ") + what);
    if (h->prof) {
        HIPCHK(h, hipEventRecord(h->ev[h->ev_used + 1], st), "hipEventRecord");
        h->ev_used += 2;
        h->ev_what.emplace_back(what, 2.0 * (double)g.M * (double)g.N * (double)g.K * (double)g.batch);
        h->prof_flops += 2.0 * (double)g.M * (double)g.N * (double)g.K * (double)g.batch;
        {   // every operand read once, every result written once
            const double out_b = (g.epi == vqs::EPI_F32 || g.epi == vqs::EPI_F32_RESID) ? 4.0 : (g.epi == vqs::EPI_RESID_RMS ? 10.0 : 2.0);
            const double out_n = (g.epi == vqs::EPI_GATED) ? 0.5 * (double)g.N : (double)g.N;
            h->prof_bytes += (double)g.batch * (2.0 * ((double)g.M + (double)g.N) * (double)g.K + out_b * (double)g.M * out_n);
        }
    }
    return VQS_OK;
}

#define RUN(expr)                    \
    do {                             \
        int _r = (expr);             \
        if (_r != VQS_OK) return _r; \
    } while (0)

// Copy an intermediate to a caller buffer registered with vqs_debug_tap (in-stream, device to device).  The workspace
// buffers are reused layer after layer, so the parity tests that check EVERY launch of a pass against the oracle on the
// engine's own inputs (tests/test_gpu_stage_locked.py) read them through this.  One empty() test when nothing is registered.
int tap(vqs_handle* h, const char* stack, int layer, const char* what, const void* src, size_t bytes, hipStream_t st) {
    if (h->taps.empty()) return VQS_OK;
    const std::string name = layer >= 0 ? std::string(stack) + "." + std::to_string(layer) + "." + what : std::string(stack) + "." + what;
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return VQS_OK;
    if (h->tap_count > 0) {        // every intermediate is outer-entry-major: the window is one contiguous byte range of it
        if (h->tap_outer <= 0 || h->tap_first + h->tap_count > h->tap_outer || bytes % (size_t)h->tap_outer != 0)
            return fail(h, VQS_ERR_INVALID, "tap " + name + ": window [" + std::to_string(h->tap_first) + ", +" + std::to_string(h->tap_count) +
                                            ") does not fit " + std::to_string(h->tap_outer) + " outer entries");
        const size_t per = bytes / (size_t)h->tap_outer;
        src = static_cast<const char*>(src) + per * (size_t)h->tap_first;
        bytes = per * (size_t)h->tap_count;
    }
    if (it->second.cap < bytes) return fail(h, VQS_ERR_WORKSPACE, "tap " + name + ": buffer too small (" + std::to_string(bytes) + " bytes needed)");
    HIPCHK(h, hipMemcpyAsync(it->second.dst, src, bytes, hipMemcpyDeviceToDevice, st), "tap copy");
    return VQS_OK;
}
#define TAP(stack, layer, what, ptr, elems) RUN(tap(h, stack, layer, what, ptr, (size_t)(elems) * sizeof(*(ptr)), st))

// A split-bf16 tensor (planes [2][rows][width], `plane` elements apart) goes to its tap buffer as two planes of the tapped rows:
// [2][window rows][width] bf16; the reader adds them (tests/gpu_util.py).
int tap_split(vqs_handle* h, const char* stack, int layer, const char* what, const bf16_t* src, size_t plane, hipStream_t st) {
    if (h->taps.empty()) return VQS_OK;
    const std::string name = layer >= 0 ? std::string(stack) + "." + std::to_string(layer) + "." + what : std::string(stack) + "." + what;
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return VQS_OK;
    size_t bytes = plane * sizeof(bf16_t), first = 0;
    if (h->tap_count > 0) {
        if (h->tap_outer <= 0 || h->tap_first + h->tap_count > h->tap_outer || bytes % (size_t)h->tap_outer != 0)
            return fail(h, VQS_ERR_INVALID, "tap " + name + ": window does not fit");
        const size_t per = bytes / (size_t)h->tap_outer;
        first = per * (size_t)h->tap_first;
        bytes = per * (size_t)h->tap_count;
    }
    if (it->second.cap < 2 * bytes) return fail(h, VQS_ERR_WORKSPACE, "tap " + name + ": buffer too small (" + std::to_string(2 * bytes) + " bytes needed)");
    for (int pl = 0; pl < 2; ++pl)
        HIPCHK(h, hipMemcpyAsync(static_cast<char*>(it->second.dst) + pl * bytes, reinterpret_cast<const char*>(src + pl * plane) + first, bytes,
                                 hipMemcpyDeviceToDevice, st), "tap copy");
    return VQS_OK;
}
#define TAP2(stack, layer, what, ptr, plane_elems) RUN(tap_split(h, stack, layer, what, ptr, (size_t)(plane_elems), st))

}  // namespace

extern "C" {

int vqs_debug_tap(vqs_handle* h, const char* name, void* d_dst, size_t bytes) {
    if (!h) return VQS_ERR_INVALID;
    if (!name) { h->taps.clear(); return VQS_OK; }
    if (!d_dst || bytes == 0) { h->taps.erase(name); return VQS_OK; }
    h->taps[name] = vqs_handle::Tap{d_dst, bytes};
    return VQS_OK;
}

int vqs_debug_tap_window(vqs_handle* h, int32_t first, int32_t count) {
    if (!h || first < 0 || count < 0) return VQS_ERR_INVALID;
    h->tap_first = count > 0 ? first : 0;
    h->tap_count = count;
    return VQS_OK;
}

int vqs_debug_gemm_form(int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t epilogue, int32_t batch, int32_t variant,
                        int32_t S, int32_t inner, int32_t inner_kv) {
    if (M <= 0 || N <= 0 || K <= 0 || epilogue < 0 || epilogue >= vqs::EPI_COUNT) return -1;
    vqs::GemmParams p{};
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = N; p.batch = batch > 0 ? batch : 1;
    p.S = S; p.inner = inner; p.inner_kv = inner_kv;
    const int form = vqs::gemm_form(p, epilogue, variant & 0xff);
    return (form == 10 && vqs::gemm_takes_slim(p, epilogue, variant & 0xff)) ? 13 : form;      // 13: a quad call site's few-row launch (gemm_slim.inc, same bits)
}

int vqs_debug_gemm_batched(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc,
                           int32_t epilogue, int32_t batch, int64_t sA, int64_t sW, int64_t sC, int64_t split_off, int32_t no_stream,
                           int32_t variant, void* stream) {
    if (!A || !W || !C || (epilogue != vqs::EPI_F32 && epilogue != vqs::EPI_BF16) || batch < 1) return VQS_ERR_INVALID;
    vqs::GemmParams p{};
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = nullptr; p.resid = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.S = 1; p.H = 0; p.inner = 1;
    p.batch = batch; p.sA = sA; p.sW = sW; p.sC = sC;
    p.split_off = split_off; p.no_stream = no_stream;
    p.f16 = (variant >> 27) & 3;           // bits 27-28 of `variant`, as in vqs_gemm: 0 bf16, 1 fp16 operands and result, 2 fp16 operands / bf16 (split) result
    if (p.f16 == 3) return VQS_ERR_INVALID;
    return vqs::launch_gemm(p, epilogue, variant & 0xff, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_debug_heads_rows(int32_t row0, int32_t S, int32_t hx, int32_t hdim, int32_t n, int64_t* off_out) {
    if (row0 < 0 || S < 8 || hx <= 0 || hdim <= 0 || n <= 0 || !off_out) return VQS_ERR_INVALID;
    int hs;
    long long off;
    vqs::heads_off_first(row0, S, hx, hdim, hs, off);
    const long long wrap = (long long)(hx - 1) * S * hdim;
    for (int k = 0; k < n; ++k) {          // exactly the use-then-step order of the GEMM epilogue
        off_out[k] = off;
        vqs::heads_off_step8(S, hdim, wrap, hs, off);
    }
    return VQS_OK;
}

int vqs_debug_tile_order(int32_t M, int32_t N, int32_t K, int32_t batch, int32_t gm, int32_t ns, int32_t grid, int32_t* out) {
    if (M <= 0 || N <= 0 || K < 0 || batch <= 0 || grid <= 0 || (grid & 7) != 0 || !out) return VQS_ERR_INVALID;
    vqs::GemmParams p{};
    p.M = M; p.N = N; p.K = K; p.batch = batch; p.tile_gm = gm; p.tile_ns = ns;
    vqs::resolve_tile_order(p, grid);
    const int tiles_m = (M + vqs::GEMM_BM - 1) / vqs::GEMM_BM, tiles_n = (N + vqs::GEMM_BN - 1) / vqs::GEMM_BN;
    const int nwg = tiles_m * tiles_n * batch;
    // the persistent kernels' walk: workgroup b takes slots b, b + grid, b + 2 grid, ... while slot < nwg
    int k = 0;
    for (int b = 0; b < grid; ++b)
        for (int pid = b; pid < nwg; pid += grid) {
            int m0, n0, bz;
            vqs::tile_of_slot(pid, nwg, tiles_m, tiles_n, p.tile_gm, p.tile_ns, m0, n0, bz);
            out[4 * k + 0] = pid; out[4 * k + 1] = m0; out[4 * k + 2] = n0; out[4 * k + 3] = bz;
            ++k;
        }
    return k == nwg ? (p.tile_gm | (p.tile_ns << 8)) : VQS_ERR_INVALID;
}

#if defined(VQS_ATTN_TIMING) && VQS_ATTN_TIMING
// lab builds only (make variant NAME=attn_timing VFLAGS=-DVQS_ATTN_TIMING=1): d_buf = 8 x uint64 on the device, zeroed by the caller
int vqs_lab_set_attn_timing(void* d_buf) { return vqs::lab_set_attn_timing((unsigned long long*)d_buf) == hipSuccess ? VQS_OK : VQS_ERR_HIP; }
#endif


int64_t vqs_attention_lds_bytes(int32_t S, int32_t has_bias, int32_t hd) {
    if (S <= 0 || (hd != 0 && hd != 64 && hd != 128)) return -1;
    return (int64_t)vqs::attention_lds_bytes(S, has_bias != 0, hd);
}

int32_t vqs_relpos_bucket(int32_t relative_position, int32_t bidirectional, int32_t num_buckets, int32_t max_distance) {
    // HF models/t5/modeling_t5.py:238-262, same fp32 operation order as torch.
    int32_t bucket = 0, nb = num_buckets, rp = relative_position;
    if (bidirectional) {
        nb /= 2;
        if (rp > 0) bucket += nb;
        if (rp < 0) rp = -rp;
    } else {
        rp = rp < 0 ? -rp : 0;
    }
    const int32_t max_exact = nb / 2;
    if (rp < max_exact) return bucket + rp;
    float v = logf((float)rp / (float)max_exact) / (float)std::log((double)max_distance / (double)max_exact);
    v = v * (float)(nb - max_exact);
    int32_t large = max_exact + (int32_t)v;
    if (large > nb - 1) large = nb - 1;
    return bucket + large;
}

int vqs_create(const vqs_config* cfg, vqs_handle** out) {
    if (!cfg || !out) return VQS_ERR_INVALID;
    vqs_handle* h = new vqs_handle();
    h->c = *cfg;
    const vqs_config& c = h->c;
    *out = h;
    auto bad = [&](const char* m) { return fail(h, VQS_ERR_INVALID, m); };
    if (c.d_kv != 64) return bad("d_kv must be 64");
    if (c.vis_hidden <= 0 || c.vis_heads <= 0 || c.vis_hidden != c.vis_heads * 64) return bad("vision head dim must be 64");
    if (c.vis_patch <= 0 || c.vis_image % c.vis_patch) return bad("image size must be a multiple of the patch size");
    if (c.vis_hidden % 64 || c.vis_mlp % 64 || c.d_model % 64 || c.d_ff % 64) return bad("hidden sizes must be multiples of 64");
    if (c.vocab % 8) return bad("vocab must be a multiple of 8");
    if (c.vis_layers_run < 1 || c.enc_layers < 1 || c.dec_layers < 1) return bad("layer counts must be positive");
    const int g = c.vis_image / c.vis_patch;
    h->P = g * g;
    h->Sv = h->P + 1;
    h->kpatch = 3 * c.vis_patch * c.vis_patch;
    h->kpad = (h->kpatch + 63) / 64 * 64;
    h->I = c.n_heads * c.d_kv;
    h->lut_len = c.rel_max_distance + 1;
    for (int n = 0; n < h->lut_len; ++n) {
        // index = |relative position|; bidirectional: the +nb/2 for rel>0 is added on the device
        h->h_lut_bidir.push_back(vqs_relpos_bucket(-n, 1, c.rel_buckets, c.rel_max_distance));
        h->h_lut_causal.push_back(vqs_relpos_bucket(-n, 0, c.rel_buckets, c.rel_max_distance));
    }
    h->gemm_variant = 3;   // persistent kernel, schedule chosen by shape (gemm.hip)
    return VQS_OK;
}

int vqs_set_option(vqs_handle* h, const char* name, int32_t value) {
    if (!h || !name) return VQS_ERR_INVALID;
    const std::string n(name);
    if (n == "cross_mode" && (value == 0 || value == 1)) h->cross_mode = value;
    else if (n == "splitk" && (value == 0 || value == 1)) h->splitk = value;
    else if (n == "fused_norm" && (value == 0 || value == 1)) h->fused_norm = value;
    else if (n == "norm_defer" && (value == 0 || value == 1)) h->norm_defer = value;
    else if (n == "dec_precise" && (value == 0 || value == 1)) h->dec_precise = value;
    else if (n == "vit_fp16" && (value == 0 || value == 1)) h->vit_fp16 = value;
    else if (n == "enc_fp16" && value >= 0 && value <= 3) h->enc_fp16 = value;      // 2 / 3: attention sub-block only / FFN input side only (A/B)
    else if (n == "proj_fp16" && (value == 0 || value == 1)) h->proj_fp16 = value;
    else if (n == "proj_fs_shift" && value >= 0 && value <= 60) h->proj_fs_shift = value;
    else if (n == "proj_mid_shift" && value >= 0 && value <= 60) h->proj_mid_shift = value;
    else if (n == "dec_fp16" && (value == 0 || value == 1)) h->dec_fp16 = value;
    else if (n == "stream_gemm" && (value == 0 || value == 1)) h->stream_gemm = value;
    else if (n == "gemm_variant" && (value == 0 || value == 2 || value == 3 || value == 5 || value == 11)) h->gemm_variant = value;
    else if (n.rfind("l2_touch:", 0) == 0 || n.rfind("nt_store:", 0) == 0 || n.rfind("tile_order:", 0) == 0) {
        // per weight shape [N, K], for the big launches of a pass; 0 removes the entry = the library's choice.  All three are
        // bitwise-neutral cache-policy knobs:
        //   "l2_touch:<N>x<K>"   1: A-panel L2 prefetch in the lock-step GEMM, 2: off
        //   "nt_store:<N>x<K>"   1: non-temporal result stores, 2: plain stores
        //   "tile_order:<N>x<K>" gm | ns << 8
        const bool order = n[0] == 't';
        const size_t colon = n.find(':');
        int N = 0, K = 0;
        const int gm = value & 0xff, ns = (value >> 8) & 0xff;
        const bool value_ok = order ? (value >= 0 && (value >> 16) == 0 && (value == 0 || (gm >= 1 && gm <= 64 && ns <= 8))) : (value >= 0 && value <= 2);
        if (std::sscanf(n.c_str() + colon + 1, "%dx%d", &N, &K) != 2 || N <= 0 || K <= 0 || !value_ok)
            return fail(h, VQS_ERR_INVALID, "set_option: bad " + n.substr(0, colon) + ": " + n + "=" + std::to_string(value));
        if (order) {
            auto& v = h->tile_orders;
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].N == N && v[i].K == K) { v.erase(v.begin() + i); break; }
            if (value != 0) v.push_back(vqs_handle::TileOrder{N, K, gm, ns});
        } else {
            auto& v = n[0] == 'l' ? h->l2_touches : h->nt_stores;
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].N == N && v[i].K == K) { v.erase(v.begin() + i); break; }
            if (value != 0) v.push_back(vqs_handle::NtStore{N, K, value});
        }
    }
    else return fail(h, VQS_ERR_INVALID, "set_option: unknown option or value: " + n + "=" + std::to_string(value));
    return VQS_OK;
}

int vqs_get_option(const vqs_handle* h, const char* name, int32_t* value) {
    if (!h || !name || !value) return VQS_ERR_INVALID;
    const std::string n(name);
    if (n == "cross_mode") *value = h->cross_mode;
    else if (n == "splitk") *value = h->splitk;
    else if (n == "fused_norm") *value = h->fused_norm;
    else if (n == "norm_defer") *value = h->norm_defer;
    else if (n == "dec_precise") *value = h->dec_precise;
    else if (n == "vit_fp16") *value = h->vit_fp16;
    else if (n == "enc_fp16") *value = h->enc_fp16;
    else if (n == "proj_fp16") *value = h->proj_fp16;
    else if (n == "proj_fs_shift") *value = h->proj_fs_shift;
    else if (n == "proj_mid_shift") *value = h->proj_mid_shift;
    else if (n == "dec_fp16") *value = h->dec_fp16;
    else if (n == "stream_gemm") *value = h->stream_gemm;
    else if (n == "gemm_variant") *value = h->gemm_variant;
    else return VQS_ERR_INVALID;
    return VQS_OK;
}

void vqs_destroy(vqs_handle* h) {
    if (!h) return;
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
}

const char* vqs_last_error(const vqs_handle* h) { return h ? h->err.c_str() : "null handle"; }

size_t vqs_packed_bytes(const vqs_handle* h) { return h ? packed_layout(h).total : 0; }

int vqs_bind_weights(vqs_handle* h, const vqs_weight_desc* weights, int32_t n, void* d_packed, size_t packed_bytes,
                     void* stream) {
    if (!h || !weights || n <= 0 || !d_packed) return fail(h, VQS_ERR_INVALID, "bind_weights: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const vqs_config& c = h->c;
    const PackedLayout pl = packed_layout(h);
    if (packed_bytes < pl.total) return fail(h, VQS_ERR_WORKSPACE, "bind_weights: packed buffer too small");
    h->bound = false;
    h->w.clear();
    for (int i = 0; i < n; ++i) {
        if (!weights[i].name || !weights[i].d_data) return fail(h, VQS_ERR_INVALID, "bind_weights: null entry");
        h->w[weights[i].name] = WEntry{(const bf16_t*)weights[i].d_data, weights[i].numel};
    }
    char* pk = (char*)d_packed;
    auto at = [&](size_t off) { return reinterpret_cast<bf16_t*>(pk + off); };
    const int hid = c.vis_hidden, D = c.d_model, I = h->I, F = c.d_ff;

    // conv kernel [hid, 3*p*p] -> [hid, kpad] zero padded (K must be a multiple of 64)
    {
        GETW(pw, "vision.embeddings.patch_embedding.weight", (int64_t)hid * h->kpatch);
        HIPCHK(h, vqs::launch_copy_rows(pw, at(pl.patch_w), hid, h->kpatch, h->kpatch, h->kpad, 0, st), "pack patch");
        h->patch_w = at(pl.patch_w);
    }
    h->vit_qkv_w.clear(); h->vit_qkv_b.clear();
    for (int i = 0; i < c.vis_layers_run; ++i) {
        const std::string p = "vision.encoder.layers." + std::to_string(i) + ".self_attn.";
        const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            GETW(ww, p + nm[j] + ".weight", (int64_t)hid * hid);
            GETW(bb, p + nm[j] + ".bias", hid);
            HIPCHK(h, vqs::launch_copy_rows(ww, at(pl.vit_qkv_w[i]), hid, hid, hid, hid, j * hid, st), "pack vit qkv");
            HIPCHK(h, vqs::launch_copy_rows(bb, at(pl.vit_qkv_b[i]), 1, hid, hid, hid, j, st), "pack vit qkv bias");
        }
        h->vit_qkv_w.push_back(at(pl.vit_qkv_w[i]));
        h->vit_qkv_b.push_back(at(pl.vit_qkv_b[i]));
    }
    // fp16 copies of the tower's and the projector's linear weights (option vit_fp16): bf16 -> fp16 is exact for 2^-14 <= |w| < 65 520
    h->vit_qkv_w16.clear(); h->vit_out_w16.clear(); h->vit_fc1_w16.clear(); h->vit_fc2_w16.clear();
    for (int i = 0; i < c.vis_layers_run; ++i) {
        const std::string p = "vision.encoder.layers." + std::to_string(i) + ".";
        const size_t mlp = (size_t)c.vis_mlp;
        GETW(ow, p + "self_attn.out_proj.weight", (int64_t)hid * hid);
        GETW(f1w, p + "mlp.fc1.weight", (int64_t)mlp * hid);
        GETW(f2w, p + "mlp.fc2.weight", (int64_t)hid * mlp);
        HIPCHK(h, vqs::launch_cast16(at(pl.vit_qkv_w[i]), at(pl.vit_qkv_w16[i]), (size_t)3 * hid * hid, true, st), "fp16 vit qkv");
        HIPCHK(h, vqs::launch_cast16(ow, at(pl.vit_out_w16[i]), (size_t)hid * hid, true, st), "fp16 vit out_proj");
        HIPCHK(h, vqs::launch_cast16(f1w, at(pl.vit_fc1_w16[i]), mlp * hid, true, st), "fp16 vit fc1");
        HIPCHK(h, vqs::launch_cast16(f2w, at(pl.vit_fc2_w16[i]), mlp * hid, true, st), "fp16 vit fc2");
        h->vit_qkv_w16.push_back(at(pl.vit_qkv_w16[i]));
        h->vit_out_w16.push_back(at(pl.vit_out_w16[i]));
        h->vit_fc1_w16.push_back(at(pl.vit_fc1_w16[i]));
        h->vit_fc2_w16.push_back(at(pl.vit_fc2_w16[i]));
    }
    {
        GETW(p0w, "mm_projector.0.weight", (int64_t)D * hid);
        GETW(p2w, "mm_projector.2.weight", (int64_t)D * D);
        HIPCHK(h, vqs::launch_cast16(p0w, at(pl.proj0_w16), (size_t)D * hid, true, st), "fp16 mm_projector.0");
        HIPCHK(h, vqs::launch_cast16(p2w, at(pl.proj2_w16), (size_t)D * D, true, st), "fp16 mm_projector.2");
        h->proj0_w16 = at(pl.proj0_w16);
        h->proj2_w16 = at(pl.proj2_w16);
    }
    auto pack_qkv = [&](const std::string& prefix, bf16_t* dst, int first, int count) -> int {
        const char* nm[3] = {"q", "k", "v"};
        for (int j = 0; j < count; ++j) {
            GETW(ww, prefix + nm[first + j] + ".weight", (int64_t)I * D);
            HIPCHK(h, vqs::launch_copy_rows(ww, dst, I, D, D, D, j * I, st), "pack t5 qkv");
        }
        return VQS_OK;
    };
    auto pack_wi = [&](const std::string& prefix, bf16_t* dst) -> int {
        GETW(w0, prefix + "wi_0.weight", (int64_t)F * D);
        GETW(w1, prefix + "wi_1.weight", (int64_t)F * D);
        HIPCHK(h, vqs::launch_interleave_gate(w0, w1, dst, F, D, st), "pack wi");
        return VQS_OK;
    };
    h->enc_qkv16.clear(); h->enc_o16.clear(); h->enc_wi16.clear(); h->dec_ckT16.clear();
    h->enc_qkv.clear(); h->enc_wi.clear(); h->dec_qkv.clear(); h->dec_ckv.clear(); h->dec_wi.clear(); h->dec_ckT.clear();
    for (int i = 0; i < c.enc_layers; ++i) {
        const std::string p = "encoder.block." + std::to_string(i) + ".";
        RUN(pack_qkv(p + "layer.0.SelfAttention.", at(pl.enc_qkv[i]), 0, 3));
        RUN(pack_wi(p + "layer.1.DenseReluDense.", at(pl.enc_wi[i])));
        h->enc_qkv.push_back(at(pl.enc_qkv[i]));
        h->enc_wi.push_back(at(pl.enc_wi[i]));
        // fp16 copies for option enc_fp16 (of the PACKED q|k|v and interleaved wi_0|wi_1, and of o): exact for 2^-14 <= |w| < 65 520
        GETW(ow, p + "layer.0.SelfAttention.o.weight", (int64_t)D * I);
        HIPCHK(h, vqs::launch_cast16(at(pl.enc_qkv[i]), at(pl.enc_qkv16[i]), (size_t)3 * I * D, true, st), "fp16 enc qkv");
        HIPCHK(h, vqs::launch_cast16(ow, at(pl.enc_o16[i]), (size_t)D * I, true, st), "fp16 enc o");
        HIPCHK(h, vqs::launch_cast16(at(pl.enc_wi[i]), at(pl.enc_wi16[i]), (size_t)2 * F * D, true, st), "fp16 enc wi");
        h->enc_qkv16.push_back(at(pl.enc_qkv16[i]));
        h->enc_o16.push_back(at(pl.enc_o16[i]));
        h->enc_wi16.push_back(at(pl.enc_wi16[i]));
    }
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = "decoder.block." + std::to_string(i) + ".";
        RUN(pack_qkv(p + "layer.0.SelfAttention.", at(pl.dec_qkv[i]), 0, 3));
        RUN(pack_qkv(p + "layer.1.EncDecAttention.", at(pl.dec_ckv[i]), 1, 2));
        RUN(pack_wi(p + "layer.2.DenseReluDense.", at(pl.dec_wi[i])));
        h->dec_qkv.push_back(at(pl.dec_qkv[i]));
        h->dec_ckv.push_back(at(pl.dec_ckv[i]));
        {   // Wk^T [D, I] for the reassociated cross-attention: q'_h = q_h . Wk_h needs Wk_h^T K-contiguous
            GETW(wk, p + "layer.1.EncDecAttention.k.weight", (int64_t)I * D);
            HIPCHK(h, vqs::launch_transpose(wk, at(pl.dec_ckT[i]), I, D, st), "pack cross k^T");
            h->dec_ckT.push_back(at(pl.dec_ckT[i]));
            HIPCHK(h, vqs::launch_cast16(at(pl.dec_ckT[i]), at(pl.dec_ckT16[i]), (size_t)I * D, true, st), "fp16 cross k^T");
            h->dec_ckT16.push_back(at(pl.dec_ckT16[i]));
        }
        h->dec_wi.push_back(at(pl.dec_wi[i]));
    }
    HIPCHK(h, hipMemcpyAsync(pk + pl.lut_bidir, h->h_lut_bidir.data(), h->lut_len * sizeof(int), hipMemcpyHostToDevice, st),
           "upload bucket lut");
    HIPCHK(h, hipMemcpyAsync(pk + pl.lut_causal, h->h_lut_causal.data(), h->lut_len * sizeof(int), hipMemcpyHostToDevice, st),
           "upload bucket lut");
    h->lut_bidir = reinterpret_cast<const int*>(pk + pl.lut_bidir);
    h->lut_causal = reinterpret_cast<const int*>(pk + pl.lut_causal);
    h->bound = true;
    return VQS_OK;
}

size_t vqs_encode_workspace_bytes(const vqs_handle* h, int32_t n_img) {
    if (!h || n_img <= 0) return 0;
    return carve_encode(h, nullptr, n_img).total;
}

int vqs_encode_images(vqs_handle* h, const void* d_pixels, int32_t N, void* d_feats, void* d_ws, size_t ws_bytes,
                      void* stream) {
    if (!h) return VQS_ERR_INVALID;
    if (!h->bound) return fail(h, VQS_ERR_STATE, "encode_images: weights not bound");
    if (!d_pixels || !d_feats || !d_ws || N <= 0) return fail(h, VQS_ERR_INVALID, "encode_images: bad arguments");
    const vqs_config& c = h->c;
    const EncodeWs w = carve_encode(h, (char*)d_ws, N);
    if (ws_bytes < w.total) return fail(h, VQS_ERR_WORKSPACE, "encode_images: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int hid = c.vis_hidden, P = h->P, Sv = h->Sv, mlp = c.vis_mlp, D = c.d_model;
    const int NP = N * P, NS = N * Sv;
    h->tap_outer = N;
    // option vit_fp16: every 16-bit tensor of the tower and the projector (norm outputs, q / k / v, attention output, deltas, FFN
    // product, selected features, projector hidden) is IEEE fp16 and every linear reads the fp16 copy of its weight; the fp32 residual
    // stream, the norm statistics, the softmax and all accumulation are what they were.  The patch embedding (fp32 result) keeps its
    // bf16 operands.
    const bool f16 = h->vit_fp16 != 0;
    if (f16 && h->gemm_variant != 3)
        return fail(h, VQS_ERR_STATE, "encode_images: the fp16 vision tower (option vit_fp16, default 1) needs gemm_variant 3 -- its linears exist in the quad form only; set vit_fp16=0 to A/B other GEMM forms");

    GETW(cls, "vision.embeddings.class_embedding", hid);
    GETW(pos, "vision.embeddings.position_embedding.weight", (int64_t)Sv * hid);
    GETW(pre_w, "vision.pre_layrnorm.weight", hid);
    GETW(pre_b, "vision.pre_layrnorm.bias", hid);

    HIPCHK(h, vqs::launch_im2col((const bf16_t*)d_pixels, w.im2col, N, c.vis_image, c.vis_patch, h->kpad, st), "im2col");
    {
        GemmCall g{w.im2col, h->patch_w, w.patch_out};
        g.M = NP; g.N = hid; g.K = h->kpad; g.lda = h->kpad; g.ldw = h->kpad; g.ldc = hid; g.epi = vqs::EPI_F32;
        RUN(run_gemm(h, g, st, "patch_embed"));
    }
    HIPCHK(h, vqs::launch_vit_assemble(w.patch_out, cls, pos, w.pre, N, P, hid, st), "vit_assemble");
    HIPCHK(h, vqs::launch_layernorm(w.pre, nullptr, pre_w, pre_b, w.hidden, 1, NS, hid, c.vis_ln_eps, st), "pre_layrnorm");
    TAP("vit", -1, "patch_out", w.patch_out, (size_t)NP * hid);
    TAP("vit", -1, "h0", w.hidden, (size_t)NS * hid);

    // Residual stream protocol: a sub-layer's output GEMM writes bf16 into `delta`; the NEXT norm kernel performs
    // hidden += delta (written back) and normalises in the same pass.  `pend` is the not-yet-added delta.
    // Deferred store (norm_defer, every layer but the last): layer_norm2 normalises hidden + delta_attn WITHOUT writing
    // the stream and the next layer_norm1 stores (hidden + delta_attn) + delta_mlp -- the same fp32 additions in the
    // same order, 22 instead of 24 bytes per element and layer.
    const bf16_t* pend = nullptr;
    const bf16_t* pend_attn = nullptr;
    for (int i = 0; i < c.vis_layers_run; ++i) {
        const bool defer = h->norm_defer != 0 && i + 1 < c.vis_layers_run;
        const std::string p = "vision.encoder.layers." + std::to_string(i) + ".";
        GETW(ln1w, p + "layer_norm1.weight", hid);
        GETW(ln1b, p + "layer_norm1.bias", hid);
        GETW(ln2w, p + "layer_norm2.weight", hid);
        GETW(ln2b, p + "layer_norm2.bias", hid);
        GETW(ow, p + "self_attn.out_proj.weight", (int64_t)hid * hid);
        GETW(ob, p + "self_attn.out_proj.bias", hid);
        GETW(f1w, p + "mlp.fc1.weight", (int64_t)mlp * hid);
        GETW(f1b, p + "mlp.fc1.bias", mlp);
        GETW(f2w, p + "mlp.fc2.weight", (int64_t)hid * mlp);
        GETW(f2b, p + "mlp.fc2.bias", hid);

        if (pend_attn)
            HIPCHK(h, vqs::launch_layernorm(w.hidden, pend_attn, ln1w, ln1b, w.xn, 0, NS, hid, c.vis_ln_eps, st, pend, true, f16), "layer_norm1");
        else
            HIPCHK(h, vqs::launch_layernorm(w.hidden, pend, ln1w, ln1b, w.xn, 0, NS, hid, c.vis_ln_eps, st, nullptr, true, f16), "layer_norm1");
        pend = nullptr;
        pend_attn = nullptr;
        TAP("vit", i, "xn0", w.xn, (size_t)NS * hid);
        {
            GemmCall g{w.xn, f16 ? h->vit_qkv_w16[i] : h->vit_qkv_w[i], nullptr};
            g.bias = h->vit_qkv_b[i];
            g.f16 = f16;
            g.M = NS; g.N = 3 * hid; g.K = hid; g.lda = hid; g.ldw = hid; g.ldc = 0; g.epi = vqs::EPI_HEADS;
            g.S = Sv; g.H = c.vis_heads; g.inner = hid;
            g.heads[0] = w.q; g.heads[1] = w.k; g.heads[2] = w.v;
            RUN(run_gemm(h, g, st, "vit qkv"));
        }
        {
            vqs::AttnParams a{w.q, w.k, w.v, w.attn, nullptr, nullptr, N, c.vis_heads, Sv, 0.125f};
            a.f16 = f16;
            TAP("vit", i, "q", w.q, (size_t)NS * hid);
            TAP("vit", i, "k", w.k, (size_t)NS * hid);
            TAP("vit", i, "v", w.v, (size_t)NS * hid);
            HIPCHK(h, vqs::launch_attention(a, st), "vit attention");
            TAP("vit", i, "attn", w.attn, (size_t)NS * hid);
        }
        {
            GemmCall g{w.attn, f16 ? h->vit_out_w16[i] : ow, w.delta};
            g.bias = ob;
            g.f16 = f16;
            g.M = NS; g.N = hid; g.K = hid; g.lda = hid; g.ldw = hid; g.ldc = hid; g.epi = vqs::EPI_BF16;
            RUN(run_gemm(h, g, st, "vit out_proj"));
            TAP("vit", i, "d_attn", w.delta, (size_t)NS * hid);
        }
        HIPCHK(h, vqs::launch_layernorm(w.hidden, w.delta, ln2w, ln2b, w.xn, 0, NS, hid, c.vis_ln_eps, st, nullptr, !defer, f16), "layer_norm2");
        if (defer) pend_attn = w.delta;
        TAP("vit", i, "xn1", w.xn, (size_t)NS * hid);
        {
            GemmCall g{w.xn, f16 ? h->vit_fc1_w16[i] : f1w, w.mid};
            g.bias = f1b;
            g.f16 = f16;
            g.M = NS; g.N = mlp; g.K = hid; g.lda = hid; g.ldw = hid; g.ldc = mlp; g.epi = vqs::EPI_BF16_QGELU;
            RUN(run_gemm(h, g, st, "vit fc1"));
            TAP("vit", i, "mid", w.mid, (size_t)NS * mlp);
        }
        {
            bf16_t* dst = defer ? w.delta2 : w.delta;
            GemmCall g{w.mid, f16 ? h->vit_fc2_w16[i] : f2w, dst};
            g.bias = f2b;
            g.f16 = f16;
            g.M = NS; g.N = hid; g.K = mlp; g.lda = mlp; g.ldw = mlp; g.ldc = hid; g.epi = vqs::EPI_BF16;
            RUN(run_gemm(h, g, st, "vit fc2"));
            TAP("vit", i, "d_mlp", dst, (size_t)NS * hid);
            pend = dst;
        }
    }
    // hidden_states[-2][:, 1:] = hidden + pending fc2 output, CLS dropped, cast to the projector's operand type
    const bool p16 = f16 && h->proj_fp16 != 0;       // operand type of the selected features and the projector
    const bool psc = p16 && (h->proj_fs_shift != 0 || h->proj_mid_shift != 0);      // scaled forms only where the proof asked for a scale
    const float s_fs = psc ? std::ldexp(1.0f, -h->proj_fs_shift) : 1.0f, s_mid = psc ? std::ldexp(1.0f, -h->proj_mid_shift) : 1.0f;
    HIPCHK(h, vqs::launch_drop_cls_cast(w.hidden, pend, w.feat_in, N, P, hid, st, f16, p16 ? 1 : 0, s_fs), "feature select");
    TAP("vit", -1, "feat_in", w.feat_in, (size_t)NP * hid);
    GETW(p0w, "mm_projector.0.weight", (int64_t)D * hid);
    GETW(p0b, "mm_projector.0.bias", D);
    GETW(p2w, "mm_projector.2.weight", (int64_t)D * D);
    GETW(p2b, "mm_projector.2.bias", D);
    {
        GemmCall g{w.feat_in, p16 ? h->proj0_w16 : p0w, w.pmid};
        g.bias = p0b;
        g.f16 = psc ? 3 : (p16 ? 1 : 0);
        g.acc_scale = 1.0f / s_fs; g.out_scale = s_mid;
        g.M = NP; g.N = D; g.K = hid; g.lda = hid; g.ldw = hid; g.ldc = D; g.epi = vqs::EPI_BF16_GELU;
        RUN(run_gemm(h, g, st, "mm_projector.0"));
        TAP("vit", -1, "pmid", w.pmid, (size_t)NP * D);
    }
    {
        // the image features are a bf16 tensor of the C ABI (vqs_score reads them as such).  Round 4 wrote the fp16 result and cast it
        // (two roundings, one more pass over 2.4 GB); the fp16-operand / bf16-result instantiation of the quad kernel (round 5,
        // gemm_f16b_quad) rounds the fp32 accumulator to bf16 ONCE and writes the feature tensor itself
        GemmCall g{w.pmid, p16 ? h->proj2_w16 : p2w, d_feats};
        g.bias = p2b;
        g.f16 = psc ? 4 : (p16 ? 2 : 0);
        g.acc_scale = 1.0f / s_mid;
        g.M = NP; g.N = D; g.K = D; g.lda = D; g.ldw = D; g.ldc = D; g.epi = vqs::EPI_BF16;
        RUN(run_gemm(h, g, st, "mm_projector.2"));
    }
    return VQS_OK;
}

size_t vqs_score_workspace_bytes(const vqs_handle* h, int32_t B, int32_t L, int32_t T) {
    if (!h || B <= 0 || L < 1 || T <= 0) return 0;
    return carve_score(h, nullptr, B, L, T).total;
}

// Encoder half of the scoring pass: prompt scan, bias table, embed + splice, 24 encoder blocks, final norm
// (+ the transposed copy the reassociated cross-attention reads).  Leaves enc_out / enc_outT / enc_len in the workspace.
// e_out_f16: the encoder's output (enc_out / enc_outT) leaves as IEEE fp16 instead of bf16 -- what the precise decoder of option dec_fp16 reads
static int encoder_pass(vqs_handle* h, const ScoreWs& w, const void* d_feats, const int32_t* d_img_index,
                        const int32_t* d_input_ids, int B, int L, hipStream_t st, bool e_out_f16 = false) {
    const int T = 1;
    const vqs_config& c = h->c;
    const int P = h->P, S = L - 1 + P;
    const int D = c.d_model, I = h->I, F = c.d_ff, H = c.n_heads, V = c.vocab;
    const int M = B * S, MT = B * T;
    (void)P; (void)M; (void)MT; (void)F; (void)I; (void)V;
    h->tap_outer = B;
    GETW(shared, "shared.weight", (int64_t)V * D);
    GETW(enc_rel, "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (int64_t)c.rel_buckets * H);

    HIPCHK(h, hipMemsetAsync(w.flags, 0, 4 * sizeof(int), st), "memset flags");
    HIPCHK(h, vqs::launch_prompt_scan(d_input_ids, B, L, P, w.sent_pos, w.enc_len, w.flags, st), "prompt_scan");
    HIPCHK(h, vqs::launch_relpos_table(enc_rel, h->lut_bidir, h->lut_causal, h->lut_len, c.rel_buckets, w.enc_table, H, S,
                                       nullptr, 1, st), "encoder bias table");
    HIPCHK(h, vqs::launch_embed_splice(d_input_ids, w.sent_pos, w.enc_len, d_img_index, shared, (const bf16_t*)d_feats,
                                       w.hidden, B, L, P, D, V, st), "embed_splice");
    TAP("enc", -1, "emb", w.hidden, (size_t)M * D);

    // ---------------- encoder (same pending-delta protocol as the vision tower)
    // Fused residual + RMSNorm (VQS_FUSED_NORM=1, off by default): the o / wo GEMM epilogue updates the fp32 stream in
    // place, writes the NEXT norm's operand x*ln_w (without the per-row 1/rms) and per-tile partial row sums of squares;
    // the consuming qkv / wi GEMM scales its accumulator rows by 1/rms.  Bit-exact repeatable and parity-tested, but the
    // read-modify-write epilogue costs the producer GEMMs +0.55 ms per launch against 0.64 ms for the norm kernel it
    // removes (+1 % end to end, and it lowers the GEMM's own roofline fraction): kept as a lab path.
    const bool fused = h->fused_norm != 0 && (h->gemm_variant == 3 || h->gemm_variant == 11 || h->gemm_variant == 5 || h->gemm_variant == 7);   // the fused epilogue's launches are not quad-eligible: they run the 8-wave forms
    const int parts = (D + 255) / 256;
    bool scaled = false;          // w.xn holds an un-normalised operand whose row sums are in w.rowss
    const bf16_t* pend = nullptr;
    const bf16_t* pend_attn = nullptr;   // deferred store (see the vision tower): the attention delta the stream has not absorbed yet
    const bool defer = h->norm_defer != 0 && !fused;
    // option enc_fp16: the attention side on IEEE fp16 tensors (see the option's comment in vqs_handle): xn (both norms), q / k / v, P, the
    // attention output are fp16 and q|k|v, o, wi read their fp16 weight copies; o's and wo's results (the deltas) and the gated product stay
    // bf16, wo runs as before.  The final norm's output (the tensor the decoder reads) stays bf16.
    // Values 2 / 3 (round 6, the per-call-site A/B of VERDICT r5 item 4): only the attention sub-block (norm0 output, q / k / v, P, attention
    // output; q|k|v and o weights) / only the FFN's input side (norm1 output; wi weights) on fp16, the other on bf16.
    const bool e16 = h->enc_fp16 != 0;
    const bool e16a = h->enc_fp16 == 1 || h->enc_fp16 == 2, e16f = h->enc_fp16 == 1 || h->enc_fp16 == 3;
    if (e16 && (h->gemm_variant != 3 || fused))
        return fail(h, VQS_ERR_STATE, "score: the fp16 encoder attention side (option enc_fp16, default 1) needs gemm_variant 3 and fused_norm 0 -- its linears "
                                      "exist in the quad form only; set enc_fp16=0 to A/B other GEMM forms");
    auto consume = [&](GemmCall& g) {
        if (scaled) {
            g.rowss_in = w.rs; g.rowss_parts = 0;
        }
    };
    auto produce = [&](GemmCall& g, const bf16_t* next_ln) {
        g.epi = vqs::EPI_RESID_RMS; g.C = w.xn; g.ldc = D; g.hres = w.hidden; g.ldh = D; g.lnw = next_ln; g.rowss_out = w.rowss;
    };
    for (int i = 0; i < c.enc_layers; ++i) {
        const std::string p = "encoder.block." + std::to_string(i) + ".";
        GETW(ln0, p + "layer.0.layer_norm.weight", D);
        GETW(ow, p + "layer.0.SelfAttention.o.weight", (int64_t)D * I);
        GETW(ln1, p + "layer.1.layer_norm.weight", D);
        GETW(wo, p + "layer.1.DenseReluDense.wo.weight", (int64_t)D * F);
        if (!scaled) {
            if (pend_attn)
                HIPCHK(h, vqs::launch_rmsnorm(w.hidden, pend_attn, ln0, w.xn, M, D, c.t5_ln_eps, st, pend, true, 0, e16a), "enc rmsnorm0");
            else
                HIPCHK(h, vqs::launch_rmsnorm(w.hidden, pend, ln0, w.xn, M, D, c.t5_ln_eps, st, nullptr, true, 0, e16a), "enc rmsnorm0");
            pend = nullptr;
            pend_attn = nullptr;
        }
        TAP("enc", i, "xn0", w.xn, (size_t)M * D);
        {
            GemmCall g{w.xn, e16a ? h->enc_qkv16[i] : h->enc_qkv[i], nullptr};
            g.M = M; g.N = 3 * I; g.K = D; g.lda = D; g.ldw = D; g.ldc = 0; g.epi = vqs::EPI_HEADS;
            g.S = S; g.H = H; g.inner = I;
            g.heads[0] = w.q; g.heads[1] = w.k; g.heads[2] = w.v;
            g.f16 = e16a ? 1 : 0;
            consume(g);
            RUN(run_gemm(h, g, st, "enc qkv"));
        }
        {
            vqs::AttnParams a{w.q, w.k, w.v, w.attn, w.enc_table, w.enc_len, B, H, S, 1.0f};
            a.f16 = e16a ? 1 : 0;
            TAP("enc", i, "q", w.q, (size_t)M * I);
            TAP("enc", i, "k", w.k, (size_t)M * I);
            TAP("enc", i, "v", w.v, (size_t)M * I);
            HIPCHK(h, vqs::launch_attention(a, st), "enc attention");
            TAP("enc", i, "attn", w.attn, (size_t)M * I);
        }
        {
            GemmCall g{w.attn, e16a ? h->enc_o16[i] : ow, w.delta};
            g.M = M; g.N = D; g.K = I; g.lda = I; g.ldw = I; g.ldc = D; g.epi = vqs::EPI_BF16;
            g.f16 = e16a ? 2 : 0;                // fp16 operands, bf16 delta
            if (fused) produce(g, ln1);
            RUN(run_gemm(h, g, st, "enc o"));
            if (!fused) TAP("enc", i, "d_attn", w.delta, (size_t)M * D);
            if (fused) HIPCHK(h, vqs::launch_rowss_to_rs(w.rowss, parts, M, 1.0f / (float)D, c.t5_ln_eps, w.rs, st), "row 1/rms");
        }
        if (fused) {
            scaled = true;
        } else {
            HIPCHK(h, vqs::launch_rmsnorm(w.hidden, w.delta, ln1, w.xn, M, D, c.t5_ln_eps, st, nullptr, !defer, 0, e16f), "enc rmsnorm1");
            if (defer) pend_attn = w.delta;
            scaled = false;
            TAP("enc", i, "xn1", w.xn, (size_t)M * D);
        }
        {
            GemmCall g{w.xn, e16f ? h->enc_wi16[i] : h->enc_wi[i], w.ff};
            g.M = M; g.N = 2 * F; g.K = D; g.lda = D; g.ldw = D; g.ldc = F; g.epi = vqs::EPI_GATED;
            g.f16 = e16f ? 2 : 0;                // fp16 operands, bf16 gated product
            consume(g);
            RUN(run_gemm(h, g, st, "enc wi"));
            TAP("enc", i, "ff", w.ff, (size_t)M * F);
        }
        {
            bf16_t* dst = defer ? w.delta2 : w.delta;
            GemmCall g{w.ff, wo, dst};
            g.M = M; g.N = D; g.K = F; g.lda = F; g.ldw = F; g.ldc = D; g.epi = vqs::EPI_BF16;
            if (fused && i + 1 < c.enc_layers) {
                GETW(ln0_next, "encoder.block." + std::to_string(i + 1) + ".layer.0.layer_norm.weight", D);
                produce(g, ln0_next);
                RUN(run_gemm(h, g, st, "enc wo"));
                HIPCHK(h, vqs::launch_rowss_to_rs(w.rowss, parts, M, 1.0f / (float)D, c.t5_ln_eps, w.rs, st), "row 1/rms");
                scaled = true;
                pend = nullptr;
            } else {
                RUN(run_gemm(h, g, st, "enc wo"));
                TAP("enc", i, "d_ff", dst, (size_t)M * D);
                scaled = false;
                pend = dst;
            }
        }
    }
    {
        GETW(fin, "encoder.final_layer_norm.weight", D);
        if (pend_attn)
            HIPCHK(h, vqs::launch_rmsnorm(w.hidden, pend_attn, fin, w.enc_out, M, D, c.t5_ln_eps, st, pend, true, 0, e_out_f16), "enc final norm");
        else
            HIPCHK(h, vqs::launch_rmsnorm(w.hidden, pend, fin, w.enc_out, M, D, c.t5_ln_eps, st, nullptr, true, 0, e_out_f16), "enc final norm");
        if (h->cross_mode != 0)
            HIPCHK(h, vqs::launch_transpose_pad(w.enc_out, w.enc_outT, B, S, D, w.S_pad, st), "enc_out transpose");
    }

    return VQS_OK;
}

// Split-K factor of a decoder nn.Linear: K-slices run as batch entries of the persistent kernel (operand pointers advance by
// K/s columns, fp32 partial tiles go to scratch), one pass then sums the slices in a fixed order -- deterministic, no atomics.
// s = the largest divisor of K/64 in 2..16 with tiles * s <= 256 CUs and >= 4 K-tiles per slice, where `tiles` counts `m_tiles`
// M-tiles per N-tile: a function of the WEIGHT's shape only.  The slicing fixes the order in which a row's fp32 partial sums
// are added, so it must not depend on how many rows the launch has -- a pair's score in a 256-pair batch has to be bit-equal to
// its score in a 4-pair batch (reference contract: independent cells, score.py:104-106).  Round 2 derived s from the launch's tile
// count (bits depended on ceil(MT / 256)); round 3 kept an MT <= 1024 cap and a scratch-size fallback that were M-dependent
// switches of the same kind: both are gone -- the partials have their own workspace region sized by the architecture.
static int dec_slices(const vqs_handle* h, int N, int K, int m_tiles) {
    const int tiles = m_tiles * ((N + 255) / 256);
    const int nt = K / 64;
    int sk = 1;
    if (h->splitk && (K % 64) == 0 && (N % 8) == 0)
        for (int s2 = 2; s2 <= 16; ++s2)
            if (nt % s2 == 0 && tiles * s2 <= 256 && nt / s2 >= 4) sk = s2;
    return sk;
}

// A decoder nn.Linear of the bf16 decoder (rounds 1-3; vqs_generate and option dec_precise=0): out[MT, N] (bf16) = A[MT, K] . W[N, K]^T.
// Two M-tiles are assumed for the slice count because that is the bench batch (MT = 512).
static int dec_linear(vqs_handle* h, const bf16_t* A, const bf16_t* W, bf16_t* out, int MT, int N, int K, float* scratch,
                      size_t scratch_bytes, hipStream_t st, const char* what) {
    const int sk = dec_slices(h, N, K, 2);
    if (sk > 1) {
        if ((size_t)sk * MT * N * sizeof(float) > scratch_bytes) return fail(h, VQS_ERR_WORKSPACE, std::string(what) + ": decoder scratch too small");
        GemmCall g{A, W, scratch};
        g.M = MT; g.N = N; g.K = K / sk; g.lda = K; g.ldw = K; g.ldc = N; g.epi = vqs::EPI_F32;
        g.batch = sk; g.sA = K / sk; g.sW = K / sk; g.sC = (long long)MT * N;
        RUN(run_gemm(h, g, st, what));
        HIPCHK(h, vqs::launch_reduce_slices(scratch, sk, (size_t)MT * N, out, st), "split-K reduce");
        return VQS_OK;
    }
    GemmCall g{A, W, out};
    g.M = MT; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = N; g.epi = vqs::EPI_BF16;
    return run_gemm(h, g, st, what);
}

// A decoder nn.Linear of the PRECISE decoder: A2 is a split-bf16 tensor, planes [2][MT][K]; the two planes run as 2*MT stacked
// rows of ONE GEMM over the same weights (and split-K slices as batch entries) into fp32 partials [slices][2*MT][N], which
// launch_sum_planes adds -- hi rows first, slices in index order -- into `out` in the form `mode` names (fp32 / split / gated split).
// The activation thus enters the bf16 MFMA with 16 significant bits and the result is never rounded to bf16.
static int dec_linear_split(vqs_handle* h, const bf16_t* A2, const bf16_t* W, int MT, int N, int K, float* scratch, size_t scratch_bytes,
                            int mode, void* out, int ld_out, long long out_plane, hipStream_t st, const char* what) {
    const int sk = dec_slices(h, N, K, 4);
    if ((size_t)sk * 2 * MT * N * sizeof(float) > scratch_bytes) return fail(h, VQS_ERR_WORKSPACE, std::string(what) + ": decoder scratch too small");
    GemmCall g{A2, W, scratch};
    g.M = 2 * MT; g.N = N; g.K = K / sk; g.lda = K; g.ldw = K; g.ldc = N; g.epi = vqs::EPI_F32;
    g.batch = sk; g.sA = K / sk; g.sW = K / sk; g.sC = (long long)2 * MT * N;
    RUN(run_gemm(h, g, st, what));
    HIPCHK(h, vqs::launch_sum_planes(scratch, sk, (long long)2 * MT * N, MT, N, N, mode, out, ld_out, out_plane, st), "sum of stacked partials");
    return VQS_OK;
}

// Decoder half: T teacher-forced rows per pair over the encoder output already in the workspace -> fp32 logits
// [B*T, ldl].  d_labels[b*ld_labels + t] are the target ids (decoder input = shift_right, HF modeling_t5.py:618-637).
// pos0 / cached (vqs_generate): the T rows are decoder positions pos0 .. pos0+T-1 of an incremental decode; their self
// attention K/V rows are appended to w.kvcache and the queries attend to positions 0 .. pos0+T-1 of it (T = 1 per step).
static int decoder_pass(vqs_handle* h, const ScoreWs& w, const int32_t* d_labels, int ld_labels, int B, int L, int T,
                        hipStream_t st, int pos0 = 0, bool cached = false) {
    const vqs_config& c = h->c;
    const int P = h->P, S = L - 1 + P;
    const int D = c.d_model, I = h->I, F = c.d_ff, H = c.n_heads, V = c.vocab;
    const int M = B * S, MT = B * T;
    (void)P; (void)M; (void)MT; (void)F; (void)I; (void)V;
    h->tap_outer = B;
    // ---------------- decoder (teacher forced, T rows per pair)
    GETW(shared, "shared.weight", (int64_t)V * D);
    float* scratch = w.dscratch;
    const size_t scratch_bytes = w.dscratch_bytes;
    GETW(dec_rel, "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (int64_t)c.rel_buckets * H);
    const int TB = cached ? w.Tc : T;                          // row length of the decoder bias table
    if (!cached || pos0 == 0)
        HIPCHK(h, vqs::launch_relpos_table(dec_rel, h->lut_bidir, h->lut_causal, h->lut_len, c.rel_buckets, nullptr, H, S,
                                           w.dec_table, TB, st), "decoder bias table");
    HIPCHK(h, vqs::launch_decoder_embed(d_labels, ld_labels, shared, w.dhid, B, T, D, V, st, pos0), "decoder embed");
    TAP("dec", -1, "emb", w.dhid, (size_t)MT * D);
    const bf16_t* dpend = nullptr;
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = "decoder.block." + std::to_string(i) + ".";
        GETW(ln0, p + "layer.0.layer_norm.weight", D);
        GETW(so, p + "layer.0.SelfAttention.o.weight", (int64_t)D * I);
        GETW(ln1, p + "layer.1.layer_norm.weight", D);
        GETW(cq, p + "layer.1.EncDecAttention.q.weight", (int64_t)I * D);
        GETW(co, p + "layer.1.EncDecAttention.o.weight", (int64_t)D * I);
        GETW(ln2, p + "layer.2.layer_norm.weight", D);
        GETW(wo, p + "layer.2.DenseReluDense.wo.weight", (int64_t)D * F);

        HIPCHK(h, vqs::launch_rmsnorm(w.dhid, dpend, ln0, w.dxn, MT, D, c.t5_ln_eps, st), "dec rmsnorm0");
        dpend = nullptr;
        TAP("dec", i, "xn0", w.dxn, (size_t)MT * D);
        RUN(dec_linear(h, w.dxn, h->dec_qkv[i], w.dqkv, MT, 3 * I, D, scratch, scratch_bytes, st, "dec self qkv"));
        TAP("dec", i, "qkv", w.dqkv, (size_t)MT * 3 * I);
        if (cached) {
            // append this step's k | v rows (columns I .. 3I of dqkv) to layer i's cache at position pos0, then attend over
            // positions 0 .. pos0+T-1 of the cache
            bf16_t* cl = w.kvcache + (size_t)i * B * w.Tc * 2 * I;
            HIPCHK(h, hipMemcpy2DAsync(cl + (size_t)pos0 * 2 * I, (size_t)w.Tc * 2 * I * sizeof(bf16_t), w.dqkv + I,
                                       (size_t)3 * I * sizeof(bf16_t), (size_t)2 * I * sizeof(bf16_t) * T, B,
                                       hipMemcpyDeviceToDevice, st), "kv cache append");
            vqs::DecAttnParams a{w.dqkv, cl, cl + I, w.dattn, w.dec_table, nullptr, B, H, T, pos0 + T, 3 * I, 2 * I, 0};
            a.kv_stride_b = (long long)w.Tc * 2 * I;
            a.qpos0 = pos0;
            a.bias_ld = TB;
            HIPCHK(h, vqs::launch_decoder_attention(a, st), "dec self attention (cached)");
            TAP("dec", i, "sattn", w.dattn, (size_t)MT * I);
        } else {
            vqs::DecAttnParams a{w.dqkv, w.dqkv + I, w.dqkv + 2 * I, w.dattn, w.dec_table, nullptr, B, H, T, T, 3 * I, 3 * I, 0};
            HIPCHK(h, vqs::launch_decoder_attention(a, st), "dec self attention");
            TAP("dec", i, "sattn", w.dattn, (size_t)MT * I);
        }
        RUN(dec_linear(h, w.dattn, so, w.ddelta, MT, D, I, scratch, scratch_bytes, st, "dec self o"));
        TAP("dec", i, "d_self", w.ddelta, (size_t)MT * D);
        HIPCHK(h, vqs::launch_rmsnorm(w.dhid, w.ddelta, ln1, w.dxn, MT, D, c.t5_ln_eps, st), "dec rmsnorm1");
        TAP("dec", i, "xn1", w.dxn, (size_t)MT * D);
        RUN(dec_linear(h, w.dxn, cq, w.dq, MT, I, D, scratch, scratch_bytes, st, "dec cross q"));
        TAP("dec", i, "cq", w.dq, (size_t)MT * I);
        if (h->cross_mode == 0) {
            // direct form (what HF executes): K|V projection of the whole encoder output for this layer
            {
                GemmCall g{w.enc_out, h->dec_ckv[i], nullptr};
                g.M = M; g.N = 2 * I; g.K = D; g.lda = D; g.ldw = D; g.ldc = 0; g.epi = vqs::EPI_HEADS;
                g.S = S; g.H = H; g.inner = I;
                g.heads[0] = w.ck; g.heads[1] = w.cv; g.heads[2] = nullptr;
                RUN(run_gemm(h, g, st, "dec cross kv"));
            }
            vqs::DecAttnParams a{w.dq, w.ck, w.cv, w.dattn, nullptr, w.enc_len, B, H, T, S, I, 0, 1};
            HIPCHK(h, vqs::launch_decoder_attention(a, st), "dec cross attention");
        } else {
            // Reassociated cross-attention.  With E = encoder output [S,D] and T (<=16) decoder rows per pair,
            //   scores_h = q_h (E Wk_h^T)^T = (q_h Wk_h) E^T          and       out_h = P_h (E Wv_h^T) = (P_h E) Wv_h^T,
            // so the per-layer [B*S, D] x [D, 2I] projection (4*S*D*I FLOPs per pair and layer, 98 % of the decoder)
            // becomes five small batched GEMMs over E itself (~30x fewer FLOPs); same function, fp32 accumulation.
            GETW(cv_w, p + "layer.1.EncDecAttention.v.weight", (int64_t)I * D);
            const int R = T * H;        // rows per pair, ordered (t, h)
            {   // q'[(b,t), h, :] = q[(b,t), h*64:(h+1)*64] . Wk_h            batched over heads
                GemmCall g{w.dq, h->dec_ckT[i], w.cqk};
                g.M = MT; g.N = D; g.K = 64; g.lda = I; g.ldw = I; g.ldc = H * D; g.epi = vqs::EPI_BF16;
                g.batch = H; g.sA = 64; g.sW = 64; g.sC = D;
                RUN(run_gemm(h, g, st, "cross q.Wk"));
                TAP("dec", i, "cqk", w.cqk, (size_t)MT * H * D);
            }
            {   // scores[b] [R, S] = q'[b] [R, D] . E[b]^T                      batched over pairs
                GemmCall g{w.cqk, w.enc_out, w.cscores};
                g.M = R; g.N = S; g.K = D; g.lda = D; g.ldw = D; g.ldc = w.S_pad; g.epi = vqs::EPI_F32;
                g.batch = B; g.sA = (long long)R * D; g.sW = (long long)S * D; g.sC = (long long)R * w.S_pad;
                RUN(run_gemm(h, g, st, "cross scores"));
                TAP("dec", i, "cscores", w.cscores, (size_t)MT * H * w.S_pad);
            }
            HIPCHK(h, vqs::launch_masked_softmax(w.cscores, w.cprobs, w.enc_len, B, R, w.S_pad, st), "cross softmax");
            TAP("dec", i, "cprobs", w.cprobs, (size_t)MT * H * w.S_pad);
            {   // ctx[b] [R, D] = P[b] [R, S_pad] . E[b]  (E^T is K-contiguous)   batched over pairs
                GemmCall g{w.cprobs, w.enc_outT, w.cctx};
                g.M = R; g.N = D; g.K = w.S_pad; g.lda = w.S_pad; g.ldw = w.S_pad; g.ldc = D; g.epi = vqs::EPI_BF16;
                g.batch = B; g.sA = (long long)R * w.S_pad; g.sW = (long long)D * w.S_pad; g.sC = (long long)R * D;
                g.no_stream = 1;      // K = S_pad = 640: ten slabs per item -- the stream form's pipeline fill per item costs more than its
                                  // deeper ring gains (0.50 vs 0.41 ms per launch, profiles/r4_call4_*); the scores launch (K = D) streams
            RUN(run_gemm(h, g, st, "cross P.E"));
                TAP("dec", i, "cctx", w.cctx, (size_t)MT * H * D);
            }
            {   // out[(b,t), h*64:(h+1)*64] = ctx[(b,t), h, :] . Wv_h^T          batched over heads
                GemmCall g{w.cctx, cv_w, w.dattn};
                g.M = MT; g.N = 64; g.K = D; g.lda = H * D; g.ldw = D; g.ldc = I; g.epi = vqs::EPI_BF16;
                g.batch = H; g.sA = D; g.sW = (long long)64 * D; g.sC = 64;
                RUN(run_gemm(h, g, st, "cross ctx.Wv"));
            }
        }
        TAP("dec", i, "cattn", w.dattn, (size_t)MT * I);
        RUN(dec_linear(h, w.dattn, co, w.ddelta, MT, D, I, scratch, scratch_bytes, st, "dec cross o"));
        TAP("dec", i, "d_cross", w.ddelta, (size_t)MT * D);
        HIPCHK(h, vqs::launch_rmsnorm(w.dhid, w.ddelta, ln2, w.dxn, MT, D, c.t5_ln_eps, st), "dec rmsnorm2");
        TAP("dec", i, "xn2", w.dxn, (size_t)MT * D);
        {
            GemmCall g{w.dxn, h->dec_wi[i], w.dff};
            g.M = MT; g.N = 2 * F; g.K = D; g.lda = D; g.ldw = D; g.ldc = F; g.epi = vqs::EPI_GATED;
            RUN(run_gemm(h, g, st, "dec wi"));
            TAP("dec", i, "ff", w.dff, (size_t)MT * F);
        }
        RUN(dec_linear(h, w.dff, wo, w.ddelta, MT, D, F, scratch, scratch_bytes, st, "dec wo"));
        TAP("dec", i, "d_ff", w.ddelta, (size_t)MT * D);
        dpend = w.ddelta;
    }
    {
        GETW(fin, "decoder.final_layer_norm.weight", D);
        GETW(head, "lm_head.weight", (int64_t)V * D);
        HIPCHK(h, vqs::launch_rmsnorm(w.dhid, dpend, fin, w.dxn, MT, D, c.t5_ln_eps, st), "dec final norm");
        GemmCall g{w.dxn, head, w.logits};
        g.M = MT; g.N = V; g.K = D; g.lda = D; g.ldw = D; g.ldc = w.ldl; g.epi = vqs::EPI_F32;
        RUN(run_gemm(h, g, st, "lm_head"));
    }
    return VQS_OK;
}

// The scoring decoder of round 4 (option dec_precise, default 1): the same T teacher-forced rows per pair, but every activation the
// error attribution (profiles/r4_error_attribution.md) shows to matter is held with 16 significant bits or in fp32:
//   split-bf16 (planes hi | lo, consumed as 2*MT stacked rows of one GEMM): norm outputs, self-attention output, P.E context,
//     ctx.Wv output, gated FFN product, final norm output (the lm_head operand);
//   fp32, never rounded: q|k|v of the self-attention and the three sub-layer outputs added into the fp32 stream;
//   bf16 as before: the cross-attention SCORE path (q, q.Wk, probabilities: <= 2e-4 of |delta log P| together), which reads the hi
//     plane of the norm output -- bit for bit what rounds 1-3 computed there.
// Same function as decoder_pass (HF modeling_t5.py:404-509), same launch sequence; the reference rounds every one of these
// tensors to bf16 (mm_utils.py:228), so this is a deviation TOWARDS fp32 arithmetic, like the fp32 residual stream.
static int decoder_pass_precise(vqs_handle* h, const ScoreWs& w, const int32_t* d_labels, int ld_labels, int B, int L, int T, hipStream_t st) {
    const vqs_config& c = h->c;
    const bool x16 = h->dec_fp16 != 0;      // the encoder pass before this call wrote enc_out / enc_outT as fp16 tensors (vqs_score)
    const int P = h->P, S = L - 1 + P;
    const int D = c.d_model, I = h->I, F = c.d_ff, H = c.n_heads, V = c.vocab;
    const int MT = B * T;
    h->tap_outer = B;
    GETW(shared, "shared.weight", (int64_t)V * D);
    GETW(dec_rel, "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", (int64_t)c.rel_buckets * H);
    float* scratch = w.dscratch;
    const size_t sb = w.dscratch_bytes;
    const long long pD = (long long)MT * D, pI = (long long)MT * I, pF = (long long)MT * F, pC = (long long)MT * H * D;   // plane sizes
    HIPCHK(h, vqs::launch_relpos_table(dec_rel, h->lut_bidir, h->lut_causal, h->lut_len, c.rel_buckets, nullptr, H, S, w.dec_table, T, st),
           "decoder bias table");
    HIPCHK(h, vqs::launch_decoder_embed(d_labels, ld_labels, shared, w.dhid, B, T, D, V, st, 0), "decoder embed");
    TAP("dec", -1, "emb", w.dhid, (size_t)MT * D);
    const float* dpend = nullptr;
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = "decoder.block." + std::to_string(i) + ".";
        GETW(ln0, p + "layer.0.layer_norm.weight", D);
        GETW(so, p + "layer.0.SelfAttention.o.weight", (int64_t)D * I);
        GETW(ln1, p + "layer.1.layer_norm.weight", D);
        GETW(cq, p + "layer.1.EncDecAttention.q.weight", (int64_t)I * D);
        GETW(cv_w, p + "layer.1.EncDecAttention.v.weight", (int64_t)I * D);
        GETW(co, p + "layer.1.EncDecAttention.o.weight", (int64_t)D * I);
        GETW(ln2, p + "layer.2.layer_norm.weight", D);
        GETW(wo, p + "layer.2.DenseReluDense.wo.weight", (int64_t)D * F);

        // ---- causal self-attention
        HIPCHK(h, vqs::launch_rmsnorm_split(w.dhid, dpend, ln0, w.dxn, pD, MT, D, c.t5_ln_eps, st), "dec rmsnorm0");
        dpend = nullptr;
        TAP2("dec", i, "xn0", w.dxn, pD);
        RUN(dec_linear_split(h, w.dxn, h->dec_qkv[i], MT, 3 * I, D, scratch, sb, vqs::SUM_F32, w.dqkv32, 3 * I, 0, st, "dec self qkv"));
        TAP("dec", i, "qkv", w.dqkv32, (size_t)MT * 3 * I);
        {
            vqs::DecAttnParams a{reinterpret_cast<const bf16_t*>(w.dqkv32), reinterpret_cast<const bf16_t*>(w.dqkv32 + I),
                                 reinterpret_cast<const bf16_t*>(w.dqkv32 + 2 * I), w.dattn, w.dec_table, nullptr, B, H, T, T, 3 * I, 3 * I, 0};
            a.precise = 1;
            a.out_plane = pI;
            HIPCHK(h, vqs::launch_decoder_attention(a, st), "dec self attention");
        }
        TAP2("dec", i, "sattn", w.dattn, pI);
        RUN(dec_linear_split(h, w.dattn, so, MT, D, I, scratch, sb, vqs::SUM_F32, w.ddelta32, D, 0, st, "dec self o"));
        TAP("dec", i, "d_self", w.ddelta32, (size_t)MT * D);
        // ---- cross-attention (reassociated, vqs_api.cpp decoder_pass): score path in bf16 off the hi plane, value path precise
        HIPCHK(h, vqs::launch_rmsnorm_split(w.dhid, w.ddelta32, ln1, w.dxn, pD, MT, D, c.t5_ln_eps, st), "dec rmsnorm1");
        TAP2("dec", i, "xn1", w.dxn, pD);
        // option dec_fp16 (x16): q from BOTH planes of the split norm output, rounded once to fp16; q.Wk, the probabilities and E are fp16 tensors
        if (x16)
            RUN(dec_linear_split(h, w.dxn, cq, MT, I, D, scratch, sb, vqs::SUM_F16, w.dq, I, 0, st, "dec cross q"));
        else
            RUN(dec_linear(h, w.dxn, cq, w.dq, MT, I, D, scratch, sb, st, "dec cross q"));
        TAP("dec", i, "cq", w.dq, (size_t)MT * I);
        const int R = T * H;        // rows per pair, ordered (t, h)
        {   // q'[(b,t), h, :] = q[(b,t), h*64:(h+1)*64] . Wk_h            batched over heads
            GemmCall g{w.dq, x16 ? h->dec_ckT16[i] : h->dec_ckT[i], w.cqk};
            g.M = MT; g.N = D; g.K = 64; g.lda = I; g.ldw = I; g.ldc = H * D; g.epi = vqs::EPI_BF16;
            g.batch = H; g.sA = 64; g.sW = 64; g.sC = D;
            g.f16 = x16 ? 1 : 0;
            RUN(run_gemm(h, g, st, "cross q.Wk"));
            TAP("dec", i, "cqk", w.cqk, (size_t)MT * H * D);
        }
        {   // scores[b] [R, S] = q'[b] [R, D] . E[b]^T                      batched over pairs
            GemmCall g{w.cqk, w.enc_out, w.cscores};
            g.M = R; g.N = S; g.K = D; g.lda = D; g.ldw = D; g.ldc = w.S_pad; g.epi = vqs::EPI_F32;
            g.batch = B; g.sA = (long long)R * D; g.sW = (long long)S * D; g.sC = (long long)R * w.S_pad;
            g.f16 = x16 ? 1 : 0;
            RUN(run_gemm(h, g, st, "cross scores"));
            TAP("dec", i, "cscores", w.cscores, (size_t)MT * H * w.S_pad);
        }
        HIPCHK(h, vqs::launch_masked_softmax(w.cscores, w.cprobs, w.enc_len, B, R, w.S_pad, st, x16), "cross softmax");
        TAP("dec", i, "cprobs", w.cprobs, (size_t)MT * H * w.S_pad);
        {   // ctx[b] [R, D] = P[b] [R, S_pad] . E[b]   -> split-bf16: hi plane = the tensor of rounds 1-3, lo plane behind it
            GemmCall g{w.cprobs, w.enc_outT, w.cctx};
            g.M = R; g.N = D; g.K = w.S_pad; g.lda = w.S_pad; g.ldw = w.S_pad; g.ldc = D; g.epi = vqs::EPI_BF16;
            g.batch = B; g.sA = (long long)R * w.S_pad; g.sW = (long long)D * w.S_pad; g.sC = (long long)R * D;
            g.split_off = pC;
            g.f16 = x16 ? 2 : 0;          // fp16 operands (P, E^T), split-bf16 result
            g.no_stream = 1;      // K = S_pad = 640: ten slabs per item -- the stream form's pipeline fill per item costs more than its
                                  // deeper ring gains (0.50 vs 0.41 ms per launch, profiles/r4_call4_*); the scores launch (K = D) streams
            RUN(run_gemm(h, g, st, "cross P.E"));
            TAP2("dec", i, "cctx", w.cctx, pC);
        }
        {   // out[(b,t), h*64:(h+1)*64] = ctx[(b,t), h, :] . Wv_h^T          batched over heads; both planes as 2*MT stacked rows
            GemmCall g{w.cctx, cv_w, scratch};
            g.M = 2 * MT; g.N = 64; g.K = D; g.lda = H * D; g.ldw = D; g.ldc = I; g.epi = vqs::EPI_F32;
            g.batch = H; g.sA = D; g.sW = (long long)64 * D; g.sC = 64;
            if ((size_t)2 * MT * I * sizeof(float) > sb) return fail(h, VQS_ERR_WORKSPACE, "cross ctx.Wv: decoder scratch too small");
            RUN(run_gemm(h, g, st, "cross ctx.Wv"));
            HIPCHK(h, vqs::launch_sum_planes(scratch, 1, 0, MT, I, I, vqs::SUM_SPLIT, w.dattn, I, pI, st), "cross ctx.Wv planes");
        }
        TAP2("dec", i, "cattn", w.dattn, pI);
        RUN(dec_linear_split(h, w.dattn, co, MT, D, I, scratch, sb, vqs::SUM_F32, w.ddelta32, D, 0, st, "dec cross o"));
        TAP("dec", i, "d_cross", w.ddelta32, (size_t)MT * D);
        // ---- gated FFN
        HIPCHK(h, vqs::launch_rmsnorm_split(w.dhid, w.ddelta32, ln2, w.dxn, pD, MT, D, c.t5_ln_eps, st), "dec rmsnorm2");
        TAP2("dec", i, "xn2", w.dxn, pD);
        RUN(dec_linear_split(h, w.dxn, h->dec_wi[i], MT, 2 * F, D, scratch, sb, vqs::SUM_GATED_SPLIT, w.dff, F, pF, st, "dec wi"));
        TAP2("dec", i, "ff", w.dff, pF);
        RUN(dec_linear_split(h, w.dff, wo, MT, D, F, scratch, sb, vqs::SUM_F32, w.ddelta32, D, 0, st, "dec wo"));
        TAP("dec", i, "d_ff", w.ddelta32, (size_t)MT * D);
        dpend = w.ddelta32;
    }
    {
        GETW(fin, "decoder.final_layer_norm.weight", D);
        GETW(head, "lm_head.weight", (int64_t)V * D);
        HIPCHK(h, vqs::launch_rmsnorm_split(w.dhid, dpend, fin, w.dxn, pD, MT, D, c.t5_ln_eps, st), "dec final norm");
        if ((w.ldl % 4) != 0 || (V % 4) != 0) return fail(h, VQS_ERR_INVALID, "precise decoder: vocabulary must be a multiple of 4");
        RUN(dec_linear_split(h, w.dxn, head, MT, V, D, scratch, sb, vqs::SUM_F32, w.logits, w.ldl, 0, st, "lm_head"));
    }
    return VQS_OK;
}

int vqs_score(vqs_handle* h, const void* d_feats, const int32_t* d_img_index, const int32_t* d_input_ids,
              const int32_t* d_labels, int32_t B, int32_t L, int32_t T, float* d_lp, float* d_scores, void* d_ws,
              size_t ws_bytes, void* stream) {
    if (!h) return VQS_ERR_INVALID;
    if (!h->bound) return fail(h, VQS_ERR_STATE, "score: weights not bound");
    if (!d_feats || !d_img_index || !d_input_ids || !d_labels || !d_lp || !d_scores || !d_ws)
        return fail(h, VQS_ERR_INVALID, "score: null argument");
    if (B <= 0 || L < 1 || T <= 0 || T > 16) return fail(h, VQS_ERR_INVALID, "score: need B>0, L>=1, 1<=T<=16");
    const vqs_config& c = h->c;
    const int P = h->P, S = L - 1 + P;
    if (L - 1 > 2048) return fail(h, VQS_ERR_INVALID, "score: prompt longer than CONTEXT_LEN (2048)");
    if ((size_t)(2 * S + 96) * 16 + 32768 > 160 * 1024) return fail(h, VQS_ERR_INVALID, "score: encoder length too large for the attention kernel's bias tables");
    const ScoreWs w = carve_score(h, (char*)d_ws, B, L, T);
    if (ws_bytes < w.total) return fail(h, VQS_ERR_WORKSPACE, "score: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int V = c.vocab;

    const bool precise = h->dec_precise && h->cross_mode != 0;
    if (precise && h->dec_fp16 && h->gemm_variant != 3)
        return fail(h, VQS_ERR_STATE, "score: the fp16 cross-attention score path (option dec_fp16, default 1) needs gemm_variant 3; set dec_fp16=0 to A/B other GEMM forms");
    RUN(encoder_pass(h, w, d_feats, d_img_index, d_input_ids, B, L, st, precise && h->dec_fp16 != 0));
    if (precise)
        RUN(decoder_pass_precise(h, w, d_labels, T, B, L, T, st));
    else                                             // bf16 decoder of rounds 1-3 (option dec_precise=0, or the direct cross-attention form)
        RUN(decoder_pass(h, w, d_labels, T, B, L, T, st));
    HIPCHK(h, vqs::launch_score_head(w.logits, w.ldl, V, d_labels, d_lp, d_scores, B, T, st, w.flags), "score head");
    return VQS_OK;
}


// Greedy decoding (the reference's `model.generate`, V_3.0_README.md:316-325; HF GenerationMixin greedy search with
// T5's decoder_start_token_id = pad = 0): the encoder runs once, then step t decodes ONE new row per pair -- its
// self-attention K/V are appended to a per-layer cache in the workspace and it attends to positions 0..t of it (the
// cross-attention is the reassociated form over the encoder output, which needs no cache) -- and appends
// argmax(logits).  No host synchronisation; all max_new steps are executed, the caller truncates at the first EOS.
// d_tokens: int32 [B, max_new].
size_t vqs_generate_workspace_bytes(const vqs_handle* h, int32_t B, int32_t L, int32_t max_new) {
    if (!h || B <= 0 || L < 1 || max_new <= 0 || max_new > VQS_MAX_NEW_TOKENS) return 0;
    return carve_score(h, nullptr, B, L, 1, nullptr, max_new).total;
}

int vqs_generate(vqs_handle* h, const void* d_feats, const int32_t* d_img_index, const int32_t* d_input_ids, int32_t B,
                 int32_t L, int32_t max_new, int32_t* d_tokens, void* d_ws, size_t ws_bytes, void* stream) {
    if (!h) return VQS_ERR_INVALID;
    if (!h->bound) return fail(h, VQS_ERR_STATE, "generate: weights not bound");
    if (!d_feats || !d_img_index || !d_input_ids || !d_tokens || !d_ws) return fail(h, VQS_ERR_INVALID, "generate: null argument");
    if (B <= 0 || L < 1 || max_new <= 0 || max_new > VQS_MAX_NEW_TOKENS)
        return fail(h, VQS_ERR_INVALID, "generate: need B>0, L>=1, 1<=max_new<=" + std::to_string(VQS_MAX_NEW_TOKENS));
    if (L - 1 > 2048) return fail(h, VQS_ERR_INVALID, "generate: prompt longer than CONTEXT_LEN (2048)");
    if (h->cross_mode == 0) return fail(h, VQS_ERR_STATE, "generate: needs the reassociated cross-attention (option cross_mode=1)");
    const ScoreWs w = carve_score(h, (char*)d_ws, B, L, 1, nullptr, max_new);
    if (ws_bytes < w.total) return fail(h, VQS_ERR_WORKSPACE, "generate: workspace too small (size it with vqs_generate_workspace_bytes(B, L, max_new))");
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(h, hipMemsetAsync(d_tokens, 0, (size_t)B * max_new * sizeof(int32_t), st), "memset tokens");
    RUN(encoder_pass(h, w, d_feats, d_img_index, d_input_ids, B, L, st));
    for (int t = 0; t < max_new; ++t) {
        // one new decoder row per pair: position t, input = token t-1 (start token for t = 0), self-attention over the cache
        RUN(decoder_pass(h, w, d_tokens, max_new, B, L, 1, st, t, true));
        HIPCHK(h, vqs::launch_argmax_append(w.logits, w.ldl, h->c.vocab, d_tokens, max_new, B, 1, st, t, w.flags), "argmax");   // + status bit 1 on a non-finite logit
    }
    return VQS_OK;
}

int64_t vqs_workspace_offset(const vqs_handle* h, const char* name, int32_t B, int32_t L, int32_t T, int64_t* ld_out) {
    if (!h || !name) return -1;
    std::unordered_map<std::string, size_t> names;
    int64_t ld = 0;
    const std::string n(name);
    if (n == "vit_hidden") {
        if (B <= 0) return -1;
        carve_encode(h, nullptr, B, &names);
        ld = h->c.vis_hidden;
    } else {
        if (B <= 0 || L < 1 || T <= 0) return -1;
        const ScoreWs w = carve_score(h, nullptr, B, L, T, &names);
        if (n == "logits") ld = w.ldl;
        else if (n == "enc_in" || n == "enc_out" || n == "dec_out") ld = h->c.d_model;
        else ld = 1;
    }
    auto it = names.find(n);
    if (it == names.end()) return -1;
    if (ld_out) *ld_out = ld;
    return (int64_t)it->second;
}

int vqs_profile_enable(vqs_handle* h, int32_t on) {
    if (!h) return VQS_ERR_INVALID;
    h->prof = on != 0;
    return VQS_OK;
}

int vqs_profile_read(vqs_handle* h, double* gemm_ms, double* gemm_flops, int32_t reset) {
    if (!h) return VQS_ERR_INVALID;
    double ms = 0.0;
    std::map<std::string, std::array<double, 3>> by;      // label -> {launches, ms, flops}
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        HIPCHK(h, hipEventSynchronize(h->ev[i + 1]), "hipEventSynchronize");
        float t = 0.f;
        HIPCHK(h, hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]), "hipEventElapsedTime");
        ms += t;
        if (i / 2 < h->ev_what.size()) {
            auto& a = by[h->ev_what[i / 2].first];
            a[0] += 1.0; a[1] += t; a[2] += h->ev_what[i / 2].second;
        }
    }
    h->prof_report.clear();
    for (const auto& kv : by) {
        char line[256];
        snprintf(line, sizeof line, "%-20s launches %6.0f  ms %10.3f  TFLOP/s %8.1f\n", kv.first.c_str(), kv.second[0], kv.second[1],
                 kv.second[1] > 0 ? kv.second[2] / kv.second[1] / 1e9 : 0.0);
        h->prof_report += line;
    }
    if (gemm_ms) *gemm_ms = ms;
    if (gemm_flops) *gemm_flops = h->prof_flops;
    const int n = (int)(h->ev_used / 2);
    if (reset) {
        h->ev_used = 0;
        h->ev_what.clear();
        h->prof_flops = 0.0;
        h->prof_bytes = 0.0;
    }
    return n;
}

const char* vqs_profile_report(vqs_handle* h) { return h ? h->prof_report.c_str() : ""; }

int vqs_profile_bytes(vqs_handle* h, double* gemm_bytes) {
    if (!h || !gemm_bytes) return VQS_ERR_INVALID;
    *gemm_bytes = h->prof_bytes;
    return VQS_OK;
}

// ---------------------------------------------------------------------------- single-kernel entry points
// GEMM with the fused residual + RMSNorm pieces (tests): epilogue 7 (producer: hres/lnw/rowss_out) and/or a consumer row
// scale (rowss_in/rowss_parts/rs_invd/rs_eps) in front of epilogues 0, 3, 5, 6.  Persistent variants (3, 5, 7) only.
int vqs_gemm_rms(const void* A, const void* W, void* C, float* hres, const void* lnw, float* rowss_out, const float* rowss_in,
                 int32_t rowss_parts, float rs_invd, float rs_eps, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                 int32_t ldc, int32_t epilogue, int32_t S, int32_t H, int32_t variant, void* stream) {
    vqs::GemmParams p;
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = nullptr; p.resid = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.S = S > 0 ? S : 1; p.H = H; p.inner = H * 64 > 0 ? H * 64 : 1;
    const size_t per = (size_t)(S > 0 ? M / S : 0) * H * S * 64;
    p.heads_out[0] = (bf16_t*)C;
    p.heads_out[1] = (bf16_t*)C + per;
    p.heads_out[2] = (bf16_t*)C + 2 * per;
    p.hres = hres; p.ldh = N; p.lnw = (const bf16_t*)lnw; p.rowss_out = rowss_out;
    p.rowss_in = rowss_in; p.rowss_parts = rowss_parts; p.rs_invd = rs_invd; p.rs_eps = rs_eps;
    p.tile_gm = (variant >> 8) & 0xff;     // same layout of the variant word as vqs_gemm
    p.tile_ns = (variant >> 16) & 0xff;
    return vqs::launch_gemm(p, epilogue, variant & 0xff, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_gemm(const void* A, const void* W, void* C, const void* bias, const float* resid, int32_t M, int32_t N, int32_t K,
             int32_t lda, int32_t ldw, int32_t ldc, int32_t epilogue, int32_t S, int32_t H, int32_t variant, void* stream) {
    vqs::GemmParams p;
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = C; p.bias = (const bf16_t*)bias; p.resid = resid;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
    p.S = S > 0 ? S : 1; p.H = H; p.inner = H * 64 > 0 ? H * 64 : 1;
    const size_t per = (size_t)(S > 0 ? M / S : 0) * H * S * 64;   // elements per head-major tensor
    p.heads_out[0] = (bf16_t*)C;
    p.heads_out[1] = (bf16_t*)C + per;
    p.heads_out[2] = (bf16_t*)C + 2 * per;
    p.tile_gm = (variant >> 8) & 0xff;     // bits 8-15 / 16-23 of `variant`: tile order (0 = default), see vqs.h
    p.tile_ns = (variant >> 16) & 0xff;
    p.nt_store = (variant >> 24) & 1;      // bit 24: non-temporal result stores; bits 25-26: A-panel L2 touch (0 by shape, 1 on, 2 off)
    p.l2_touch = (variant >> 25) & 3;
    p.f16 = (variant >> 27) & 3;           // bits 27-28: 0 bf16 operands and result, 1 IEEE fp16 operands and result, 2 fp16 operands / bf16 result
    if (p.f16 == 3) return VQS_ERR_INVALID;
    return vqs::launch_gemm(p, epilogue, variant & 0xff, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_attention(const void* q, const void* k, const void* v, void* out, const float* bias_table, const int32_t* key_len,
                  int32_t B, int32_t H, int32_t S, float scale, void* stream) {
    vqs::AttnParams a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, bias_table, key_len, B, H, S, scale};
    return vqs::launch_attention(a, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_debug_attention_f16(const void* q, const void* k, const void* v, void* out, const float* bias_table, const int32_t* key_len,
                            int32_t B, int32_t H, int32_t S, float scale, void* stream) {
    vqs::AttnParams a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, bias_table, key_len, B, H, S, scale};
    a.f16 = 1;
    return vqs::launch_attention(a, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_debug_norm16(int32_t kind, float* x, const void* delta, const void* delta2, int32_t store_x, const void* w, const void* b,
                     void* out, int32_t M, int32_t D, float eps, int32_t types, void* stream) {
    if (!x || !w || !out || (kind == 1 && !b) || (kind != 0 && kind != 1)) return VQS_ERR_INVALID;
    hipError_t e;
    if (kind == 0 && types == 1)          // RMSNorm, bf16 deltas in, fp16 operand out: the T5 encoder of option enc_fp16
        e = vqs::launch_rmsnorm(x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)out, M, D, eps, (hipStream_t)stream, (const bf16_t*)delta2,
                                store_x != 0, 0, true);
    else if (kind == 1 && types == 3)     // LayerNorm, fp16 deltas in, fp16 operand out: the vision tower of option vit_fp16
        e = vqs::launch_layernorm(x, (const bf16_t*)delta, (const bf16_t*)w, (const bf16_t*)b, out, 0, M, D, eps, (hipStream_t)stream,
                                  (const bf16_t*)delta2, store_x != 0, true);
    else
        return VQS_ERR_INVALID;
    return e == hipSuccess ? VQS_OK : (e == hipErrorInvalidValue ? VQS_ERR_INVALID : VQS_ERR_HIP);
}

int vqs_normalize_u8(const void* d_u8, void* d_out, int32_t N, int32_t H, int32_t W, const float* mean3, const float* std3,
                     void* stream) {
    if (!d_u8 || !d_out || !mean3 || !std3) return VQS_ERR_INVALID;
    return vqs::launch_u8_to_norm_bf16((const unsigned char*)d_u8, (bf16_t*)d_out, N, H, W, mean3, std3, (hipStream_t)stream) == hipSuccess
               ? VQS_OK : VQS_ERR_HIP;
}

int vqs_rope(void* x, const float* cos_t, const float* sin_t, int32_t B, int32_t H, int32_t S, int32_t hd, int32_t half,
             void* stream) {
    return vqs::launch_rope((bf16_t*)x, cos_t, sin_t, B, H, S, hd, half, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

// Attention with head_dim 128, grouped-query heads and an optional causal mask (the Qwen2.5-VL row)
int vqs_attention_hd(const void* q, const void* k, const void* v, void* out, const int32_t* key_len, int32_t B, int32_t H,
                     int32_t Hkv, int32_t S, int32_t hd, float scale, int32_t causal, void* stream) {
    vqs::AttnParams a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, nullptr, key_len, B, H, S, scale};
    a.hd = hd; a.Hkv = Hkv; a.causal = causal;
    return vqs::launch_attention(a, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_decoder_attention(const void* q, const void* k, const void* v, void* out, const float* bias_table,
                          const int32_t* key_len, int32_t B, int32_t H, int32_t T, int32_t S, int32_t ldq, int32_t ldk,
                          int32_t cross, void* stream) {
    vqs::DecAttnParams a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, bias_table, key_len,
                         B, H, T, S, ldq, ldk, cross};
    return vqs::launch_decoder_attention(a, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_rmsnorm(float* x, const void* delta, const void* w, void* out, int32_t M, int32_t D, float eps, void* stream) {
    return vqs::launch_rmsnorm(x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)out, M, D, eps, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

int vqs_norm_deferred(int32_t kind, float* x, const void* delta, const void* delta2, int32_t store_x, const void* w, const void* b,
                      void* out, int32_t M, int32_t D, float eps, void* stream) {
    if (!x || !delta || !w || !out || (kind == 1 && !b) || (kind != 0 && kind != 1)) return VQS_ERR_INVALID;
    const hipError_t e = kind == 0
        ? vqs::launch_rmsnorm(x, (const bf16_t*)delta, (const bf16_t*)w, (bf16_t*)out, M, D, eps, (hipStream_t)stream,
                              (const bf16_t*)delta2, store_x != 0)
        : vqs::launch_layernorm(x, (const bf16_t*)delta, (const bf16_t*)w, (const bf16_t*)b, out, 0, M, D, eps, (hipStream_t)stream,
                                (const bf16_t*)delta2, store_x != 0);
    return e == hipSuccess ? VQS_OK : (e == hipErrorInvalidValue ? VQS_ERR_INVALID : VQS_ERR_HIP);
}

int vqs_layernorm(float* x, const void* delta, const void* w, const void* b, void* out, int32_t out_f32, int32_t M, int32_t D,
                  float eps, void* stream) {
    return vqs::launch_layernorm(x, (const bf16_t*)delta, (const bf16_t*)w, (const bf16_t*)b, out, out_f32, M, D, eps, (hipStream_t)stream) == hipSuccess
               ? VQS_OK : VQS_ERR_HIP;
}

int vqs_score_head(const float* logits, int32_t ldl, int32_t V, const int32_t* labels, float* lp, float* scores, int32_t B,
                   int32_t T, void* stream) {
    return vqs::launch_score_head(logits, ldl, V, labels, lp, scores, B, T, (hipStream_t)stream) == hipSuccess ? VQS_OK : VQS_ERR_HIP;
}

}  // extern "C"
