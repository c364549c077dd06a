// Host side of the Qwen2.5-VL row of libvqs_hip (include/vqs_qwen.h): weight packing and the launch sequences of the vision
// tower and of the language-model prefill.  Restates what HF executes for the reference's one-prefill scoring pass
// (models/qwen2_5_vl/modeling_qwen2_5_vl.py; the oracle oracle/qwen25vl_oracle.py cites the lines):
//   Qwen2_5_VisionTransformerPretrainedModel.forward :408-471  -> vqs_qwen_encode_vision
//   Qwen2_5_VLModel.forward (splice) :1185-1255, Qwen2_5_VLTextModel.forward :790-873, lm_head -> vqs_qwen_score
// Heads narrower than 128 (the tower's 80) are zero-padded to 128 lanes at bind time so that one attention kernel serves
// both stacks; K dimensions are zero-padded to multiples of 64, gate|up rows interleaved in blocks of 32 for the gated
// epilogue.  No device allocation, no stream synchronisation, no CPU fallback.
#include "../../include/vqs_qwen.h"
#include "vqs_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

using vqs::bf16_t;

namespace {
constexpr int HDP = 128;   // padded head width
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct WEnt {
    const bf16_t* p;
    int64_t numel;
};
}  // namespace

struct vqs_qwen_handle {
    vqs_qwen_config c;
    std::string err;
    std::unordered_map<std::string, WEnt> w;
    bool bound = false;
    // tower heads narrower than 128 lanes whose q | k | v ranges are whole 128-column blocks (7B: 16 x 80): the qkv product keeps its
    // 3 x hidden columns (the GEMM scatters every head into the first v_hd lanes of a 128-lane slot, the other lanes are zeroed once
    // per call), attention writes its output compact and proj contracts over hidden -- 37.5 % fewer flops in both GEMMs at 80 lanes.
    // Otherwise weights are packed with every head padded to 128 rows / columns.
    bool v_compact = false;
    // Row pitch (elements) of the language model's normalised activations and of its packed gate|up weight rows.  Measured
    // (profiles/r3_call28_qwen_gemm_pitch.jsonl): the 51712 x 37888 x 3584 gate|up GEMM runs at 1.25 PFLOP/s with both operands at
    // their natural 7 KiB pitch and at 1.37 with 8 KiB rows; the other launches do not care.  Rows of 2^k bytes once the hidden
    // size is past 4 KiB; vqs_qwen_debug_option("x_pitch") overrides it before the weights are bound (tests at small sizes).
    int t_xld = 0;
    int v_hd = 0, t_hd = 0, v_kpatch = 0, v_mlp_p = 0, v_ffld = 0, t_mlp_p = 0, t_ffld = 0, t_iq = 0, t_ikv = 0, merge_hidden = 0;
    // host copies of the packing maps (must outlive the async uploads)
    std::vector<int> m_vheads, m_theads, m_tkv, m_vgate, m_tgate;
    // packed device pointers
    const int *d_vheads = nullptr, *d_theads = nullptr, *d_tkv = nullptr, *d_vgate = nullptr, *d_tgate = nullptr;
    const bf16_t* patch_w = nullptr;
    std::vector<const bf16_t*> v_qkv_w, v_qkv_b, v_proj_w, v_gu_w, v_gu_b, v_down_w;
    std::vector<const bf16_t*> t_qkv_w, t_qkv_b, t_o_w, t_gu_w, t_down_w;
    // GEMM launch timing for bench roofline numbers (HIP events on the launch stream, as in vqs_api.cpp)
    bool prof = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double prof_flops = 0.0, prof_bytes = 0.0;
    // stage taps (vqs_qwen_debug_tap): point name -> (caller buffer, capacity); a pass copies the named intermediate there
    struct Tap { void* dst; size_t cap; };
    std::unordered_map<std::string, Tap> taps;
    // 1 (default, round 5) = vqs_qwen_score / vqs_qwen_prefill re-evaluate every sample's LAST prompt position -- the one row that feeds the
    // head -- layer by layer with 16 significant bits (split-bf16 operands as stacked rows of the same GEMMs, fp32 partial sums, fp32
    // q and softmax over the layer's bf16 K / V) and take the logits from that row: the "precise tail" (tail_pass below; the analogue of the
    // CLIP-FlanT5 row's precise decoder).  The error attribution (profiles/r5_qwen_error_attribution.md) puts two thirds of the row's
    // |delta log P| into this one row's bf16 roundings.  0 = logits from the bf16 prefill's last row (rounds 2-4)
    int tail_precise = 1;
    // ---- the range-safe fp16 forms (round 6; include/vqs_qwen.h "fp16").  fp16_req: what the caller asked for (default 1 where every
    // contraction of the model can run the quad GEMM form); fp16_active: what vqs_qwen_bind_weights established -- fp16 copies of the packed
    // weights exist, every weight fits fp16, and every 16-bit activation site has a finite proven bound (compute_ranges) and with it a
    // power-of-two scale sigma = 2^-s under which fp16(T * sigma) cannot overflow.  Sites per block / layer: x0 = norm1 output, qkv = q | k | v
    // (after the rotary embedding), dattn = attention sub-layer output, x1 = norm2 output, act = gated product, dmlp = FFN sub-layer output.
    int fp16_req = 0, fp16_active = 0;
    bool fp16_eligible = false;
    // with the fp16 forms and 128-lane language-model heads the rotary embedding of q / k runs inside the q|k|v GEMM's
    // epilogue (on the unrounded projection: one rounding instead of two, one launch less per layer); 0 = the separate rope_qk_kernel (A/B, tests).
    // The tower's 80-lane heads straddle the epilogue's 128-column blocks -- a rotation partner can sit in another wave's block -- and keep the kernel.
    int rope_fused = 1;
    struct Site { float bound = 0.0f, sigma = 1.0f; };
    struct LayerSites { Site x0, qkv, dattn, x1, act, dmlp; };
    std::vector<LayerSites> v_sc, t_sc;
    Site m_x, m_mid, m_out;                     // merger: norm output, GELU(mlp.0), merged tokens
    float w_absmax = 0.0f;                      // largest |weight| among the tensors that get fp16 copies
    std::vector<const bf16_t*> v_qkv_h, v_proj_h, v_gu_h, v_down_h, t_qkv_h, t_o_h, t_gu_h, t_down_h;
    const bf16_t *m0_h = nullptr, *m2_h = nullptr;
    float *d_rb = nullptr, *d_ucol = nullptr, *d_slots = nullptr;     // bind-time scratch of compute_ranges (inside the packed buffer)
    size_t n_rb = 0, n_ucol = 0, n_slots = 0;
};

namespace {

int qfail(vqs_qwen_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define QHIP(h, expr, what)                                                                                         \
    do {                                                                                                            \
        hipError_t _e = (expr);                                                                                     \
        if (_e != hipSuccess) return qfail((h), VQS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(_e));     \
    } while (0)
#define QRUN(expr)                   \
    do {                             \
        int _r = (expr);             \
        if (_r != VQS_OK) return _r; \
    } while (0)

// Copy an intermediate to a caller buffer registered with vqs_qwen_debug_tap (in-stream, device to device): the workspace buffers
// are reused block after block, so this is how tests/test_gpu_qwen.py checks EVERY launch of a pass against the rounding-matched
// oracle on the engine's own inputs.  One empty() test when nothing is registered.
int qtap(vqs_qwen_handle* h, const char* stack, int layer, const char* what, const void* src, size_t bytes, hipStream_t st) {
    if (h->taps.empty()) return VQS_OK;
    const std::string name = layer >= 0 ? std::string(stack) + "." + std::to_string(layer) + "." + what : std::string(stack) + "." + what;
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return VQS_OK;
    if (it->second.cap < bytes) return qfail(h, VQS_ERR_WORKSPACE, "tap " + name + ": buffer too small (" + std::to_string(bytes) + " bytes needed)");
    QHIP(h, hipMemcpyAsync(it->second.dst, src, bytes, hipMemcpyDeviceToDevice, st), "tap copy");
    return VQS_OK;
}
// the same for a tensor whose rows sit at a pitch: the tap receives the dense [rows, width] tensor
int qtap2d(vqs_qwen_handle* h, const char* stack, int layer, const char* what, const bf16_t* src, int width, int rows, int pitch, hipStream_t st) {
    if (h->taps.empty()) return VQS_OK;
    if (pitch == width) return qtap(h, stack, layer, what, src, (size_t)rows * width * sizeof(bf16_t), st);
    const std::string name = layer >= 0 ? std::string(stack) + "." + std::to_string(layer) + "." + what : std::string(stack) + "." + what;
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return VQS_OK;
    const size_t bytes = (size_t)rows * width * sizeof(bf16_t);
    if (it->second.cap < bytes) return qfail(h, VQS_ERR_WORKSPACE, "tap " + name + ": buffer too small (" + std::to_string(bytes) + " bytes needed)");
    QHIP(h, hipMemcpy2DAsync(it->second.dst, (size_t)width * sizeof(bf16_t), src, (size_t)pitch * sizeof(bf16_t), (size_t)width * sizeof(bf16_t),
                             (size_t)rows, hipMemcpyDeviceToDevice, st), "tap copy");
    return VQS_OK;
}
#define QTAP(stack, layer, what, ptr, elems) QRUN(qtap(h, stack, layer, what, ptr, (size_t)(elems) * sizeof(*(ptr)), st))

int get_w(vqs_qwen_handle* h, const std::string& name, int64_t numel, const bf16_t** out) {
    auto it = h->w.find(name);
    if (it == h->w.end()) return qfail(h, VQS_ERR_MISSING_WEIGHT, "missing weight " + name);
    if (it->second.numel != numel)
        return qfail(h, VQS_ERR_MISSING_WEIGHT, "weight " + name + " has " + std::to_string(it->second.numel) + " elements, expected " + std::to_string(numel));
    *out = it->second.p;
    return VQS_OK;
}
#define QW(var, name, numel)                                      \
    const bf16_t* var = nullptr;                                  \
    QRUN(get_w(h, (name), (int64_t)(numel), &var))

// rows (or columns) of a head-major [n heads x hd] range laid into n x 128 lanes, -1 = zero
std::vector<int> head_pad_map(int n, int hd) {
    std::vector<int> m((size_t)n * HDP);
    for (int head = 0; head < n; ++head)
        for (int d = 0; d < HDP; ++d) m[(size_t)head * HDP + d] = d < hd ? head * hd + d : -1;
    return m;
}
// interleaved gate|up rows in blocks of 32 (gate block, up block), mlp padded to mlp_p: >= 0 gate row, <= -2 up row, -1 zero
std::vector<int> gate_up_map(int mlp, int mlp_p) {
    std::vector<int> m((size_t)2 * mlp_p);
    for (int r = 0; r < 2 * mlp_p; ++r) {
        const int blk = r >> 6, within = r & 63, j = blk * 32 + (within & 31);
        m[r] = j < mlp ? (within < 32 ? j : -j - 2) : -1;
    }
    return m;
}

struct Carver {
    char* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off = align_up(off + n * sizeof(T));
        return p;
    }
};

struct GCall {
    const bf16_t* A;
    const bf16_t* W;
    void* C;
    const bf16_t* bias = nullptr;
    int M = 0, N = 0, K = 0, lda = 0, ldw = 0, ldc = 0, epi = 0;
    int S = 0, H = 0, inner = 0, hd = 0, inner_kv = 0, Hkv = 0, gate_act = 0, hd_src = 0;
    bf16_t* heads[3] = {nullptr, nullptr, nullptr};
    int f16 = 0;                         // 3: the scaled fp16 family (A, W, result fp16; acc_scale / out_scale)
    float acc_scale = 1.0f, out_scale = 1.0f;
    const float *rope_cos = nullptr, *rope_sin = nullptr;     // f16 = 3, EPI_HEADS: rotary embedding in the epilogue (vqs_kernels.h)
};

int qgemm(vqs_qwen_handle* h, const GCall& g, hipStream_t st, const char* what) {
    vqs::GemmParams p;
    p.A = g.A; p.W = g.W; p.C = g.C; p.bias = g.bias; p.resid = nullptr;
    p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldw = g.ldw; p.ldc = g.ldc;
    p.S = g.S > 0 ? g.S : 1; p.H = g.H; p.inner = g.inner > 0 ? g.inner : 1;
    p.heads_out[0] = g.heads[0]; p.heads_out[1] = g.heads[1]; p.heads_out[2] = g.heads[2];
    p.hd = g.hd; p.inner_kv = g.inner_kv; p.Hkv = g.Hkv; p.gate_act = g.gate_act; p.hd_src = g.hd_src;
    p.f16 = g.f16; p.acc_scale = g.acc_scale; p.out_scale = g.out_scale;
    p.rope_cos = g.rope_cos; p.rope_sin = g.rope_sin;
    if (h->prof) {
        while (h->ev.size() < h->ev_used + 2) {
            hipEvent_t e;
            QHIP(h, hipEventCreate(&e), "hipEventCreate");
            h->ev.push_back(e);
        }
        QHIP(h, hipEventRecord(h->ev[h->ev_used], st), "hipEventRecord");
    }
    QHIP(h, vqs::launch_gemm(p, g.epi, 3, st), std::string("gemm ") + what);
    if (h->prof) {
        QHIP(h, hipEventRecord(h->ev[h->ev_used + 1], st), "hipEventRecord");
        h->ev_used += 2;
        h->prof_flops += 2.0 * (double)g.M * (double)g.N * (double)g.K;
        const double out_n = (g.epi == vqs::EPI_GATED) ? 0.5 * (double)g.N : (double)g.N;
        h->prof_bytes += 2.0 * ((double)g.M + (double)g.N) * (double)g.K + ((g.epi == vqs::EPI_F32) ? 4.0 : 2.0) * (double)g.M * out_n;
    }
    return VQS_OK;
}

struct PackPlan {
    size_t total = 0;
};

// One pass that either sizes (base == nullptr) or performs the packing.
int pack(vqs_qwen_handle* h, char* base, size_t* total, hipStream_t st) {
    const vqs_qwen_config& c = h->c;
    Carver cv{base};
    auto up_map = [&](const std::vector<int>& m, const int** dptr) -> int {
        int* d = cv.take<int>(m.size());
        if (base) {
            QHIP(h, hipMemcpyAsync(d, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice, st), "upload packing map");
            *dptr = d;
        }
        return VQS_OK;
    };
    QRUN(up_map(h->m_vheads, &h->d_vheads));
    QRUN(up_map(h->m_theads, &h->d_theads));
    QRUN(up_map(h->m_tkv, &h->d_tkv));
    QRUN(up_map(h->m_vgate, &h->d_vgate));
    QRUN(up_map(h->m_tgate, &h->d_tgate));

    const int VH = c.v_hidden, VHD = h->v_hd, VNH = c.v_heads, VPK = VNH * HDP;
    {   // patch embed: [hidden, patch_dim] -> K padded to 64
        bf16_t* d = cv.take<bf16_t>((size_t)VH * h->v_kpatch);
        if (base) {
            QW(w, "model.visual.patch_embed.proj.weight", (int64_t)VH * c.v_patch_dim);
            QHIP(h, vqs::launch_gather_rows_bf16(w, nullptr, nullptr, d, VH, c.v_patch_dim, c.v_patch_dim, h->v_kpatch, st), "pack patch embed");
            h->patch_w = d;
        }
    }
    if (base) {
        h->v_qkv_w.assign(c.v_depth, nullptr); h->v_qkv_b.assign(c.v_depth, nullptr); h->v_proj_w.assign(c.v_depth, nullptr);
        h->v_gu_w.assign(c.v_depth, nullptr); h->v_gu_b.assign(c.v_depth, nullptr); h->v_down_w.assign(c.v_depth, nullptr);
        h->t_qkv_w.assign(c.t_layers, nullptr); h->t_qkv_b.assign(c.t_layers, nullptr); h->t_o_w.assign(c.t_layers, nullptr);
        h->t_gu_w.assign(c.t_layers, nullptr); h->t_down_w.assign(c.t_layers, nullptr);
        h->v_qkv_h.assign(c.v_depth, nullptr); h->v_proj_h.assign(c.v_depth, nullptr); h->v_gu_h.assign(c.v_depth, nullptr); h->v_down_h.assign(c.v_depth, nullptr);
        h->t_qkv_h.assign(c.t_layers, nullptr); h->t_o_h.assign(c.t_layers, nullptr); h->t_gu_h.assign(c.t_layers, nullptr); h->t_down_h.assign(c.t_layers, nullptr);
    }
    for (int i = 0; i < c.v_depth; ++i) {
        const std::string p = "model.visual.blocks." + std::to_string(i) + ".";
        const bool cp = h->v_compact;            // compact heads: the checkpoint's qkv / proj tensors are used where they lie
        bf16_t* qkv = cp ? nullptr : cv.take<bf16_t>((size_t)3 * VPK * VH);
        bf16_t* qkvb = cp ? nullptr : cv.take<bf16_t>((size_t)3 * VPK);
        bf16_t* proj = cp ? nullptr : cv.take<bf16_t>((size_t)VH * VPK);
        bf16_t* gu = cv.take<bf16_t>((size_t)2 * h->v_mlp_p * VH);
        bf16_t* gub = cv.take<bf16_t>((size_t)2 * h->v_mlp_p);
        bf16_t* down = cv.take<bf16_t>((size_t)VH * h->v_ffld);
        // fp16 copies of the four GEMM weights (option fp16): the packed layouts above, cast element for element
        const size_t n_qkv = cp ? (size_t)3 * VH * VH : (size_t)3 * VPK * VH, n_proj = cp ? (size_t)VH * VH : (size_t)VH * VPK;
        const size_t n_gu = (size_t)2 * h->v_mlp_p * VH, n_down = (size_t)VH * h->v_ffld;
        bf16_t *qkv_h = nullptr, *proj_h = nullptr, *gu_h = nullptr, *down_h = nullptr;
        if (h->fp16_req) {
            qkv_h = cv.take<bf16_t>(n_qkv); proj_h = cv.take<bf16_t>(n_proj); gu_h = cv.take<bf16_t>(n_gu); down_h = cv.take<bf16_t>(n_down);
        }
        if (!base) continue;
        QW(wqkv, p + "attn.qkv.weight", (int64_t)3 * VH * VH);
        QW(bqkv, p + "attn.qkv.bias", 3 * VH);
        QW(wproj, p + "attn.proj.weight", (int64_t)VH * VH);
        QW(wg, p + "mlp.gate_proj.weight", (int64_t)c.v_mlp * VH);
        QW(bg, p + "mlp.gate_proj.bias", c.v_mlp);
        QW(wu, p + "mlp.up_proj.weight", (int64_t)c.v_mlp * VH);
        QW(bu, p + "mlp.up_proj.bias", c.v_mlp);
        QW(wd, p + "mlp.down_proj.weight", (int64_t)VH * c.v_mlp);
        for (int which = 0; which < 3 && !cp; ++which) {   // q | k | v row ranges, every head padded to 128 rows
            QHIP(h, vqs::launch_gather_rows_bf16(wqkv + (size_t)which * VH * VH, nullptr, h->d_vheads, qkv + (size_t)which * VPK * VH,
                                                  VPK, VH, VH, VH, st), "pack vision qkv");
            QHIP(h, vqs::launch_gather_rows_bf16(bqkv + (size_t)which * VH, nullptr, h->d_vheads, qkvb + (size_t)which * VPK, VPK, 1, 1, 1, st),
                 "pack vision qkv bias");
        }
        if (!cp) QHIP(h, vqs::launch_gather_cols_bf16(wproj, h->d_vheads, proj, VH, VH, VPK, st), "pack vision proj");
        if (cp) { qkv = const_cast<bf16_t*>(wqkv); qkvb = const_cast<bf16_t*>(bqkv); proj = const_cast<bf16_t*>(wproj); }
        QHIP(h, vqs::launch_gather_rows_bf16(wg, wu, h->d_vgate, gu, 2 * h->v_mlp_p, VH, VH, VH, st), "pack vision gate|up");
        QHIP(h, vqs::launch_gather_rows_bf16(bg, bu, h->d_vgate, gub, 2 * h->v_mlp_p, 1, 1, 1, st), "pack vision gate|up bias");
        QHIP(h, vqs::launch_gather_rows_bf16(wd, nullptr, nullptr, down, VH, c.v_mlp, c.v_mlp, h->v_ffld, st), "pack vision down");
        (void)VHD;
        h->v_qkv_w[i] = qkv; h->v_qkv_b[i] = qkvb; h->v_proj_w[i] = proj; h->v_gu_w[i] = gu; h->v_gu_b[i] = gub; h->v_down_w[i] = down;
        if (h->fp16_req) {
            QHIP(h, vqs::launch_cast16(qkv, qkv_h, n_qkv, true, st), "fp16 vision qkv");
            QHIP(h, vqs::launch_cast16(proj, proj_h, n_proj, true, st), "fp16 vision proj");
            QHIP(h, vqs::launch_cast16(gu, gu_h, n_gu, true, st), "fp16 vision gate|up");
            QHIP(h, vqs::launch_cast16(down, down_h, n_down, true, st), "fp16 vision down");
            h->v_qkv_h[i] = qkv_h; h->v_proj_h[i] = proj_h; h->v_gu_h[i] = gu_h; h->v_down_h[i] = down_h;
        }
    }
    {   // merger weights: used where they lie in bf16; fp16 copies for the fp16 forms
        const size_t n0 = (size_t)h->merge_hidden * h->merge_hidden, n2 = (size_t)c.v_out_hidden * h->merge_hidden;
        bf16_t *m0 = nullptr, *m2 = nullptr;
        if (h->fp16_req) { m0 = cv.take<bf16_t>(n0); m2 = cv.take<bf16_t>(n2); }
        if (base && h->fp16_req) {
            QW(m0w, "model.visual.merger.mlp.0.weight", (int64_t)n0);
            QW(m2w, "model.visual.merger.mlp.2.weight", (int64_t)n2);
            QHIP(h, vqs::launch_cast16(m0w, m0, n0, true, st), "fp16 merger mlp.0");
            QHIP(h, vqs::launch_cast16(m2w, m2, n2, true, st), "fp16 merger mlp.2");
            h->m0_h = m0; h->m2_h = m2;
        }
    }
    const int TH = c.t_hidden, IQ = h->t_iq, IKV = h->t_ikv, QN = IQ + 2 * IKV, thd = h->t_hd;
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.language_model.layers." + std::to_string(i) + ".";
        bf16_t* qkv = cv.take<bf16_t>((size_t)QN * TH);
        bf16_t* qkvb = cv.take<bf16_t>((size_t)QN);
        bf16_t* ow = cv.take<bf16_t>((size_t)TH * IQ);
        bf16_t* gu = cv.take<bf16_t>((size_t)2 * h->t_mlp_p * h->t_xld);
        bf16_t* down = cv.take<bf16_t>((size_t)TH * h->t_ffld);
        const size_t n_qkv = (size_t)QN * TH, n_o = (size_t)TH * IQ, n_gu = (size_t)2 * h->t_mlp_p * h->t_xld, n_down = (size_t)TH * h->t_ffld;
        bf16_t *qkv_h = nullptr, *o_h = nullptr, *gu_h = nullptr, *down_h = nullptr;
        if (h->fp16_req) {
            qkv_h = cv.take<bf16_t>(n_qkv); o_h = cv.take<bf16_t>(n_o); gu_h = cv.take<bf16_t>(n_gu); down_h = cv.take<bf16_t>(n_down);
        }
        if (!base) continue;
        const int64_t qn = (int64_t)c.t_heads * thd, kn = (int64_t)c.t_kv_heads * thd;
        QW(wq, p + "self_attn.q_proj.weight", qn * TH);
        QW(bq, p + "self_attn.q_proj.bias", qn);
        QW(wk, p + "self_attn.k_proj.weight", kn * TH);
        QW(bk, p + "self_attn.k_proj.bias", kn);
        QW(wv, p + "self_attn.v_proj.weight", kn * TH);
        QW(bv, p + "self_attn.v_proj.bias", kn);
        QW(wo, p + "self_attn.o_proj.weight", (int64_t)TH * qn);
        QW(wg, p + "mlp.gate_proj.weight", (int64_t)c.t_mlp * TH);
        QW(wu, p + "mlp.up_proj.weight", (int64_t)c.t_mlp * TH);
        QW(wd, p + "mlp.down_proj.weight", (int64_t)TH * c.t_mlp);
        QHIP(h, vqs::launch_gather_rows_bf16(wq, nullptr, h->d_theads, qkv, IQ, TH, TH, TH, st), "pack q");
        QHIP(h, vqs::launch_gather_rows_bf16(wk, nullptr, h->d_tkv, qkv + (size_t)IQ * TH, IKV, TH, TH, TH, st), "pack k");
        QHIP(h, vqs::launch_gather_rows_bf16(wv, nullptr, h->d_tkv, qkv + (size_t)(IQ + IKV) * TH, IKV, TH, TH, TH, st), "pack v");
        QHIP(h, vqs::launch_gather_rows_bf16(bq, nullptr, h->d_theads, qkvb, IQ, 1, 1, 1, st), "pack q bias");
        QHIP(h, vqs::launch_gather_rows_bf16(bk, nullptr, h->d_tkv, qkvb + IQ, IKV, 1, 1, 1, st), "pack k bias");
        QHIP(h, vqs::launch_gather_rows_bf16(bv, nullptr, h->d_tkv, qkvb + IQ + IKV, IKV, 1, 1, 1, st), "pack v bias");
        QHIP(h, vqs::launch_gather_cols_bf16(wo, h->d_theads, ow, TH, (int)qn, IQ, st), "pack o");
        QHIP(h, vqs::launch_gather_rows_bf16(wg, wu, h->d_tgate, gu, 2 * h->t_mlp_p, TH, TH, h->t_xld, st), "pack gate|up");
        QHIP(h, vqs::launch_gather_rows_bf16(wd, nullptr, nullptr, down, TH, c.t_mlp, c.t_mlp, h->t_ffld, st), "pack down");
        h->t_qkv_w[i] = qkv; h->t_qkv_b[i] = qkvb; h->t_o_w[i] = ow; h->t_gu_w[i] = gu; h->t_down_w[i] = down;
        if (h->fp16_req) {
            QHIP(h, vqs::launch_cast16(qkv, qkv_h, n_qkv, true, st), "fp16 qkv");
            QHIP(h, vqs::launch_cast16(ow, o_h, n_o, true, st), "fp16 o");
            QHIP(h, vqs::launch_cast16(gu, gu_h, n_gu, true, st), "fp16 gate|up");
            QHIP(h, vqs::launch_cast16(down, down_h, n_down, true, st), "fp16 down");
            h->t_qkv_h[i] = qkv_h; h->t_o_h[i] = o_h; h->t_gu_h[i] = gu_h; h->t_down_h[i] = down_h;
        }
    }
    if (h->fp16_req) {   // scratch of compute_ranges: row bounds of the widest packed weight, column bounds of the widest contraction, the sites' maxima
        const size_t QNp = (size_t)h->t_iq + 2 * h->t_ikv;
        h->n_rb = std::max({(size_t)2 * h->t_mlp_p, (size_t)2 * h->v_mlp_p, QNp, (size_t)3 * c.v_heads * HDP, (size_t)h->merge_hidden, (size_t)c.v_out_hidden});
        h->n_ucol = std::max({(size_t)h->t_ffld, (size_t)h->v_ffld, (size_t)h->merge_hidden});
        h->n_slots = (size_t)8 * (c.v_depth + c.t_layers) + 8;
        float* rb = cv.take<float>(h->n_rb);
        float* uc = cv.take<float>(h->n_ucol);
        float* sl = cv.take<float>(h->n_slots);
        if (base) { h->d_rb = rb; h->d_ucol = uc; h->d_slots = sl; }
    }
    *total = align_up(cv.off);
    return VQS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Bind-time range proof of the fp16 forms (kernels + the inequalities: qwen_decode.hip "Bind-time range proof").  For every 16-bit
// activation site of the tower, the merger and the language model a bound of |T| is computed from the weights alone, its maximum lands
// in a slot, one copy + stream synchronisation brings the slots to the host (the ONE synchronisation of vqs_qwen_bind_weights, only with
// option fp16), and the host turns a bound B into sigma = 2^-s, the largest power of two <= 1 with B * sigma <= FP16_HEAD (half of the
// fp16 maximum: room for the fp32 accumulation order and the final rounding).  A non-finite bound (NaN / inf weight) or a weight beyond
// the fp16 range switches the fp16 forms off for this handle (fp16_active = 0, reason in vqs_qwen_last_error): the bf16 forms run.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr float FP16_HEAD = 32768.0f;
bool site_from_bound(vqs_qwen_handle::Site& s, float B) {
    s.bound = B;
    s.sigma = 1.0f;
    if (!(B >= 0.0f) || std::isinf(B)) return false;
    int e = 0;
    while (B * std::ldexp(1.0f, -e) > FP16_HEAD && e < 120) ++e;
    s.sigma = std::ldexp(1.0f, -e);
    return e < 120;
}

int compute_ranges(vqs_qwen_handle* h, hipStream_t st) {
    const vqs_qwen_config& c = h->c;
    const int VH = c.v_hidden, VNH = c.v_heads, VPK = VNH * HDP, TH = c.t_hidden, IQ = h->t_iq, IKV = h->t_ikv, QN = IQ + 2 * IKV;
    const bool cp = h->v_compact;
    const int VAK = cp ? VH : VPK, VQN = cp ? 3 * VH : 3 * VPK;
    float* sl = h->d_slots;
    QHIP(h, hipMemsetAsync(sl, 0, h->n_slots * sizeof(float), st), "clear range slots");
    // slot layout: block / layer l at 8 l: [x0, qkv (before the rotary embedding), dattn, x1, gate|up rows (unused), act, dmlp, -]; then the
    // merger's [x, mid, out] and the largest |weight|
    auto layer_chain = [&](float* s8, const bf16_t* g1, const bf16_t* g2, int D, const bf16_t* wqkv, int nq, const bf16_t* bqkv, const bf16_t* wo,
                           int ko, const bf16_t* bo, const bf16_t* wgu, long long ld_gu, int mlp_p, const bf16_t* bgu, const bf16_t* wd, int ffld,
                           const bf16_t* bd) -> int {
        const float R = std::sqrt((float)D);
        QHIP(h, vqs::launch_absmax_bf16(g1, (size_t)D, R, s8 + 0, st), "range x0");
        QHIP(h, vqs::launch_rowbound(wqkv, D, nq, D, 0, g1, D, R, nullptr, nullptr, bqkv, nullptr, s8 + 1, st), "range qkv");
        QHIP(h, vqs::launch_rowbound(wo, ko, D, ko, 1, nullptr, 0, 0.0f, nullptr, s8 + 1, bo, nullptr, s8 + 2, st), "range attention output");
        QHIP(h, vqs::launch_absmax_bf16(g2, (size_t)D, R, s8 + 3, st), "range x1");
        QHIP(h, vqs::launch_rowbound(wgu, ld_gu, 2 * mlp_p, D, 0, g2, D, R, nullptr, nullptr, bgu, h->d_rb, s8 + 4, st), "range gate|up");
        QHIP(h, vqs::launch_gate_pair_bound(h->d_rb, mlp_p, ffld, h->d_ucol, s8 + 5, st), "range gated product");
        QHIP(h, vqs::launch_rowbound(wd, ffld, D, ffld, 1, nullptr, 0, 0.0f, h->d_ucol, nullptr, bd, nullptr, s8 + 6, st), "range FFN output");
        return VQS_OK;
    };
    float* wmax = sl + (size_t)8 * (c.v_depth + c.t_layers) + 3;
    auto wrange = [&](const bf16_t* w, size_t n) -> int {
        QHIP(h, vqs::launch_absmax_bf16(w, n, 1.0f, wmax, st), "weight range");
        return VQS_OK;
    };
    for (int i = 0; i < c.v_depth; ++i) {
        const std::string p = "model.visual.blocks." + std::to_string(i) + ".";
        QW(n1, p + "norm1.weight", VH);
        QW(n2, p + "norm2.weight", VH);
        QW(pb, p + "attn.proj.bias", VH);
        QW(db, p + "mlp.down_proj.bias", VH);
        QRUN(layer_chain(sl + (size_t)8 * i, n1, n2, VH, h->v_qkv_w[i], VQN, h->v_qkv_b[i], h->v_proj_w[i], VAK, pb, h->v_gu_w[i], VH, h->v_mlp_p,
                         h->v_gu_b[i], h->v_down_w[i], h->v_ffld, db));
        QRUN(wrange(h->v_qkv_w[i], (size_t)VQN * VH));
        QRUN(wrange(h->v_proj_w[i], (size_t)VH * VAK));
        QRUN(wrange(h->v_gu_w[i], (size_t)2 * h->v_mlp_p * VH));
        QRUN(wrange(h->v_down_w[i], (size_t)VH * h->v_ffld));
    }
    {   // merger: RMSNorm over VH, merge_unit rows concatenated (||x^cat||_2 <= sqrt(merge_unit VH)), Linear-GELU-Linear
        float* sm = sl + (size_t)8 * (c.v_depth + c.t_layers);
        QW(lnq, "model.visual.merger.ln_q.weight", VH);
        QW(m0w, "model.visual.merger.mlp.0.weight", (int64_t)h->merge_hidden * h->merge_hidden);
        QW(m0b, "model.visual.merger.mlp.0.bias", h->merge_hidden);
        QW(m2w, "model.visual.merger.mlp.2.weight", (int64_t)c.v_out_hidden * h->merge_hidden);
        QW(m2b, "model.visual.merger.mlp.2.bias", c.v_out_hidden);
        QHIP(h, vqs::launch_absmax_bf16(lnq, (size_t)VH, std::sqrt((float)VH), sm + 0, st), "range merger norm");
        QHIP(h, vqs::launch_rowbound(m0w, h->merge_hidden, h->merge_hidden, h->merge_hidden, 0, lnq, VH, std::sqrt((float)h->merge_hidden), nullptr, nullptr,
                                     m0b, h->d_ucol, sm + 1, st), "range merger mlp.0");
        QHIP(h, vqs::launch_rowbound(m2w, h->merge_hidden, c.v_out_hidden, h->merge_hidden, 1, nullptr, 0, 0.0f, h->d_ucol, nullptr, m2b, nullptr, sm + 2,
                                     st), "range merger mlp.2");
        QRUN(wrange(m0w, (size_t)h->merge_hidden * h->merge_hidden));
        QRUN(wrange(m2w, (size_t)c.v_out_hidden * h->merge_hidden));
    }
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.language_model.layers." + std::to_string(i) + ".";
        QW(ln1, p + "input_layernorm.weight", TH);
        QW(ln2, p + "post_attention_layernorm.weight", TH);
        QRUN(layer_chain(sl + (size_t)8 * (c.v_depth + i), ln1, ln2, TH, h->t_qkv_w[i], QN, h->t_qkv_b[i], h->t_o_w[i], IQ, nullptr, h->t_gu_w[i],
                         h->t_xld, h->t_mlp_p, nullptr, h->t_down_w[i], h->t_ffld, nullptr));
        QRUN(wrange(h->t_qkv_w[i], (size_t)QN * TH));
        QRUN(wrange(h->t_o_w[i], (size_t)TH * IQ));
        QRUN(wrange(h->t_gu_w[i], (size_t)2 * h->t_mlp_p * h->t_xld));
        QRUN(wrange(h->t_down_w[i], (size_t)TH * h->t_ffld));
    }
    std::vector<float> host(h->n_slots);
    QHIP(h, hipMemcpyAsync(host.data(), sl, h->n_slots * sizeof(float), hipMemcpyDeviceToHost, st), "read range slots");
    QHIP(h, hipStreamSynchronize(st), "range proof synchronise");
    bool ok = true;
    auto layer_sites = [&](vqs_qwen_handle::LayerSites& L, const float* s8) {
        ok &= site_from_bound(L.x0, s8[0]);
        ok &= site_from_bound(L.qkv, 1.41421357f * s8[1]);           // after the rotary embedding: |x'| <= sqrt(2) max|x| (v, unrotated, is below it)
        ok &= site_from_bound(L.dattn, s8[2]);
        ok &= site_from_bound(L.x1, s8[3]);
        ok &= site_from_bound(L.act, s8[5]);
        ok &= site_from_bound(L.dmlp, s8[6]);
    };
    h->v_sc.assign(c.v_depth, vqs_qwen_handle::LayerSites{});
    h->t_sc.assign(c.t_layers, vqs_qwen_handle::LayerSites{});
    for (int i = 0; i < c.v_depth; ++i) layer_sites(h->v_sc[i], host.data() + (size_t)8 * i);
    for (int i = 0; i < c.t_layers; ++i) layer_sites(h->t_sc[i], host.data() + (size_t)8 * (c.v_depth + i));
    const float* sm = host.data() + (size_t)8 * (c.v_depth + c.t_layers);
    ok &= site_from_bound(h->m_x, sm[0]);
    ok &= site_from_bound(h->m_mid, sm[1]);
    ok &= site_from_bound(h->m_out, sm[2]);
    h->w_absmax = sm[3];
    // d_attn's bound used the PRE-rotation q|k|v maximum as the value bound: v is not rotated, so that is the right constant
    if (!ok) {
        h->fp16_active = 0;
        h->err = "fp16 forms off: a range bound is not finite (NaN / inf in the weights?); the bf16 forms run";
    } else if (!(h->w_absmax < 65504.0f)) {
        h->fp16_active = 0;
        h->err = "fp16 forms off: a weight of magnitude " + std::to_string(h->w_absmax) + " does not fit IEEE fp16; the bf16 forms run";
    } else {
        h->fp16_active = 1;
    }
    return VQS_OK;
}

struct VisWs {
    bf16_t *patches, *xn, *delta, *delta2, *q, *k, *v, *attn, *ff, *mid, *merged_w, *xc, *dc;
    float *pre, *hidden;
    size_t total;
};
// N real patches, Np >= N rows of the windowed layout (partial windows padded to win_len slots)
VisWs carve_vision(const vqs_qwen_handle* h, char* base, int N, int Np) {
    const vqs_qwen_config& c = h->c;
    Carver cv{base};
    VisWs w{};
    const size_t n = (size_t)N, np = (size_t)Np, VPK = (size_t)c.v_heads * HDP;
    w.patches = cv.take<bf16_t>(n * h->v_kpatch);
    w.pre = cv.take<float>(n * c.v_hidden);
    w.hidden = cv.take<float>(np * c.v_hidden);
    w.xn = cv.take<bf16_t>(np * c.v_hidden);
    w.delta = cv.take<bf16_t>(np * c.v_hidden);
    w.delta2 = cv.take<bf16_t>(np * c.v_hidden);   // second pending delta (deferred stream store, see elementwise.hip)
    w.q = cv.take<bf16_t>(np * VPK);
    w.k = cv.take<bf16_t>(np * VPK);
    w.v = cv.take<bf16_t>(np * VPK);
    w.attn = cv.take<bf16_t>(np * VPK);
    w.ff = cv.take<bf16_t>(np * h->v_ffld);
    w.xc = cv.take<bf16_t>(n * c.v_hidden);        // frame-compact copies around the full-attention blocks
    w.dc = cv.take<bf16_t>(n * c.v_hidden);
    w.mid = cv.take<bf16_t>(np / c.v_merge_unit * h->merge_hidden);
    w.merged_w = cv.take<bf16_t>(np / c.v_merge_unit * c.v_out_hidden);
    w.total = align_up(cv.off);
    return w;
}

// K slices of a decode-step linear (decode_linear below): the fewest slices (a divisor of K / 64, <= 16) that put >= 192
// (slice, 128-column) items on the chip -- a function of the weight's shape only.  carve_decode sizes the partial buffer with it.
int decode_slices(int N, int K) {
    const int nblk = (N + 127) / 128, nsl = K / 64;
    if (nblk >= 192 || (K % 64) != 0) return 1;
    int best = 1;
    for (int s2 = 2; s2 <= 16; ++s2)
        if (nsl % s2 == 0) {
            best = s2;
            if (nblk * s2 >= 192) break;
        }
    return best;
}

static constexpr int TAIL_ROWS = 64;      // samples per chunk of the precise tail: 2 x 64 stacked rows = the stream GEMM form's 128-row limit
struct TxtWs {
    bf16_t *xn, *delta, *delta2, *q, *k, *v, *attn, *ff, *last;
    float* hidden;
    // precise tail (vqs_qwen_handle::tail_precise): per-sample state for all B samples, chunk-sized operands
    float *t_h, *t_delta;         // fp32 stream row and pending sub-layer output of every sample's last position [B, hidden]
    float *t_qkv, *t_q;           // fp32 q|k|v row and rotated q of a chunk
    bf16_t *t_xn, *t_attn, *t_ff; // split-bf16 operands of a chunk: planes [2][rows][width]
    float* t_part;                // fp32 partials of a chunk's stacked launch
    size_t t_part_bytes;
    size_t total;
};
TxtWs carve_text(const vqs_qwen_handle* h, char* base, int B, int L) {
    const vqs_qwen_config& c = h->c;
    Carver cv{base};
    TxtWs w{};
    const size_t M = (size_t)B * L;
    w.hidden = cv.take<float>(M * c.t_hidden);
    w.xn = cv.take<bf16_t>(M * h->t_xld);
    w.delta = cv.take<bf16_t>(M * c.t_hidden);
    w.delta2 = cv.take<bf16_t>(M * c.t_hidden);
    w.q = cv.take<bf16_t>(M * h->t_iq);
    w.k = cv.take<bf16_t>(M * h->t_ikv);
    w.v = cv.take<bf16_t>(M * h->t_ikv);
    w.attn = cv.take<bf16_t>(M * h->t_iq);
    w.ff = cv.take<bf16_t>(M * h->t_ffld);
    w.last = cv.take<bf16_t>((size_t)B * c.t_hidden);
    {   // precise tail: always laid out (a few hundred MB at 7B), so that the option can be toggled on a bound handle
        const size_t Bc = (size_t)std::min(B, TAIL_ROWS), QN = (size_t)h->t_iq + 2 * h->t_ikv;
        w.t_h = cv.take<float>((size_t)B * c.t_hidden);
        w.t_delta = cv.take<float>((size_t)B * c.t_hidden);
        w.t_qkv = cv.take<float>(Bc * QN);
        w.t_q = cv.take<float>(Bc * h->t_iq);
        w.t_xn = cv.take<bf16_t>(2 * Bc * c.t_hidden);
        w.t_attn = cv.take<bf16_t>(2 * Bc * h->t_iq);
        w.t_ff = cv.take<bf16_t>(2 * Bc * h->t_ffld);
        size_t widest = (size_t)decode_slices((int)QN, c.t_hidden) * QN;
        widest = std::max(widest, (size_t)decode_slices(c.t_hidden, h->t_iq) * c.t_hidden);
        widest = std::max(widest, (size_t)decode_slices(2 * h->t_mlp_p, c.t_hidden) * 2 * h->t_mlp_p);
        widest = std::max(widest, (size_t)decode_slices(c.t_hidden, h->t_ffld) * c.t_hidden);
        widest = std::max(widest, (size_t)decode_slices(c.t_vocab, c.t_hidden) * c.t_vocab);
        w.t_part_bytes = 2 * Bc * widest * sizeof(float);
        w.t_part = cv.take<float>(2 * Bc * widest);
    }
    w.total = align_up(cv.off);
    return w;
}

}  // namespace

extern "C" {

int vqs_qwen_create(const vqs_qwen_config* cfg, vqs_qwen_handle** out) {
    if (!cfg || !out) return VQS_ERR_INVALID;
    const vqs_qwen_config& c = *cfg;
    if (c.v_depth <= 0 || c.v_depth > 32 || c.v_heads <= 0 || c.v_hidden % c.v_heads || c.v_hidden % 64 || c.v_merge_unit <= 0 ||
        c.t_layers <= 0 || c.t_heads <= 0 || c.t_kv_heads <= 0 || c.t_heads % c.t_kv_heads || c.t_hidden % c.t_heads || c.t_hidden % 64 ||
        c.v_out_hidden != c.t_hidden || c.t_vocab % 8)
        return VQS_ERR_INVALID;
    vqs_qwen_handle* h = new vqs_qwen_handle();
    h->c = c;
    h->v_hd = c.v_hidden / c.v_heads;
    h->t_hd = c.t_hidden / c.t_heads;
    if (h->v_hd > HDP || h->t_hd > HDP || (h->v_hd & 3) || (h->t_hd & 3)) {
        delete h;
        return VQS_ERR_INVALID;
    }
    h->v_kpatch = round_up(c.v_patch_dim, 64);
    h->v_mlp_p = round_up(c.v_mlp, 32);
    h->v_ffld = round_up(h->v_mlp_p, 64);
    h->t_mlp_p = round_up(c.t_mlp, 32);
    h->t_ffld = round_up(h->t_mlp_p, 64);
    h->t_iq = c.t_heads * HDP;
    h->t_ikv = c.t_kv_heads * HDP;
    h->merge_hidden = c.v_hidden * c.v_merge_unit;
    h->t_xld = c.t_hidden;
    if (c.t_hidden > 2048) {
        int p2 = 4096;
        while (p2 < c.t_hidden) p2 *= 2;
        h->t_xld = p2;
    }
    h->v_compact = h->v_hd < HDP && (h->v_hd % 8) == 0 && (c.v_hidden % 128) == 0;
    h->m_vheads = head_pad_map(c.v_heads, h->v_hd);
    h->m_theads = head_pad_map(c.t_heads, h->t_hd);
    h->m_tkv = head_pad_map(c.t_kv_heads, h->t_hd);
    h->m_vgate = gate_up_map(c.v_mlp, h->v_mlp_p);
    h->m_tgate = gate_up_map(c.t_mlp, h->t_mlp_p);
    // fp16 forms: every 16-bit-result GEMM must resolve to the quad form (K >= 128; the scaled family exists only there) and both rotary
    // embeddings to the 16-byte kernel
    {
        const int VAK = h->v_compact ? c.v_hidden : c.v_heads * HDP;
        h->fp16_eligible = c.v_hidden >= 128 && VAK >= 128 && h->v_ffld >= 128 && h->merge_hidden >= 128 && c.t_hidden >= 128 && h->t_ffld >= 128 &&
                           ((h->v_hd / 2) % 8) == 0 && ((h->t_hd / 2) % 8) == 0;
        h->fp16_req = h->fp16_eligible ? 1 : 0;
    }
    *out = h;
    return VQS_OK;
}

void vqs_qwen_destroy(vqs_qwen_handle* h) {
    if (!h) return;
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
}

int vqs_qwen_profile_enable(vqs_qwen_handle* h, int32_t on) {
    if (!h) return VQS_ERR_INVALID;
    h->prof = on != 0;
    return VQS_OK;
}

int vqs_qwen_profile_read(vqs_qwen_handle* h, double* gemm_ms, double* gemm_flops, double* gemm_bytes, int32_t reset) {
    if (!h) return VQS_ERR_INVALID;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        QHIP(h, hipEventSynchronize(h->ev[i + 1]), "hipEventSynchronize");
        float t = 0.f;
        QHIP(h, hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]), "hipEventElapsedTime");
        ms += t;
    }
    if (gemm_ms) *gemm_ms = ms;
    if (gemm_flops) *gemm_flops = h->prof_flops;
    if (gemm_bytes) *gemm_bytes = h->prof_bytes;
    const int n = (int)(h->ev_used / 2);
    if (reset) { h->ev_used = 0; h->prof_flops = 0.0; h->prof_bytes = 0.0; }
    return n;
}

int vqs_qwen_debug_option(vqs_qwen_handle* h, const char* name, int64_t value) {
    if (!h || !name) return VQS_ERR_INVALID;
    if (std::string(name) == "x_pitch") {        // before vqs_qwen_bind_weights: the packed gate|up rows carry the pitch
        if (h->bound) return qfail(h, VQS_ERR_STATE, "x_pitch must be set before the weights are bound");
        if (value < h->c.t_hidden || (value % 64) != 0 || value > 8 * (int64_t)h->c.t_hidden + 4096)
            return qfail(h, VQS_ERR_INVALID, "x_pitch: a multiple of 64 elements, >= hidden");
        h->t_xld = (int)value;
        return VQS_OK;
    }
    if (std::string(name) == "fp16") {           // see include/vqs_qwen.h: 1 needs an eligible model and must be asked for BEFORE the weights are bound
        if (value != 0 && value != 1) return qfail(h, VQS_ERR_INVALID, "fp16: 0 or 1");
        if (value == 1 && !h->fp16_eligible) return qfail(h, VQS_ERR_INVALID, "fp16: this configuration has contractions narrower than 128 or rotary halves that are not multiples of 8");
        if (!h->bound) { h->fp16_req = (int)value; return VQS_OK; }
        if (value == 0) { h->fp16_active = 0; return VQS_OK; }              // the bf16 forms are always there
        if (!h->fp16_req) return qfail(h, VQS_ERR_STATE, "fp16: the weights were bound without fp16 copies; set the option before vqs_qwen_bind_weights");
        if (h->t_sc.empty()) return qfail(h, VQS_ERR_STATE, "fp16: the range proof did not complete at bind time");
        if (!(h->w_absmax < 65504.0f)) return qfail(h, VQS_ERR_STATE, "fp16: a weight does not fit IEEE fp16");
        h->fp16_active = 1;
        return VQS_OK;
    }
    if (std::string(name) == "rope_fused") {     // see vqs_qwen_handle::rope_fused
        if (value != 0 && value != 1) return qfail(h, VQS_ERR_INVALID, "rope_fused: 0 or 1");
        h->rope_fused = (int)value;
        return VQS_OK;
    }
    if (std::string(name) == "tail_precise") {   // see vqs_qwen_handle::tail_precise
        if (value != 0 && value != 1) return qfail(h, VQS_ERR_INVALID, "tail_precise: 0 or 1");
        h->tail_precise = (int)value;
        return VQS_OK;
    }
    return qfail(h, VQS_ERR_INVALID, std::string("unknown option ") + name);
}

int vqs_qwen_get_option(const vqs_qwen_handle* h, const char* name, int64_t* value) {
    if (!h || !name || !value) return VQS_ERR_INVALID;
    const std::string n(name);
    if (n == "fp16") *value = h->fp16_active;                  // what runs (0 before the weights are bound)
    else if (n == "fp16_requested") *value = h->fp16_req;
    else if (n == "fp16_eligible") *value = h->fp16_eligible ? 1 : 0;
    else if (n == "tail_precise") *value = h->tail_precise;
    else if (n == "rope_fused") *value = (h->fp16_active && h->rope_fused && h->t_hd == HDP) ? 1 : 0;      // what runs
    else if (n == "x_pitch") *value = h->t_xld;
    else return VQS_ERR_INVALID;
    return VQS_OK;
}

int vqs_qwen_range_report(const vqs_qwen_handle* h, float* bounds, float* sigmas, int32_t cap) {
    if (!h) return VQS_ERR_INVALID;
    if (h->t_sc.empty()) return 0;                              // no proof was made (option fp16 off at bind time)
    std::vector<float> b, s;
    auto push = [&](const vqs_qwen_handle::Site& x) { b.push_back(x.bound); s.push_back(x.sigma); };
    auto layer = [&](const vqs_qwen_handle::LayerSites& L) { push(L.x0); push(L.qkv); push(L.dattn); push(L.x1); push(L.act); push(L.dmlp); };
    for (const auto& L : h->v_sc) layer(L);
    push(h->m_x); push(h->m_mid); push(h->m_out);
    for (const auto& L : h->t_sc) layer(L);
    const int n = (int)b.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (bounds) bounds[i] = b[i];
        if (sigmas) sigmas[i] = s[i];
    }
    return n;
}

int vqs_qwen_debug_tap(vqs_qwen_handle* h, const char* name, void* d_dst, size_t bytes) {
    if (!h) return VQS_ERR_INVALID;
    if (!name) { h->taps.clear(); return VQS_OK; }
    if (!d_dst || bytes == 0) { h->taps.erase(name); return VQS_OK; }
    h->taps[name] = vqs_qwen_handle::Tap{d_dst, bytes};
    return VQS_OK;
}

const char* vqs_qwen_last_error(const vqs_qwen_handle* h) { return h ? h->err.c_str() : "null handle"; }

size_t vqs_qwen_packed_bytes(const vqs_qwen_handle* h) {
    if (!h) return 0;
    size_t total = 0;
    (void)pack(const_cast<vqs_qwen_handle*>(h), nullptr, &total, nullptr);
    return total;
}

int vqs_qwen_bind_weights(vqs_qwen_handle* h, const vqs_weight_desc* descs, int32_t n, void* d_packed, size_t packed_bytes,
                          void* stream) {
    if (!h || !descs || n <= 0 || !d_packed) return VQS_ERR_INVALID;
    h->w.clear();
    for (int i = 0; i < n; ++i) {
        if (!descs[i].name || !descs[i].d_data) return qfail(h, VQS_ERR_INVALID, "bind: null name or pointer");
        h->w[descs[i].name] = WEnt{(const bf16_t*)descs[i].d_data, descs[i].numel};
    }
    size_t need = 0;
    QRUN(pack(h, nullptr, &need, nullptr));
    if (packed_bytes < need) return qfail(h, VQS_ERR_WORKSPACE, "bind: packed buffer too small");
    QRUN(pack(h, (char*)d_packed, &need, (hipStream_t)stream));
    // every weight the launch sequences read directly must exist
    const vqs_qwen_config& c = h->c;
    const bf16_t* dummy;
    for (int i = 0; i < c.v_depth; ++i) {
        const std::string p = "model.visual.blocks." + std::to_string(i) + ".";
        QRUN(get_w(h, p + "norm1.weight", c.v_hidden, &dummy));
        QRUN(get_w(h, p + "norm2.weight", c.v_hidden, &dummy));
        QRUN(get_w(h, p + "attn.proj.bias", c.v_hidden, &dummy));
        QRUN(get_w(h, p + "mlp.down_proj.bias", c.v_hidden, &dummy));
    }
    QRUN(get_w(h, "model.visual.merger.ln_q.weight", c.v_hidden, &dummy));
    QRUN(get_w(h, "model.visual.merger.mlp.0.weight", (int64_t)h->merge_hidden * h->merge_hidden, &dummy));
    QRUN(get_w(h, "model.visual.merger.mlp.0.bias", h->merge_hidden, &dummy));
    QRUN(get_w(h, "model.visual.merger.mlp.2.weight", (int64_t)c.v_out_hidden * h->merge_hidden, &dummy));
    QRUN(get_w(h, "model.visual.merger.mlp.2.bias", c.v_out_hidden, &dummy));
    QRUN(get_w(h, "model.language_model.embed_tokens.weight", (int64_t)c.t_vocab * c.t_hidden, &dummy));
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.language_model.layers." + std::to_string(i) + ".";
        QRUN(get_w(h, p + "input_layernorm.weight", c.t_hidden, &dummy));
        QRUN(get_w(h, p + "post_attention_layernorm.weight", c.t_hidden, &dummy));
    }
    QRUN(get_w(h, "model.language_model.norm.weight", c.t_hidden, &dummy));
    QRUN(get_w(h, "lm_head.weight", (int64_t)c.t_vocab * c.t_hidden, &dummy));
    h->fp16_active = 0;
    if (h->fp16_req) QRUN(compute_ranges(h, (hipStream_t)stream));
    h->bound = true;
    return VQS_OK;
}

size_t vqs_qwen_vision_workspace_bytes(const vqs_qwen_handle* h, int32_t N, int32_t Np) {
    if (!h || N <= 0 || Np < N || N % h->c.v_merge_unit || Np % h->c.v_merge_unit) return 0;
    return carve_vision(h, nullptr, N, Np).total;
}

int vqs_qwen_encode_vision(vqs_qwen_handle* h, const void* d_patches, int32_t N, const int32_t* d_row_map,
                           const int32_t* d_inv_row, const int32_t* d_win_valid, const int32_t* d_cell_inv, const float* d_cos_w,
                           const float* d_sin_w, const float* d_cos_f, const float* d_sin_f, int32_t Np, int32_t win_len,
                           int32_t frame_len, void* d_merged, void* d_ws, size_t ws_bytes, void* stream) {
    if (!h) return VQS_ERR_INVALID;
    if (!h->bound) return qfail(h, VQS_ERR_STATE, "encode_vision: weights not bound");
    if (!d_patches || !d_row_map || !d_inv_row || !d_win_valid || !d_cell_inv || !d_cos_w || !d_sin_w || !d_cos_f || !d_sin_f ||
        !d_merged || !d_ws)
        return qfail(h, VQS_ERR_INVALID, "encode_vision: null argument");
    const vqs_qwen_config& c = h->c;
    if (N <= 0 || Np < N || N % c.v_merge_unit || win_len <= 0 || frame_len <= 0 || Np % win_len || N % frame_len ||
        win_len % c.v_merge_unit)
        return qfail(h, VQS_ERR_INVALID, "encode_vision: need N % merge_unit == 0, Np % win_len == 0, N % frame_len == 0");
    const VisWs w = carve_vision(h, (char*)d_ws, N, Np);
    if (ws_bytes < w.total) return qfail(h, VQS_ERR_WORKSPACE, "encode_vision: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int VH = c.v_hidden, VNH = c.v_heads, VPK = VNH * HDP, NC = N / c.v_merge_unit;
    const float scale = 1.0f / sqrtf((float)h->v_hd);

    // patch embed = matmul over the flattened receptive field (K padded to 64), then the window permutation into the
    // padded windowed layout (slots of partial windows are zero rows)
    QHIP(h, vqs::launch_gather_rows_bf16((const bf16_t*)d_patches, nullptr, nullptr, w.patches, N, c.v_patch_dim, c.v_patch_dim, h->v_kpatch, st), "pad patches");
    {
        GCall g{w.patches, h->patch_w, w.pre};
        g.M = N; g.N = VH; g.K = h->v_kpatch; g.lda = h->v_kpatch; g.ldw = h->v_kpatch; g.ldc = VH; g.epi = vqs::EPI_F32;
        QRUN(qgemm(h, g, st, "patch embed"));
    }
    QTAP("vis", -1, "pre", w.pre, (size_t)N * VH);
    QHIP(h, vqs::launch_gather_rows_f32(w.pre, d_row_map, w.hidden, Np, VH, st), "window permutation");
    QHIP(h, hipMemsetAsync(w.ff, 0, (size_t)Np * h->v_ffld * sizeof(bf16_t), st), "clear ff padding");
    const bool cp = h->v_compact;
    const int VAK = cp ? VH : VPK;       // width of the attention output = K of the proj GEMM
    if (cp)   // lanes [v_hd, 128) of every head slot: written by nobody, read by the attention kernel (q | k | v are contiguous)
        QHIP(h, hipMemsetAsync(w.q, 0, (size_t)((char*)w.attn - (char*)w.q), st), "clear head padding");
    // F: the range-safe fp16 forms (every 16-bit tensor below is fp16(T * sigma_T), sigma_T from the bind-time proof); else bf16
    const bool F = h->fp16_active != 0;
    if (F && (win_len < 8 || frame_len < 8)) return qfail(h, VQS_ERR_INVALID, "encode_vision: the fp16 forms need windows and frames of >= 8 patches (option fp16 = 0 runs any)");
    const vqs_qwen_handle::LayerSites one{};
    // RMSNorm whose pending deltas sit behind sd1 / sd2 and whose operand leaves behind so
    auto norm = [&](const bf16_t* d, const bf16_t* g, const bf16_t* d2, bool store, float eps, float sd1, float sd2, float so, const char* what) -> int {
        if (F) QHIP(h, vqs::launch_rmsnorm_f16s(w.hidden, d, g, w.xn, Np, VH, eps, st, d2, store, 0, 1.0f / sd1, 1.0f / sd2, so), what);
        else QHIP(h, vqs::launch_rmsnorm(w.hidden, d, g, w.xn, Np, VH, eps, st, d2, store), what);
        return VQS_OK;
    };

    const bf16_t* pend = nullptr;
    const bf16_t* pend_attn = nullptr;   // attention delta the fp32 stream has not absorbed yet
    float s_pend = 1.0f, s_pend_attn = 1.0f;     // their scales
    for (int i = 0; i < c.v_depth; ++i) {
        const std::string p = "model.visual.blocks." + std::to_string(i) + ".";
        QW(n1, p + "norm1.weight", VH);
        QW(n2, p + "norm2.weight", VH);
        QW(pb, p + "attn.proj.bias", VH);
        QW(db, p + "mlp.down_proj.bias", VH);
        const vqs_qwen_handle::LayerSites& sc = F ? h->v_sc[i] : one;
        const bool full = ((c.v_fullatt_mask >> i) & 1) != 0;
        // deferred store: norm2 normalises hidden + attention delta without writing the stream; the next norm1 (or the
        // merger norm) stores (hidden + attention delta) + mlp delta -- same fp32 sums, 22 instead of 24 B per element
        QRUN(norm(pend_attn ? pend_attn : pend, n1, pend_attn ? pend : nullptr, true, c.v_eps, pend_attn ? s_pend_attn : s_pend, s_pend, sc.x0.sigma, "vision norm1"));
        QTAP("vis", i, "h", w.hidden, (size_t)Np * VH);          // the fp32 stream this block starts from (norm1 stored it)
        QTAP("vis", i, "xn0", w.xn, (size_t)Np * VH);
        // window blocks run on the padded windowed layout (every window = win_len slots, d_win_valid of them real);
        // full-attention blocks on a frame-compact copy (original patch order), scattered back afterwards
        const bf16_t* xin = w.xn;
        int rows = Np, S = win_len;
        if (full) {
            QHIP(h, vqs::launch_gather_rows_bf16(w.xn, nullptr, d_inv_row, w.xc, N, VH, VH, VH, st), "compact frame rows");
            xin = w.xc; rows = N; S = frame_len;
        }
        const int Bseg = rows / S;
        {
            GCall g{xin, F ? h->v_qkv_h[i] : h->v_qkv_w[i], nullptr};
            g.bias = h->v_qkv_b[i];
            g.M = rows; g.N = cp ? 3 * VH : 3 * VPK; g.K = VH; g.lda = VH; g.ldw = VH; g.epi = vqs::EPI_HEADS;
            g.S = S; g.H = VNH; g.inner = cp ? VH : VPK; g.hd = HDP; g.hd_src = cp ? h->v_hd : 0;
            g.heads[0] = w.q; g.heads[1] = w.k; g.heads[2] = w.v;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.x0.sigma; g.out_scale = sc.qkv.sigma; }
            QRUN(qgemm(h, g, st, "vision qkv"));
        }
        QTAP("vis", i, "q0", w.q, (size_t)rows * VPK);           // projection output, before the rotary embedding
        QTAP("vis", i, "k0", w.k, (size_t)rows * VPK);
        QHIP(h, vqs::launch_rope_qk(w.q, w.k, full ? d_cos_f : d_cos_w, full ? d_sin_f : d_sin_w, Bseg, VNH, VNH, S, HDP, h->v_hd / 2, st, F), "vision rope");
        QTAP("vis", i, "q", w.q, (size_t)rows * VPK);            // after the rotary embedding, head-major [Bseg, heads, S, 128]
        QTAP("vis", i, "k", w.k, (size_t)rows * VPK);
        QTAP("vis", i, "v", w.v, (size_t)rows * VPK);
        {   // q and k sit behind sigma_qkv each: the scores' scale takes 1 / sigma^2; the output inherits v's sigma
            vqs::AttnParams a{w.q, w.k, w.v, w.attn, nullptr, full ? nullptr : d_win_valid, Bseg, VNH, S, scale / (sc.qkv.sigma * sc.qkv.sigma)};
            a.hd = HDP; a.out_hd = cp ? h->v_hd : 0; a.f16 = F ? 1 : 0;
            QHIP(h, vqs::launch_attention(a, st), "vision attention");
        }
        QTAP("vis", i, "attn", w.attn, (size_t)rows * VAK);
        {
            GCall g{w.attn, F ? h->v_proj_h[i] : h->v_proj_w[i], full ? (void*)w.dc : (void*)w.delta};
            g.bias = pb;
            g.M = rows; g.N = VH; g.K = VAK; g.lda = VAK; g.ldw = VAK; g.ldc = VH; g.epi = vqs::EPI_BF16;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.qkv.sigma; g.out_scale = sc.dattn.sigma; }
            QRUN(qgemm(h, g, st, "vision proj"));
        }
        if (full)
            QHIP(h, vqs::launch_gather_rows_bf16(w.dc, nullptr, d_row_map, w.delta, Np, VH, VH, VH, st), "scatter frame rows back");
        QTAP("vis", i, "d_attn", w.delta, (size_t)Np * VH);
        QRUN(norm(w.delta, n2, nullptr, false, c.v_eps, sc.dattn.sigma, 1.0f, sc.x1.sigma, "vision norm2"));
        QTAP("vis", i, "xn1", w.xn, (size_t)Np * VH);
        pend_attn = w.delta; s_pend_attn = sc.dattn.sigma;
        {
            GCall g{w.xn, F ? h->v_gu_h[i] : h->v_gu_w[i], w.ff};
            g.bias = h->v_gu_b[i];
            g.M = Np; g.N = 2 * h->v_mlp_p; g.K = VH; g.lda = VH; g.ldw = VH; g.ldc = h->v_ffld; g.epi = vqs::EPI_GATED; g.gate_act = 1;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.x1.sigma; g.out_scale = sc.act.sigma; }
            QRUN(qgemm(h, g, st, "vision gate|up"));
        }
        QTAP("vis", i, "ff", w.ff, (size_t)Np * h->v_ffld);
        {
            GCall g{w.ff, F ? h->v_down_h[i] : h->v_down_w[i], w.delta2};
            g.bias = db;
            g.M = Np; g.N = VH; g.K = h->v_ffld; g.lda = h->v_ffld; g.ldw = h->v_ffld; g.ldc = VH; g.epi = vqs::EPI_BF16;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.act.sigma; g.out_scale = sc.dmlp.sigma; }
            QRUN(qgemm(h, g, st, "vision down"));
            pend = w.delta2; s_pend = sc.dmlp.sigma;
        }
        QTAP("vis", i, "d_mlp", w.delta2, (size_t)Np * VH);
    }
    // merger: RMSNorm, the 4 patches of a cell (consecutive rows in the windowed layout) concatenated, Linear-GELU-Linear,
    // then original cell order (padding cells are simply never gathered)
    QW(lnq, "model.visual.merger.ln_q.weight", VH);
    QW(m0w, "model.visual.merger.mlp.0.weight", (int64_t)h->merge_hidden * h->merge_hidden);
    QW(m0b, "model.visual.merger.mlp.0.bias", h->merge_hidden);
    QW(m2w, "model.visual.merger.mlp.2.weight", (int64_t)c.v_out_hidden * h->merge_hidden);
    QW(m2b, "model.visual.merger.mlp.2.bias", c.v_out_hidden);
    const int NCp = Np / c.v_merge_unit;
    const float s_mx = F ? h->m_x.sigma : 1.0f, s_mid = F ? h->m_mid.sigma : 1.0f, s_mo = F ? h->m_out.sigma : 1.0f;
    QRUN(norm(pend_attn ? pend_attn : pend, lnq, pend_attn ? pend : nullptr, true, 1e-6f, pend_attn ? s_pend_attn : s_pend, s_pend, s_mx, "merger norm"));
    QTAP("vis", -1, "h_out", w.hidden, (size_t)Np * VH);
    QTAP("vis", -1, "xnm", w.xn, (size_t)Np * VH);
    {
        GCall g{w.xn, F ? h->m0_h : m0w, w.mid};
        g.bias = m0b;
        g.M = NCp; g.N = h->merge_hidden; g.K = h->merge_hidden; g.lda = h->merge_hidden; g.ldw = h->merge_hidden; g.ldc = h->merge_hidden;
        g.epi = vqs::EPI_BF16_GELU;
        if (F) { g.f16 = 3; g.acc_scale = 1.0f / s_mx; g.out_scale = s_mid; }
        QRUN(qgemm(h, g, st, "merger mlp.0"));
    }
    QTAP("vis", -1, "mid", w.mid, (size_t)NCp * h->merge_hidden);
    {
        GCall g{w.mid, F ? h->m2_h : m2w, w.merged_w};
        g.bias = m2b;
        g.M = NCp; g.N = c.v_out_hidden; g.K = h->merge_hidden; g.lda = h->merge_hidden; g.ldw = h->merge_hidden; g.ldc = c.v_out_hidden;
        g.epi = vqs::EPI_BF16;
        if (F) { g.f16 = 3; g.acc_scale = 1.0f / s_mid; g.out_scale = s_mo; }
        QRUN(qgemm(h, g, st, "merger mlp.2"));
    }
    QTAP("vis", -1, "merged_w", w.merged_w, (size_t)NCp * c.v_out_hidden);
    // d_merged leaves in the handle's operand format: bf16, or (fp16 forms) fp16 behind the merger's output scale -- vqs_qwen_score reads it back the same way
    QHIP(h, vqs::launch_gather_rows_bf16(w.merged_w, nullptr, d_cell_inv, (bf16_t*)d_merged, NC, c.v_out_hidden, c.v_out_hidden,
                                          c.v_out_hidden, st), "undo window permutation");
    return VQS_OK;
}

size_t vqs_qwen_score_workspace_bytes(const vqs_qwen_handle* h, int32_t B, int32_t L) {
    if (!h || B <= 0 || L <= 0) return 0;
    return carve_text(h, nullptr, B, L).total;
}

}  // extern "C"

namespace {
// KV cache: layer i holds K at slab 2i and V at slab 2i + 1, each [B, kv_heads, Lmax, 128] bf16 (K after the rotary embedding)
inline size_t kv_slab(const vqs_qwen_handle* h, int B, int Lmax) { return (size_t)B * h->c.t_kv_heads * (size_t)Lmax * HDP; }

// ---------------------------------------------------------------------------------------------------------------------------------
// The precise tail (vqs_qwen_handle::tail_precise).  The reference reads ONE row of the prefill -- the logits of the last prompt position
// (qwen2vl_model.py:222-301: scores[0] of generate) -- and in a decoder-only model that row's own 28-layer residual path feeds the head
// directly, while every other row reaches it only through attention (averaged over ~800 keys).  So that one row per sample is evaluated
// again, layer by layer next to the bf16 prefill, with 16 significant bits: RMSNorm outputs, the attention output and the gated product
// are split-bf16 tensors (two bf16 planes, consumed as 2 x rows STACKED rows of the same GEMM over the same weights, fp32 partial sums
// added afterwards), q|k|v, q after the rotary embedding, the softmax and the three sub-layer outputs stay fp32.  Its attention reads the
// layer's K / V as the prefill left them (bf16, head-major, the row's own position included).  Same function as the prefill's last row
// (HF Qwen2_5_VLDecoderLayer, modeling_qwen2_5_vl.py:720-787), evaluated closer to fp32 -- like the CLIP-FlanT5 row's precise decoder.
// Cost: the language model's weights streamed once more per chunk of 64 samples (~1 % of the 7B pass at B = 64).
// ---------------------------------------------------------------------------------------------------------------------------------
// One nn.Linear of the tail: A2 = split-bf16 [2][rows][lda] (planes rows * lda apart) as 2 * rows stacked rows, K slices as batch
// entries (decode_slices: the decode step's rule), fp32 partials, then launch_sum_planes adds planes and slices (+ bias / SiLU gate).
int tail_linear(vqs_qwen_handle* h, const bf16_t* A2, int lda, const bf16_t* W, int ldw, const bf16_t* bias, int rows, int N, int K, int mode,
                void* out, int ld_out, long long out_plane, const TxtWs& w, hipStream_t st, const char* what) {
    const int sk = decode_slices(N, K);
    if ((size_t)sk * 2 * rows * N * sizeof(float) > w.t_part_bytes) return qfail(h, VQS_ERR_WORKSPACE, std::string(what) + ": tail scratch too small");
    vqs::GemmParams p{};
    p.A = A2; p.W = W; p.C = w.t_part; p.bias = nullptr; p.resid = nullptr;
    p.M = 2 * rows; p.N = N; p.K = K / sk; p.lda = lda; p.ldw = ldw; p.ldc = N;
    p.S = 1; p.H = 0; p.inner = 1;
    p.batch = sk; p.sA = K / sk; p.sW = K / sk; p.sC = (long long)2 * rows * N;
    if (h->prof) {
        while (h->ev.size() < h->ev_used + 2) {
            hipEvent_t e;
            QHIP(h, hipEventCreate(&e), "hipEventCreate");
            h->ev.push_back(e);
        }
        QHIP(h, hipEventRecord(h->ev[h->ev_used], st), "hipEventRecord");
    }
    QHIP(h, vqs::launch_gemm(p, vqs::EPI_F32, 3, st), std::string("gemm ") + what);
    if (h->prof) {
        QHIP(h, hipEventRecord(h->ev[h->ev_used + 1], st), "hipEventRecord");
        h->ev_used += 2;
        h->prof_flops += 2.0 * 2.0 * (double)rows * (double)N * (double)K;
        h->prof_bytes += 2.0 * (2.0 * (double)rows + (double)N) * (double)K + 4.0 * (double)sk * 2.0 * (double)rows * (double)N;
    }
    QHIP(h, vqs::launch_sum_planes(w.t_part, sk, (long long)2 * rows * N, rows, N, N, mode, out, ld_out, out_plane, st, bias,
                                   mode == vqs::SUM_GATED_SPLIT ? 1 : 0), std::string("sum ") + what);
    return VQS_OK;
}

// One decoder layer of the tail for every chunk of <= TAIL_ROWS samples; w.k / w.v hold THIS layer's K (rotated) and V of the prefill.
int tail_layer(vqs_qwen_handle* h, const TxtWs& w, int i, const bf16_t* ln1, const bf16_t* ln2, const int32_t* d_seq_len, const int32_t* d_last_row,
               const float* d_cos, const float* d_sin, int B, int L, bool first, hipStream_t st, bool kv_f16 = false, float kv_sigma = 1.0f) {
    const vqs_qwen_config& c = h->c;
    const int TH = c.t_hidden, IQ = h->t_iq, IKV = h->t_ikv, QN = IQ + 2 * IKV;
    const float scale = 1.0f / sqrtf((float)h->t_hd);
    for (int c0 = 0; c0 < B; c0 += TAIL_ROWS) {
        const int rows = std::min(TAIL_ROWS, B - c0);
        float* th = w.t_h + (size_t)c0 * TH;
        float* td = w.t_delta + (size_t)c0 * TH;
        // ---- attention sub-layer
        QHIP(h, vqs::launch_rmsnorm_split(th, first ? nullptr : td, ln1, w.t_xn, (long long)rows * TH, rows, TH, c.t_eps, st), "tail input_layernorm");
        QRUN(tail_linear(h, w.t_xn, TH, h->t_qkv_w[i], TH, h->t_qkv_b[i], rows, QN, TH, vqs::SUM_F32, w.t_qkv, QN, 0, w, st, "tail qkv"));
        QHIP(h, vqs::launch_qwen_tail_rope_q(w.t_qkv, QN, d_cos, d_sin, d_last_row + c0, w.t_q, rows, c.t_heads, HDP, h->t_hd / 2, st), "tail rope");
        // kv_f16: the layer's K / V are the fp16 prefill's tensors behind kv_sigma (the tail's own q is in true units)
        QHIP(h, vqs::launch_qwen_tail_attn(w.t_q, w.k + (size_t)c0 * L * IKV, w.v + (size_t)c0 * L * IKV, d_seq_len + c0, w.t_attn, (long long)rows * IQ, rows,
                                           c.t_heads, c.t_kv_heads, L, scale / kv_sigma, st, kv_f16, 1.0f / kv_sigma), "tail attention");
        QRUN(tail_linear(h, w.t_attn, IQ, h->t_o_w[i], IQ, nullptr, rows, TH, IQ, vqs::SUM_F32, td, TH, 0, w, st, "tail o_proj"));
        // ---- gated FFN
        QHIP(h, vqs::launch_rmsnorm_split(th, td, ln2, w.t_xn, (long long)rows * TH, rows, TH, c.t_eps, st), "tail post_attention_layernorm");
        QHIP(h, hipMemsetAsync(w.t_ff, 0, (size_t)2 * rows * h->t_ffld * sizeof(bf16_t), st), "tail clear ff padding");
        QRUN(tail_linear(h, w.t_xn, TH, h->t_gu_w[i], h->t_xld, nullptr, rows, 2 * h->t_mlp_p, TH, vqs::SUM_GATED_SPLIT, w.t_ff, h->t_ffld,
                         (long long)rows * h->t_ffld, w, st, "tail gate|up"));
        QRUN(tail_linear(h, w.t_ff, h->t_ffld, h->t_down_w[i], h->t_ffld, nullptr, rows, TH, h->t_ffld, vqs::SUM_F32, td, TH, 0, w, st, "tail down_proj"));
    }
    return VQS_OK;
}

// final norm + lm_head of the tail -> d_logits [B, vocab] fp32
int tail_head(vqs_qwen_handle* h, const TxtWs& w, const bf16_t* fin, const bf16_t* head, float* d_logits, int B, hipStream_t st) {
    const vqs_qwen_config& c = h->c;
    const int TH = c.t_hidden;
    for (int c0 = 0; c0 < B; c0 += TAIL_ROWS) {
        const int rows = std::min(TAIL_ROWS, B - c0);
        QHIP(h, vqs::launch_rmsnorm_split(w.t_h + (size_t)c0 * TH, w.t_delta + (size_t)c0 * TH, fin, w.t_xn, (long long)rows * TH, rows, TH, c.t_eps, st), "tail final norm");
        QRUN(tail_linear(h, w.t_xn, TH, head, TH, nullptr, rows, c.t_vocab, TH, vqs::SUM_F32, d_logits + (size_t)c0 * c.t_vocab, c.t_vocab, 0, w, st, "tail lm_head"));
    }
    return VQS_OK;
}

// vqs_qwen_score / vqs_qwen_prefill: d_kv != nullptr also keeps every layer's K and V
int score_impl(vqs_qwen_handle* h, const void* d_merged, const int32_t* d_input_ids, const int32_t* d_vis_slot,
               const int32_t* d_seq_len, const int32_t* d_last_row, const float* d_cos, const float* d_sin, int32_t B,
               int32_t L, float* d_logits, void* d_ws, size_t ws_bytes, void* stream, void* d_kv, int32_t Lmax) {
    if (!h) return VQS_ERR_INVALID;
    if (!h->bound) return qfail(h, VQS_ERR_STATE, "score: weights not bound");
    if (!d_merged || !d_input_ids || !d_vis_slot || !d_seq_len || !d_last_row || !d_cos || !d_sin || !d_logits || !d_ws)
        return qfail(h, VQS_ERR_INVALID, "score: null argument");
    if (B <= 0 || L <= 0) return qfail(h, VQS_ERR_INVALID, "score: need B > 0, L > 0");
    const vqs_qwen_config& c = h->c;
    const TxtWs w = carve_text(h, (char*)d_ws, B, L);
    if (ws_bytes < w.total) return qfail(h, VQS_ERR_WORKSPACE, "score: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int TH = c.t_hidden, M = B * L, IQ = h->t_iq, IKV = h->t_ikv, XLD = h->t_xld;
    const float scale = 1.0f / sqrtf((float)h->t_hd);
    QW(embed, "model.language_model.embed_tokens.weight", (int64_t)c.t_vocab * TH);
    // F: the range-safe fp16 forms (bind-time proof; every 16-bit tensor is fp16(T * sigma_T)); d_merged arrives in the same format
    const bool F = h->fp16_active != 0;
    if (F && L < 8) return qfail(h, VQS_ERR_INVALID, "score: the fp16 forms need L >= 8 (option fp16 = 0 runs any)");
    QHIP(h, vqs::launch_qwen_embed(d_input_ids, d_vis_slot, embed, (const bf16_t*)d_merged, w.hidden, M, TH, c.t_vocab, st, F, F ? 1.0f / h->m_out.sigma : 1.0f), "embed + splice");
    QHIP(h, hipMemsetAsync(w.ff, 0, (size_t)M * h->t_ffld * sizeof(bf16_t), st), "clear ff padding");
    QTAP("txt", -1, "emb", w.hidden, (size_t)M * TH);
    const bool tail = h->tail_precise != 0 && (c.t_vocab % 4) == 0 && (c.t_hidden % 4) == 0;
    if (tail)      // the tail's starting state: the embedding rows of the last positions (fp32 copies of bf16 values: exact)
        QHIP(h, vqs::launch_gather_rows_f32(w.hidden, d_last_row, w.t_h, B, TH, TH, st), "tail embed rows");
    const vqs_qwen_handle::LayerSites one{};
    const bool rope_in_gemm = F && h->rope_fused != 0 && h->t_hd == HDP;      // K12: 128-lane heads, scaled fp16 family (gemm_quad.inc ROPE)
    auto norm = [&](const bf16_t* d, const bf16_t* g, const bf16_t* d2, bool store, float sd1, float sd2, float so, bool out_bf16, const char* what) -> int {
        if (F) QHIP(h, vqs::launch_rmsnorm_f16s(w.hidden, d, g, w.xn, M, TH, c.t_eps, st, d2, store, XLD, 1.0f / sd1, 1.0f / sd2, so, out_bf16), what);
        else QHIP(h, vqs::launch_rmsnorm(w.hidden, d, g, w.xn, M, TH, c.t_eps, st, d2, store, XLD), what);
        return VQS_OK;
    };

    const bf16_t* pend = nullptr;
    const bf16_t* pend_attn = nullptr;   // attention delta the fp32 stream has not absorbed yet
    float s_pend = 1.0f, s_pend_attn = 1.0f;
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.language_model.layers." + std::to_string(i) + ".";
        QW(ln1, p + "input_layernorm.weight", TH);
        QW(ln2, p + "post_attention_layernorm.weight", TH);
        const vqs_qwen_handle::LayerSites& sc = F ? h->t_sc[i] : one;
        QRUN(norm(pend_attn ? pend_attn : pend, ln1, pend_attn ? pend : nullptr, true, pend_attn ? s_pend_attn : s_pend, s_pend, sc.x0.sigma, false, "input_layernorm"));
        QTAP("txt", i, "h", w.hidden, (size_t)M * TH);            // the fp32 stream this layer starts from (input_layernorm stored it)
        QRUN(qtap2d(h, "txt", i, "xn0", w.xn, TH, M, XLD, st));
        {
            GCall g{w.xn, F ? h->t_qkv_h[i] : h->t_qkv_w[i], nullptr};
            g.bias = h->t_qkv_b[i];
            g.M = M; g.N = IQ + 2 * IKV; g.K = TH; g.lda = XLD; g.ldw = TH; g.epi = vqs::EPI_HEADS;
            g.S = L; g.H = c.t_heads; g.inner = IQ; g.hd = HDP; g.inner_kv = IKV; g.Hkv = c.t_kv_heads;
            g.heads[0] = w.q; g.heads[1] = w.k; g.heads[2] = w.v;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.x0.sigma; g.out_scale = sc.qkv.sigma; }
            if (rope_in_gemm) { g.rope_cos = d_cos; g.rope_sin = d_sin; }
            QRUN(qgemm(h, g, st, "qkv"));
        }
        if (!rope_in_gemm) {
            QTAP("txt", i, "q0", w.q, (size_t)M * IQ);            // projection output, before the rotary embedding
            QTAP("txt", i, "k0", w.k, (size_t)M * IKV);
            QHIP(h, vqs::launch_rope_qk(w.q, w.k, d_cos, d_sin, B, c.t_heads, c.t_kv_heads, L, HDP, h->t_hd / 2, st, F), "rope");
        }
        QTAP("txt", i, "q", w.q, (size_t)M * IQ);                 // after the rotary embedding, head-major [B, heads, L, 128]
        QTAP("txt", i, "k", w.k, (size_t)M * IKV);
        QTAP("txt", i, "v", w.v, (size_t)M * IKV);
        if (d_kv) {   // rows [b, head, 0 .. L) of the cache slabs (pitch Lmax positions); the cache (and the decode step) is bf16 in true units
            bf16_t* kc = (bf16_t*)d_kv + (size_t)(2 * i) * kv_slab(h, B, Lmax);
            bf16_t* vc = kc + kv_slab(h, B, Lmax);
            const size_t wb = (size_t)L * HDP * sizeof(bf16_t), pb = (size_t)Lmax * HDP * sizeof(bf16_t), nr = (size_t)B * c.t_kv_heads;
            if (F) {
                if (nr > 65535) return qfail(h, VQS_ERR_INVALID, "prefill: B * kv_heads exceeds 65535");
                QHIP(h, vqs::launch_f16_to_bf16_rows(w.k, kc, (int)nr, (long long)L * HDP, (long long)L * HDP, (long long)Lmax * HDP, 1.0f / sc.qkv.sigma, st), "keep K");
                QHIP(h, vqs::launch_f16_to_bf16_rows(w.v, vc, (int)nr, (long long)L * HDP, (long long)L * HDP, (long long)Lmax * HDP, 1.0f / sc.qkv.sigma, st), "keep V");
            } else {
                QHIP(h, hipMemcpy2DAsync(kc, pb, w.k, wb, wb, nr, hipMemcpyDeviceToDevice, st), "keep K");
                QHIP(h, hipMemcpy2DAsync(vc, pb, w.v, wb, wb, nr, hipMemcpyDeviceToDevice, st), "keep V");
            }
        }
        {
            vqs::AttnParams a{w.q, w.k, w.v, w.attn, nullptr, d_seq_len, B, c.t_heads, L, scale / (sc.qkv.sigma * sc.qkv.sigma)};
            a.hd = HDP; a.Hkv = c.t_kv_heads; a.causal = 1; a.f16 = F ? 1 : 0;
            QHIP(h, vqs::launch_attention(a, st), "attention");
        }
        QTAP("txt", i, "attn", w.attn, (size_t)M * IQ);
        if (tail)      // this layer's K / V are in w.k / w.v until the next layer's qkv launch
            QRUN(tail_layer(h, w, i, ln1, ln2, d_seq_len, d_last_row, d_cos, d_sin, B, L, i == 0, st, F, sc.qkv.sigma));
        {
            GCall g{w.attn, F ? h->t_o_h[i] : h->t_o_w[i], w.delta};
            g.M = M; g.N = TH; g.K = IQ; g.lda = IQ; g.ldw = IQ; g.ldc = TH; g.epi = vqs::EPI_BF16;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.qkv.sigma; g.out_scale = sc.dattn.sigma; }
            QRUN(qgemm(h, g, st, "o_proj"));
        }
        QTAP("txt", i, "d_attn", w.delta, (size_t)M * TH);
        QRUN(norm(w.delta, ln2, nullptr, false, sc.dattn.sigma, 1.0f, sc.x1.sigma, false, "post_attention_layernorm"));
        QRUN(qtap2d(h, "txt", i, "xn1", w.xn, TH, M, XLD, st));
        pend_attn = w.delta; s_pend_attn = sc.dattn.sigma;
        {
            GCall g{w.xn, F ? h->t_gu_h[i] : h->t_gu_w[i], w.ff};
            g.M = M; g.N = 2 * h->t_mlp_p; g.K = TH; g.lda = XLD; g.ldw = XLD; g.ldc = h->t_ffld; g.epi = vqs::EPI_GATED; g.gate_act = 1;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.x1.sigma; g.out_scale = sc.act.sigma; }
            QRUN(qgemm(h, g, st, "gate|up"));
        }
        QTAP("txt", i, "ff", w.ff, (size_t)M * h->t_ffld);
        {
            GCall g{w.ff, F ? h->t_down_h[i] : h->t_down_w[i], w.delta2};
            g.M = M; g.N = TH; g.K = h->t_ffld; g.lda = h->t_ffld; g.ldw = h->t_ffld; g.ldc = TH; g.epi = vqs::EPI_BF16;
            if (F) { g.f16 = 3; g.acc_scale = 1.0f / sc.act.sigma; g.out_scale = sc.dmlp.sigma; }
            QRUN(qgemm(h, g, st, "down_proj"));
            pend = w.delta2; s_pend = sc.dmlp.sigma;
        }
        QTAP("txt", i, "d_mlp", w.delta2, (size_t)M * TH);
    }
    QW(fin, "model.language_model.norm.weight", TH);
    QW(head, "lm_head.weight", (int64_t)c.t_vocab * TH);
    // the final norm's operand is bf16 in true units either way: lm_head (bf16 weights, fp32 logits) reads it when the tail is off
    QRUN(norm(pend_attn ? pend_attn : pend, fin, pend_attn ? pend : nullptr, true, pend_attn ? s_pend_attn : s_pend, s_pend, 1.0f, true, "final norm"));
    QTAP("txt", -1, "h_out", w.hidden, (size_t)M * TH);
    QRUN(qtap2d(h, "txt", -1, "xnf", w.xn, TH, M, XLD, st));
    if (tail) {
        QRUN(tail_head(h, w, fin, head, d_logits, B, st));
        return VQS_OK;
    }
    QHIP(h, vqs::launch_gather_rows_bf16(w.xn, nullptr, d_last_row, w.last, B, TH, XLD, TH, st), "last positions");
    {
        GCall g{w.last, head, d_logits};
        g.M = B; g.N = c.t_vocab; g.K = TH; g.lda = TH; g.ldw = TH; g.ldc = c.t_vocab; g.epi = vqs::EPI_F32;
        QRUN(qgemm(h, g, st, "lm_head"));
    }
    return VQS_OK;
}

struct DecWs {
    float* hidden;
    bf16_t *xn, *delta, *delta2, *qkv, *q, *attn, *ff;
    int* neg1;
    float* part;              // fp32 partials of one decode nn.Linear: [K slices][B][N]
    size_t part_bytes;
    size_t total;
};
DecWs carve_decode(const vqs_qwen_handle* h, char* base, int B) {
    const vqs_qwen_config& c = h->c;
    Carver cv{base};
    DecWs w{};
    const size_t b = (size_t)B;
    w.hidden = cv.take<float>(b * c.t_hidden);
    w.xn = cv.take<bf16_t>(b * h->t_xld);
    w.delta = cv.take<bf16_t>(b * c.t_hidden);
    w.delta2 = cv.take<bf16_t>(b * c.t_hidden);
    w.qkv = cv.take<bf16_t>(b * (h->t_iq + 2 * h->t_ikv));
    w.q = cv.take<bf16_t>(b * h->t_iq);
    w.attn = cv.take<bf16_t>(b * h->t_iq);
    w.ff = cv.take<bf16_t>(b * h->t_ffld);
    w.neg1 = cv.take<int>(b);
    {   // widest partial set of the step's four linears: slices(N, K) x N fp32 per row, by the SAME rule decode_linear launches with
        // (a hand-written bound missed gate|up with two slices -- Qwen2.5-VL-3B's 2048 / 11008 -- and every step failed)
        const int QN = h->t_iq + 2 * h->t_ikv;
        size_t widest = (size_t)decode_slices(QN, c.t_hidden) * QN;                                       // qkv
        widest = std::max(widest, (size_t)decode_slices(c.t_hidden, h->t_iq) * c.t_hidden);               // o_proj
        widest = std::max(widest, (size_t)decode_slices(2 * h->t_mlp_p, c.t_hidden) * 2 * h->t_mlp_p);    // gate|up
        widest = std::max(widest, (size_t)decode_slices(c.t_hidden, h->t_ffld) * c.t_hidden);             // down_proj
        w.part_bytes = b * widest * sizeof(float);
        w.part = cv.take<float>(b * widest);
    }
    w.total = align_up(cv.off);
    return w;
}

// A decode step's nn.Linear (M = B rows, one per sample): the 256-row quad tiles of the prefill put one M-tile x N/256 N-tiles = 14-18
// workgroups on the chip for qkv / o / down and computed on 192 padding rows; the step ran at 7 % of the HBM roofline (round 3).
// Here the launch is the stream form (gemm_stream.inc: <= 128 rows, W streamed through a five-stage LDS ring) over K slices as batch
// entries -- the fewest slices (a divisor of K / 64, <= 16) that put >= 192 (slice, 128-column) items on the chip: a function of the
// weight's shape only -- into fp32 partials, and one reducer pass applies the epilogue (bias, or SiLU(gate) * up on the packed
// gate|up columns) and rounds to bf16.  Falls back to the persistent kernel (same bits) when the form does not apply (B > 128).
int decode_linear(vqs_qwen_handle* h, const bf16_t* A, int lda, const bf16_t* W, int ldw, const bf16_t* bias, bf16_t* out, int ldo, int B,
                  int N, int K, int gated, const DecWs& w, hipStream_t st, const char* what) {
    const int sk = decode_slices(N, K);
    if ((size_t)sk * B * N * sizeof(float) > w.part_bytes) return qfail(h, VQS_ERR_WORKSPACE, std::string(what) + ": decode scratch too small");
    vqs::GemmParams p{};
    p.A = A; p.W = W; p.C = w.part; p.bias = nullptr; p.resid = nullptr;
    p.M = B; p.N = N; p.K = K / sk; p.lda = lda; p.ldw = ldw; p.ldc = N;
    p.S = 1; p.H = 0; p.inner = 1;
    p.batch = sk; p.sA = K / sk; p.sW = K / sk; p.sC = (long long)B * N;
    if (h->prof) {
        while (h->ev.size() < h->ev_used + 2) {
            hipEvent_t e;
            QHIP(h, hipEventCreate(&e), "hipEventCreate");
            h->ev.push_back(e);
        }
        QHIP(h, hipEventRecord(h->ev[h->ev_used], st), "hipEventRecord");
    }
    QHIP(h, vqs::launch_gemm(p, vqs::EPI_F32, 3, st), std::string("gemm ") + what);
    if (h->prof) {
        QHIP(h, hipEventRecord(h->ev[h->ev_used + 1], st), "hipEventRecord");
        h->ev_used += 2;
        h->prof_flops += 2.0 * (double)B * (double)N * (double)K;
        h->prof_bytes += 2.0 * ((double)B + (double)N) * (double)K + 4.0 * (double)sk * (double)B * (double)N;
    }
    QHIP(h, vqs::launch_reduce_slices_act(w.part, sk, (long long)B * N, B, N, N, bias, gated, out, ldo, st), std::string("reduce ") + what);
    return VQS_OK;
}
}  // namespace

extern "C" {

int vqs_qwen_score(vqs_qwen_handle* h, const void* d_merged, const int32_t* d_input_ids, const int32_t* d_vis_slot,
                   const int32_t* d_seq_len, const int32_t* d_last_row, const float* d_cos, const float* d_sin, int32_t B,
                   int32_t L, float* d_logits, void* d_ws, size_t ws_bytes, void* stream) {
    return score_impl(h, d_merged, d_input_ids, d_vis_slot, d_seq_len, d_last_row, d_cos, d_sin, B, L, d_logits, d_ws, ws_bytes, stream,
                      nullptr, 0);
}

size_t vqs_qwen_kv_bytes(const vqs_qwen_handle* h, int32_t B, int32_t Lmax) {
    if (!h || B <= 0 || Lmax <= 0) return 0;
    return (size_t)2 * h->c.t_layers * kv_slab(h, B, Lmax) * sizeof(bf16_t);
}

int vqs_qwen_prefill(vqs_qwen_handle* h, const void* d_merged, const int32_t* d_input_ids, const int32_t* d_vis_slot,
                     const int32_t* d_seq_len, const int32_t* d_last_row, const float* d_cos, const float* d_sin, int32_t B,
                     int32_t L, float* d_logits, void* d_ws, size_t ws_bytes, void* d_kv, size_t kv_bytes, int32_t Lmax, void* stream) {
    if (!h) return VQS_ERR_INVALID;
    if (!d_kv || Lmax < L || kv_bytes < vqs_qwen_kv_bytes(h, B, Lmax)) return qfail(h, VQS_ERR_WORKSPACE, "prefill: KV cache missing or too small");
    if (Lmax > VQS_QWEN_MAX_CACHE_POSITIONS)      // the decode step's one-row attention keeps Lmax fp32 scores in LDS: refuse BEFORE the prefill is spent
        return qfail(h, VQS_ERR_INVALID, "prefill: Lmax exceeds the decode kernel's limit (" + std::to_string(VQS_QWEN_MAX_CACHE_POSITIONS) + " cache positions)");
    return score_impl(h, d_merged, d_input_ids, d_vis_slot, d_seq_len, d_last_row, d_cos, d_sin, B, L, d_logits, d_ws, ws_bytes, stream, d_kv,
                      Lmax);
}

size_t vqs_qwen_decode_workspace_bytes(const vqs_qwen_handle* h, int32_t B) {
    if (!h || B <= 0) return 0;
    return carve_decode(h, nullptr, B).total;
}

int vqs_qwen_decode(vqs_qwen_handle* h, const int32_t* d_ids, const int32_t* d_len, const float* d_cos, const float* d_sin, int32_t B,
                    int32_t Lmax, void* d_kv, size_t kv_bytes, float* d_logits, void* d_ws, size_t ws_bytes, void* stream) {
    if (!h) return VQS_ERR_INVALID;
    if (!h->bound) return qfail(h, VQS_ERR_STATE, "decode: weights not bound");
    if (!d_ids || !d_len || !d_cos || !d_sin || !d_kv || !d_logits || !d_ws) return qfail(h, VQS_ERR_INVALID, "decode: null argument");
    if (B <= 0 || Lmax <= 0) return qfail(h, VQS_ERR_INVALID, "decode: need B > 0, Lmax > 0");
    if (Lmax > VQS_QWEN_MAX_CACHE_POSITIONS) return qfail(h, VQS_ERR_INVALID, "decode: Lmax exceeds " + std::to_string(VQS_QWEN_MAX_CACHE_POSITIONS) + " cache positions");
    if (kv_bytes < vqs_qwen_kv_bytes(h, B, Lmax)) return qfail(h, VQS_ERR_WORKSPACE, "decode: KV cache too small");
    const vqs_qwen_config& c = h->c;
    const DecWs w = carve_decode(h, (char*)d_ws, B);
    if (ws_bytes < w.total) return qfail(h, VQS_ERR_WORKSPACE, "decode: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int TH = c.t_hidden, IQ = h->t_iq, IKV = h->t_ikv, XLD = h->t_xld, QN = IQ + 2 * IKV;
    const float scale = 1.0f / sqrtf((float)h->t_hd);
    QW(embed, "model.language_model.embed_tokens.weight", (int64_t)c.t_vocab * TH);
    QHIP(h, hipMemsetAsync(w.neg1, 0xff, (size_t)B * sizeof(int), st), "no placeholder rows");
    QHIP(h, vqs::launch_qwen_embed(d_ids, w.neg1, embed, embed, w.hidden, B, TH, c.t_vocab, st), "embed");
    QHIP(h, hipMemsetAsync(w.ff, 0, (size_t)B * h->t_ffld * sizeof(bf16_t), st), "clear ff padding");
    QTAP("dec", -1, "emb", w.hidden, (size_t)B * TH);
    const bf16_t* pend = nullptr;
    const bf16_t* pend_attn = nullptr;
    for (int i = 0; i < c.t_layers; ++i) {
        const std::string p = "model.language_model.layers." + std::to_string(i) + ".";
        QW(ln1, p + "input_layernorm.weight", TH);
        QW(ln2, p + "post_attention_layernorm.weight", TH);
        bf16_t* kc = (bf16_t*)d_kv + (size_t)(2 * i) * kv_slab(h, B, Lmax);
        bf16_t* vc = kc + kv_slab(h, B, Lmax);
        QHIP(h, vqs::launch_rmsnorm(w.hidden, pend_attn ? pend_attn : pend, ln1, w.xn, B, TH, c.t_eps, st, pend_attn ? pend : nullptr, true, XLD), "input_layernorm");
        QTAP("dec", i, "h", w.hidden, (size_t)B * TH);
        QRUN(qtap2d(h, "dec", i, "xn0", w.xn, TH, B, XLD, st));
        // one position per sample: head-major [B, heads, 1, 128] IS token-major [B, heads * 128]
        QRUN(decode_linear(h, w.xn, XLD, h->t_qkv_w[i], TH, h->t_qkv_b[i], w.qkv, QN, B, QN, TH, 0, w, st, "decode qkv"));
        QTAP("dec", i, "qkv", w.qkv, (size_t)B * QN);
        QHIP(h, vqs::launch_qwen_decode_rope_append(w.qkv, d_cos, d_sin, d_len, w.q, kc, vc, B, c.t_heads, c.t_kv_heads, HDP, h->t_hd / 2, Lmax, st),
             "decode rope + append");
        QTAP("dec", i, "q", w.q, (size_t)B * IQ);
        QHIP(h, vqs::launch_qwen_decode_attn(w.q, kc, vc, d_len, w.attn, B, c.t_heads, c.t_kv_heads, Lmax, scale, st), "decode attention");
        QTAP("dec", i, "attn", w.attn, (size_t)B * IQ);
        QRUN(decode_linear(h, w.attn, IQ, h->t_o_w[i], IQ, nullptr, w.delta, TH, B, TH, IQ, 0, w, st, "decode o_proj"));
        QTAP("dec", i, "d_attn", w.delta, (size_t)B * TH);
        QHIP(h, vqs::launch_rmsnorm(w.hidden, w.delta, ln2, w.xn, B, TH, c.t_eps, st, nullptr, false, XLD), "post_attention_layernorm");
        QRUN(qtap2d(h, "dec", i, "xn1", w.xn, TH, B, XLD, st));
        pend_attn = w.delta;
        QRUN(decode_linear(h, w.xn, XLD, h->t_gu_w[i], XLD, nullptr, w.ff, h->t_ffld, B, 2 * h->t_mlp_p, TH, 1, w, st, "decode gate|up"));
        QTAP("dec", i, "ff", w.ff, (size_t)B * h->t_ffld);
        QRUN(decode_linear(h, w.ff, h->t_ffld, h->t_down_w[i], h->t_ffld, nullptr, w.delta2, TH, B, TH, h->t_ffld, 0, w, st, "decode down_proj"));
        pend = w.delta2;
        QTAP("dec", i, "d_mlp", w.delta2, (size_t)B * TH);
    }
    QW(fin, "model.language_model.norm.weight", TH);
    QW(head, "lm_head.weight", (int64_t)c.t_vocab * TH);
    QHIP(h, vqs::launch_rmsnorm(w.hidden, pend_attn ? pend_attn : pend, fin, w.xn, B, TH, c.t_eps, st, pend_attn ? pend : nullptr, true, XLD), "final norm");
    QTAP("dec", -1, "h_out", w.hidden, (size_t)B * TH);
    QRUN(qtap2d(h, "dec", -1, "xnf", w.xn, TH, B, XLD, st));
    {
        GCall g{w.xn, head, d_logits};
        g.M = B; g.N = c.t_vocab; g.K = TH; g.lda = XLD; g.ldw = TH; g.ldc = c.t_vocab; g.epi = vqs::EPI_F32;
        QRUN(qgemm(h, g, st, "lm_head"));
    }
    return VQS_OK;
}

}  // extern "C"
