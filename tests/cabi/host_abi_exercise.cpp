// Host-side exercise of the C ABI under AddressSanitizer + UBSan: everything that runs without touching a device
// (creation, sizing, options, taps registration, error paths, integer helpers).  No GPU needed.
#include "vqs.h"
#include "vqs_debug.h"
#include "vqs_qwen.h"
#include <cassert>
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
    vqs_config c;
    memset(&c, 0, sizeof c);
    c.vis_hidden = 1024; c.vis_layers_run = 23; c.vis_heads = 16; c.vis_mlp = 4096; c.vis_patch = 14; c.vis_image = 336; c.vis_ln_eps = 1e-5f;
    c.d_model = 4096; c.n_heads = 64; c.d_kv = 64; c.d_ff = 10240; c.enc_layers = 24; c.dec_layers = 24; c.vocab = 32128;
    c.rel_buckets = 32; c.rel_max_distance = 128; c.t5_ln_eps = 1e-6f;
    vqs_handle* h = nullptr;
    assert(vqs_create(&c, &h) == 0 && h);
    assert(vqs_packed_bytes(h) > 0);
    assert(vqs_encode_workspace_bytes(h, 256) > 0 && vqs_encode_workspace_bytes(h, 0) == 0);
    assert(vqs_score_workspace_bytes(h, 256, 33, 2) > 0 && vqs_score_workspace_bytes(h, 0, 33, 2) == 0);
    assert(vqs_generate_workspace_bytes(h, 4, 33, 64) > 0 && vqs_generate_workspace_bytes(h, 4, 33, 100000) == 0);
    assert(vqs_set_option(h, "cross_mode", 0) == 0 && vqs_set_option(h, "cross_mode", 1) == 0);
    assert(vqs_set_option(h, "cross_mode", 7) != 0 && vqs_set_option(h, "nonsense", 1) != 0 && vqs_set_option(h, nullptr, 1) != 0);
    {
        int32_t v = -1;                                   // defaults are readable, a set is read back, unknown names and null arguments are refused
        assert(vqs_get_option(h, "vit_fp16", &v) == 0 && v == 1 && vqs_get_option(h, "dec_precise", &v) == 0 && v == 1);
        assert(vqs_get_option(h, "gemm_variant", &v) == 0 && v == 3 && vqs_get_option(h, "cross_mode", &v) == 0 && v == 1);
        assert(vqs_set_option(h, "vit_fp16", 0) == 0 && vqs_get_option(h, "vit_fp16", &v) == 0 && v == 0 && vqs_set_option(h, "vit_fp16", 1) == 0);
        assert(vqs_set_option(h, "vit_fp16", 2) != 0 && vqs_get_option(h, "nonsense", &v) != 0 && vqs_get_option(h, "vit_fp16", nullptr) != 0);
        assert(vqs_get_option(nullptr, "vit_fp16", &v) != 0 && vqs_get_option(h, nullptr, &v) != 0 && vqs_get_option(h, "tile_order:64x64", &v) != 0);
    }
    assert(strlen(vqs_last_error(h)) > 0);
    int dummy = 0;
    assert(vqs_debug_tap(h, "enc.3.xn0", &dummy, 4) == 0 && vqs_debug_tap(h, "enc.3.xn0", nullptr, 0) == 0 && vqs_debug_tap(h, nullptr, nullptr, 0) == 0);
    assert(vqs_debug_tap_window(h, 3, 2) == 0 && vqs_debug_tap_window(h, 0, 0) == 0 && vqs_debug_tap_window(h, -1, 2) != 0);
    // GEMM form resolution (host arithmetic): the quad form for bf16 results whatever M, the 32-bit kernels refuse >= 4 GiB operands
    // (13 = the quad family's few-row launch shape, gemm_slim.inc, round 6: same bits; variant 10 = the quad kernel whatever the shape)
    assert(vqs_debug_gemm_form(155648, 20480, 4096, 4096, 4096, 5, 1, 3, 0, 0, 0) == 10 && vqs_debug_gemm_form(2, 20480, 4096, 4096, 4096, 5, 1, 3, 0, 0, 0) == 13 &&
           vqs_debug_gemm_form(2, 20480, 4096, 4096, 4096, 5, 1, 10, 0, 0, 0) == 10);
    // round 4: the stream form's rule (<= 128 rows per entry, >= 4 items since round 6 (192 before), not a quad call site) and the batched debug launch's argument checks
    assert(vqs_debug_gemm_form(128, 608, 4096, 4096, 4096, 3, 256, 3, 0, 0, 0) == 12 && vqs_debug_gemm_form(128, 608, 4096, 4096, 4096, 3, 16, 3, 0, 0, 0) == 12 &&
           vqs_debug_gemm_form(64, 384, 4096, 4096, 4096, 3, 1, 3, 0, 0, 0) == 3);
    assert(vqs_debug_gemm_form(64, 32768, 4096, 4096, 4096, 0, 1, 3, 0, 0, 0) == 13 && vqs_debug_gemm_form(64, 32768, 4096, 4096, 4096, 0, 1, 10, 0, 0, 0) == 10);
    assert(vqs_debug_gemm_batched(nullptr, nullptr, nullptr, 128, 608, 4096, 4096, 4096, 640, 3, 256, 0, 0, 0, 0, 0, 3, nullptr) == VQS_ERR_INVALID);
    {
        int dummy2 = 0;
        assert(vqs_debug_gemm_batched(&dummy2, &dummy2, &dummy2, 128, 608, 4096, 4096, 4096, 640, 5, 256, 0, 0, 0, 0, 0, 3, nullptr) == VQS_ERR_INVALID);   // gated: not a batched debug epilogue
        assert(vqs_debug_gemm_batched(&dummy2, &dummy2, &dummy2, 128, 608, 4096, 4096, 4096, 640, 3, 0, 0, 0, 0, 0, 0, 3, nullptr) == VQS_ERR_INVALID);     // batch < 1
    }
    assert(vqs_debug_gemm_form(400000, 4096, 10240, 10240, 10240, 3, 1, 3, 0, 0, 0) == 0 && vqs_debug_gemm_form(400000, 4096, 10240, 10240, 10240, 3, 2, 3, 0, 0, 0) == -1);
    int64_t ld = 0;
    assert(vqs_workspace_offset(h, "logits", 4, 33, 2, &ld) >= 0 && ld == 32128);
    assert(vqs_workspace_offset(h, "dec_out", 4, 33, 2, &ld) >= 0 && vqs_workspace_offset(h, "nope", 4, 33, 2, &ld) < 0);
    assert(vqs_workspace_offset(h, "vit_hidden", 4, 0, 0, &ld) >= 0);
    // calls that must fail cleanly before any device access
    assert(vqs_score(h, &dummy, nullptr, nullptr, nullptr, 1, 33, 2, nullptr, nullptr, nullptr, 0, nullptr) != 0);
    assert(vqs_encode_images(h, &dummy, 1, &dummy, &dummy, 16, nullptr) != 0);      // weights not bound
    assert(vqs_bind_weights(h, nullptr, 0, nullptr, 0, nullptr) != 0);
    for (int rel = -700; rel <= 700; ++rel) {
        const int b = vqs_relpos_bucket(rel, 1, 32, 128), u = vqs_relpos_bucket(rel, 0, 32, 128);
        assert(b >= 0 && b < 32 && u >= 0 && u < 32);
    }
    std::vector<int64_t> off(64);
    assert(vqs_debug_heads_rows(5, 608, 64, 64, 64, off.data()) == 0 && vqs_debug_heads_rows(5, 7, 64, 64, 4, off.data()) != 0);
    assert(vqs_attention_lds_bytes(608, 1, 64) > 0 && vqs_attention_lds_bytes(608, 1, 77) < 0);
    // tile order: option parsing (per-shape table, removal, malformed names / values) and the host walk of the tile map
    assert(vqs_set_option(h, "tile_order:20480x4096", 4 | 2 << 8) == 0 && vqs_set_option(h, "tile_order:20480x4096", 2) == 0);
    assert(vqs_set_option(h, "tile_order:20480x4096", 0) == 0 && vqs_set_option(h, "tile_order:4096x10240", 0) == 0);
    assert(vqs_set_option(h, "tile_order:", 8) != 0 && vqs_set_option(h, "tile_order:12x", 8) != 0 && vqs_set_option(h, "tile_order:-4x64", 8) != 0);
    assert(vqs_set_option(h, "tile_order:64x64", 65) != 0 && vqs_set_option(h, "tile_order:64x64", 1 << 16) != 0 && vqs_set_option(h, "tile_order:64x64", 8 | 9 << 8) != 0);
    assert(vqs_set_option(h, "nt_store:4096x10240", 1) == 0 && vqs_set_option(h, "nt_store:4096x10240", 2) == 0 && vqs_set_option(h, "nt_store:4096x10240", 0) == 0);
    assert(vqs_set_option(h, "nt_store:4096x10240", 3) != 0 && vqs_set_option(h, "nt_store:x", 1) != 0);
    assert(vqs_set_option(h, "l2_touch:12288x4096", 1) == 0 && vqs_set_option(h, "l2_touch:12288x4096", 0) == 0 && vqs_set_option(h, "l2_touch:12288x4096", 5) != 0);
    assert(vqs_set_option(h, "gemm_variant", 11) == 0 && vqs_set_option(h, "gemm_variant", 3) == 0 && vqs_set_option(h, "gemm_variant", 6) != 0 && vqs_set_option(h, "gemm_variant", 8) != 0);
    {
        const int M = 155648, N = 20480, nwg = (M / 256) * (N / 256);
        std::vector<int32_t> tiles(4 * (size_t)nwg);
        assert(vqs_debug_tile_order(M, N, 4096, 1, 4, 2, 256, tiles.data()) == (4 | 2 << 8));
        assert(vqs_debug_tile_order(M, N, 4096, 1, 0, 0, 256, tiles.data()) == (4 | 2 << 8));
        assert(vqs_debug_tile_order(577 * 256, 1024, 1024, 1, 8, 2, 256, tiles.data()) == (8 | 1 << 8));     // 577 x 2 tiles per range: remainders -> ns = 1
        assert(vqs_debug_tile_order(M, N, 4096, 1, 4, 2, 250, tiles.data()) < 0 && vqs_debug_tile_order(0, N, 4096, 1, 4, 2, 256, tiles.data()) < 0);
    }
    vqs_destroy(h);
    vqs_config bad = c;
    bad.d_kv = 32;
    vqs_handle* hb = nullptr;
    assert(vqs_create(&bad, &hb) != 0);
    vqs_destroy(hb);
    assert(vqs_create(nullptr, &hb) != 0);
    {   // Qwen2.5-VL row: sizes, options and the calls that must fail cleanly before any device access
        vqs_qwen_config q;
        memset(&q, 0, sizeof q);
        q.v_depth = 32; q.v_hidden = 1280; q.v_heads = 16; q.v_mlp = 3420; q.v_patch_dim = 1176; q.v_merge_unit = 4; q.v_out_hidden = 3584;
        q.v_fullatt_mask = (1 << 7) | (1 << 15) | (1 << 23) | (int32_t)(1u << 31); q.v_eps = 1e-6f;
        q.t_vocab = 152064; q.t_hidden = 3584; q.t_layers = 28; q.t_heads = 28; q.t_kv_heads = 4; q.t_mlp = 18944; q.t_eps = 1e-6f;
        vqs_qwen_handle* qh = nullptr;
        assert(vqs_qwen_create(&q, &qh) == 0 && qh);
        assert(vqs_qwen_packed_bytes(qh) > 0);
        assert(vqs_qwen_vision_workspace_bytes(qh, 3072, 3072) > 0 && vqs_qwen_vision_workspace_bytes(qh, 3071, 3072) == 0);
        assert(vqs_qwen_score_workspace_bytes(qh, 64, 808) > 0 && vqs_qwen_score_workspace_bytes(qh, 0, 808) == 0);
        // KV cache: 28 layers x (K, V) x [B, 4, Lmax, 128] bf16
        assert(vqs_qwen_kv_bytes(qh, 64, 816) == (size_t)2 * 28 * 64 * 4 * 816 * 128 * 2 && vqs_qwen_kv_bytes(qh, 0, 816) == 0);
        assert(vqs_qwen_decode_workspace_bytes(qh, 64) > 0 && vqs_qwen_decode_workspace_bytes(qh, 0) == 0);
        assert(vqs_qwen_debug_option(qh, "x_pitch", 4096) == 0 && vqs_qwen_debug_option(qh, "x_pitch", 3584) == 0);
        assert(vqs_qwen_debug_option(qh, "x_pitch", 3000) != 0 && vqs_qwen_debug_option(qh, "x_pitch", 3600) != 0 && vqs_qwen_debug_option(qh, "nope", 1) != 0);
        assert(strlen(vqs_qwen_last_error(qh)) > 0);
        assert(vqs_qwen_debug_tap(qh, "txt.3.xn0", &dummy, 4) == 0 && vqs_qwen_debug_tap(qh, nullptr, nullptr, 0) == 0);
        assert(vqs_qwen_decode(qh, &dummy, &dummy, nullptr, nullptr, 1, 16, &dummy, 1 << 20, nullptr, &dummy, 1 << 20, nullptr) != 0);   // not bound
        assert(vqs_qwen_prefill(qh, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr, 1, 16, nullptr, &dummy, 16, nullptr, 0, 16, nullptr) != 0);
        assert(vqs_qwen_prefill(qh, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr, 1, 16, nullptr, &dummy, 16, &dummy, 16, 8, nullptr) != 0);   // Lmax < L
        // a cache longer than the decode kernel can address is refused by the prefill itself, before anything is launched
        assert(vqs_qwen_prefill(qh, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr, 1, 16, nullptr, &dummy, 16, &dummy, (size_t)-1,
                                VQS_QWEN_MAX_CACHE_POSITIONS + 1, nullptr) == VQS_ERR_INVALID);
        assert(strstr(vqs_qwen_last_error(qh), "cache positions") != nullptr);
        assert(vqs_qwen_score(qh, &dummy, &dummy, &dummy, &dummy, &dummy, nullptr, nullptr, 1, 16, nullptr, &dummy, 16, nullptr) != 0);
        vqs_qwen_destroy(qh);
        q.t_heads = 27;        // hidden not divisible by heads
        assert(vqs_qwen_create(&q, &qh) != 0);
    }
    printf("host ABI exercise under ASan/UBSan: ok\n");
    return 0;
}
