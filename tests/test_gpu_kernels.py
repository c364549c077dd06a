"""Per-kernel parity (-m gpu): every HIP kernel, called through the C ABI, against a plain fp32 torch
reference of the same op on the same seeded inputs."""
import numpy as np
import pytest
import torch

from tests.gpu_util import (assert_close, attention_ref, gelu_erf, gelu_new, interleave_gate, quick_gelu, randn_bf16)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    from t2v_metrics_amd import engine
    engine.load_library()
    return engine


GEMM_SHAPES = [(256, 256, 64), (512, 768, 128), (300, 264, 192), (2065, 1024, 1024), (512, 1000, 256), (40, 64, 640)]


@pytest.mark.parametrize("variant", [0, 2, 3, 5, 11])   # the shipped kernels (3 = the library's rule: the quad form wherever it is eligible;
                                                        # 11 = the 8-wave forms by shape; 1 / 4 / 6-9 exist in -DVQS_LAB builds only)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bf16_and_f32(eng, M, N, K, variant):
    A = randn_bf16(M, K, seed=1)
    W = randn_bf16(N, K, seed=2, scale=K ** -0.5)
    bias = randn_bf16(N, seed=3)
    ref = A.float() @ W.float().t()
    out = eng.gemm(A, W, 0, variant=variant)
    assert_close(out, ref, 2e-2, 1e-2, f"gemm bf16 {M}x{N}x{K} v{variant}")
    out = eng.gemm(A, W, 0, bias=bias, variant=variant)
    assert_close(out, ref + bias.float(), 2e-2, 1e-2, "gemm bf16+bias")
    out = eng.gemm(A, W, 3, bias=bias, variant=variant)
    assert_close(out, ref + bias.float(), 2e-3, 1e-4, "gemm f32+bias")
    out = eng.gemm(A, W, 1, bias=bias, variant=variant)
    assert_close(out, quick_gelu(ref + bias.float()), 2e-2, 1e-2, "gemm quick_gelu")
    out = eng.gemm(A, W, 2, bias=bias, variant=variant)
    assert_close(out, gelu_erf(ref + bias.float()), 2e-2, 1e-2, "gemm gelu_erf")
    resid = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    out = eng.gemm(A, W, 4, resid=resid, variant=variant)
    assert_close(out, ref + resid, 2e-3, 1e-4, "gemm f32+resid")
    # in place on the residual stream (how the engine uses it)
    buf = resid.clone()
    eng.gemm(A, W, 4, bias=bias, resid=buf, out=buf, variant=variant)
    assert_close(buf, ref + bias.float() + resid, 2e-3, 1e-4, "gemm f32+resid in place")


@pytest.mark.parametrize("M,N,K,epi", [(5000, 768, 256, 0), (70000, 512, 128, 4), (33000, 1024, 64, 3), (9000, 640, 192, 1),
                                       (40000, 512, 320, 0), (33000, 1024, 256, 1), (70000, 2048, 512, 3),
                                       (20000, 4096, 8192, 0)])
def test_gemm_persistent_many_tiles(eng, M, N, K, epi):
    """More tiles than workgroups: the persistent kernel's cross-tile pipeline (prefetch of the next tile's first
    K-tile, counted vmcnt behind the epilogue stores, edge tiles in the middle of a run) against variant 0, bitwise.
    The last three shapes (>= 256 tiles, K >= 256, N <= 2048) run the L2-touch form of the lock-step kernel, whose
    prefetch crosses tile boundaries too; the last one is the XXL wo rule (N = 4096, K >= 8192)."""
    A = randn_bf16(M, K, seed=31)
    W = randn_bf16(N, K, seed=32, scale=K ** -0.5)
    bias = randn_bf16(N, seed=33)
    resid = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(34)) if epi == 4 else None
    ref = eng.gemm(A, W, epi, bias=bias, resid=resid, variant=0)
    for variant in (3, 5, 11):         # 3 = the quad form for the bf16-result epilogues (16x16x32 MFMAs: the same bits), 11 = the 8-wave rule
        for _ in range(3):
            out = eng.gemm(A, W, epi, bias=bias, resid=resid, variant=variant)
            assert torch.equal(out, ref), describe(out, ref)


def describe(out, ref):
    from tests.gpu_util import describe_mismatch
    return describe_mismatch(out, ref, 0.0, 0.0, "persistent vs variant 0")


@pytest.mark.parametrize("M,N,K,epi,S,H", [(16384 - 40, 8192, 128, 0, 0, 0), (16384, 8192, 192, 5, 0, 0), (608 * 27, 3 * 2048 + 2048, 128, 0, 0, 0),
                                           (608 * 28, 3 * 2048, 128, 6, 608, 32), (40000, 512, 320, 0, 0, 0)])
def test_gemm_tile_order_is_bitwise_neutral(eng, M, N, K, epi, S, H):
    """The workgroup -> tile order (gm M-tiles per group, N cut into ns column ranges walked one after the other) permutes
    which workgroup computes a tile when and nothing else: every order gives the bits of the default order and of the
    one-tile-per-workgroup kernel -- persistent lock-step (incl. its L2-touch form on the last shape), ping-pong and
    variant 0, full and ragged edges, plain / gated / head-major epilogues."""
    A = randn_bf16(M, K, seed=61)
    W = randn_bf16(N, K, seed=62, scale=K ** -0.5)
    ref = eng.gemm(A, W, epi, S=S, H=H, variant=0)
    for variant in (3, 5, 11, 0):
        for order in ((8, 1), (4, 1), (2, 1), (1, 1), (16, 1), (3, 1), (8, 2), (4, 2), (2, 4), (64, 4)):
            out = eng.gemm(A, W, epi, S=S, H=H, variant=variant, tile_order=order)
            assert torch.equal(out, ref), (variant, order, describe(out, ref))
        for nt, touch in ((True, 0), (False, 1), (False, 2), (True, 1)):      # the other two cache-policy hints of a launch
            out = eng.gemm(A, W, epi, S=S, H=H, variant=variant, nt_store=nt, l2_touch=touch)
            assert torch.equal(out, ref), (variant, nt, touch, describe(out, ref))


@pytest.mark.parametrize("M,N,K,epi,S,H", [
    (33000, 2048, 64, 0, 0, 0),            # one K-tile per output tile: every K-tile is a tile's first AND last
    (33000 - 7, 2048 - 8, 128, 0, 0, 0),   # ragged M and N edges in the middle of a workgroup's run
    (20000, 4096, 1024, 1, 0, 0),          # quick_gelu + bias (ViT fc1)
    (147456 // 4, 2048, 1024, 2, 0, 0),    # erf-GELU + bias (projector)
    (16384, 8192, 256, 5, 0, 0),           # gated gelu_new over interleaved wi_0 / wi_1 blocks
    (608 * 40, 3 * 1024, 192, 6, 608, 16), # head-major scatter, sample boundaries inside tiles
    (577 * 64, 3 * 1024, 128, 6, 577, 16), # ... with the ViT's odd sequence length
    (131072, 512, 2048, 0, 0, 0)])         # long K, many tiles per workgroup
def test_gemm_quad_form_is_bitwise_the_other_forms(eng, M, N, K, epi, S, H):
    """gemm_bf16_quad (the library's default for bf16-result launches, variant 3): 256x256x64 tiles computed by four waves of
    128x128 on v_mfma_f32_16x16x32_bf16 with the accumulators in hand-named AGPRs, a table-scheduled K loop and the
    stream two K-tiles ahead through range-checked descriptors.  A 16x16x32 MFMA adds a step's 32 products in the same order
    as two 32x32x16 MFMAs do: bit for bit the one-tile-per-workgroup kernel (variant 0) and the 8-wave persistent kernels
    (variant 11), on every epilogue it carries, with ragged edges (rows beyond M / N come back as zeros from the descriptor's
    range check instead of clamped copies), repeated (no dependence on what the previous launch left in LDS / registers)."""
    A = randn_bf16(M, K, seed=71)
    W = randn_bf16(N, K, seed=72, scale=K ** -0.5)
    bias = randn_bf16(N, seed=73) if epi in (0, 1, 2, 6) else None
    ref = eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=0)
    assert torch.equal(eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=11), ref)
    for _ in range(3):
        out = eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=3)
        assert torch.equal(out, ref), describe(out, ref)
    if epi in (0, 1, 2):                   # and against fp32 torch, so that the three forms cannot be wrong together
        full = A.float() @ W.float().t() + bias.float()
        full = quick_gelu(full) if epi == 1 else (gelu_erf(full) if epi == 2 else full)
        assert_close(out, full, 2e-2, 1e-2, "quad form vs fp32")


@pytest.mark.parametrize("M,N,K,epi,S,H,ftype,with_bias", [
    (608, 4096, 4096, 0, 0, 0, 2, False),      # T5-XXL encoder o of one pair (fp16 operands, bf16 delta)
    (608, 4096, 10240, 0, 0, 0, 0, False),     # ... wo (bf16 operands, K = 10 240)
    (1216, 2048, 1024, 0, 0, 0, 0, True),      # two pairs at T5-XL's width; plain + bias
    (608, 3 * 1024, 1024, 6, 608, 16, 1, False),   # q|k|v of one pair: head-major scatter, fp16 in / out
    (2 * 608 - 600, 3 * 1024, 320, 6, 8, 16, 0, True),   # S = 8: sample boundaries inside a row block; ragged M; bf16
    (577, 3 * 1024, 1024, 6, 577, 16, 1, True),    # the tower's q|k|v of one image (odd sequence length, bias)
    (577, 4096, 1024, 1, 0, 0, 1, True),       # the tower's fc1: quick-GELU + bias, fp16
    (577, 1024, 4096, 0, 0, 0, 1, True),       # ... fc2
    (576, 4096, 1024, 2, 0, 0, 1, True),       # the projector's first GEMM: erf-GELU + bias
    (300, 2048, 512, 5, 0, 0, 2, False),       # gated gelu_new over interleaved wi_0 | wi_1 blocks, fp16 operands -> bf16
    (130, 1024, 256, 5, 0, 0, 0, False),       # ... bf16 operands, one full and one 2-row block
    (1, 1024, 256, 0, 0, 0, 0, False), (129, 128, 2048, 0, 0, 0, 2, False)])
def test_slim_form_is_bitwise_the_quad_form(eng, M, N, K, epi, S, H, ftype, with_bias):
    """gemm_slim.inc (round 6): the few-row launches of a quad call site -- 128 x 128 tiles, four waves of 64 x 64 on the quad form's MFMA, a five-stage
    LDS ring -- against the quad kernel on the same launch (variant 10: the quad kernel whatever the shape): torch.equal on every epilogue and operand type
    the path carries, ragged M, sample boundaries inside row blocks, rows beyond M untouched, repeated.  The launcher's own host function says which launches
    take the form (13)."""
    from t2v_metrics_amd.engine import load_library
    lib = load_library()
    inner = N // 3 if epi == 6 else 0
    assert lib.vqs_debug_gemm_form(M, N, K, K, K, epi, 1, 3, S, inner, 0) == 13, "this shape must take the slim form"
    assert lib.vqs_debug_gemm_form(M, N, K, K, K, epi, 1, 10, S, inner, 0) == 10
    g = torch.Generator(device="cuda").manual_seed(163)
    dt = torch.float16 if ftype else torch.bfloat16
    A = torch.randn(M, K, device="cuda", generator=g).to(dt)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    bias = randn_bf16(N, seed=173) if with_bias else None
    quad = eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=10, ftype=ftype)
    for _ in range(2):
        got = torch.full_like(quad, 3.0)
        eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=3, ftype=ftype, out=got)
        assert torch.equal(got.view(torch.int16), quad.view(torch.int16)), describe(got, quad)
    if epi in (0, 1, 2):                   # and against fp32 torch, so that the two forms cannot be wrong together
        full = A.float() @ W.float().t() + (bias.float() if with_bias else 0.0)
        full = quick_gelu(full) if epi == 1 else (gelu_erf(full) if epi == 2 else full)
        assert_close(got, full, 2e-2, 1e-2, "slim form vs fp32")
    if epi == 0:                           # rows past M are not touched
        pad = torch.full((M + 16, N), 5.0, dtype=quad.dtype, device="cuda")
        eng.gemm(A, W, epi, bias=bias, variant=3, ftype=ftype, out=pad[:M])
        assert torch.equal(pad[:M], quad) and float((pad[M:].float() - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize("Z,M,N,K,epi", [(256, 128, 608, 256, 3), (256, 128, 512, 640, 0), (300, 64, 96, 64, 3), (200, 100, 264, 192, 0),
                                         (1, 128, 24576 + 132, 128, 3), (1500, 33, 40, 320, 0), (40, 128, 4096, 128, 0),
                                         # round 6 (VQS_STREAM_MIN_ITEMS 192 -> 4): the decoder's linears at B = 1 .. 32 -- 4 K-slices of a 4 096-wide weight with 4 stacked
                                         # rows (128 items), q|k|v (96 items), the cross scores of ONE pair (5 items), 128 stacked rows of B = 32
                                         (4, 4, 4096, 1024, 3), (1, 4, 12288, 512, 3), (1, 128, 608, 1024, 3), (4, 128, 4096, 640, 3), (2, 16, 520, 192, 0)])
def test_gemm_stream_form_is_bitwise_the_persistent_kernel(eng, Z, M, N, K, epi):
    """gemm_stream.inc (<= 128 rows per batch entry, W streamed through a five-stage LDS ring) against the 256-row persistent kernel
    on the same batched launch (option no_stream) and against one-tile-per-workgroup launches entry by entry: torch.equal -- same
    MFMA, operand roles, k order.  Ragged M / N, a single K slab, fewer slabs than ring stages, the split-bf16 store (hi + lo of the
    fp32 accumulator), a result pitch wider than N; every launch's form is checked through the launcher's own host function."""
    from t2v_metrics_amd.engine import load_library
    g = torch.Generator(device="cuda").manual_seed(97)
    A = torch.randn(Z, M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(Z, N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    lib = load_library()
    assert lib.vqs_debug_gemm_form(M, N, K, K, K, epi, Z, 3, 0, 0, 0) == 12, "this shape must take the stream form"
    ldc = ((N + 63) // 64) * 64
    split = epi == 0
    got = eng.gemm_batched(A, W, epi, split=split, ldc=ldc)
    ref = eng.gemm_batched(A, W, epi, split=split, ldc=ldc, no_stream=True)
    for a, b in zip(got if split else (got,), ref if split else (ref,)):
        assert torch.equal(a[..., :N], b[..., :N]), f"stream vs persistent {Z}x{M}x{N}x{K} epi {epi}: {(a[..., :N].float() - b[..., :N].float()).abs().max().item()}"
        assert float(a[..., N:].abs().max()) == 0.0 if ldc > N else True          # nothing written beyond N
    hi = got[0] if split else got
    for z in (0, Z // 2, Z - 1):
        one = eng.gemm(A[z], W[z], epi, variant=0)
        assert torch.equal(hi[z, :, :N], one), f"entry {z} vs the one-tile kernel"
    if split:                                                                    # hi + lo carries 16 bits of the fp32 accumulator
        acc = eng.gemm_batched(A, W, 3, ldc=ldc)
        two = got[0].float() + got[1].float()
        rel = ((two - acc)[..., :N].abs() / acc[..., :N].abs().clamp(min=1e-3)).max().item()
        assert rel <= 2.0 ** -15, rel


@pytest.mark.parametrize("M,N,K", [(300, 512, 128), (1000, 768, 256), (70000, 512, 64)])
@pytest.mark.parametrize("variant", [3, 5])
def test_gemm_fused_residual_rmsnorm(eng, M, N, K, variant):
    """Producer epilogue (hres += A.W^T in place, xhat = hres*ln_w, per-tile row sums of squares) and consumer row scale
    against fp32 torch: together they must equal  rmsnorm(h + A.W^T) @ W2^T  (HF modeling_t5.py:59-72,140)."""
    g = torch.Generator(device="cuda").manual_seed(41)
    A = randn_bf16(M, K, seed=42)
    W = randn_bf16(N, K, seed=43, scale=K ** -0.5)
    lnw = (1.0 + 0.25 * torch.randn(N, device="cuda", generator=g)).to(torch.bfloat16)
    h0 = torch.randn(M, N, device="cuda", generator=g) * 2.0
    h_ref = h0 + A.float() @ W.float().t()
    h = h0.clone()
    xhat, rowss = eng.gemm_resid_rms(A, W, h, lnw, variant=variant)
    assert_close(h, h_ref, 1e-3, 1e-4, "residual stream update")
    assert_close(xhat, h_ref * lnw.float(), 1e-2, 1e-2, "un-normalised norm operand")
    assert_close(rowss.sum(0), h_ref.pow(2).sum(-1), 1e-2, 1e-3, "row sums of squares")
    for _ in range(2):            # deterministic (no atomics): bitwise repeatable
        h2 = h0.clone()
        x2, r2 = eng.gemm_resid_rms(A, W, h2, lnw, variant=variant)
        assert torch.equal(x2, xhat) and torch.equal(r2, rowss) and torch.equal(h2, h)
    # consumer: plain bf16 and gated epilogues
    N2 = 256
    W2 = randn_bf16(N2, N, seed=44, scale=N ** -0.5)
    rs = torch.rsqrt(h_ref.pow(2).mean(-1, keepdim=True) + 1e-6)
    ref = (xhat.float() * rs) @ W2.float().t()
    out = eng.gemm_rowscaled(xhat, W2, 0, rowss, N, 1e-6, variant=variant)
    assert_close(out, ref, 2e-2, 1e-2, "row-scaled consumer (bf16)")
    out = eng.gemm_rowscaled(xhat, W2, 3, rowss, N, 1e-6, variant=variant)
    assert_close(out, ref, 2e-3, 1e-3, "row-scaled consumer (fp32)")


def test_gemm_transpose_detecting(eng):
    """A = I-like structure with an asymmetric W catches swapped C layouts."""
    M = N = K = 256
    A = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    A[torch.arange(M), torch.arange(K)] = 1.0
    W = (torch.arange(N, device="cuda")[:, None] * 0.25 + torch.arange(K, device="cuda")[None, :] * 0.001953125)
    W = W.to(torch.bfloat16)
    out = eng.gemm(A, W, 3)
    assert_close(out, W.float().t(), 1e-6, 0, "gemm identity")


@pytest.mark.parametrize("M,F,K", [(256, 128, 64), (300, 192, 128), (1029, 640, 256)])
def test_gemm_gated(eng, M, F, K):
    A = randn_bf16(M, K, seed=5)
    w0 = randn_bf16(F, K, seed=6, scale=K ** -0.5)
    w1 = randn_bf16(F, K, seed=7, scale=K ** -0.5)
    W = interleave_gate(w0, w1)
    ref = gelu_new(A.float() @ w0.float().t()) * (A.float() @ w1.float().t())
    for variant in (0, 2, 3, 5):
        out = eng.gemm(A, W, 5, variant=variant)
        assert_close(out, ref, 2e-2, 1e-2, f"gemm gated {M}x{F}x{K} v{variant}")


@pytest.mark.parametrize("B,S,H,K,nsel", [(2, 17, 2, 128, 3), (3, 100, 4, 256, 3), (2, 577, 2, 128, 2)])
def test_gemm_heads(eng, B, S, H, K, nsel):
    M, I = B * S, H * 64
    A = randn_bf16(M, K, seed=8)
    W = randn_bf16(nsel * I, K, seed=9, scale=K ** -0.5)
    bias = randn_bf16(nsel * I, seed=10)
    ref = (A.float() @ W.float().t() + bias.float()).reshape(B, S, nsel, H, 64).permute(2, 0, 3, 1, 4)
    for variant in (0, 2, 3, 5):
        out = eng.gemm(A, W, 6, bias=bias, S=S, H=H, variant=variant)
        assert_close(out, ref, 2e-2, 1e-2, f"gemm heads v{variant}")


@pytest.mark.parametrize("B,H,S,use_bias,ragged", [(1, 1, 64, False, False), (2, 2, 100, False, False),
                                                   (2, 3, 577, False, False), (2, 2, 128, True, False),
                                                   (3, 2, 200, True, True), (2, 4, 608, True, True)])
def test_attention(eng, B, H, S, use_bias, ragged):
    q = randn_bf16(B, H, S, 64, seed=11, scale=0.5 if use_bias else 1.0)
    k = randn_bf16(B, H, S, 64, seed=12, scale=0.5 if use_bias else 1.0)
    v = randn_bf16(B, H, S, 64, seed=13)
    scale = 1.0 if use_bias else 0.125
    table = bias = key_len = None
    if use_bias:
        table = torch.randn(H, 2 * S - 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(14))
        idx = (torch.arange(S)[None, :] - torch.arange(S)[:, None] + S - 1).to("cuda")     # key - query + S-1
        bias = table[:, idx]
    if ragged:
        key_len = torch.tensor([S, max(1, S // 3), S - 1][:B] + [S] * max(0, B - 3), dtype=torch.int32, device="cuda")
    ref = attention_ref(q, k, v, scale, bias, key_len)
    out = eng.attention(q, k, v, scale, bias_table=table, key_len=key_len)
    assert_close(out, ref, 2e-2, 2e-2, f"attention B{B} H{H} S{S} bias={use_bias} ragged={ragged}")


def test_attention_spiked_max(eng):
    """Forces the running-max rescale: one key dominates late in the sequence."""
    B, H, S = 1, 1, 256
    q = randn_bf16(B, H, S, 64, seed=15, scale=0.3)
    k = randn_bf16(B, H, S, 64, seed=16, scale=0.3)
    v = randn_bf16(B, H, S, 64, seed=17)
    k[0, 0, 200] = q[0, 0, 5] * 8.0
    ref = attention_ref(q, k, v, 1.0)
    out = eng.attention(q, k, v, 1.0)
    assert_close(out, ref, 2e-2, 2e-2, "attention spiked")


@pytest.mark.parametrize("B,H,S,hd,half", [(2, 3, 37, 128, 64), (1, 4, 64, 128, 40), (3, 2, 5, 128, 64)])
def test_rope_rotate_half(eng, B, H, S, hd, half):
    """rotate-half RoPE in place (HF modeling_qwen2_5_vl.py:153-172): fp32 math, bf16 storage; padded dims untouched."""
    g = torch.Generator(device="cuda").manual_seed(71)
    x = torch.randn(B, H, S, hd, device="cuda", generator=g).to(torch.bfloat16)
    ang = torch.rand(B * S, half, device="cuda", generator=g) * 20.0
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    xf = x.float()
    c, s_ = cos.view(B, 1, S, half), sin.view(B, 1, S, half)
    ref = xf.clone()
    ref[..., :half] = xf[..., :half] * c - xf[..., half:2 * half] * s_
    ref[..., half:2 * half] = xf[..., half:2 * half] * c + xf[..., :half] * s_
    out = eng.rope_(x.clone(), cos, sin)
    assert_close(out, ref, 1e-2, 1e-2, "rope")
    assert torch.equal(out[..., 2 * half:], x[..., 2 * half:])


@pytest.mark.parametrize("B,H,Hkv,S,causal,ragged", [(2, 4, 4, 64, False, False), (3, 4, 2, 200, True, False),
                                                     (2, 7, 1, 333, True, True), (4, 2, 2, 768, False, False),
                                                     (1, 28, 4, 808, True, False)])
def test_attention_hd128_gqa_causal(eng, B, H, Hkv, S, causal, ragged):
    """head_dim 128, grouped-query heads, causal / key-padding masks vs fp32 torch (HF eager_attention_forward,
    models/qwen2_5_vl/modeling_qwen2_5_vl.py:187-208).  The 80-wide tower heads run through the same kernel zero-padded."""
    g = torch.Generator(device="cuda").manual_seed(61)
    q = (torch.randn(B, H, S, 128, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(B, Hkv, S, 128, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(B, Hkv, S, 128, device="cuda", generator=g).to(torch.bfloat16)
    if B >= 4:
        q[..., 80:] = 0; k[..., 80:] = 0; v[..., 80:] = 0            # the padded vision-head case
    kl = None
    if ragged:
        kl = torch.tensor([S, max(1, S // 3)][:B] + [S] * max(0, B - 2), dtype=torch.int32, device="cuda")
    scale = 128 ** -0.5 if B < 4 else 80 ** -0.5
    out = eng.attention_hd(q, k, v, scale, causal=causal, key_len=kl).float().view(B, S, H, 128)
    rep = H // Hkv
    kf, vf = k.float().repeat_interleave(rep, 1), v.float().repeat_interleave(rep, 1)
    s = torch.einsum("bhqd,bhkd->bhqk", q.float(), kf) * scale
    mask = torch.ones(S, S, dtype=torch.bool, device="cuda").tril() if causal else torch.ones(S, S, dtype=torch.bool, device="cuda")
    mask = mask[None, None].expand(B, 1, S, S).clone()
    if kl is not None:
        mask &= (torch.arange(S, device="cuda")[None, None, None, :] < kl[:, None, None, None])
    s = s.masked_fill(~mask, float("-inf"))
    ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(s, -1), vf)
    valid = torch.ones(B, S, dtype=torch.bool, device="cuda")
    if kl is not None and not causal:
        pass
    if kl is not None:
        valid = torch.arange(S, device="cuda")[None, :] < kl[:, None]   # padded queries are don't-care rows
    assert_close(out[valid], ref[valid], 2e-2, 2e-2, f"attention hd128 B{B} H{H}/{Hkv} S{S} causal={causal}")


@pytest.mark.parametrize("T", [1, 2, 5])
def test_decoder_attention(eng, T):
    B, H, S = 3, 2, 77
    I = H * 64
    qkv = randn_bf16(B * T, 3 * I, seed=18, scale=0.5)
    table = torch.randn(H, T, device="cuda", generator=torch.Generator(device="cuda").manual_seed(19))
    out = eng.decoder_attention(qkv, qkv[:, I:], qkv[:, 2 * I:], B, H, T, T, 3 * I, 3 * I, False, bias_table=table)
    x = qkv.float().reshape(B, T, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    dist = torch.arange(T)[:, None] - torch.arange(T)[None, :]
    bias = table[:, dist.clamp(min=0).to("cuda")]
    s = s + bias[None]
    s = s.masked_fill((dist < 0).to("cuda")[None, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * T, I)
    assert_close(out, ref, 2e-2, 2e-2, f"decoder self attention T{T}")

    qc = randn_bf16(B * T, I, seed=20, scale=0.5)
    kc = randn_bf16(B, H, S, 64, seed=21, scale=0.5)
    vc = randn_bf16(B, H, S, 64, seed=22)
    key_len = torch.tensor([S, 10, 50], dtype=torch.int32, device="cuda")
    out = eng.decoder_attention(qc, kc, vc, B, H, T, S, I, 0, True, key_len=key_len)
    qh = qc.float().reshape(B, T, H, 64).transpose(1, 2)
    s = qh @ kc.float().transpose(-1, -2)
    mask = torch.arange(S, device="cuda")[None, :] >= key_len[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vc.float()).transpose(1, 2).reshape(B * T, I)
    assert_close(out, ref, 2e-2, 2e-2, f"decoder cross attention T{T}")


@pytest.mark.parametrize("M,D", [(7, 128), (1000, 2048), (33, 4096), (5, 1024), (301, 1280), (77, 3584)])
def test_norms(eng, M, D):
    x = torch.randn(M, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(23)) * 3.0 + 0.5
    w = randn_bf16(D, seed=24)
    b = randn_bf16(D, seed=25)
    ref = w.float() * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert_close(eng.rmsnorm(x, w, 1e-6), ref, 1e-2, 1e-2, "rmsnorm")
    ref = torch.nn.functional.layer_norm(x, (D,), w.float(), b.float(), 1e-5)
    assert_close(eng.layernorm(x, w, b, 1e-5), ref, 1e-2, 1e-2, "layernorm bf16")
    assert_close(eng.layernorm(x, w, b, 1e-5, out_f32=True), ref, 1e-4, 1e-5, "layernorm f32")
    # fused residual update: x += delta (in place) then normalise
    delta = torch.randn(M, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(26)).to(torch.bfloat16)
    xs = x + delta.float()
    x1 = x.clone()
    out = eng.rmsnorm(x1, w, 1e-6, delta=delta)
    assert_close(x1, xs, 1e-6, 1e-6, "rmsnorm residual write-back")
    assert_close(out, w.float() * (xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-6)), 1e-2, 1e-2, "rmsnorm+add")
    x2 = x.clone()
    out = eng.layernorm(x2, w, b, 1e-5, delta=delta)
    assert_close(x2, xs, 1e-6, 1e-6, "layernorm residual write-back")
    assert_close(out, torch.nn.functional.layer_norm(xs, (D,), w.float(), b.float(), 1e-5), 1e-2, 1e-2, "layernorm+add")


@pytest.mark.parametrize("M,D", [(9, 96), (257, 1024), (130, 2048), (33, 4096), (65, 1280), (40, 3584)])
@pytest.mark.parametrize("kind", ["rms", "layer"])
def test_norm_deferred_store_is_bitwise_the_stored_form(eng, M, D, kind):
    """vqs_norm_deferred: a layer's two norms as (x + d1, not stored) then (x = (x + d1) + d2, stored) must reproduce --
    bit for bit, stream and outputs -- the two stored single-delta updates they replace (HF modeling_t5.py:140,400;
    modeling_clip.py:366,371), and the not-stored form must leave the stream untouched."""
    g = torch.Generator(device="cuda").manual_seed(91)
    x0 = torch.randn(M, D, device="cuda", generator=g) * 3.0 + 0.25
    d1 = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
    d2 = (torch.randn(M, D, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = randn_bf16(D, seed=92)
    b = randn_bf16(D, seed=93) if kind == "layer" else None
    eps = 1e-5 if kind == "layer" else 1e-6

    def stored(x, delta):
        return eng.layernorm(x, w, b, eps, delta=delta) if kind == "layer" else eng.rmsnorm(x, w, eps, delta=delta)

    xa = x0.clone()
    out_a1 = stored(xa, d1)
    mid = xa.clone()
    out_a2 = stored(xa, d2)
    xb = x0.clone()
    out_b1 = eng.norm_deferred(xb, w, eps, d1, store_x=False, b=b)
    assert torch.equal(xb, x0), "the not-stored form modified the stream"
    assert torch.equal(out_b1, out_a1)
    out_b2 = eng.norm_deferred(xb, w, eps, d1, delta2=d2, b=b)
    assert torch.equal(xb, xa) and torch.equal(out_b2, out_a2)
    assert_close(mid, x0 + d1.float(), 1e-6, 1e-6, "first stored update")
    assert_close(xa, (x0 + d1.float()) + d2.float(), 1e-6, 1e-6, "second stored update")
    with pytest.raises(eng.VqsError):           # two deltas cannot be combined with "do not store"
        eng.norm_deferred(xb, w, eps, d1, delta2=d2, store_x=False, b=b)


def test_normalize_u8_matches_the_host_processor_arithmetic(eng):
    """vqs_normalize_u8 == ((x * 1/255) - mean) / std in fp32, rounded to bf16: bit-exact against the host path
    (t2v_metrics_amd/preprocess.py, itself pinned to HF's CLIPImageProcessor in tests/golden/clip_preprocess.npz)."""
    from t2v_metrics_amd.preprocess import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
    g = torch.Generator().manual_seed(81)
    x = torch.randint(0, 256, (3, 37, 53, 3), generator=g, dtype=torch.uint8)
    out = eng.normalize_u8(x.cuda(), OPENAI_CLIP_MEAN, OPENAI_CLIP_STD).cpu()
    arr = x.numpy().astype(np.float32) * np.float32(1.0 / 255.0)
    ref = (arr - np.asarray(OPENAI_CLIP_MEAN, dtype=np.float32)) / np.asarray(OPENAI_CLIP_STD, dtype=np.float32)
    ref = torch.from_numpy(ref).permute(0, 3, 1, 2).contiguous().to(torch.bfloat16)
    assert torch.equal(out, ref)


def test_score_head(eng):
    B, T, V = 5, 3, 1000
    logits = torch.randn(B, T, V, device="cuda", generator=torch.Generator(device="cuda").manual_seed(26)) * 4.0
    labels = torch.randint(0, V, (B, T), generator=torch.Generator().manual_seed(27))
    labels[1, 2] = -100
    labels[3, 1:] = -100
    lp, sc = eng.score_head(logits, labels.cuda())
    ref = torch.log_softmax(logits, -1).gather(-1, labels.clamp(min=0).cuda()[..., None])[..., 0]
    valid = (labels != -100).cuda()
    ref = torch.where(valid, ref, torch.zeros_like(ref))
    assert_close(lp, ref, 1e-4, 1e-5, "label logprobs")
    ref_sc = torch.exp((ref * valid).sum(-1) / valid.sum(-1))
    assert_close(sc, ref_sc, 1e-6, 1e-4, "scores")


def test_relpos_bucket_host_function_matches_golden(eng, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "relpos_buckets.npz"))
    for r, vb, vc in zip(g["relative_position"], g["bidirectional"], g["causal"]):
        assert eng.relpos_bucket(int(r), True) == int(vb)
        assert eng.relpos_bucket(int(r), False) == int(vc)


# ---------------------------------------------------------------------------------------------------------------------------
# fp16 instantiations (options vit_fp16 / enc_fp16): the same kernels on IEEE fp16 fragments.  In whole passes they only ever see the
# tower's / the encoder's own shapes; here: shapes of their own, edge tiles, ragged key lengths, every epilogue they carry
# (VERDICT r4 item 5).  Reference = fp32 torch on the SAME fp16 inputs; the result is held to fp16 (ftype 1) or bf16 (ftype 2) rounding.
# ---------------------------------------------------------------------------------------------------------------------------
def _randn_f16(*shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(torch.float16)


@pytest.mark.parametrize("M,N,K,epi,S,H", [
    (300, 264, 192, 0, 0, 0),               # ragged M and N edges inside one tile
    (2065, 1024, 1024, 0, 0, 0),
    (33000 - 7, 2048 - 8, 128, 0, 0, 0),    # many tiles per workgroup, ragged edges in the middle of a run
    (9000, 640, 192, 1, 0, 0),              # quick_gelu + bias
    (5000, 768, 256, 2, 0, 0),              # erf-GELU + bias
    (608 * 12, 3 * 1024, 192, 6, 608, 16),  # head-major scatter, sample boundaries inside tiles (T5 encoder: no bias)
    (577 * 9, 3 * 1024, 128, 6, 577, 16),   # ... with the ViT's odd sequence length
    (131072, 512, 2048, 0, 0, 0)])          # long K
def test_gemm_fp16_operands_fp16_result(eng, M, N, K, epi, S, H):
    A = _randn_f16(M, K, seed=81)
    W = _randn_f16(N, K, seed=82, scale=K ** -0.5)
    bias = randn_bf16(N, seed=83) if epi != 6 else None
    out = eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=3, ftype=1)
    assert out.dtype == torch.float16
    assert torch.equal(out, eng.gemm(A, W, epi, bias=bias, S=S, H=H, variant=3, ftype=1))          # repeatable
    ref = A.float() @ W.float().t() + (bias.float() if bias is not None else 0.0)
    ref = quick_gelu(ref) if epi == 1 else (gelu_erf(ref) if epi == 2 else ref)
    if epi == 6:
        ref = ref.reshape(M // S, S, N // (H * 64), H, 64).permute(2, 0, 3, 1, 4)                   # [q|k|v, B, H, S, 64]
    # an fp16 result: 2^-11 relative rounding + fp32 summation order; three bits tighter than the bf16 kernels' bound
    assert_close(out, ref, 2.5e-3, 1.5e-3, f"gemm f16/f16 epi {epi} {M}x{N}x{K}")
    # the bf16 kernel on the same VALUES rounded to bf16 cannot be this close: the test would pass on a silently-bf16 launch otherwise
    if epi == 0 and K >= 1024:
        coarse = eng.gemm(A.to(torch.bfloat16), W.to(torch.bfloat16), 0, bias=bias, variant=3)
        assert (coarse.float() - ref).abs().mean() > 3 * (out.float() - ref).abs().mean()


@pytest.mark.parametrize("M,N,K,epi", [(300, 264, 192, 0), (33000 - 7, 2048 - 8, 128, 0), (20000, 2048, 1024, 0),
                                       (16384, 8192, 256, 5), (9000 - 3, 1024, 2048, 5), (608 * 9, 5120 * 2, 2048, 5)])
def test_gemm_fp16_operands_bf16_result(eng, M, N, K, epi):
    """gemm_f16b_quad: fp16 operands, bf16 result -- the T5 encoder's o projection (plain) and gated wi of option enc_fp16."""
    A = _randn_f16(M, K, seed=84)
    if epi == 5:
        F = N // 2
        w0, w1 = _randn_f16(F, K, seed=85, scale=K ** -0.5), _randn_f16(F, K, seed=86, scale=K ** -0.5)
        W = interleave_gate(w0, w1)
        ref = gelu_new(A.float() @ w0.float().t()) * (A.float() @ w1.float().t())
    else:
        W = _randn_f16(N, K, seed=85, scale=K ** -0.5)
        ref = A.float() @ W.float().t()
    out = eng.gemm(A, W, epi, variant=3, ftype=2)
    assert out.dtype == torch.bfloat16 and torch.equal(out, eng.gemm(A, W, epi, variant=3, ftype=2))
    assert_close(out, ref, 2e-2, 1e-2, f"gemm f16/bf16 epi {epi} {M}x{N}x{K}")
    # exactly the bf16 rounding of the fp16-operand product: against the fp16-result kernel on the same operands (same MFMAs, same k
    # order, another pack) the plain epilogue must agree to the last bit after rounding that result's fp32 ... it cannot be read back,
    # so instead: the result must be within ONE bf16 ulp of the fp32 reference wherever the reference is not at a rounding boundary
    if epi == 0:
        err = (out.float() - ref).abs()
        ulp = torch.ldexp(torch.ones_like(ref), torch.frexp(ref).exponent - 8)
        assert float((err > 0.51 * ulp + 1e-6 * ref.abs().max()).float().mean()) < 2e-3


def test_fp16_gemm_refuses_what_it_has_no_kernel_for(eng):
    A, W = _randn_f16(512, 256, seed=87), _randn_f16(512, 256, seed=88)
    from t2v_metrics_amd.engine import VqsError
    with pytest.raises(VqsError):
        eng.gemm(A, W, 5, variant=3, ftype=1)          # no fp16 tensor leaves a gated FFN
    with pytest.raises(VqsError):
        eng.gemm(A, W, 1, variant=3, ftype=2)          # fp16 in / bf16 out: plain and gated only
    with pytest.raises(VqsError):
        eng.gemm(A, W, 0, variant=11, ftype=1)         # the 8-wave forms have no fp16 instantiation
    # K < 128 is not a quad launch: since option dec_fp16 it runs the persistent 8-wave kernel's fp16 instantiation (plain epilogues only)
    a64, w64 = A[:, :64].contiguous(), W[:, :64].contiguous()
    assert_close(eng.gemm(a64, w64, 0, variant=3, ftype=1), a64.float() @ w64.float().t(), 2.5e-3, 1.5e-3, "K = 64 on the fp16 persistent kernel")
    with pytest.raises(VqsError):
        eng.gemm(a64, w64, 1, variant=3, ftype=1)      # ... which carries no activation epilogue


@pytest.mark.parametrize("B,H,S,use_bias,ragged", [(2, 3, 577, False, False), (3, 2, 608, True, True), (2, 4, 333, True, False),
                                                   (3, 2, 200, False, True), (1, 2, 64, True, False), (4, 1, 648, True, True)])
def test_attention_fp16(eng, B, H, S, use_bias, ragged):
    """attn_fwd_dma_f16_kernel<false> (vision tower) and <true> (T5 encoder of option enc_fp16: position-bias table + key mask)."""
    q = _randn_f16(B, H, S, 64, seed=91, scale=0.5 if use_bias else 1.0)
    k = _randn_f16(B, H, S, 64, seed=92, scale=0.5 if use_bias else 1.0)
    v = _randn_f16(B, H, S, 64, seed=93)
    scale = 1.0 if use_bias else 0.125
    table = bias = key_len = None
    if use_bias:
        table = torch.randn(H, 2 * S - 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(94))
        idx = (torch.arange(S)[None, :] - torch.arange(S)[:, None] + S - 1).to("cuda")
        bias = table[:, idx]
    if ragged:
        key_len = torch.tensor([S, max(1, S // 3), S - 1][:B] + [S] * max(0, B - 3), dtype=torch.int32, device="cuda")
    ref = attention_ref(q, k, v, scale, bias, key_len)
    out = eng.attention(q, k, v, scale, bias_table=table, key_len=key_len)
    assert out.dtype == torch.float16 and torch.equal(out, eng.attention(q, k, v, scale, bias_table=table, key_len=key_len))
    assert_close(out, ref, 4e-3, 3e-3, f"attention fp16 B{B} H{H} S{S} bias={use_bias} ragged={ragged}")   # P is an fp16 tensor: 8 x below the bf16 kernel's bound
    coarse = eng.attention(q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16), scale, bias_table=table, key_len=key_len)
    assert (coarse.float() - ref).abs().mean() > 2.5 * (out.float() - ref).abs().mean()


@pytest.mark.parametrize("M,D", [(9, 96), (257, 1024), (130, 2048), (33, 4096)])
def test_norms_with_fp16_operand_out(eng, M, D):
    """RMSNorm with bf16 deltas in and an fp16 operand out (T5 encoder, option enc_fp16): every add mode, against fp32 torch and -- the
    stream -- bit for bit against the bf16-output kernel (same additions, only the output's pack differs); LayerNorm with fp16 deltas
    (vision tower)."""
    g = torch.Generator(device="cuda").manual_seed(95)
    x = torch.randn(M, D, device="cuda", generator=g) * 3.0 + 0.5
    w, b = randn_bf16(D, seed=96), randn_bf16(D, seed=97)
    d1 = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
    d2 = torch.randn(M, D, device="cuda", generator=g).to(torch.bfloat16)
    rms = lambda t: w.float() * (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert_close(eng.norm16(0, x.clone(), None, w, types=1), rms(x), 1.5e-3, 1e-3, "rmsnorm -> fp16")
    x1 = x.clone()
    out = eng.norm16(0, x1, d1, w, types=1)
    xb = x.clone()
    eng.rmsnorm(xb, w, 1e-6, delta=d1)
    assert torch.equal(x1, xb)                                                       # the stream: the bf16-output kernel's bits
    assert_close(out, rms(x + d1.float()), 1.5e-3, 1e-3, "rmsnorm + add -> fp16")
    x2 = x.clone()
    out = eng.norm16(0, x2, d1, w, types=1, store_x=False)
    assert torch.equal(x2, x) and torch.equal(out, eng.norm16(0, x.clone(), d1, w, types=1))   # not stored: same operand, stream untouched
    x3 = x.clone()
    out = eng.norm16(0, x3, d1, w, delta2=d2, types=1)
    assert_close(x3, (x + d1.float()) + d2.float(), 1e-6, 1e-6, "two pending deltas")
    assert_close(out, rms((x + d1.float()) + d2.float()), 1.5e-3, 1e-3, "rmsnorm + two deltas -> fp16")
    h1 = d1.to(torch.float16)
    x4 = x.clone()
    out = eng.norm16(1, x4, h1, w, b=b, types=3, eps=1e-5)
    xs = x + h1.float()
    assert_close(x4, xs, 1e-6, 1e-6, "layernorm fp16 delta write-back")
    assert_close(out, torch.nn.functional.layer_norm(xs, (D,), w.float(), b.float(), 1e-5), 1.5e-3, 1e-3, "layernorm fp16")


@pytest.mark.parametrize("Z,M,N,K,epi,ftype", [
    (256, 128, 608, 512, 3, 1),      # the cross scores' form: rows of q.Wk x the fp16 encoder output -> fp32 (stream form)
    (200, 100, 600, 192, 3, 1),      # ragged M / N
    (64, 16, 512, 64, 0, 1),         # q.Wk's form: K = 64, fp16 result (batched over heads: the persistent kernel)
    (256, 128, 512, 640, 0, 2),      # P.E's form: fp16 operands, split-bf16 result (kept off the stream form by the engine)
    (3, 40, 72, 128, 3, 1), (5, 130, 264, 192, 0, 2)])   # too few items for the stream form / more than 128 rows: the persistent kernel
def test_batched_gemms_on_fp16_operands(eng, Z, M, N, K, epi, ftype):
    """Option dec_fp16: gemm_bf16_stream<EPI, FT> and gemm_bf16_persistent<EPI, 0, FT> on IEEE fp16 operands -- the decoder's cross-attention score
    path.  Against fp32 torch on the same fp16 inputs at the result type's rounding; stream form == persistent kernel bit for bit where both
    apply; the split result carries 16 bits of the fp32 accumulator; the fp16-result form is eight times closer than the bf16 kernel."""
    from t2v_metrics_amd.engine import load_library
    g = torch.Generator(device="cuda").manual_seed(131)
    A = torch.randn(Z, M, K, device="cuda", generator=g).to(torch.float16)
    W = (torch.randn(Z, N, K, device="cuda", generator=g) * K ** -0.5).to(torch.float16)
    ref = torch.einsum("zmk,znk->zmn", A.float(), W.float())
    split = epi == 0 and ftype == 2
    got = eng.gemm_batched(A, W, epi, split=split, ftype=ftype)
    again = eng.gemm_batched(A, W, epi, split=split, ftype=ftype)
    hi = got[0] if split else got
    assert torch.equal(hi, again[0] if split else again)
    if epi == 3:
        assert hi.dtype == torch.float32
        assert_close(hi, ref, 2e-4, 2e-5, f"f16 batched -> fp32 {Z}x{M}x{N}x{K}")
    elif ftype == 1:
        assert hi.dtype == torch.float16
        assert_close(hi, ref, 2.5e-3, 1.5e-3, "f16 batched -> fp16")
    else:
        assert hi.dtype == torch.bfloat16
        assert_close(hi, ref, 2e-2, 1e-2, "f16 batched -> bf16 hi plane")
        two = got[0].float() + got[1].float()
        rel = ((two - ref).abs() / ref.abs().clamp(min=1e-2)).max().item()
        assert rel <= 1e-3, rel                                      # hi + lo = the fp32 accumulator to 2^-16; ref differs by fp32 summation order
        assert ((two - ref).abs().mean() * 20 < (got[0].float() - ref).abs().mean())
    lib = load_library()
    if lib.vqs_debug_gemm_form(M, N, K, K, K, epi, Z, 3, 0, 0, 0) == 12 and not split:      # the bf16 rule; the fp16 launch follows the same rule
        other = eng.gemm_batched(A, W, epi, no_stream=True, ftype=ftype)
        assert torch.equal(hi, other)
