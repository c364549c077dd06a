#!/usr/bin/env python3
"""Per-op check of the rounding-matched oracle against the HIP kernels (run on the GPU box): for every op the
fraction of bf16 outputs that differ and the largest |difference|.  Expected: ~1e-4 fractions (fp32 summation order)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine as E
from oracle.clip_t5_engine_rounding import bf16_round, tiled_attention, gelu_new_sigmoid
from oracle.clip_t5_oracle import layer_norm, t5_rms_norm, quick_gelu, gelu_erf
from tests.gpu_util import randn_bf16, interleave_gate

E.load_library()


def rep(tag, out, ref):
    o, r = out.detach().float().cpu(), ref.detach().float().cpu()
    d = (o - r).abs()
    print(f"{tag:40s} frac_diff {float((d > 0).float().mean()):.2e}  max|d| {float(d.max()):.3e}  absmax {float(r.abs().max()):.3g}", flush=True)


M, N, K = 1154, 512, 256
A = randn_bf16(M, K, seed=1); W = randn_bf16(N, K, seed=2, scale=K ** -0.5); b = randn_bf16(N, seed=3, scale=0.1)
acc = (A.double().cpu() @ W.double().cpu().t()).float()
accb = acc + b.float().cpu()
for v in (0, 3):
    rep(f"gemm bf16 v{v}", E.gemm(A, W, 0, variant=v), bf16_round(acc))
    rep(f"gemm bf16+bias v{v}", E.gemm(A, W, 0, bias=b, variant=v), bf16_round(accb))
    rep(f"gemm quick_gelu v{v}", E.gemm(A, W, 1, bias=b, variant=v), bf16_round(quick_gelu(accb)))
    rep(f"gemm gelu_erf v{v}", E.gemm(A, W, 2, bias=b, variant=v), bf16_round(gelu_erf(accb)))
    rep(f"gemm f32 v{v}", E.gemm(A, W, 3, bias=b, variant=v), accb)
F = 384
w0 = randn_bf16(F, K, seed=6, scale=K ** -0.5); w1 = randn_bf16(F, K, seed=7, scale=K ** -0.5)
g0 = (A.double().cpu() @ w0.double().cpu().t()).float(); g1 = (A.double().cpu() @ w1.double().cpu().t()).float()
rep("gemm gated v3", E.gemm(A, interleave_gate(w0, w1), 5, variant=3), bf16_round(gelu_new_sigmoid(g0) * g1))

# norms
x = torch.randn(300, 1024, generator=torch.Generator().manual_seed(5)) * 2
d1 = randn_bf16(300, 1024, seed=8); wn = (1 + 0.1 * torch.randn(1024)).to(torch.bfloat16); bn = (0.02 * torch.randn(1024)).to(torch.bfloat16)
xs = x.cuda().clone()
out = E.rmsnorm(xs, wn.cuda(), 1e-6, delta=d1)
h = x + d1.float().cpu()
rep("rmsnorm(+delta) bf16", out, bf16_round(t5_rms_norm(h, wn.float(), 1e-6)))
rep("  stream after", xs, h)
xs = x.cuda().clone()
out = E.layernorm(xs, wn.cuda(), bn.cuda(), 1e-5, delta=d1)
rep("layernorm(+delta) bf16", out, bf16_round(layer_norm(h, wn.float(), bn.float(), 1e-5)))
xs = x.cuda().clone()
out = E.layernorm(xs, wn.cuda(), bn.cuda(), 1e-5, out_f32=True)
rep("layernorm f32", out, layer_norm(x, wn.float(), bn.float(), 1e-5))

# attention
for (B, H, S, bias, kl) in [(2, 2, 17, False, None), (2, 3, 577, False, None), (2, 2, 150, True, [150, 97]), (1, 4, 608, True, [601])]:
    q = randn_bf16(B, H, S, 64, seed=11, scale=1.5); k = randn_bf16(B, H, S, 64, seed=12); v = randn_bf16(B, H, S, 64, seed=13)
    table = (torch.randn(H, 2 * S - 1, generator=torch.Generator().manual_seed(14)) * 0.5).to(torch.bfloat16).float() if bias else None
    klt = torch.tensor(kl, dtype=torch.int32) if kl else None
    scale = 1.0 if bias else 0.125
    if bias:
        q = (q.float() * 0.3).to(torch.bfloat16)
    out = E.attention(q, k, v, scale, bias_table=table.cuda() if bias else None, key_len=klt.cuda() if kl else None)
    ref = tiled_attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), scale, table, klt.long() if kl else None)
    o = out.reshape(B, S, H * 64)
    if kl:
        for bi, n in enumerate(kl):
            rep(f"attention S={S} bias={bias} sample{bi} valid rows", o[bi, :n], ref[bi, :n])
    else:
        rep(f"attention S={S} bias={bias}", o, ref)
