"""Helpers for the -m gpu parity tests: references in plain fp32 torch and mismatch diagnostics."""
import math

import torch


def randn_bf16(*shape, seed=0, scale=1.0, device="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(device)


def describe_mismatch(out: torch.Tensor, ref: torch.Tensor, atol: float, rtol: float, name: str = "") -> str:
    """Readable summary of where two tensors differ (row/col periodicity exposes fragment-layout bugs)."""
    o = out.detach().float().cpu()
    r = ref.detach().float().cpu()
    if o.shape != r.shape:
        return f"{name}: shape {tuple(o.shape)} vs {tuple(r.shape)}"
    err = (o - r).abs()
    tol = atol + rtol * r.abs()
    bad = err > tol
    nbad = int(bad.sum())
    msg = [f"{name}: max|err|={err.max().item():.4g} at {tuple(int(i) for i in torch.nonzero(err == err.max())[0])} "
           f"ref_absmax={r.abs().max().item():.4g} bad={nbad}/{o.numel()} nan_out={int(torch.isnan(o).sum())}"]
    if nbad and o.dim() >= 2:
        b2 = bad.reshape(-1, bad.shape[-1])
        rows = torch.nonzero(b2.any(1)).flatten()
        cols = torch.nonzero(b2.any(0)).flatten()
        msg.append(f"  bad rows: n={rows.numel()} first={rows[:12].tolist()} rows%32 hist={torch.bincount(rows % 32, minlength=32).tolist()}")
        msg.append(f"  bad cols: n={cols.numel()} first={cols[:12].tolist()} cols%32 hist={torch.bincount(cols % 32, minlength=32).tolist()}")
        i = torch.nonzero(b2)[0]
        msg.append(f"  first bad [{int(i[0])},{int(i[1])}] out={o.reshape(-1, o.shape[-1])[i[0], i[1]].item():.5g} "
                   f"ref={r.reshape(-1, r.shape[-1])[i[0], i[1]].item():.5g}")
    return "\n".join(msg)


def assert_close(out, ref, atol, rtol, name=""):
    o = out.detach().float().cpu()
    r = ref.detach().float().cpu()
    ok = o.shape == r.shape and bool(((o - r).abs() <= atol + rtol * r.abs()).all()) and not bool(torch.isnan(o).any())
    assert ok, describe_mismatch(out, ref, atol, rtol, name)


# ---- fp32 references of the fused epilogues (HF activations.py:59-66,117-123; torch GELU)
def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def interleave_gate(wi0, wi1):
    """rows [64j,64j+32) = wi_0[32j:32j+32), rows [64j+32,64j+64) = wi_1[32j:32j+32)."""
    F, D = wi0.shape
    a = wi0.reshape(F // 32, 32, D)
    b = wi1.reshape(F // 32, 32, D)
    return torch.stack([a, b], dim=1).reshape(2 * F, D).contiguous()


def attention_ref(q, k, v, scale, bias=None, key_len=None):
    """q,k,v [B,H,S,64] -> [B*S, H*64] fp32.  bias [H,S,S] additive; key_len [B]."""
    B, H, S, d = q.shape
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias[None].float()
    if key_len is not None:
        mask = torch.arange(S, device=q.device)[None, :] >= key_len[:, None].to(q.device)
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = p @ v.float()
    return o.transpose(1, 2).reshape(B * S, H * d)
