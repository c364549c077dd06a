"""ROUNDING-MATCHED CPU ORACLE for the Qwen2.5-VL row (vision tower + language-model prefill).  TEST INFRASTRUCTURE ONLY.

``QwenEngineRounded`` computes the SAME function as ``oracle/qwen25vl_oracle.py::QwenOracle`` -- the forward pass the
reference runs through HuggingFace (citations there) -- but (a) in the LAYOUTS the HIP engine
(``t2v_metrics_amd/csrc/vqs_qwen.cpp``) holds its tensors in and (b) with a round-to-bf16 at exactly the points where the
engine holds a bf16 tensor, fp32 wherever the engine keeps fp32:

  * vision rows live in the padded windowed layout (``t2v_metrics_amd/qwen/layout.py::vision_layout``: every attention
    window = ``win_len`` slots, the real ones first, zero rows in the slots of partial windows); full-attention blocks run
    on a frame-compact copy in the original patch order and scatter their delta back (vqs_qwen.cpp:494-527);
  * heads are 128 lanes wide, the real ``head_dim`` lanes first, zeros behind (vqs_qwen.cpp HDP); Q / K / V are
    bf16(acc + bias), head-major ``[segment, head, position, 128]``;
  * rotary embedding on the bf16 Q / K in fp32, result bf16 (elementwise.hip:474-517), tables per token from the host;
  * attention = attn_fwd_hd_kernel (attn.hip:468-672): ``tiled_attention`` of clip_t5_engine_rounding.py (64-key tiles,
    log2 domain, wave-wide deferred running max, P rounded to bf16 before P.V and the row sum), grouped-query heads,
    ``key_len`` = real slots of the window / tokens of the sample, causal for the language model;
  * residual stream fp32; every sub-layer output GEMM is rounded to bf16 (the engine's delta) BEFORE the fp32 add, added in
    the engine's order ``(h + d_attn) + d_mlp`` at the next storing norm (vqs_qwen.cpp:487-489,529,610,638);
  * RMSNorm: fp32 mean of squares, rsqrtf, ``(x * rs) * w`` -> bf16 (elementwise.hip:150-215);
  * SwiGLU: ``silu(g + bg) * (u + bu)`` on the fp32 accumulators -> bf16 (gemm_quad.inc:109-133); merger: erf-GELU on the
    fp32 accumulator -> bf16; patch embed and lm_head logits fp32.

What is NOT matched: the summation order inside an fp32 accumulation and the last ulp of v_exp_f32 / v_rcp_f32 / rsqrtf
(see clip_t5_engine_rounding.py) -- about one bf16 rounding in 10^4 flips.  FREE-RUNNING (``forward``) therefore sits at the
bf16 noise floor from the engine; STAGE-LOCKED (``locked`` = the engine's own intermediates read through
``vqs_qwen_debug_tap``) evaluates every launch on the ENGINE's inputs and compares it with the engine's output of that
launch -- the bit-level check ``tests/test_gpu_qwen.py`` asserts.

Pinning: ``tests/test_qwen_rounding_oracle.py`` -- with every rounding switched off (``round_fn`` = identity) this class
must reproduce ``QwenOracle`` (itself pinned to the HF modules by ``tests/test_qwen_oracle_golden.py``) to fp32 accuracy on
grids with and without partial windows, which pins the layout handling, the padded heads, the grouped-query / causal
attention and the deferred residual adds; the roundings themselves are the ones pinned for the CLIP-FlanT5 row.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence

import torch
import torch.nn.functional as F

from .clip_t5_engine_rounding import bf16_round, compare_tap, tiled_attention


def fp16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to IEEE fp16 (overflow -> inf, like v_cvt_pk_f16_f32), returned as fp32."""
    return x.to(torch.float32).to(torch.float16).to(torch.float32)


_SITE_OF_TAP = {"xn0": "x0", "q0": "qkv", "k0": "qkv", "q": "qkv", "k": "qkv", "v": "qkv", "attn": "qkv", "d_attn": "dattn", "xn1": "x1", "ff": "act",
                "d_mlp": "dmlp", "xnm": "m_x", "mid": "m_mid", "merged_w": "m_out", "merged": "m_out"}
SITE_KINDS = ("x0", "qkv", "dattn", "x1", "act", "dmlp")


def sites_from_report(cfg, values) -> Dict[tuple, float]:
    """vqs_qwen_range_report's flat order (vision blocks x six sites, merger x three, language-model layers x six) -> {(stack, layer, kind): value}."""
    v = [float(x) for x in values]
    out, k = {}, 0
    for i in range(cfg.vision.depth):
        for kind in SITE_KINDS:
            out[("vis", i, kind)] = v[k]
            k += 1
    for kind in ("m_x", "m_mid", "m_out"):
        out[("vis", -1, kind)] = v[k]
        k += 1
    for i in range(cfg.text.layers):
        for kind in SITE_KINDS:
            out[("txt", i, kind)] = v[k]
            k += 1
    assert k == len(v), (k, len(v))
    return out


def range_bounds(cfg, weights: Dict[str, torch.Tensor]) -> Dict[tuple, float]:
    """The bind-time range proof of the fp16 forms restated in torch (csrc/qwen_decode.hip "Bind-time range proof"; vqs_qwen.cpp
    compute_ranges): for every 16-bit activation site a bound of |T| from the weights alone.  Same inequalities, computed independently on the
    UNPACKED checkpoint tensors in fp64: RMSNorm output <= sqrt(D) |g|; a norm-fed linear <= ||W_j o g||_2 sqrt(D) + |b_j|; rotary <= sqrt 2;
    attention output <= the q|k|v maximum; a sum-fed linear <= sum_k |W_jk| u_k + |b_j|; |SiLU(g) u| <= |g| |u|."""
    W = lambda n: weights[n].detach().to("cpu", torch.float64)  # noqa: E731
    out = {}

    def chain(stack, i, D, g1, g2, qkv_w, qkv_b, o_w, o_b, gate_w, gate_b, up_w, up_b, down_w, down_b):
        R = D ** 0.5
        out[(stack, i, "x0")] = float(R * g1.abs().max())
        pre = R * (qkv_w * g1[None, :]).norm(dim=1) + (qkv_b.abs() if qkv_b is not None else 0.0)
        out[(stack, i, "qkv")] = float(2.0 ** 0.5 * pre.max())
        out[(stack, i, "dattn")] = float((o_w.abs().sum(dim=1) * pre.max() + (o_b.abs() if o_b is not None else 0.0)).max())
        out[(stack, i, "x1")] = float(R * g2.abs().max())
        gb = R * (gate_w * g2[None, :]).norm(dim=1) + (gate_b.abs() if gate_b is not None else 0.0)
        ub = R * (up_w * g2[None, :]).norm(dim=1) + (up_b.abs() if up_b is not None else 0.0)
        act = gb * ub
        out[(stack, i, "act")] = float(act.max())
        out[(stack, i, "dmlp")] = float((down_w.abs() @ act + (down_b.abs() if down_b is not None else 0.0)).max())

    v, t = cfg.vision, cfg.text
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        chain("vis", i, v.hidden, W(p + "norm1.weight"), W(p + "norm2.weight"), W(p + "attn.qkv.weight"), W(p + "attn.qkv.bias"), W(p + "attn.proj.weight"),
              W(p + "attn.proj.bias"), W(p + "mlp.gate_proj.weight"), W(p + "mlp.gate_proj.bias"), W(p + "mlp.up_proj.weight"), W(p + "mlp.up_proj.bias"),
              W(p + "mlp.down_proj.weight"), W(p + "mlp.down_proj.bias"))
    g = W("model.visual.merger.ln_q.weight")
    mh = v.hidden * v.merge_unit
    out[("vis", -1, "m_x")] = float(v.hidden ** 0.5 * g.abs().max())
    mid = mh ** 0.5 * (W("model.visual.merger.mlp.0.weight") * g.repeat(v.merge_unit)[None, :]).norm(dim=1) + W("model.visual.merger.mlp.0.bias").abs()
    out[("vis", -1, "m_mid")] = float(mid.max())
    out[("vis", -1, "m_out")] = float((W("model.visual.merger.mlp.2.weight").abs() @ mid + W("model.visual.merger.mlp.2.bias").abs()).max())
    for i in range(t.layers):
        p = f"model.language_model.layers.{i}."
        qkv_w = torch.cat([W(p + f"self_attn.{n}_proj.weight") for n in "qkv"])
        qkv_b = torch.cat([W(p + f"self_attn.{n}_proj.bias") for n in "qkv"])
        chain("txt", i, t.hidden, W(p + "input_layernorm.weight"), W(p + "post_attention_layernorm.weight"), qkv_w, qkv_b, W(p + "self_attn.o_proj.weight"), None,
              W(p + "mlp.gate_proj.weight"), None, W(p + "mlp.up_proj.weight"), None, W(p + "mlp.down_proj.weight"), None)
    return out


def sigma_of_bound(B: float, head: float = 32768.0) -> float:
    """vqs_qwen.cpp site_from_bound: the largest power of two <= 1 with B * sigma <= head (half of the fp16 maximum)."""
    e = 0
    while B * 2.0 ** -e > head:
        e += 1
    return 2.0 ** -e

HDP = 128          # lanes per head in the engine's Q / K / V / attention tensors (vqs_qwen.cpp)


def vision_heads_compact(cfg) -> bool:
    """vqs_qwen.cpp v_compact: tower heads narrower than 128 lanes with whole 128-column q | k | v ranges (7B: 16 x 80) keep the
    checkpoint's qkv / proj tensors -- Q / K / V still sit in 128-lane slots, but the attention output leaves compact
    ``[rows, heads * head_dim]``."""
    v = cfg.vision
    return v.head_dim < HDP and v.head_dim % 8 == 0 and v.hidden % 128 == 0


def _identity(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float32)


class QwenEngineRounded:
    """See the module docstring.  ``layers`` (stage-locked runs only): restrict the check to these block / layer indices
    (None = all) -- every launch is evaluated on the engine's own inputs, so a subset is a valid, cheaper check."""

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], round_fn=bf16_round, acc=torch.float64, sites: Optional[Dict[tuple, float]] = None,
                 rope_fused: bool = False):
        """``sites`` (round 6): {(stack, layer, kind): sigma} of the engine's range-safe fp16 forms (``sites_from_report(cfg, sigmas)``) --
        the tower, merger and prefill then hold every 16-bit tensor as fp16(T * sigma) (computed here in TRUE units: fp16_round(x * sigma) /
        sigma), P as plain fp16, the language model's final norm output as bf16; taps are fp16 tensors divided by their sigma on read.
        None = every tensor rounded with ``round_fn`` (bf16: the reference's dtype)."""
        self.cfg = cfg
        self._raw = weights                      # converted on use: the 7B weight set is 33 GB in fp32
        self.r = round_fn
        self.sites = sites
        # the engine's option of that name (what vqs_qwen_get_option "rope_fused" reads): the language model's q / k are rotated inside the q|k|v
        # GEMM's epilogue on the UNROUNDED projection and rounded once; there are no q0 / k0 tensors
        self.rope_fused = bool(rope_fused)
        self.acc = acc
        self.locked: Optional[Dict[str, torch.Tensor]] = None
        self.record: Optional[Dict[str, torch.Tensor]] = None
        self.report: Dict[str, dict] = {}
        self.layers: Optional[Iterable[int]] = None

    # ------------------------------------------------------------------------------------------------ plumbing
    class _W:
        def __init__(self, raw):
            self.raw = raw

        def __getitem__(self, name: str) -> torch.Tensor:
            return self.raw[name].detach().to("cpu", torch.float32)

    @property
    def w(self):
        return QwenEngineRounded._W(self._raw)

    def _rf(self, stack: str, i: int, kind: str):
        """rounding function (true units in and out) of a 16-bit site"""
        if self.sites is None:
            return self.r
        s = self.sites[(stack, i, kind)]
        return lambda x: fp16_round(x.to(torch.float32) * s) / s

    def _tap_sigma_bits(self, name: str):
        """(sigma, stored mantissa bits) of the engine tensor behind a tap name"""
        if self.sites is None:
            return 1.0, 7
        parts = name.split(".")
        kind = _SITE_OF_TAP.get(parts[-1])
        if kind is None or parts[0] == "dec":
            return 1.0, 7                        # fp32 tensors, the final norm's bf16 output, the decode step (bf16)
        i = int(parts[1]) if len(parts) == 3 else -1
        return self.sites[(parts[0], i, kind)], 10

    def _emit(self, name: str, y: torch.Tensor, cols: Optional[int] = None) -> torch.Tensor:
        """Free-running: record and pass through.  Stage-locked: compare with the engine's tensor of that name and return the
        ENGINE's tensor.  ``cols``: the engine's rows are wider than ``y``'s (zero padding behind ``cols`` columns): compare
        the first ``cols`` columns, count non-zero padding, return the unpadded view."""
        if self.locked is None:
            if self.record is not None:
                self.record[name] = y
            return y
        if name not in self.locked:
            raise KeyError(f"stage-locked run needs the engine tap {name!r}")
        sigma, bits = self._tap_sigma_bits(name)
        e = self.locked[name].detach().to("cpu", torch.float32) / sigma
        pad_nonzero = 0
        if cols is not None:
            e = e.reshape(y.shape[0], -1)
            pad_nonzero = int((e[:, cols:] != 0).sum())
            e = e[:, :cols]
        else:
            e = e.reshape(y.shape)
        self.report[name] = compare_tap(y, e, mant_bits=bits)
        self.report[name]["pad_nonzero"] = pad_nonzero
        self.report[name]["stored_absmax"] = float(e.abs().max()) * sigma       # what the 16-bit tensor itself holds (fp16 forms: <= 65504)
        return e

    def _have(self, name: str) -> bool:
        return self.locked is None or name in self.locked

    def _lin(self, x: torch.Tensor, wname: str, bname: Optional[str] = None) -> torch.Tensor:
        """fp32 accumulator of an nn.Linear over bf16 operands, bias added in fp32 (the GEMM epilogues)."""
        w = self.w[wname]
        y = (x.to(self.acc) @ w.reshape(w.shape[0], -1).to(self.acc).t()).float()
        return y + self.w[bname] if bname is not None else y

    def _norm(self, h: torch.Tensor, wname: str, eps: float, rf=None) -> torch.Tensor:
        ms = (h.double() ** 2).mean(-1, keepdim=True).float()
        rs = torch.rsqrt(ms + eps)
        return (rf or self.r)((h * rs) * self.w[wname])

    def _heads(self, y: torch.Tensor, nseg: int, S: int, H: int, hd: int, rf=None) -> torch.Tensor:
        """[nseg*S, H*hd] -> bf16, head-major, 128-lane heads: [nseg, H, S, 128]."""
        y = (rf or self.r)(y).reshape(nseg, S, H, hd)
        return F.pad(y, (0, HDP - hd)).permute(0, 2, 1, 3).contiguous()

    def _rope(self, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, hd: int, rf=None) -> torch.Tensor:
        """x [nseg, H, S, 128], cos / sin [nseg*S, hd/2] -> rotated, bf16 (rope_kernel: a*c - b*s, b*c + a*s in fp32)."""
        nseg, H, S, _ = x.shape
        half = hd // 2
        c = cos.reshape(nseg, 1, S, half).double()
        s = sin.reshape(nseg, 1, S, half).double()
        a, b = x[..., :half].double(), x[..., half: 2 * half].double()
        out = x.clone()
        out[..., :half] = (rf or self.r)((a * c - b * s).float())
        out[..., half: 2 * half] = (rf or self.r)((b * c + a * s).float())
        return out

    def _unpad_heads(self, a: torch.Tensor, H: int, hd: int) -> torch.Tensor:
        """attention output [rows, H*128] -> the real lanes [rows, H*hd] (what the unpadded o_proj / proj weight multiplies)."""
        return a.reshape(a.shape[0], H, HDP)[..., :hd].reshape(a.shape[0], H * hd)

    def _stream(self, name: str, h, d_attn, d_mlp, shape, first: bool = False) -> torch.Tensor:
        """The fp32 residual stream a storing norm writes: ``(h + d_attn) + d_mlp`` of the block before (``h`` itself for the
        first block).  ``h`` None = the block before was skipped in a stage-locked run: take the engine's stream unchecked."""
        if h is None:
            return self.locked[name].detach().to("cpu", torch.float32).reshape(shape)
        return self._emit(name, h if first else (h + d_attn) + d_mlp)

    def _want(self, i: int) -> bool:
        return self.locked is None or self.layers is None or i in self.layers

    # ------------------------------------------------------------------------------------------------ vision tower
    def vision_tower(self, pixel_values: torch.Tensor, lay: Dict[str, torch.Tensor]) -> torch.Tensor:
        """vqs_qwen_encode_vision (vqs_qwen.cpp:447-578): patches [N, patch_dim] (bf16 values) + the arrays of
        ``vision_layout`` -> merged tokens [N/4, out_hidden] in the original cell order."""
        v = self.cfg.vision
        r, N, Np, S_w, S_f = self.r, lay["N"], lay["Np"], lay["win_len"], lay["frame_len"]
        row_map, inv_row = lay["row_map"].long(), lay["inv_row"].long()
        real = row_map >= 0
        H, hd, VH = v.heads, v.head_dim, v.hidden
        scale = hd ** -0.5
        F16 = self.sites is not None
        rp = fp16_round if F16 else r            # probabilities: plain fp16 in the fp16 forms (P <= 1 needs no scale)
        pre = self._emit("vis.pre", self._lin(r(pixel_values), "model.visual.patch_embed.proj.weight"))
        h = torch.where(real[:, None], pre[row_map.clamp(min=0)], torch.zeros(1, VH))        # window permutation, zero padding rows
        d_attn = d_mlp = None
        for i in range(v.depth):
            p, t = f"model.visual.blocks.{i}.", f"vis.{i}."
            if not self._want(i):
                h = None             # skipped block of a stage-locked run: the next checked block starts from the engine's stream
                continue
            h = self._stream(t + "h", h, d_attn, d_mlp, (Np, VH), first=(i == 0))
            r_x0, r_qkv, r_da, r_x1, r_act, r_dm = (self._rf("vis", i, k) for k in SITE_KINDS)
            xn = self._emit(t + "xn0", self._norm(h, p + "norm1.weight", v.rms_eps, r_x0))
            full = i in v.fullatt_blocks
            if full:
                xin, rows, S, cos, sin, klen = xn[inv_row], N, S_f, lay["cos_f"], lay["sin_f"], None
            else:
                xin, rows, S, cos, sin, klen = xn, Np, S_w, lay["cos_w"], lay["sin_w"], lay["win_valid"].long()
            nseg = rows // S
            qkv = self._lin(xin, p + "attn.qkv.weight", p + "attn.qkv.bias").reshape(rows, 3, H * hd)
            q0 = self._emit(t + "q0", self._heads(qkv[:, 0], nseg, S, H, hd, r_qkv))
            k0 = self._emit(t + "k0", self._heads(qkv[:, 1], nseg, S, H, hd, r_qkv))
            val = self._emit(t + "v", self._heads(qkv[:, 2], nseg, S, H, hd, r_qkv))
            q = self._emit(t + "q", self._rope(q0, cos, sin, hd, r_qkv))
            k = self._emit(t + "k", self._rope(k0, cos, sin, hd, r_qkv))
            a = tiled_attention(q, k, val, scale, key_len=klen, round_fn=rp, round_out=r_qkv, acc=self.acc).reshape(rows, H * HDP)
            if vision_heads_compact(self.cfg):
                a = self._emit(t + "attn", self._unpad_heads(a, H, hd))
            else:
                a = self._unpad_heads(self._emit(t + "attn", a), H, hd)
            d = r_da(self._lin(a, p + "attn.proj.weight", p + "attn.proj.bias"))
            if full:
                d = torch.where(real[:, None], d[row_map.clamp(min=0)], torch.zeros(1, VH))   # scatter back, zero padding rows
            d_attn = self._emit(t + "d_attn", d)
            xn = self._emit(t + "xn1", self._norm(h + d_attn, p + "norm2.weight", v.rms_eps, r_x1))
            g = self._lin(xn, p + "mlp.gate_proj.weight", p + "mlp.gate_proj.bias")
            u = self._lin(xn, p + "mlp.up_proj.weight", p + "mlp.up_proj.bias")
            ff = self._emit(t + "ff", r_act(g * torch.sigmoid(g) * u), cols=v.mlp)
            d_mlp = self._emit(t + "d_mlp", r_dm(self._lin(ff, p + "mlp.down_proj.weight", p + "mlp.down_proj.bias")))
        h = self._stream("vis.h_out", h, d_attn, d_mlp, (Np, VH))
        xn = self._emit("vis.xnm", self._norm(h, "model.visual.merger.ln_q.weight", 1e-6, self._rf("vis", -1, "m_x")))
        x = xn.reshape(Np // v.merge_unit, v.merge_unit * VH)
        mid = self._emit("vis.mid", self._rf("vis", -1, "m_mid")(F.gelu(self._lin(x, "model.visual.merger.mlp.0.weight", "model.visual.merger.mlp.0.bias"))))
        mw = self._emit("vis.merged_w", self._rf("vis", -1, "m_out")(self._lin(mid, "model.visual.merger.mlp.2.weight", "model.visual.merger.mlp.2.bias")))
        return self._emit("vis.merged", mw[lay["cell_inv"].long()])

    # ------------------------------------------------------------------------------------------------ language model
    def text_logits(self, merged: torch.Tensor, input_ids: torch.Tensor, lay: Dict[str, torch.Tensor]) -> torch.Tensor:
        """vqs_qwen_score (vqs_qwen.cpp:585-667): merged vision tokens (bf16 values) + right-padded ids + the arrays of
        ``text_layout`` -> fp32 logits [B, vocab] of the last valid position."""
        t_ = self.cfg.text
        r = self.r
        B, L = input_ids.shape
        M, TH, H, Hkv, hd = B * L, t_.hidden, t_.heads, t_.kv_heads, t_.head_dim
        scale = hd ** -0.5
        slot = lay["vis_slot"].reshape(-1).long()
        ids = input_ids.reshape(-1).long().clamp(0, t_.vocab - 1)
        F16 = self.sites is not None
        rp = fp16_round if F16 else r
        emb = r(self.w["model.language_model.embed_tokens.weight"])[ids]
        h = self._emit("txt.emb", torch.where((slot >= 0)[:, None], (self._rf("vis", -1, "m_out") if F16 else r)(merged)[slot.clamp(min=0)], emb))
        klen = lay["seq_len"].long()
        d_attn = d_mlp = None
        for i in range(t_.layers):
            p, t = f"model.language_model.layers.{i}.", f"txt.{i}."
            if not self._want(i):
                h = None
                continue
            h = self._stream(t + "h", h, d_attn, d_mlp, (M, TH), first=(i == 0))
            r_x0, r_qkv, r_da, r_x1, r_act, r_dm = (self._rf("txt", i, k) for k in SITE_KINDS)
            xn = self._emit(t + "xn0", self._norm(h, p + "input_layernorm.weight", t_.rms_eps, r_x0))
            if self.rope_fused:
                q0 = self._heads(self._lin(xn, p + "self_attn.q_proj.weight", p + "self_attn.q_proj.bias"), B, L, H, hd, _identity)
                k0 = self._heads(self._lin(xn, p + "self_attn.k_proj.weight", p + "self_attn.k_proj.bias"), B, L, Hkv, hd, _identity)
            else:
                q0 = self._emit(t + "q0", self._heads(self._lin(xn, p + "self_attn.q_proj.weight", p + "self_attn.q_proj.bias"), B, L, H, hd, r_qkv))
                k0 = self._emit(t + "k0", self._heads(self._lin(xn, p + "self_attn.k_proj.weight", p + "self_attn.k_proj.bias"), B, L, Hkv, hd, r_qkv))
            val = self._emit(t + "v", self._heads(self._lin(xn, p + "self_attn.v_proj.weight", p + "self_attn.v_proj.bias"), B, L, Hkv, hd, r_qkv))
            q = self._emit(t + "q", self._rope(q0, lay["cos"], lay["sin"], hd, r_qkv))
            k = self._emit(t + "k", self._rope(k0, lay["cos"], lay["sin"], hd, r_qkv))
            rep = H // Hkv
            a = tiled_attention(q, k.repeat_interleave(rep, dim=1), val.repeat_interleave(rep, dim=1), scale, key_len=klen, round_fn=rp,
                                round_out=r_qkv, acc=self.acc, causal=True).reshape(M, H * HDP)
            a = self._emit(t + "attn", a)
            d_attn = self._emit(t + "d_attn", r_da(self._lin(self._unpad_heads(a, H, hd), p + "self_attn.o_proj.weight")))
            xn = self._emit(t + "xn1", self._norm(h + d_attn, p + "post_attention_layernorm.weight", t_.rms_eps, r_x1))
            g = self._lin(xn, p + "mlp.gate_proj.weight")
            u = self._lin(xn, p + "mlp.up_proj.weight")
            ff = self._emit(t + "ff", r_act(g * torch.sigmoid(g) * u), cols=t_.mlp)
            d_mlp = self._emit(t + "d_mlp", r_dm(self._lin(ff, p + "mlp.down_proj.weight")))
        h = self._stream("txt.h_out", h, d_attn, d_mlp, (M, TH))
        xn = self._emit("txt.xnf", self._norm(h, "model.language_model.norm.weight", t_.rms_eps, bf16_round if F16 else None))
        last = xn[lay["last_row"].long()]
        return self._emit("txt.logits", self._lin(last, "lm_head.weight"))

    def text_decode(self, token_ids: torch.Tensor, kc, vc, length: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        """vqs_qwen_decode (vqs_qwen.cpp; kernels in qwen_decode.hip): ONE further position per sample against the KV cache.
        token_ids long [B]; kc / vc: per layer [B, kv_heads, Lmax, 128] fp32 holding bf16 values; length long [B] = index of the new
        position; cos / sin [B, head_dim/2].  Free-running: the new K / V rows are written into kc / vc.  Stage-locked: kc / vc are the
        ENGINE's cache after the step -- its new rows are compared with the oracle's ("dec.<i>.k_row" / "v_row") and used.
        One-row attention: fp32 scores * scale, p = exp(t - max) NOT rounded, sum(p v) / sum(p) -> bf16 (qwen_decode_attn_kernel)."""
        t_ = self.cfg.text
        r = self.r
        B = token_ids.shape[0]
        TH, H, Hkv, hd = t_.hidden, t_.heads, t_.kv_heads, t_.head_dim
        scale, rep = hd ** -0.5, H // Hkv
        bi = torch.arange(B)
        h = self._emit("dec.emb", r(self.w["model.language_model.embed_tokens.weight"])[token_ids.long().clamp(0, t_.vocab - 1)])
        d_attn = d_mlp = None
        for i in range(t_.layers):
            p, t = f"model.language_model.layers.{i}.", f"dec.{i}."
            if not self._want(i):
                h = None
                continue
            h = self._stream(t + "h", h, d_attn, d_mlp, (B, TH), first=(i == 0))
            xn = self._emit(t + "xn0", self._norm(h, p + "input_layernorm.weight", t_.rms_eps))
            parts = [self._heads(self._lin(xn, p + f"self_attn.{n}_proj.weight", p + f"self_attn.{n}_proj.bias"), B, 1, nh, hd).reshape(B, nh * HDP)
                     for n, nh in (("q", H), ("k", Hkv), ("v", Hkv))]
            qkv = self._emit(t + "qkv", torch.cat(parts, dim=1))
            q0 = qkv[:, : H * HDP].reshape(B, H, 1, HDP)
            k0 = qkv[:, H * HDP: (H + Hkv) * HDP].reshape(B, Hkv, 1, HDP)
            v0 = qkv[:, (H + Hkv) * HDP:].reshape(B, Hkv, HDP)
            q = self._emit(t + "q", self._rope(q0, cos, sin, hd).reshape(B, H * HDP)).reshape(B, H, HDP)
            k_new = self._rope(k0, cos, sin, hd).reshape(B, Hkv, HDP)
            if self.locked is None:
                kc[i][bi, :, length.long()] = k_new
                vc[i][bi, :, length.long()] = v0
            else:
                self.report[t + "k_row"] = compare_tap(k_new, kc[i][bi, :, length.long()])
                self.report[t + "v_row"] = compare_tap(v0, vc[i][bi, :, length.long()])
            a = torch.zeros(B, H, HDP)
            for b in range(B):
                n = int(length[b]) + 1
                K = kc[i][b, :, :n].repeat_interleave(rep, dim=0).to(self.acc)                    # [H, n, 128]
                V = vc[i][b, :, :n].repeat_interleave(rep, dim=0).to(self.acc)
                sc = torch.einsum("hd,hnd->hn", q[b].to(self.acc), K).float() * scale
                pr = torch.exp(sc - sc.max(-1, keepdim=True).values)
                a[b] = (torch.einsum("hn,hnd->hd", pr.to(self.acc), V).float() / pr.to(self.acc).sum(-1, keepdim=True).float())
            a = self._emit(t + "attn", r(a).reshape(B, H * HDP))
            d_attn = self._emit(t + "d_attn", r(self._lin(self._unpad_heads(a, H, hd), p + "self_attn.o_proj.weight")))
            xn = self._emit(t + "xn1", self._norm(h + d_attn, p + "post_attention_layernorm.weight", t_.rms_eps))
            g = self._lin(xn, p + "mlp.gate_proj.weight")
            u = self._lin(xn, p + "mlp.up_proj.weight")
            ff = self._emit(t + "ff", r(g * torch.sigmoid(g) * u), cols=t_.mlp)
            d_mlp = self._emit(t + "d_mlp", r(self._lin(ff, p + "mlp.down_proj.weight")))
        h = self._stream("dec.h_out", h, d_attn, d_mlp, (B, TH))
        xn = self._emit("dec.xnf", self._norm(h, "model.language_model.norm.weight", t_.rms_eps))
        return self._emit("dec.logits", self._lin(xn, "lm_head.weight"))

    def decode_locked(self, taps, token_ids, kc, vc, length, cos, sin, layers: Optional[Sequence[int]] = None) -> Dict[str, dict]:
        """Stage-locked check of one vqs_qwen_decode call; kc / vc = the engine's cache after the call, "dec.logits" its output."""
        self.locked, self.layers, self.report = taps, (None if layers is None else set(layers)), {}
        with torch.no_grad():
            self.text_decode(token_ids, kc, vc, length, cos, sin)
        self.locked = None
        return self.report

    # ------------------------------------------------------------------------------------------------ whole passes
    def forward(self, pixel_values: torch.Tensor, vis_lay: Dict[str, torch.Tensor], input_ids: torch.Tensor,
                txt_lay: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Free-running pass of ONE vision call (videos of one grid) followed by the prefill: fp32 logits [B, vocab]."""
        self.locked = None
        with torch.no_grad():
            return self.text_logits(self.vision_tower(pixel_values, vis_lay), input_ids, txt_lay)

    def vision_locked(self, taps: Dict[str, torch.Tensor], pixel_values, lay, layers: Optional[Sequence[int]] = None) -> Dict[str, dict]:
        """Stage-locked check of one vqs_qwen_encode_vision call; ``taps`` must hold every vis.* name of the checked blocks
        (plus "vis.merged" = the call's output).  -> report {tap name: compare_tap(...)}."""
        self.locked, self.layers, self.report = taps, (None if layers is None else set(layers)), {}
        with torch.no_grad():
            self.vision_tower(pixel_values, lay)
        self.locked = None
        return self.report

    def text_locked(self, taps: Dict[str, torch.Tensor], merged, input_ids, lay, layers: Optional[Sequence[int]] = None) -> Dict[str, dict]:
        """Stage-locked check of one vqs_qwen_score call ("txt.logits" = the call's output)."""
        self.locked, self.layers, self.report = taps, (None if layers is None else set(layers)), {}
        with torch.no_grad():
            self.text_logits(merged, input_ids, lay)
        self.locked = None
        return self.report


_PER_LAYER = ("h", "xn0", "q0", "k0", "q", "k", "v", "attn", "d_attn", "xn1", "ff", "d_mlp")
FP32_TAPS = ("pre", "h", "h_out", "emb", "logits")           # last name component of the fp32 tensors; all others are bf16


def _ffld(mlp: int) -> int:
    return -(-mlp // 64) * 64                                  # vqs_qwen.cpp:341-344 (gate|up blocks of 32, rows of 64)


def vision_tap_shapes(cfg, lay, layers: Optional[Sequence[int]] = None):
    """{tap name: (shape, dtype)} of one vqs_qwen_encode_vision call in the engine's layouts (include/vqs_qwen.h)."""
    v = cfg.vision
    N, Np, VH, H = lay["N"], lay["Np"], v.hidden, v.heads
    f32, b16 = torch.float32, torch.bfloat16
    out = {"vis.pre": ((N, VH), f32)}
    for i in (range(v.depth) if layers is None else layers):
        rows, S = (N, lay["frame_len"]) if i in v.fullatt_blocks else (Np, lay["win_len"])
        hs = ((rows // S, H, S, HDP), b16)
        out.update({f"vis.{i}.h": ((Np, VH), f32), f"vis.{i}.xn0": ((Np, VH), b16), f"vis.{i}.q0": hs, f"vis.{i}.k0": hs, f"vis.{i}.q": hs,
                    f"vis.{i}.k": hs, f"vis.{i}.v": hs, f"vis.{i}.attn": ((rows, VH if vision_heads_compact(cfg) else H * HDP), b16), f"vis.{i}.d_attn": ((Np, VH), b16),
                    f"vis.{i}.xn1": ((Np, VH), b16), f"vis.{i}.ff": ((Np, _ffld(v.mlp)), b16), f"vis.{i}.d_mlp": ((Np, VH), b16)})
    ncp = Np // v.merge_unit
    out.update({"vis.h_out": ((Np, VH), f32), "vis.xnm": ((Np, VH), b16), "vis.mid": ((ncp, v.merge_unit * VH), b16),
                "vis.merged_w": ((ncp, v.out_hidden), b16)})
    return out


def text_tap_shapes(cfg, B: int, L: int, layers: Optional[Sequence[int]] = None, rope_fused: bool = False):
    """{tap name: (shape, dtype)} of one vqs_qwen_score call (rope_fused: the engine rotates q / k in the q|k|v epilogue: no q0 / k0)."""
    t = cfg.text
    M, TH = B * L, t.hidden
    f32, b16 = torch.float32, torch.bfloat16
    out = {"txt.emb": ((M, TH), f32)}
    for i in (range(t.layers) if layers is None else layers):
        hq, hk = ((B, t.heads, L, HDP), b16), ((B, t.kv_heads, L, HDP), b16)
        out.update({f"txt.{i}.h": ((M, TH), f32), f"txt.{i}.xn0": ((M, TH), b16), f"txt.{i}.q0": hq, f"txt.{i}.k0": hk, f"txt.{i}.q": hq,
                    f"txt.{i}.k": hk, f"txt.{i}.v": hk, f"txt.{i}.attn": ((M, t.heads * HDP), b16), f"txt.{i}.d_attn": ((M, TH), b16),
                    f"txt.{i}.xn1": ((M, TH), b16), f"txt.{i}.ff": ((M, _ffld(t.mlp)), b16), f"txt.{i}.d_mlp": ((M, TH), b16)})
    out.update({"txt.h_out": ((M, TH), f32), "txt.xnf": ((M, TH), b16)})
    if rope_fused:
        out = {n: v for n, v in out.items() if not (n.endswith(".q0") or n.endswith(".k0"))}
    return out


def decode_tap_shapes(cfg, B: int, layers: Optional[Sequence[int]] = None):
    """{tap name: (shape, dtype)} of one vqs_qwen_decode call."""
    t = cfg.text
    TH, IQ, QN = t.hidden, t.heads * HDP, (t.heads + 2 * t.kv_heads) * HDP
    f32, b16 = torch.float32, torch.bfloat16
    out = {"dec.emb": ((B, TH), f32)}
    for i in (range(t.layers) if layers is None else layers):
        out.update({f"dec.{i}.h": ((B, TH), f32), f"dec.{i}.xn0": ((B, TH), b16), f"dec.{i}.qkv": ((B, QN), b16), f"dec.{i}.q": ((B, IQ), b16),
                    f"dec.{i}.attn": ((B, IQ), b16), f"dec.{i}.d_attn": ((B, TH), b16), f"dec.{i}.xn1": ((B, TH), b16),
                    f"dec.{i}.ff": ((B, _ffld(t.mlp)), b16), f"dec.{i}.d_mlp": ((B, TH), b16)})
    out.update({"dec.h_out": ((B, TH), f32), "dec.xnf": ((B, TH), b16)})
    return out


def vision_tap_names(cfg, layers: Optional[Sequence[int]] = None):
    blocks = range(cfg.vision.depth) if layers is None else layers
    return ["vis.pre"] + [f"vis.{i}.{n}" for i in blocks for n in _PER_LAYER] + ["vis.h_out", "vis.xnm", "vis.mid", "vis.merged_w"]


def text_tap_names(cfg, layers: Optional[Sequence[int]] = None):
    ls = range(cfg.text.layers) if layers is None else layers
    return ["txt.emb"] + [f"txt.{i}.{n}" for i in ls for n in _PER_LAYER] + ["txt.h_out", "txt.xnf"]
