"""ROUNDING-MATCHED CPU ORACLE for the CLIP-FlanT5 VQAScore hot path.  TEST INFRASTRUCTURE ONLY.

``EngineRoundedOracle`` (= ``Oracle(cfg, weights, emulate="engine")``) computes the SAME function as
``oracle/clip_t5_oracle.py::Oracle`` -- the forward pass the reference runs through HuggingFace
(citations there) -- but places a round-to-bf16 at exactly the points where the HIP engine
(``t2v_metrics_amd/csrc/vqs_api.cpp``) holds a bf16 tensor, and keeps fp32 everywhere the engine does:

  * residual stream fp32; a sub-layer's output GEMM is rounded to bf16 (the engine's ``delta``) BEFORE the
    fp32 add into the stream, added in the engine's order ``(h + d_attn) + d_mlp`` (vqs_api.cpp:501-561,609-691);
  * norm outputs, Q/K/V (bias added in fp32 first), the attention output, activation outputs
    (quick_gelu / erf-GELU / gated gelu_new product, computed in fp32 on the fp32 accumulator), the projected image
    features and the encoder output are bf16 (GEMM epilogues, gemm.hip:65-74,286-372; elementwise.hip:93-164);
  * self-attention (attn.hip:366-595): 64-key tiles, scores in the log2 domain (one FMA with scale*log2e and the
    bias table pre-multiplied by log2e), a running max per query row that is only refreshed when some row of the
    32-query wave jumps by more than 2^6 (``RESCALE_THR``), P = 2^(t - m) rounded to bf16 BEFORE both the P.V
    product and the row sum (the sum is taken on the matrix pipe over the rounded P), fp32 accumulators, one
    division at the end, bf16 output;
  * decoder self-attention (attn.hip dec_attn_kernel): fp32 probabilities, bf16 output;
  * decoder cross-attention in the reassociated form the engine executes (vqs_api.cpp:781-812): q.Wk -> bf16,
    scores fp32, masked softmax normalised in fp32 THEN rounded to bf16, P.E -> bf16, ctx.Wv^T -> bf16;
  * lm_head logits, log-softmax and the score fold fp32.

What is NOT matched (and cannot be from the outside): the summation order inside an fp32 accumulation (MFMA k-order,
split-K slices, wave reductions) and the last ulp of v_exp_f32 / v_rcp_f32 / rsqrtf.  Those move a result by ~1e-7
relative, which flips the bf16 rounding of roughly one element in 10^4; everything else is bit-identical.  Matrix
products are therefore evaluated in float64 here (``acc``), so that the only summation noise is the engine's own.
The distance HIP <-> this oracle isolates "is the kernel arithmetic right" from "bf16 is bf16": the tests hold it
to north_star's 1e-3 on the label log-probs, where the distance to the fp32 oracle is the bf16 operand noise
(4e-3 .. 5e-2).

Two ways to use it.  FREE-RUNNING (``forward``): the whole pass on the CPU -- its log-probs sit at the bf16 noise
floor from the fp32 oracle, and so does the engine; but the two are NOT within 1e-3 of each other, and no two
evaluations of this arithmetic can be: one flipped bf16 rounding in a GEMM operand moves every output of that row by
~2^-8 |x| |w|, which flips ~sqrt(D) of the next stage's roundings -- the flips multiply by 30-100x per GEMM stage until
they are dense (measured: tools/rounding_chaos.py).  STAGE-LOCKED (``locked`` = the engine's own intermediates, read
through vqs_debug_tap): every op is evaluated on the ENGINE's inputs, its result compared with the engine's output of
that launch, and the engine's tensor -- not the oracle's -- is handed to the next op.  That checks every launch of a
pass at the bit level (expected: ~1e-4 of the elements off by one bf16 ulp) and is what the -m gpu tests assert at
full XL / XXL size (tests/test_gpu_stage_locked.py).

Pinning: this module adds roundings to a restatement that is itself pinned to the HF modules
(``tests/test_oracle_golden.py``); ``tests/test_engine_rounding_oracle.py`` pins the additions -- with every
rounding switched off (``round_fn`` = identity) it must reproduce ``Oracle`` to fp32 accuracy, the tiled attention
must equal a plain softmax when P is not rounded, and the reassociated cross-attention must equal the direct form.
"""
from __future__ import annotations

import numpy as np
import torch

from .clip_t5_oracle import Oracle, gelu_erf, layer_norm, quick_gelu, t5_rms_norm

LOG2E = float(np.float32(1.4426950408889634))      # the kernels' constexpr float LOG2E
NEG_BIG = -1.0e30                                  # attn.hip:69 (finite mask value)
RESCALE_THR = 6.0                                  # attn.hip:70, log2 units
KT = 64                                            # keys per tile
WAVE_ROWS = 32                                     # queries per wave (attn.hip:399)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round-to-nearest-even to bf16, returned as fp32 (v_cvt_pk_bf16_f32 is RNE)."""
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float32)


def split_bf16_round(x: torch.Tensor) -> torch.Tensor:
    """The value a SPLIT-bf16 tensor holds: hi = bf16(x), lo = bf16(x - hi), stored as two bf16 planes and consumed as
    hi + lo (two MFMA passes over the same weights, or one fp32 add in an elementwise kernel) -- 16 significant bits instead
    of 8 (the engine's precise decoder, elementwise.hip split_store)."""
    x = x.to(torch.float32)
    hi = bf16_round(x)
    return hi + bf16_round(x - hi)


def _fma32(a: torch.Tensor, b, c) -> torch.Tensor:
    """fp32 fused multiply-add: the product of two fp32 is exact in fp64, so one fp64 add + one rounding to fp32
    reproduces v_fma_f32 except for rare double-rounding ties."""
    return (a.double() * b + (c.double() if torch.is_tensor(c) else c)).float()


def gelu_new_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """gemm.hip:71-74 act_gelu_new: 0.5 x (1 + tanh(u)) written as x * sigmoid(2u) (HF activations.py:59-66)."""
    u2 = 2.0 * 0.7978845608028654 * (x + 0.044715 * x * x * x)
    return x * torch.sigmoid(u2)


def tiled_attention(q, k, v, scale: float, bias_table=None, key_len=None, round_fn=bf16_round, acc=torch.float64, causal=False,
                    round_out=None):
    """attn_fwd_dma_kernel<HAS_BIAS> (attn.hip:366-595) and attn_fwd_hd_kernel<128, CAUSAL> (attn.hip:468-672; same tiles,
    same softmax, key > query masked when ``causal``) restated on the CPU.  A tile the causal kernel skips for a 128-query
    block is a tile whose scores are all masked there: it moves neither the running max nor the sums, so masking is enough.

    q, k, v: [B, H, S, d] fp32 tensors holding bf16 values.  bias_table: [H, 2S-1] fp32, entry (key - query + S - 1),
    natural-log units (the kernel multiplies by log2e when it fills its LDS copies).  key_len: [B] or None.
    Returns [B, S, H*64] fp32 holding bf16 values (``round_fn`` applied to P, ``round_out`` -- default ``round_fn`` -- to
    the output: the two are separate rounding classes of the error attribution, tools/error_attribution.py)."""
    B, H, S, d = q.shape
    round_out = round_fn if round_out is None else round_out
    sl2 = float(np.float32(scale) * np.float32(LOG2E))
    s_raw = (q.to(acc) @ k.to(acc).transpose(-1, -2)).float()              # [B,H,S,S] fp32 accumulators of K.Q^T
    if bias_table is not None:
        rel = torch.arange(S)[None, :] - torch.arange(S)[:, None] + (S - 1)          # [query, key]
        b_l2 = (bias_table.float() * np.float32(LOG2E))[:, rel]                      # [H,S,S] fp32 (fill_bias_copies)
        t_all = _fma32(s_raw, sl2, b_l2[None])                                       # log2-domain scores
    else:
        t_all = None
    klen = torch.full((B,), S, dtype=torch.long) if key_len is None else key_len.clamp(max=S).long()
    keymask = torch.arange(S)[None, :] >= klen[:, None]                              # [B,S] True = masked
    cmask = (torch.arange(S)[None, :] > torch.arange(S)[:, None]) if causal else None  # [query, key] True = masked
    G = (S + WAVE_ROWS - 1) // WAVE_ROWS
    pad = G * WAVE_ROWS - S

    m_run = torch.full((B, H, S), NEG_BIG, dtype=torch.float32)
    osum = torch.zeros(B, H, S, dtype=torch.float32)
    o = torch.zeros(B, H, S, d, dtype=torch.float32)
    ntiles = (int(klen.max()) + KT - 1) // KT
    for kt in range(ntiles):
        kb, ke = kt * KT, min(kt * KT + KT, S)
        km = keymask[:, None, None, kb:ke]
        if cmask is not None:
            km = km | cmask[None, None, :, kb:ke]
        if t_all is not None:
            t = t_all[..., kb:ke].masked_fill(km, NEG_BIG)
            mx = t.max(-1).values
        else:
            sr = s_raw[..., kb:ke].masked_fill(km, NEG_BIG)
            mx = sr.max(-1).values * np.float32(sl2)
        # wave-wide decision: rows [32g, 32g+32) of one (sample, head) share the "somebody jumped" flag; rows >= S are
        # clamped copies of row S-1 and add nothing
        jump = mx > m_run + RESCALE_THR
        jp = torch.nn.functional.pad(jump, (0, pad)).reshape(B, H, G, WAVE_ROWS).any(-1, keepdim=True)
        jp = jp.expand(B, H, G, WAVE_ROWS).reshape(B, H, G * WAVE_ROWS)[..., :S]
        # samples whose tile loop has already ended (kb >= klen) execute nothing
        jp = jp & (kb < klen)[:, None, None]
        m_new = torch.maximum(m_run, mx)
        alpha = torch.exp2(m_run - m_new)
        alpha = torch.where(jp, alpha, torch.ones_like(alpha))
        m_run = torch.where(jp, m_new, m_run)
        o = o * alpha[..., None]
        osum = osum * alpha
        if t_all is not None:
            p = torch.exp2(t - m_run[..., None])
        else:
            p = torch.exp2(_fma32(sr, sl2, -m_run[..., None]))
        p = round_fn(p)
        live = (kb < klen)[:, None, None]
        p = torch.where(live[..., None], p, torch.zeros_like(p))
        osum = osum + p.to(acc).sum(-1).float()
        o = o + (p.to(acc) @ v[..., kb:ke, :].to(acc)).float()
    inv = torch.where(osum > 0, 1.0 / osum, torch.zeros_like(osum))
    out = round_out(o * inv[..., None])
    return out.transpose(1, 2).reshape(B, S, H * d)


def compare_tap(y: torch.Tensor, e: torch.Tensor, valid: torch.Tensor | None = None, mant_bits: int = 7) -> dict:
    """Distance of an oracle result `y` from the engine's tensor `e` of the same launch (both fp32, same shape), over the
    `valid` elements if given.  Per-element measure: the difference in units of the element's OWN ulp in the tensor's 16-bit type
    (bf16: 2^(exponent - 7) of the larger of the two values; `mant_bits` = 10 for an fp16 tensor) plus an absolute floor of 2^-18 of the tensor's top value -- the fp32 summation noise a
    result that cancels to nearly nothing still carries.  An absmax-relative bound alone lets a defect confined to small
    elements through."""
    d = (y - e).abs()
    if valid is not None:
        d = torch.where(valid.expand_as(d), d, torch.zeros_like(d))
        n = int(valid.expand_as(d).sum())
        ref_abs = float(torch.where(valid.expand_as(e), e.abs(), torch.zeros_like(e)).max())
    else:
        n = d.numel()
        ref_abs = float(e.abs().max())
    big = torch.maximum(y.abs(), e.abs())
    # bf16: 7 stored mantissa bits; an fp16 tensor (mant_bits = 10, the fp16 vision tower): 10, and nothing finer than its subnormal grid 2^-24
    own_ulp = torch.ldexp(torch.ones_like(big), torch.frexp(big).exponent - 1 - mant_bits)
    if mant_bits == 10:
        own_ulp = own_ulp.clamp_min(2.0 ** -24)
    own = d / (own_ulp + ref_abs * 2.0 ** -18 + 1e-37)
    return {"frac_diff": float((d > 0).sum()) / max(n, 1), "max_abs": float(d.max()), "ref_absmax": ref_abs, "n": n, "mant_bits": mant_bits,
            "max_own_ulps": float(own.max()), "frac_over_1_own_ulp": float((own > 1.0).sum()) / max(n, 1)}


class EngineRoundedOracle(Oracle):
    """See the module docstring.  ``round_fn`` = identity turns every rounding off (used to pin this class against
    ``Oracle``); ``acc`` is the dtype matrix products are accumulated in."""

    # every rounding site carries a class name "<stack>.<what>"; `classes` = the set of classes that round (None = all,
    # the engine's arithmetic).  One class at a time = the error attribution of tools/error_attribution.py.
    CLASSES = ("vit.norm", "vit.qkv", "vit.p", "vit.attn", "vit.act", "vit.delta", "vit.feat", "proj.mid", "proj.out",
               "enc.norm", "enc.qkv", "enc.p", "enc.attn", "enc.act", "enc.delta", "enc.out",
               "dec.norm", "dec.qkv", "dec.sattn", "dec.delta", "dec.cq", "dec.cqk", "dec.cprobs", "dec.cctx", "dec.cattn",
               "dec.act", "dec.out")

    # the engine's precise decoder (vqs_set_option "dec_precise", default 1; round 4): every decoder activation that the error
    # attribution (profiles/r4_error_attribution.md) shows to matter is a split-bf16 tensor or stays fp32 --
    #   split (hi + lo planes): norm outputs, self-attention output, P.E context, ctx.Wv output, gated FFN product, final norm output
    #   fp32, never rounded:    q|k|v of the self attention, the three deltas into the fp32 stream
    #   bf16 as before:         the cross-attention score path (cq, q.Wk, probabilities), which reads the hi plane of the norm output
    # option enc_fp16: the encoder's tensor classes held in IEEE fp16, and the linears that read fp16 weight copies
    ENC_FP16_CLASSES = ("enc.norm", "enc.qkv", "enc.p", "enc.attn")
    ENC_FP16_LINEARS = ("SelfAttention.q.weight", "SelfAttention.k.weight", "SelfAttention.v.weight", "SelfAttention.o.weight",
                        "DenseReluDense.wi_0.weight", "DenseReluDense.wi_1.weight")
    # option dec_fp16: the encoder's output and the precise decoder's cross-attention score path in IEEE fp16
    DEC_FP16_CLASSES = ("enc.out", "dec.cq", "dec.cqk", "dec.cprobs")
    DEC_SPLIT = ("dec.norm", "dec.sattn", "dec.cctx", "dec.cattn", "dec.act", "dec.out")
    DEC_FP32 = ("dec.qkv", "dec.delta")

    def __init__(self, cfg, weights, emulate="engine", round_fn=bf16_round, acc=torch.float64, classes=None, dec_precise=True,
                 split_classes=(), half_classes=(), vit_fp16=True, device="cpu", enc_fp16=True, dec_fp16=True, proj_shifts=(0, 0)):
        """`split_classes`: classes (of the tower / projector / encoder) to model as split-bf16 tensors instead of bf16 ones -- a
        what-if for tools/error_attribution.py, nothing the engine does today; the extra name "vit.v" splits the value heads only
        (q and k stay bf16: the score path).  `half_classes`: classes to model as IEEE fp16 tensors (11 significant bits instead of
        8, same MFMA rate); the weights of a stack with a half class are then fp16 too (bf16 -> fp16 is exact above 2^-14).
        `vit_fp16` = the engine's option of that name (vqs_set_option "vit_fp16"): every class of the tower and the projector's hidden
        tensor are fp16, the linear weights of both are the fp16 copies (the patch embedding keeps bf16 operands), and the projector's
        output is rounded ONCE, from the fp32 accumulator to the C ABI's bf16 feature tensor (round 4 rounded to fp16 first).
        `enc_fp16` = the engine's option of that name (round 5, default 1): the ATTENTION SIDE of the T5 encoder -- both norm outputs, q / k / v,
        the probabilities, the attention output -- is IEEE fp16 and q / k / v / o / wi_0 / wi_1 are read as fp16 copies; the sub-layer outputs,
        the gated product, the wo GEMM and the encoder's final output stay bf16 (vqs_api.cpp encoder_pass).
        `dec_fp16` = the engine's option of that name (round 5, default 1; only with the precise decoder): the encoder's OUTPUT and the decoder's
        cross-attention score path -- q (from the full split norm output, rounded once), q.Wk (fp16 copy of Wk), the probabilities -- are IEEE fp16
        tensors; P.E runs on fp16 operands and still leaves as a split tensor (vqs_api.cpp decoder_pass_precise)."""
        # proj_shifts = the engine's options (proj_fs_shift, proj_mid_shift) (round 6): with the fp16 tower the selected features are held as
        # fp16(x * 2^-fs) and the projector's hidden tensor as fp16(x * 2^-mid) -- the scales the bind-time range proof asks for (engine.py)
        self.proj_sigma = (2.0 ** -int(proj_shifts[0]), 2.0 ** -int(proj_shifts[1]))
        super().__init__(cfg, weights, device=device)     # device != "cpu": the what-if runs of tools/error_attribution.py evaluated on the GPU (under `with torch.device(dev)`)
        self.r = round_fn
        self.acc = acc
        self.dec_precise = bool(dec_precise)
        self.r2 = split_bf16_round if round_fn is bf16_round else round_fn
        unknown = set(split_classes) - set(self.CLASSES) - {"vit.v"}
        if unknown:
            raise ValueError(f"unknown split classes {sorted(unknown)}")
        self.split_extra = frozenset(split_classes)
        unknown = set(half_classes) - set(self.CLASSES)
        if unknown:
            raise ValueError(f"unknown half classes {sorted(unknown)}")
        self.vit_fp16 = bool(vit_fp16)           # default = what ships (the engine's default); False = the bf16 tower of rounds 1-3
        if self.vit_fp16:
            half_classes = tuple(half_classes) + tuple(c for c in self.CLASSES if c.startswith("vit.")) + ("proj.mid",)
        # enc_fp16 == "attn" (a what-if of tools/error_attribution.py, nothing the engine does): only the attention sub-block -- the FFN's norm
        # output and the wi weights stay bf16
        self.enc_ffn_bf16 = enc_fp16 == "attn"
        self.enc_fp16 = bool(enc_fp16)
        if self.enc_fp16:
            half_classes = tuple(half_classes) + self.ENC_FP16_CLASSES
        self.dec_fp16 = bool(dec_fp16) and self.dec_precise
        if self.dec_fp16:
            half_classes = tuple(half_classes) + self.DEC_FP16_CLASSES
        self.half_extra = frozenset(half_classes)
        self.half_stacks = frozenset(c.split(".")[0] for c in self.half_extra)
        self.rh = (lambda x: x.to(torch.float16).to(torch.float32)) if round_fn is bf16_round else round_fn
        if classes is not None:
            unknown = set(classes) - set(self.CLASSES)
            if unknown:
                raise ValueError(f"unknown rounding classes {sorted(unknown)}")
        self.classes = None if classes is None else frozenset(classes)
        self.locked = None       # dict tap name -> engine tensor: stage-locked mode
        self.record = None       # dict: free-running mode stores every named intermediate here
        self.report = {}         # stage-locked mode: tap name -> {"frac_diff", "max_abs", "ref_absmax", "n"}

    def _emit(self, name: str, y: torch.Tensor, valid: torch.Tensor | None = None) -> torch.Tensor:
        """Name an op result.  Free-running: record and pass through.  Stage-locked: compare with the engine's tensor of
        that name (over `valid` elements if given) and return the ENGINE's tensor, so the next op consumes what the
        engine's next launch consumed."""
        if self.locked is None:
            if self.record is not None:
                self.record[name] = y
            return y
        if name not in self.locked:
            raise KeyError(f"stage-locked run needs the engine tap {name!r}")
        mant = 10 if self.locked[name].dtype == torch.float16 else 7
        e = self.locked[name].detach().to(self.device, torch.float32)
        if mant == 10 and name in ("vit.feat_in", "vit.pmid"):          # fp16 tensors behind the projector's scales: back to true units
            e = e / self.proj_sigma[0 if name == "vit.feat_in" else 1]
        e = e.reshape(-1)[: y.numel()].reshape(y.shape) if e.numel() >= y.numel() and e.shape != y.shape else e
        self.report[name] = compare_tap(y, e, valid, mant)
        return e

    # ---------------------------------------------------------------------------------------------- helpers
    def rc(self, cls: str, x: torch.Tensor) -> torch.Tensor:
        """The rounding of class `cls`: ``round_fn`` when the class is switched on, identity otherwise."""
        assert cls in self.CLASSES, cls
        if self.classes is not None and cls not in self.classes:
            return x
        if self.dec_precise and cls in self.DEC_FP32:
            return x
        if (self.dec_precise and cls in self.DEC_SPLIT) or cls in self.split_extra:
            return self.r2(x)
        if cls in self.half_extra:
            return self.rh(x)
        return self.r(x)

    def _rc_scaled(self, cls: str, x: torch.Tensor, sigma: float) -> torch.Tensor:
        """rc() of a class the engine holds behind a power-of-two scale when it is an fp16 tensor (true units in and out)"""
        if sigma == 1.0 or cls not in self.half_extra:
            return self.rc(cls, x)
        return self.rc(cls, x * sigma) / sigma

    def rcf(self, cls: str):
        if self.classes is not None and cls not in self.classes:
            return lambda x: x
        return self.r2 if cls in self.split_extra else self.rh if cls in self.half_extra else self.r

    def _mm(self, x: torch.Tensor, wname: str, bname: str | None = None) -> torch.Tensor:
        """fp32 accumulator of an nn.Linear over bf16 operands (+ bias added in fp32, as the GEMM epilogues do)."""
        w = self.w[wname].detach().to(self.device)
        stack = {"vision": "vit", "mm_projector": "proj"}.get(wname.split(".")[0])
        if stack in self.half_stacks and "patch_embedding" not in wname:
            w = w.to(torch.float16)                 # what a bf16 checkpoint becomes in an fp16 tower
        if self.enc_fp16 and wname.startswith("encoder.") and wname.endswith(self.ENC_FP16_LINEARS) and not (self.enc_ffn_bf16 and "DenseReluDense" in wname):
            w = w.to(torch.float16)                 # ... and in the encoder's fp16 attention side (wo keeps bf16 operands)
        w = w.reshape(w.shape[0], -1).to(self.acc)
        y = (x.to(self.acc) @ w.t()).float()
        if bname is not None:
            y = y + self._w(bname)
        return y

    # ---------------------------------------------------------------------------------------------- vision tower
    def vision_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        v, rc = self.cfg.vision, self.rc
        B = pixel_values.shape[0]
        g, p = v.grid, v.patch
        x = self.r(pixel_values.to(torch.float32))                  # the engine receives bf16 pixels (an input, not a class)
        patches = x.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
        pe = self._emit("vit.patch_out", self._mm(patches, "vision.embeddings.patch_embedding.weight"))   # EPI_F32
        cls = self._w("vision.embeddings.class_embedding").reshape(1, 1, -1).expand(B, 1, -1)
        pre = torch.cat([cls, pe], dim=1) + self._w("vision.embeddings.position_embedding.weight")[None]
        h = self._emit("vit.h0", layer_norm(pre, self._w("vision.pre_layrnorm.weight"), self._w("vision.pre_layrnorm.bias"), v.ln_eps))   # fp32 out
        S = h.shape[1]
        for i in range(v.layers_run):
            pfx = f"vision.encoder.layers.{i}."
            t = f"vit.{i}."
            xn = self._emit(t + "xn0", rc("vit.norm", layer_norm(h, self._w(pfx + "layer_norm1.weight"), self._w(pfx + "layer_norm1.bias"), v.ln_eps)))

            def heads(nm):
                y = self._mm(xn, pfx + f"self_attn.{nm}_proj.weight", pfx + f"self_attn.{nm}_proj.bias")
                value_split = nm == "v" and "vit.v" in self.split_extra and (self.classes is None or "vit.qkv" in self.classes)
                y = self.r2(y) if value_split else rc("vit.qkv", y)
                return self._emit(t + nm, y.reshape(B, S, v.heads, v.head_dim).transpose(1, 2))

            att = tiled_attention(heads("q"), heads("k"), heads("v"), v.head_dim ** -0.5, None, None, self.rcf("vit.p"), self.acc,
                                  round_out=self.rcf("vit.attn"))
            att = self._emit(t + "attn", att)
            d_attn = self._emit(t + "d_attn", rc("vit.delta", self._mm(att, pfx + "self_attn.out_proj.weight", pfx + "self_attn.out_proj.bias")))
            h1 = h + d_attn
            xn = self._emit(t + "xn1", rc("vit.norm", layer_norm(h1, self._w(pfx + "layer_norm2.weight"), self._w(pfx + "layer_norm2.bias"), v.ln_eps)))
            mid = self._emit(t + "mid", rc("vit.act", quick_gelu(self._mm(xn, pfx + "mlp.fc1.weight", pfx + "mlp.fc1.bias"))))
            d_mlp = self._emit(t + "d_mlp", rc("vit.delta", self._mm(mid, pfx + "mlp.fc2.weight", pfx + "mlp.fc2.bias")))
            h = h1 + d_mlp
        return self._emit("vit.feat_in", self._rc_scaled("vit.feat", h[:, 1:], self.proj_sigma[0]))      # drop_cls_cast: 16-bit operand of the projector

    def projector(self, feats: torch.Tensor) -> torch.Tensor:
        rc = self.rc
        x = self._emit("vit.pmid", self._rc_scaled("proj.mid", gelu_erf(self._mm(feats, "mm_projector.0.weight", "mm_projector.0.bias")), self.proj_sigma[1]))
        y = self._mm(x, "mm_projector.2.weight", "mm_projector.2.bias")
        # round 5: the fp16-operand GEMM rounds its fp32 accumulator to the bf16 feature tensor directly (round 4: to fp16, then a cast)
        return self._emit("proj", rc("proj.out", y))

    # ---------------------------------------------------------------------------------------------- T5
    def _gated_ff(self, prefix: str, xn: torch.Tensor, cls: str = "enc.act") -> torch.Tensor:
        g = self._mm(xn, prefix + "wi_0.weight")
        l = self._mm(xn, prefix + "wi_1.weight")
        return self.rc(cls, gelu_new_sigmoid(g) * l)                     # EPI_GATED: fp32 accumulators -> one bf16 product

    def enc_bias_table(self, S: int) -> torch.Tensor:
        """[H, 2S-1] fp32, entry (key - query + S - 1): relpos_table_kernel (elementwise.hip:590-609)."""
        from .clip_t5_oracle import relative_position_bucket
        t = self.cfg.t5
        rel = np.arange(-(S - 1), S)
        bucket = relative_position_bucket(rel, True, t.rel_buckets, t.rel_max_distance)
        table = self._w("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")    # [buckets, H]
        return table[torch.from_numpy(bucket)].t().contiguous()

    def t5_encoder(self, emb: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
        t, rc = self.cfg.t5, self.rc
        B, S, _ = emb.shape
        klen = key_mask.sum(1)
        table = self.enc_bias_table(S)
        h = self._emit("enc.emb", emb)
        for i in range(t.layers):
            p = f"encoder.block.{i}."
            a = p + "layer.0.SelfAttention."
            n = f"enc.{i}."
            xn = self._emit(n + "xn0", rc("enc.norm", t5_rms_norm(h, self._w(p + "layer.0.layer_norm.weight"), t.ln_eps)))

            def heads(nm):
                return self._emit(n + nm, rc("enc.qkv", self._mm(xn, a + nm + ".weight")).reshape(B, S, t.heads, t.d_kv).transpose(1, 2))

            att = self._emit(n + "attn", tiled_attention(heads("q"), heads("k"), heads("v"), 1.0, table, klen, self.rcf("enc.p"), self.acc,
                                                         round_out=self.rcf("enc.attn")))
            d_attn = self._emit(n + "d_attn", rc("enc.delta", self._mm(att, a + "o.weight")))
            h1 = h + d_attn
            y1 = t5_rms_norm(h1, self._w(p + "layer.1.layer_norm.weight"), t.ln_eps)
            xn = self._emit(n + "xn1", self.r(y1) if (self.enc_ffn_bf16 and (self.classes is None or "enc.norm" in self.classes)) else rc("enc.norm", y1))
            ff = self._emit(n + "ff", self._gated_ff(p + "layer.1.DenseReluDense.", xn, "enc.act"))
            d_ff = self._emit(n + "d_ff", rc("enc.delta", self._mm(ff, p + "layer.1.DenseReluDense.wo.weight")))
            h = h1 + d_ff
        return self._emit("enc_out", rc("enc.out", t5_rms_norm(h, self._w("encoder.final_layer_norm.weight"), t.ln_eps)))

    def _dec_self_attention(self, qkv: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
        """dec_attn_kernel, self form: fp32 scores + bias, causal, fp32 softmax and P.V, bf16 output."""
        t = self.cfg.t5
        B, T, _ = qkv.shape
        q, k, v = (qkv[..., j * t.inner:(j + 1) * t.inner].reshape(B, T, t.heads, t.d_kv).transpose(1, 2) for j in range(3))
        s = (q.to(self.acc) @ k.to(self.acc).transpose(-1, -2)).float() + bias[None]
        causal = torch.tril(torch.ones(T, T, dtype=torch.bool))
        s = s.masked_fill(~causal, NEG_BIG)
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (e.to(self.acc) @ v.to(self.acc)).float() / e.sum(-1, keepdim=True)
        return self.rc("dec.sattn", o).transpose(1, 2).reshape(B, T, t.inner)

    def _s_pad(self, S: int) -> int:
        return (S + 63) // 64 * 64

    def _hi_plane(self, name: str, y: torch.Tensor):
        """HI plane of the split tensor `name` whose unrounded fp32 value is y: stage-locked = the engine's own plane 0 (handed over
        as "<name>#hi" by taps_to_values), free-running = bf16(y).  None when the tensor is not split (bf16 decoder / class off)."""
        if not self.dec_precise or (self.classes is not None and "dec.norm" not in self.classes):
            return None
        if self.locked is not None:
            return self.locked[name + "#hi"].detach().to(self.device, torch.float32).reshape(y.shape)
        hi = self.r(y)
        if self.record is not None:
            self.record[name + "#hi"] = hi
        return hi

    def _dec_cross_attention(self, p: str, xn: torch.Tensor, enc_out: torch.Tensor, klen: torch.Tensor, n: str = "", xn_hi=None) -> torch.Tensor:
        """Reassociated cross-attention (vqs_api.cpp:781-812): (q_h Wk_h) E^T, masked softmax, (P E) Wv_h^T."""
        t, rc = self.cfg.t5, self.rc
        B, T, D = xn.shape
        S = enc_out.shape[1]
        H, dk = t.heads, t.d_kv
        # precise decoder: the score path reads the HI plane of the split norm output (xn_hi = bf16 of the norm's fp32 result: what
        # it read before round 4).  Not bf16(hi + lo): lo is itself rounded and can land hi + lo exactly on a tie.
        xq = xn if (xn_hi is None or self.dec_fp16) else xn_hi       # option dec_fp16: q from BOTH planes (stacked-plane GEMM), rounded once to fp16
        q = self._emit(n + "cq", rc("dec.cq", self._mm(xq, p + "q.weight"))).reshape(B, T, H, dk)
        wk = self.w[p + "k.weight"].detach().to(self.device)
        if self.dec_fp16:
            wk = wk.to(torch.float16)               # the fp16 copy of Wk^T made at bind time
        wk = wk.to(self.acc).reshape(H, dk, D)
        wv = self.w[p + "v.weight"].detach().to(self.device).to(self.acc).reshape(H, dk, D)
        qk = self._emit(n + "cqk", rc("dec.cqk", torch.einsum("bthd,hdD->bthD", q.to(self.acc), wk).float()))      # "cross q.Wk" -> bf16
        # the engine's score / probability rows are S_pad wide (keys >= S: zero-padded E^T columns; keys >= enc_len: masked)
        Sp = self._s_pad(S)
        sc = torch.einsum("bthD,bsD->bths", qk.to(self.acc), enc_out.to(self.acc)).float()  # "cross scores" fp32
        sc = torch.nn.functional.pad(sc, (0, Sp - S))
        live = (torch.arange(Sp)[None, :] < klen[:, None])[:, None, None, :]
        sc = self._emit(n + "cscores", sc, valid=live)
        sc = sc.masked_fill(~live, float("-inf"))
        e = torch.exp(sc - sc.max(-1, keepdim=True).values)
        pr = self._emit(n + "cprobs", rc("dec.cprobs", e * (1.0 / e.sum(-1, keepdim=True))))                # masked_softmax_kernel
        ctx = rc("dec.cctx", torch.einsum("bths,bsD->bthD", pr[..., :S].to(self.acc), enc_out.to(self.acc)).float())   # "cross P.E" -> bf16
        ctx = self._emit(n + "cctx", ctx)
        out = rc("dec.cattn", torch.einsum("bthD,hdD->bthd", ctx.to(self.acc), wv).float())               # "cross ctx.Wv" -> bf16
        return self._emit(n + "cattn", out.reshape(B, T, H * dk))

    def t5_decoder(self, dec_ids: torch.Tensor, enc_out: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
        from .clip_t5_oracle import relative_position_bucket
        t, rc = self.cfg.t5, self.rc
        B, T = dec_ids.shape
        klen = key_mask.sum(1)
        h = self._emit("dec.emb", self._w("shared.weight")[dec_ids])
        dist = np.arange(T)                                           # query - key
        bucket = relative_position_bucket(-dist, False, t.rel_buckets, t.rel_max_distance)
        tab = self._w("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")[torch.from_numpy(bucket)]   # [T,H]
        tq = torch.arange(T)
        idx = (tq[:, None] - tq[None, :]).clamp(min=0)
        self_bias = tab[idx].permute(2, 0, 1)                         # [H,Tq,Tk]; entries above the diagonal are masked
        for i in range(t.dec_layers):
            p = f"decoder.block.{i}."
            sa = p + "layer.0.SelfAttention."
            n = f"dec.{i}."
            xn = self._emit(n + "xn0", rc("dec.norm", t5_rms_norm(h, self._w(p + "layer.0.layer_norm.weight"), t.ln_eps)))
            qkv = self._emit(n + "qkv", torch.cat([rc("dec.qkv", self._mm(xn, sa + nm + ".weight")) for nm in ("q", "k", "v")], dim=-1))
            sattn = self._emit(n + "sattn", self._dec_self_attention(qkv, self_bias))
            h = h + self._emit(n + "d_self", rc("dec.delta", self._mm(sattn, sa + "o.weight")))
            y1 = t5_rms_norm(h, self._w(p + "layer.1.layer_norm.weight"), t.ln_eps)
            xn = self._emit(n + "xn1", rc("dec.norm", y1))
            ca = p + "layer.1.EncDecAttention."
            h = h + self._emit(n + "d_cross", rc("dec.delta", self._mm(self._dec_cross_attention(ca, xn, enc_out, klen, n, self._hi_plane(n + "xn1", y1)),
                                                                        ca + "o.weight")))
            xn = self._emit(n + "xn2", rc("dec.norm", t5_rms_norm(h, self._w(p + "layer.2.layer_norm.weight"), t.ln_eps)))
            ffp = p + "layer.2.DenseReluDense."
            ff = self._emit(n + "ff", self._gated_ff(ffp, xn, "dec.act"))
            h = h + self._emit(n + "d_ff", rc("dec.delta", self._mm(ff, ffp + "wo.weight")))
        return self._emit("dec_out", rc("dec.out", t5_rms_norm(h, self._w("decoder.final_layer_norm.weight"), t.ln_eps)))

    def lm_logits(self, dec_out: torch.Tensor) -> torch.Tensor:
        return self._emit("logits", self._mm(dec_out, "lm_head.weight"))                    # EPI_F32

    # ---------------------------------------------------------------------------------------------- stage-locked run
    TAP_NAMES_VIT = ("xn0", "q", "k", "v", "attn", "d_attn", "xn1", "mid", "d_mlp")
    TAP_NAMES_ENC = ("xn0", "q", "k", "v", "attn", "d_attn", "xn1", "ff", "d_ff")
    TAP_NAMES_DEC = ("xn0", "qkv", "sattn", "d_self", "xn1", "cq", "cqk", "cscores", "cprobs", "cctx", "cattn", "d_cross", "xn2", "ff", "d_ff")

    def tap_shapes(self, n_img: int, B: int, L: int, T: int):
        """name -> (shape, dtype) of every engine tap a stage-locked run consumes (include/vqs.h vqs_debug_tap)."""
        v, t = self.cfg.vision, self.cfg.t5
        NS, NP, S = n_img * v.seq, n_img * v.n_patches, L - 1 + v.n_patches
        M, MT, Sp = B * S, B * T, self._s_pad(S)
        bf, f32 = torch.bfloat16, torch.float32
        vt = torch.float16 if self.vit_fp16 else bf             # the tower's 16-bit tensors
        out = {"vit.patch_out": ((NP, v.hidden), f32), "vit.h0": ((NS, v.hidden), f32),
               "vit.feat_in": ((NP, v.hidden), vt), "vit.pmid": ((NP, t.d_model), vt),
               "enc.emb": ((M, t.d_model), f32), "dec.emb": ((MT, t.d_model), f32)}
        for i in range(v.layers_run):
            for nm in self.TAP_NAMES_VIT:
                out[f"vit.{i}.{nm}"] = ((NS, v.mlp if nm == "mid" else v.hidden), vt)
        et = torch.float16 if self.enc_fp16 else bf            # the encoder's attention-side tensors (option enc_fp16)
        for i in range(t.layers):
            for nm in self.TAP_NAMES_ENC:
                out[f"enc.{i}.{nm}"] = ((M, {"ff": t.d_ff, "xn0": t.d_model, "xn1": t.d_model, "d_attn": t.d_model, "d_ff": t.d_model}.get(nm, t.inner)),
                                        et if nm in ("xn0", "q", "k", "v", "attn", "xn1") else bf)
        width = {"xn0": t.d_model, "xn1": t.d_model, "xn2": t.d_model, "d_self": t.d_model, "d_cross": t.d_model, "d_ff": t.d_model,
                 "qkv": 3 * t.inner, "sattn": t.inner, "cq": t.inner, "cattn": t.inner, "ff": t.d_ff,
                 "cqk": t.heads * t.d_model, "cctx": t.heads * t.d_model, "cscores": t.heads * Sp, "cprobs": t.heads * Sp}
        # precise decoder: "split" = the engine taps the two bf16 planes [2][rows][width] (value = plane 0 + plane 1, see
        # taps_to_values); q|k|v and the three sub-layer outputs are fp32
        split = {"xn0", "xn1", "xn2", "sattn", "cctx", "cattn", "ff"} if self.dec_precise else set()
        f32_names = {"cscores"} | ({"qkv", "d_self", "d_cross", "d_ff"} if self.dec_precise else set())
        for i in range(t.dec_layers):
            for nm in self.TAP_NAMES_DEC:
                d16 = torch.float16 if (self.dec_fp16 and nm in ("cq", "cqk", "cprobs")) else bf       # option dec_fp16: the score path's tensors
                out[f"dec.{i}.{nm}"] = ((MT, width[nm]), "split" if nm in split else (f32 if nm in f32_names else d16))
        return out

    @staticmethod
    def tap_alloc(shapes, device="cpu"):
        """Tap buffers for tap_shapes(): a "split" entry is two bf16 planes [2, rows, width]."""
        return {n: (torch.empty((2,) + tuple(shape), dtype=torch.bfloat16, device=device) if dt == "split"
                    else torch.empty(shape, dtype=dt, device=device)) for n, (shape, dt) in shapes.items()}

    @staticmethod
    def taps_to_values(shapes, bufs, device="cpu"):
        """Tap buffers -> the tensors' values on `device` (own copies): split entries become fp32 hi + lo."""
        out = {}
        for n, b in bufs.items():
            if shapes[n][1] == "split":
                out[n] = (b[0].float() + b[1].float()).to(device)
                out[n + "#hi"] = b[0].float().to(device)
            else:
                out[n] = b.to(device).clone() if torch.device(device).type != "cpu" else b.cpu()
        return out

    def forward_locked(self, taps, pixel_values, img_index, input_ids, labels):
        """Stage-locked pass: `taps` = the engine's intermediates of ITS pass over the same inputs (every name of
        tap_shapes() plus "proj", "enc_out", "dec_out", "logits").  Returns (report, label_logprobs computed from the
        engine's logits).  report[name] = how far the engine's tensor is from this oracle's evaluation of that one op on
        the engine's own inputs."""
        self.locked, self.report = taps, {}
        try:
            out = self.forward(pixel_values, img_index, input_ids, labels)
        finally:
            self.locked = None
        return self.report, out["label_logprobs"]
