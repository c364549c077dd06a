"""CPU ORACLE for the CLIP-FlanT5 VQAScore hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 tensor arithmetic (matmul / softmax / elementwise on CPU
tensors; no HF modules, no fused attention, no GPU), the forward pass that the reference
runs through HuggingFace for ``clip-flant5-{xl,xxl}``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and
only as the checker.  The product path (``t2v_metrics_amd``) never imports this module.

Provenance.  The CLIP-FlanT5 model code is NOT in /root/reference (v3.1 dropped it; see
SURVEY.md §0), so the algorithm is restated from
  * the third-party dependency that actually holds the arithmetic: ``transformers`` (installed
    5.15.0; reference pin ``transformers>=4.52.0``, /root/reference/pyproject.toml:26) --
    cited below as ``HF:<path>:<lines>`` relative to the transformers package root,
  * the reference's surviving call sites and helpers
    (/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:128-139,164-179,182-241;
    /root/reference/t2v_metrics/constants.py:3-8; /root/reference/t2v_metrics/score.py:104-106),
  * the public v3.0 recipe (select layer -2 / drop CLS, mlp2x_gelu projector, splice at -200,
    score = exp(-mean CE)) as recorded in SURVEY.md §8a rows a10-a12, a21 ([RECALLED] there).

Pinning.  The reference holds no golden vector for this path.  The oracle is
therefore pinned against outputs of the HF modules themselves, run in this container by
``oracle/make_golden.py`` on seeded tiny configurations and committed under ``tests/golden/``
(``tests/test_oracle_golden.py``).  Parity against a real checkpoint remains unpinned.

Numerics.  "Truth" = bf16-rounded weights (what the reference holds after
``model.to(dtype=torch.bfloat16)``, mm_utils.py:228) up-cast to fp32, all arithmetic in fp32.
Weights are up-cast one tensor at a time (``_w``) so that XXL fits in host RAM.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

IGNORE_INDEX = -100        # /root/reference/t2v_metrics/constants.py:6
IMAGE_TOKEN_INDEX = -200   # /root/reference/t2v_metrics/constants.py:7


# ----------------------------------------------------------------------------------------------
# integer helpers
# ----------------------------------------------------------------------------------------------
def relative_position_bucket(relative_position: np.ndarray, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> np.ndarray:
    """HF:models/t5/modeling_t5.py:216-262 (`_relative_position_bucket`), integer for integer.

    relative_position = memory_position - query_position.  The logarithmic branch is computed
    in fp32 exactly as torch does (log of an fp32 quotient, divided by a Python-double constant
    that torch applies as an fp32 scalar, times an integer, truncated toward zero).
    """
    rp = np.asarray(relative_position, dtype=np.int64)
    buckets = np.zeros_like(rp)
    nb = num_buckets
    if bidirectional:
        nb //= 2
        buckets = buckets + (rp > 0).astype(np.int64) * nb
        rp = np.abs(rp)
    else:
        rp = -np.minimum(rp, 0)
    max_exact = nb // 2
    is_small = rp < max_exact
    with np.errstate(divide="ignore"):
        # torch: log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
        q = rp.astype(np.float32) / np.float32(max_exact)
        lg = np.log(q, dtype=np.float32)
        val = lg / np.float32(math.log(max_distance / max_exact))
        val = val * np.float32(nb - max_exact)
    val = np.where(np.isfinite(val), val, np.float32(0.0))
    if_large = max_exact + val.astype(np.int64)       # .to(torch.long) truncates toward zero
    if_large = np.minimum(if_large, nb - 1)
    return buckets + np.where(is_small, rp, if_large)


def shift_right(labels: torch.Tensor, decoder_start_id: int = 0, pad_id: int = 0) -> torch.Tensor:
    """HF:models/t5/modeling_t5.py:618-637."""
    out = labels.new_zeros(labels.shape)
    out[..., 1:] = labels[..., :-1]
    out[..., 0] = decoder_start_id
    out = out.masked_fill(out == IGNORE_INDEX, pad_id)
    return out


# ----------------------------------------------------------------------------------------------
# elementwise pieces
# ----------------------------------------------------------------------------------------------
def quick_gelu(x):      # HF:activations.py:117-123
    return x * torch.sigmoid(1.702 * x)


def gelu_new(x):        # HF:activations.py:59-66 (tanh form)
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def gelu_erf(x):        # torch.nn.GELU() of the mlp2x_gelu projector (exact erf form)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps):   # torch.nn.LayerNorm as used at HF:models/clip/modeling_clip.py:357-360
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def t5_rms_norm(x, w, eps):     # HF:models/t5/modeling_t5.py:59-72
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


class Oracle:
    """fp32 restatement of CLIP-FlanT5 scoring over a dict of (bf16) weights.

    ``Oracle(cfg, weights, emulate="engine")`` returns the rounding-matched variant
    (``oracle/clip_t5_engine_rounding.py``): the same function with a round-to-bf16 wherever the HIP engine
    holds a bf16 tensor -- the checker for north_star's 1e-3 bound."""

    def __new__(cls, cfg=None, weights=None, emulate: Optional[str] = None, **kw):
        if emulate is not None and cls is Oracle:
            if emulate != "engine":
                raise ValueError(f"unknown emulation mode {emulate!r}")
            from .clip_t5_engine_rounding import EngineRoundedOracle
            return object.__new__(EngineRoundedOracle)
        return object.__new__(cls)

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], emulate: Optional[str] = None, device="cpu"):
        """device: where the fp32 arithmetic is evaluated.  "cpu" = the oracle proper.  bench.py's parity sample also evaluates
        this same code in torch fp32 on the GPU (under ``with torch.device(dev)``) to get fp32 truth for more pairs than the
        host cores can do in the bench's time budget; that evaluation is cross-checked against the CPU one on the pairs both do."""
        self.cfg = cfg
        self.w = weights
        self.device = torch.device(device)

    def _w(self, name: str) -> torch.Tensor:
        return self.w[name].detach().to(self.device, torch.float32)

    # ------------------------------------------------------------------------------------------
    # vision tower: HF:models/clip/modeling_clip.py
    # ------------------------------------------------------------------------------------------
    def vision_embeddings(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """:202-218 -- conv(14x14, stride 14, no bias) as a matmul over (c,ky,kx)-ordered patches,
        prepend class embedding, add learned position table; then pre_layrnorm (:642)."""
        v = self.cfg.vision
        B = pixel_values.shape[0]
        x = pixel_values.to(torch.float32)
        g, p = v.grid, v.patch
        # [B,3,g,p,g,p] -> [B,g,g,3,p,p] -> [B, g*g, 3*p*p]
        patches = x.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
        wpe = self._w("vision.embeddings.patch_embedding.weight").reshape(v.hidden, -1)
        pe = patches @ wpe.t()
        cls = self._w("vision.embeddings.class_embedding").reshape(1, 1, -1).expand(B, 1, -1)
        h = torch.cat([cls, pe], dim=1) + self._w("vision.embeddings.position_embedding.weight")[None]
        return layer_norm(h, self._w("vision.pre_layrnorm.weight"), self._w("vision.pre_layrnorm.bias"), v.ln_eps)

    def vision_layer(self, h: torch.Tensor, i: int) -> torch.Tensor:
        """CLIPEncoderLayer :353-383; attention :280-335 with eager math :259-277."""
        v = self.cfg.vision
        p = f"vision.encoder.layers.{i}."
        B, S, _ = h.shape
        x = layer_norm(h, self._w(p + "layer_norm1.weight"), self._w(p + "layer_norm1.bias"), v.ln_eps)

        def proj(nm):
            y = x @ self._w(p + f"self_attn.{nm}.weight").t() + self._w(p + f"self_attn.{nm}.bias")
            return y.reshape(B, S, v.heads, v.head_dim).transpose(1, 2)

        q, k, val = proj("q_proj"), proj("k_proj"), proj("v_proj")
        att = torch.softmax((q @ k.transpose(-1, -2)) * (v.head_dim ** -0.5), dim=-1)
        o = (att @ val).transpose(1, 2).reshape(B, S, v.hidden)
        o = o @ self._w(p + "self_attn.out_proj.weight").t() + self._w(p + "self_attn.out_proj.bias")
        h = h + o
        x = layer_norm(h, self._w(p + "layer_norm2.weight"), self._w(p + "layer_norm2.bias"), v.ln_eps)
        x = quick_gelu(x @ self._w(p + "mlp.fc1.weight").t() + self._w(p + "mlp.fc1.bias"))
        x = x @ self._w(p + "mlp.fc2.weight").t() + self._w(p + "mlp.fc2.bias")
        return h + x

    def vision_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """hidden_states[select_layer][:, 1:]  (v3.0 ``feature_select``: layer -2, 'patch')."""
        h = self.vision_embeddings(pixel_values)
        for i in range(self.cfg.vision.layers_run):
            h = self.vision_layer(h, i)
        return h[:, 1:]

    def projector(self, feats: torch.Tensor) -> torch.Tensor:
        """mlp2x_gelu: Linear(1024->D) -> GELU(erf) -> Linear(D->D)  (SURVEY.md §8a row a11)."""
        x = feats @ self._w("mm_projector.0.weight").t() + self._w("mm_projector.0.bias")
        x = gelu_erf(x)
        return x @ self._w("mm_projector.2.weight").t() + self._w("mm_projector.2.bias")

    # ------------------------------------------------------------------------------------------
    # embed + splice (sentinel from mm_utils.py:164-179)
    # ------------------------------------------------------------------------------------------
    def splice(self, proj: torch.Tensor, img_index: torch.Tensor, input_ids: torch.Tensor):
        """Returns (inputs_embeds [B,S_e,D], key_mask [B,S_e] bool, lengths [B]).

        Per sample: embed(ids before sentinel) ++ 576 projected rows ++ embed(ids after sentinel);
        ids equal to pad (0) are removed first (``attention_mask = input_ids.ne(pad)``), the
        result is right-padded with zero rows to S_e = max length and masked."""
        t = self.cfg.t5
        shared = self._w("shared.weight")
        B, L = input_ids.shape
        P = proj.shape[1]
        rows, lens = [], []
        for b in range(B):
            ids = input_ids[b]
            ids = ids[ids != t.pad_id]
            pos = (ids == IMAGE_TOKEN_INDEX).nonzero()
            if pos.numel() != 1:
                raise ValueError("each prompt must contain exactly one image sentinel")
            pos = int(pos[0, 0])
            parts = [shared[ids[:pos]], proj[int(img_index[b])], shared[ids[pos + 1:]]]
            r = torch.cat(parts, dim=0)
            rows.append(r)
            lens.append(r.shape[0])
        S_e = L - 1 + P           # the engine's static layout: L ids incl. one sentinel
        emb = torch.zeros(B, S_e, t.d_model)
        mask = torch.zeros(B, S_e, dtype=torch.bool)
        for b, r in enumerate(rows):
            emb[b, : r.shape[0]] = r
            mask[b, : r.shape[0]] = True
        return emb, mask, torch.tensor(lens)

    # ------------------------------------------------------------------------------------------
    # T5: HF:models/t5/modeling_t5.py
    # ------------------------------------------------------------------------------------------
    def rel_bias(self, prefix: str, q_len: int, k_len: int, bidirectional: bool) -> torch.Tensor:
        """compute_bias :264-279 -> [H, q_len, k_len]."""
        t = self.cfg.t5
        ctx = np.arange(q_len)[:, None]
        mem = np.arange(k_len)[None, :]
        bucket = relative_position_bucket(mem - ctx, bidirectional, t.rel_buckets, t.rel_max_distance)
        table = self._w(prefix + "relative_attention_bias.weight")      # [buckets, H]
        return table[torch.from_numpy(bucket)].permute(2, 0, 1)

    def t5_attention(self, prefix, x, kv, add_bias):
        """T5Attention.forward :281-369 with eager math :144-173; scaling = 1.0 (:196-197).
        add_bias: [B or 1, H or 1, Tq, Tk] additive (position bias + mask)."""
        t = self.cfg.t5
        B, Tq, _ = x.shape
        Tk = kv.shape[1]
        q = (x @ self._w(prefix + "q.weight").t()).reshape(B, Tq, t.heads, t.d_kv).transpose(1, 2)
        k = (kv @ self._w(prefix + "k.weight").t()).reshape(B, Tk, t.heads, t.d_kv).transpose(1, 2)
        v = (kv @ self._w(prefix + "v.weight").t()).reshape(B, Tk, t.heads, t.d_kv).transpose(1, 2)
        s = q @ k.transpose(2, 3) + add_bias
        a = torch.softmax(s, dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, Tq, t.inner)
        return o @ self._w(prefix + "o.weight").t()

    def t5_ff(self, prefix, x):
        """T5DenseGatedActDense :97-127."""
        g = gelu_new(x @ self._w(prefix + "wi_0.weight").t())
        l = x @ self._w(prefix + "wi_1.weight").t()
        return (g * l) @ self._w(prefix + "wo.weight").t()

    def t5_encoder(self, emb: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
        """T5Stack (encoder) :663-750; bias from block 0 shared by all layers (:739-742)."""
        t = self.cfg.t5
        B, S, _ = emb.shape
        neg = torch.finfo(torch.float32).min
        bias = self.rel_bias("encoder.block.0.layer.0.SelfAttention.", S, S, True)[None]
        add = bias + torch.where(key_mask, 0.0, neg)[:, None, None, :]
        h = emb
        for i in range(t.layers):
            p = f"encoder.block.{i}."
            x = t5_rms_norm(h, self._w(p + "layer.0.layer_norm.weight"), t.ln_eps)
            h = h + self.t5_attention(p + "layer.0.SelfAttention.", x, x, add)
            x = t5_rms_norm(h, self._w(p + "layer.1.layer_norm.weight"), t.ln_eps)
            h = h + self.t5_ff(p + "layer.1.DenseReluDense.", x)
        return t5_rms_norm(h, self._w("encoder.final_layer_norm.weight"), t.ln_eps)

    def t5_decoder(self, dec_ids: torch.Tensor, enc_out: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
        """T5Stack (decoder) :663-750 with T5Block :435-509: causal self-attention with the
        unidirectional bucket bias, cross-attention with zero bias (:337-342) + key mask, gated FFN."""
        t = self.cfg.t5
        B, T = dec_ids.shape
        neg = torch.finfo(torch.float32).min
        h = self._w("shared.weight")[dec_ids]
        causal = torch.where(torch.tril(torch.ones(T, T, dtype=torch.bool)), 0.0, neg)
        self_add = (self.rel_bias("decoder.block.0.layer.0.SelfAttention.", T, T, False) + causal)[None]
        cross_add = torch.where(key_mask, 0.0, neg)[:, None, None, :]
        for i in range(t.dec_layers):
            p = f"decoder.block.{i}."
            x = t5_rms_norm(h, self._w(p + "layer.0.layer_norm.weight"), t.ln_eps)
            h = h + self.t5_attention(p + "layer.0.SelfAttention.", x, x, self_add)
            x = t5_rms_norm(h, self._w(p + "layer.1.layer_norm.weight"), t.ln_eps)
            h = h + self.t5_attention(p + "layer.1.EncDecAttention.", x, enc_out, cross_add)
            x = t5_rms_norm(h, self._w(p + "layer.2.layer_norm.weight"), t.ln_eps)
            h = h + self.t5_ff(p + "layer.2.DenseReluDense.", x)
        return t5_rms_norm(h, self._w("decoder.final_layer_norm.weight"), t.ln_eps)

    def lm_logits(self, dec_out: torch.Tensor) -> torch.Tensor:
        """:1042-1047 with scale_decoder_outputs False (untied head, flan-t5)."""
        return dec_out @ self._w("lm_head.weight").t()

    # ------------------------------------------------------------------------------------------
    # scoring tail
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def label_logprobs(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """log_softmax(logits)[label] per position; 0 where label == -100."""
        lp = torch.log_softmax(logits.to(torch.float32), dim=-1)
        safe = labels.clamp(min=0)
        out = lp.gather(-1, safe[..., None])[..., 0]
        return torch.where(labels == IGNORE_INDEX, torch.zeros_like(out), out)

    @staticmethod
    def scores_from_logprobs(lp: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """exp(-CrossEntropyLoss(mean over non-ignored positions)) per sample."""
        valid = (labels != IGNORE_INDEX).to(torch.float32)
        return torch.exp((lp * valid).sum(-1) / valid.sum(-1).clamp(min=1.0))

    # ------------------------------------------------------------------------------------------
    def forward(self, pixel_values: torch.Tensor, img_index: torch.Tensor, input_ids: torch.Tensor,
                labels: torch.Tensor, return_stages: bool = False):
        """pixel_values [N_img,3,H,W]; img_index [B] -> rows of pixel_values; input_ids [B,L] with one
        -200 each (0 = pad); labels [B,T] (-100 = pad)."""
        t = self.cfg.t5
        with torch.no_grad():
            feats = self.vision_features(pixel_values)
            proj = self.projector(feats)
            emb, mask, lens = self.splice(proj, img_index, input_ids)
            enc = self.t5_encoder(emb, mask)
            dec_ids = shift_right(labels, t.decoder_start_id, t.pad_id)
            dec = self.t5_decoder(dec_ids, enc, mask)
            logits = self.lm_logits(dec)
            lp = self.label_logprobs(logits, labels)
            scores = self.scores_from_logprobs(lp, labels)
        out = {"label_logprobs": lp, "scores": scores}
        if return_stages:
            out.update(vit_feats=feats, proj=proj, enc_in=emb, enc_mask=mask, enc_out=enc, dec_out=dec, logits=logits)
        return out

    def generate(self, pixel_values: torch.Tensor, img_index: torch.Tensor, input_ids: torch.Tensor, max_new_tokens: int,
                 return_margins: bool = False):
        """Greedy search as HF GenerationMixin runs it for T5ForConditionalGeneration (generation/utils.py greedy branch:
        decoder_start_token_id = 0, next token = argmax of the last position's logits, no KV cache needed for the
        arithmetic).  Returns int64 [B, max_new_tokens] WITHOUT the start token and without EOS truncation; with
        return_margins also the top-1 minus top-2 logit gap of every step (tests skip near-ties)."""
        t = self.cfg.t5
        with torch.no_grad():
            feats = self.vision_features(pixel_values)
            proj = self.projector(feats)
            emb, mask, lens = self.splice(proj, img_index, input_ids)
            enc = self.t5_encoder(emb, mask)
            B = input_ids.shape[0]
            dec_ids = torch.full((B, 1), t.decoder_start_id, dtype=torch.long)
            margins = []
            for _ in range(max_new_tokens):
                logits = self.lm_logits(self.t5_decoder(dec_ids, enc, mask))[:, -1]
                top2 = logits.topk(2, dim=-1).values
                margins.append(top2[:, 0] - top2[:, 1])
                dec_ids = torch.cat([dec_ids, logits.argmax(-1, keepdim=True)], dim=1)
        if return_margins:
            return dec_ids[:, 1:], torch.stack(margins, dim=1)
        return dec_ids[:, 1:]

