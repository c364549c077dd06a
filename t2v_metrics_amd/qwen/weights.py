"""Weight inventory of Qwen2.5-VL (names = HF ``Qwen2_5_VLForConditionalGeneration.state_dict()`` keys) and a seeded
generator for checkpoint-free runs (no checkpoint is reachable offline); same per-tensor seeding scheme as
t2v_metrics_amd/weights.py."""
from __future__ import annotations

import zlib
from typing import Dict, List, Tuple

import torch

from .config import Qwen25VLConfig

Spec = Tuple[str, Tuple[int, ...], Tuple[str, float]]


def qwen_weight_specs(cfg: Qwen25VLConfig, lm_head_gain: float = 1.0) -> List[Spec]:
    v, t = cfg.vision, cfg.text
    s: List[Spec] = [("model.visual.patch_embed.proj.weight", (v.hidden, v.in_channels, v.temporal_patch, v.patch, v.patch),
                      ("normal", v.patch_dim ** -0.5))]
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        s += [(p + "norm1.weight", (v.hidden,), ("ones", 0.1)), (p + "norm2.weight", (v.hidden,), ("ones", 0.1)),
              (p + "attn.qkv.weight", (3 * v.hidden, v.hidden), ("normal", v.hidden ** -0.5)),
              (p + "attn.qkv.bias", (3 * v.hidden,), ("zeros", 0.05)),
              (p + "attn.proj.weight", (v.hidden, v.hidden), ("normal", 0.5 * v.hidden ** -0.5)),
              (p + "attn.proj.bias", (v.hidden,), ("zeros", 0.02)),
              (p + "mlp.gate_proj.weight", (v.mlp, v.hidden), ("normal", v.hidden ** -0.5)),
              (p + "mlp.gate_proj.bias", (v.mlp,), ("zeros", 0.02)),
              (p + "mlp.up_proj.weight", (v.mlp, v.hidden), ("normal", v.hidden ** -0.5)),
              (p + "mlp.up_proj.bias", (v.mlp,), ("zeros", 0.02)),
              (p + "mlp.down_proj.weight", (v.hidden, v.mlp), ("normal", 0.5 * v.mlp ** -0.5)),
              (p + "mlp.down_proj.bias", (v.hidden,), ("zeros", 0.02))]
    mh = v.hidden * v.merge_unit
    s += [("model.visual.merger.ln_q.weight", (v.hidden,), ("ones", 0.1)),
          ("model.visual.merger.mlp.0.weight", (mh, mh), ("normal", mh ** -0.5)),
          ("model.visual.merger.mlp.0.bias", (mh,), ("zeros", 0.02)),
          ("model.visual.merger.mlp.2.weight", (v.out_hidden, mh), ("normal", mh ** -0.5)),
          ("model.visual.merger.mlp.2.bias", (v.out_hidden,), ("zeros", 0.02)),
          ("model.language_model.embed_tokens.weight", (t.vocab, t.hidden), ("normal", 1.0))]
    kv = t.kv_heads * t.head_dim
    for i in range(t.layers):
        p = f"model.language_model.layers.{i}."
        s += [(p + "input_layernorm.weight", (t.hidden,), ("ones", 0.1)),
              (p + "self_attn.q_proj.weight", (t.hidden, t.hidden), ("normal", t.hidden ** -0.5)),
              (p + "self_attn.q_proj.bias", (t.hidden,), ("zeros", 0.05)),
              (p + "self_attn.k_proj.weight", (kv, t.hidden), ("normal", t.hidden ** -0.5)),
              (p + "self_attn.k_proj.bias", (kv,), ("zeros", 0.05)),
              (p + "self_attn.v_proj.weight", (kv, t.hidden), ("normal", t.hidden ** -0.5)),
              (p + "self_attn.v_proj.bias", (kv,), ("zeros", 0.05)),
              (p + "self_attn.o_proj.weight", (t.hidden, t.hidden), ("normal", 0.5 * t.hidden ** -0.5)),
              (p + "post_attention_layernorm.weight", (t.hidden,), ("ones", 0.1)),
              (p + "mlp.gate_proj.weight", (t.mlp, t.hidden), ("normal", t.hidden ** -0.5)),
              (p + "mlp.up_proj.weight", (t.mlp, t.hidden), ("normal", t.hidden ** -0.5)),
              (p + "mlp.down_proj.weight", (t.hidden, t.mlp), ("normal", 0.5 * t.mlp ** -0.5))]
    s += [("model.language_model.norm.weight", (t.hidden,), ("ones", 0.1)),
          ("lm_head.weight", (t.vocab, t.hidden), ("normal", lm_head_gain * t.hidden ** -0.5))]
    return s


def make_seeded_qwen_weights(cfg: Qwen25VLConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16,
                             lm_head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for name, shape, (kind, scale) in qwen_weight_specs(cfg, lm_head_gain):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        x = torch.randn(shape, generator=g, dtype=torch.float32) * scale
        if kind == "ones":
            x = x + 1.0
        out[name] = x.to(dtype).to(device)
    return out


# --- mapping a published Qwen2.5-VL checkpoint onto the inventory -----------------------------------------------------
# The safetensors shards of Qwen/Qwen2.5-VL-*-Instruct carry the pre-4.52 key layout (``visual.*``, ``model.layers.*``,
# ``model.embed_tokens.weight``, ``model.norm.weight``, ``lm_head.weight``); HF renames them on load to the in-memory
# names this inventory uses (transformers conversion_mapping for Qwen2_5_VLForConditionalGeneration:
# ``^visual`` -> ``model.visual``, ``^model(?!\.(language_model|visual))`` -> ``model.language_model``).  Both layouts
# are accepted.
import re as _re

_LEGACY_VISUAL = _re.compile(r"^visual\.")
_LEGACY_TEXT = _re.compile(r"^model\.(?!language_model\.|visual\.)")


def canonical_qwen_name(key: str) -> str:
    if _LEGACY_VISUAL.match(key):
        return "model." + key
    if _LEGACY_TEXT.match(key):
        return "model.language_model." + key[len("model."):]
    return key


def load_qwen_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Every ``*.safetensors`` shard of a local HF directory, keys canonicalised (legacy or in-memory layout)."""
    import os
    from safetensors.torch import load_file
    out: Dict[str, torch.Tensor] = {}
    for f in sorted(os.listdir(path)):
        if f.endswith(".safetensors"):
            for k, w in load_file(os.path.join(path, f)).items():
                out[canonical_qwen_name(k)] = w
    if not out:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    return out
