"""Weight inventory for CLIP-FlanT5 and a seeded generator for checkpoint-free runs.

Tensor names are the HF ``state_dict`` keys of the three modules the reference assembles
(/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:201,222-229):

* ``vision.*``        -- ``CLIPVisionModel`` (HF models/clip/modeling_clip.py:594-656),
* ``mm_projector.*``  -- the ``mlp2x_gelu`` projector (Linear, GELU, Linear),
* everything else     -- ``T5ForConditionalGeneration`` (HF models/t5/modeling_t5.py:898-1069)
  with an *untied* ``lm_head.weight`` (flan-t5 checkpoints are untied).

No checkpoint is reachable offline, so benchmarks and parity tests draw every tensor from
a per-tensor seeded normal distribution whose scales follow HF's ``_init_weights``
(HF models/t5/modeling_t5.py:563-616).  The same materialised tensors are handed to the
HIP engine and to the CPU oracle, so parity never depends on the generator itself.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, List, Tuple

import torch

from .config import ClipT5Config

# kind -> how the generator scales it
#   ("normal", std) | ("ones", jitter) | ("zeros", jitter)
Spec = Tuple[str, Tuple[int, ...], Tuple[str, float]]


def weight_specs(cfg: ClipT5Config, lm_head_gain: float = 1.0) -> List[Spec]:
    v, t = cfg.vision, cfg.t5
    specs: List[Spec] = []
    hd = v.hidden
    kpatch = 3 * v.patch * v.patch
    specs += [
        ("vision.embeddings.class_embedding", (hd,), ("normal", 0.5)),
        ("vision.embeddings.patch_embedding.weight", (hd, 3, v.patch, v.patch), ("normal", kpatch ** -0.5)),
        ("vision.embeddings.position_embedding.weight", (v.seq, hd), ("normal", 0.1)),
        ("vision.pre_layrnorm.weight", (hd,), ("ones", 0.1)),
        ("vision.pre_layrnorm.bias", (hd,), ("zeros", 0.02)),
    ]
    for i in range(v.layers):
        p = f"vision.encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            specs.append((p + f"self_attn.{nm}.weight", (hd, hd), ("normal", hd ** -0.5)))
            specs.append((p + f"self_attn.{nm}.bias", (hd,), ("zeros", 0.02)))
        specs += [
            (p + "layer_norm1.weight", (hd,), ("ones", 0.1)),
            (p + "layer_norm1.bias", (hd,), ("zeros", 0.02)),
            (p + "mlp.fc1.weight", (v.mlp, hd), ("normal", hd ** -0.5)),
            (p + "mlp.fc1.bias", (v.mlp,), ("zeros", 0.02)),
            (p + "mlp.fc2.weight", (hd, v.mlp), ("normal", v.mlp ** -0.5)),
            (p + "mlp.fc2.bias", (hd,), ("zeros", 0.02)),
            (p + "layer_norm2.weight", (hd,), ("ones", 0.1)),
            (p + "layer_norm2.bias", (hd,), ("zeros", 0.02)),
        ]
    D, I, F = t.d_model, t.inner, t.d_ff
    specs += [
        ("mm_projector.0.weight", (D, hd), ("normal", hd ** -0.5)),
        ("mm_projector.0.bias", (D,), ("zeros", 0.02)),
        ("mm_projector.2.weight", (D, D), ("normal", D ** -0.5)),
        ("mm_projector.2.bias", (D,), ("zeros", 0.02)),
        ("shared.weight", (t.vocab, D), ("normal", 1.0)),
    ]

    def attn(prefix: str, rel: bool):
        out = [
            (prefix + "q.weight", (I, D), ("normal", (D * t.d_kv) ** -0.5)),
            (prefix + "k.weight", (I, D), ("normal", D ** -0.5)),
            (prefix + "v.weight", (I, D), ("normal", D ** -0.5)),
            (prefix + "o.weight", (D, I), ("normal", I ** -0.5)),
        ]
        if rel:
            out.append((prefix + "relative_attention_bias.weight", (t.rel_buckets, t.heads), ("normal", 0.5)))
        return out

    def ff(prefix: str):
        return [
            (prefix + "wi_0.weight", (F, D), ("normal", D ** -0.5)),
            (prefix + "wi_1.weight", (F, D), ("normal", D ** -0.5)),
            (prefix + "wo.weight", (D, F), ("normal", F ** -0.5)),
        ]

    for i in range(t.layers):
        p = f"encoder.block.{i}."
        specs += attn(p + "layer.0.SelfAttention.", i == 0)
        specs.append((p + "layer.0.layer_norm.weight", (D,), ("ones", 0.1)))
        specs += ff(p + "layer.1.DenseReluDense.")
        specs.append((p + "layer.1.layer_norm.weight", (D,), ("ones", 0.1)))
    specs.append(("encoder.final_layer_norm.weight", (D,), ("ones", 0.1)))
    for i in range(t.dec_layers):
        p = f"decoder.block.{i}."
        specs += attn(p + "layer.0.SelfAttention.", i == 0)
        specs.append((p + "layer.0.layer_norm.weight", (D,), ("ones", 0.1)))
        specs += attn(p + "layer.1.EncDecAttention.", False)
        specs.append((p + "layer.1.layer_norm.weight", (D,), ("ones", 0.1)))
        specs += ff(p + "layer.2.DenseReluDense.")
        specs.append((p + "layer.2.layer_norm.weight", (D,), ("ones", 0.1)))
    specs.append(("decoder.final_layer_norm.weight", (D,), ("ones", 0.1)))
    specs.append(("lm_head.weight", (t.vocab, D), ("normal", lm_head_gain * D ** -0.5)))
    return specs


def param_count(cfg: ClipT5Config) -> int:
    n = 0
    for _, shape, _ in weight_specs(cfg):
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


def _tensor_seed(seed: int, name: str) -> int:
    return (seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF


def iter_seeded_weights(cfg: ClipT5Config, seed: int = 0, device="cpu", dtype=torch.bfloat16,
                        lm_head_gain: float = 1.0) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (name, tensor) in inventory order.  Each tensor has its own seed, so any
    subset can be regenerated independently (order-independent)."""
    dev = torch.device(device)
    for name, shape, (kind, scale) in weight_specs(cfg, lm_head_gain):
        g = torch.Generator(device=dev)
        g.manual_seed(_tensor_seed(seed, name))
        x = torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        if kind == "normal":
            x.mul_(scale)
        elif kind == "ones":
            x.mul_(scale).add_(1.0)
        else:
            x.mul_(scale)
        yield name, x.to(dtype)


def make_seeded_weights(cfg: ClipT5Config, seed: int = 0, device="cpu", dtype=torch.bfloat16,
                        lm_head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    return dict(iter_seeded_weights(cfg, seed, device, dtype, lm_head_gain))


# --- mapping a real CLIP-FlanT5 checkpoint onto the inventory ------------------------------
# zhiqiulin/clip-flant5-* state dicts keep the T5 keys as-is, nest the projector under
# ``encoder.mm_projector`` and the CLIP tower under ``encoder.vision_tower.vision_tower``
# (v3.0 layout, not present in /root/reference -- see SURVEY.md §0; unverified offline).
_CKPT_PREFIXES = (
    ("encoder.vision_tower.vision_tower.vision_model.", "vision."),
    ("vision_tower.vision_model.", "vision."),
    ("vision_model.", "vision."),
    ("encoder.mm_projector.", "mm_projector."),
)


def read_checkpoint_dir(path: str) -> Dict[str, torch.Tensor]:
    """Every weight shard of a local HF directory: ``*.safetensors`` and/or ``pytorch_model*.bin`` (the published
    zhiqiulin/clip-flant5-* repos ship .bin shards; layout [RECALLED], unverified offline).  .bin shards are read with
    ``torch.load(weights_only=True, mmap=True)`` -- tensors only, no pickled code."""
    import os
    sd: Dict[str, torch.Tensor] = {}
    names = sorted(os.listdir(path))
    for f in names:
        if f.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd.update(load_file(os.path.join(path, f)))
    for f in names:
        if f.endswith(".bin") and f.startswith("pytorch_model"):
            part = torch.load(os.path.join(path, f), map_location="cpu", weights_only=True, mmap=True)
            for k, v in part.items():
                sd.setdefault(k, v)
    if not sd:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    return sd


def canonical_name(ckpt_key: str) -> str:
    for src, dst in _CKPT_PREFIXES:
        if ckpt_key.startswith(src):
            return dst + ckpt_key[len(src):]
    return ckpt_key


def load_checkpoint_weights(cfg: ClipT5Config, state_dict: Dict[str, torch.Tensor], device,
                            dtype=torch.bfloat16, vision_state_dict: Dict[str, torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Map a loaded ``state_dict`` (shards of a local HF dir) onto the inventory; raises KeyError listing what is
    missing.  ``vision_state_dict``: the CLIP tower when it lives in its own directory (the reference loads it
    separately -- ``get_vision_tower().load_model()``, mm_utils.py:236-237 -- from openai/clip-vit-large-patch14-336,
    keys ``vision_model.*``); tensors of the main checkpoint win on a clash.  The whole model is cast to bf16, as the
    reference does (mm_utils.py:228)."""
    canon = {canonical_name(k): v for k, v in (vision_state_dict or {}).items() if canonical_name(k).startswith("vision.")}
    canon.update({canonical_name(k): v for k, v in state_dict.items()})
    out, missing = {}, []
    for name, shape, _ in weight_specs(cfg):
        if name not in canon:
            missing.append(name)
            continue
        w = canon[name]
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {tuple(w.shape)} != expected {shape}")
        out[name] = w.to(device=device, dtype=dtype).contiguous()
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from checkpoint, e.g. {missing[:5]}")
    return out
