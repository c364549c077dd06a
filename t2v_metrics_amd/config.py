"""Architecture constants of the CLIP-FlanT5 VQAScore path.

The reference loads these from HF configs at run time (``model_cls.from_pretrained``,
/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:201,222-229).  They are
restated here as plain data so that neither the HIP engine nor the oracle needs a
checkpoint directory to know its shapes.

Vision tower = openai/clip-vit-large-patch14-336 (field list:
HF models/clip/configuration_clip.py:97-109); language model = google/flan-t5-{xl,xxl}
(field list: HF models/t5/configuration_t5.py:44-62).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict


@dataclass(frozen=True)
class VisionConfig:
    hidden: int = 1024
    layers: int = 24            # layers present in the checkpoint
    select_layer: int = -2      # hidden_states[-2]  -> layers-1 transformer layers are run
    heads: int = 16
    mlp: int = 4096
    patch: int = 14
    image: int = 336
    ln_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def seq(self) -> int:          # CLS + patches
        return self.n_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def layers_run(self) -> int:
        # hidden_states has layers+1 entries (entry 0 = output of pre_layrnorm)
        return self.layers + 1 + self.select_layer if self.select_layer < 0 else self.select_layer


@dataclass(frozen=True)
class T5Config:
    d_model: int = 2048
    heads: int = 32
    d_kv: int = 64
    d_ff: int = 5120
    layers: int = 24
    dec_layers: int = 24
    vocab: int = 32128
    rel_buckets: int = 32
    rel_max_distance: int = 128
    ln_eps: float = 1e-6
    pad_id: int = 0
    eos_id: int = 1
    decoder_start_id: int = 0

    @property
    def inner(self) -> int:
        return self.heads * self.d_kv


@dataclass(frozen=True)
class ClipT5Config:
    name: str = "clip-flant5-xl"
    vision: VisionConfig = field(default_factory=VisionConfig)
    t5: T5Config = field(default_factory=T5Config)

    def to_dict(self):
        return asdict(self)

    # ---- algorithmic FLOPs (2 per MAC; GEMMs + QK^T + PV only) -- SURVEY.md §8(d) ----
    def flops_vit(self) -> float:
        v = self.vision
        s, d, f = v.seq, v.hidden, v.mlp
        per_layer = 8 * s * d * d + 4 * s * d * f + 4 * s * s * d
        patch = 2 * v.n_patches * (3 * v.patch * v.patch) * d
        return v.layers_run * per_layer + patch

    def flops_projector(self) -> float:
        return 2 * self.vision.n_patches * (self.vision.hidden * self.t5.d_model + self.t5.d_model ** 2)

    def flops_encoder(self, s_e: int) -> float:
        t = self.t5
        return t.layers * (8 * s_e * t.d_model * t.inner + 6 * s_e * t.d_model * t.d_ff + 4 * s_e * s_e * t.inner)

    def flops_decoder(self, s_e: int, T: int) -> float:
        t = self.t5
        per = (12 * T * t.d_model * t.inner + 4 * s_e * t.d_model * t.inner + 6 * T * t.d_model * t.d_ff
               + 4 * T * T * t.inner + 4 * T * s_e * t.inner)
        return t.dec_layers * per + 2 * T * t.d_model * t.vocab

    def flops_pair(self, s_e: int, T: int) -> float:
        return self.flops_vit() + self.flops_projector() + self.flops_encoder(s_e) + self.flops_decoder(s_e, T)


CLIP_FLANT5_XL = ClipT5Config(name="clip-flant5-xl", vision=VisionConfig(),
                              t5=T5Config(d_model=2048, heads=32, d_kv=64, d_ff=5120))
CLIP_FLANT5_XXL = ClipT5Config(name="clip-flant5-xxl", vision=VisionConfig(),
                               t5=T5Config(d_model=4096, heads=64, d_kv=64, d_ff=10240))

# Small configurations used by parity tests (same structure, every dimension shrunk but
# kept legal for the HIP kernels: head_dim 64, hidden sizes multiples of 64).
TINY = ClipT5Config(
    name="tiny",
    vision=VisionConfig(hidden=128, layers=3, heads=2, mlp=256, patch=14, image=56),
    t5=T5Config(d_model=128, heads=2, d_kv=64, d_ff=192, layers=2, dec_layers=2, vocab=512),
)
SMALL = ClipT5Config(
    name="small",
    vision=VisionConfig(hidden=256, layers=4, heads=4, mlp=512, patch=14, image=112),
    t5=T5Config(d_model=256, heads=4, d_kv=64, d_ff=640, layers=3, dec_layers=3, vocab=1024),
)

CONFIGS = {c.name: c for c in (CLIP_FLANT5_XL, CLIP_FLANT5_XXL, TINY, SMALL)}


def get_config(name: str) -> ClipT5Config:
    return CONFIGS[name]
