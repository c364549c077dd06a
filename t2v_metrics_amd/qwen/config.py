"""Architecture constants of Qwen2.5-VL as the reference scores with it
(/root/reference/t2v_metrics/models/vqascore_models/qwen2vl_model.py:110-133 loads
``Qwen2_5_VLForConditionalGeneration.from_pretrained``; field lists: HF
models/qwen2_5_vl/configuration_qwen2_5_vl.py).  Plain data, no checkpoint directory needed.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class QwenVisionConfig:
    depth: int = 32
    hidden: int = 1280
    heads: int = 16
    mlp: int = 3420                     # SwiGLU intermediate, biases on all three projections
    in_channels: int = 3
    patch: int = 14
    temporal_patch: int = 2
    spatial_merge: int = 2
    window: int = 112                   # pixels; 112 / 2 / 14 = 4 merged cells = 8 x 8 patches per window
    fullatt_blocks: Tuple[int, ...] = (7, 15, 23, 31)
    out_hidden: int = 3584
    tokens_per_second: int = 2
    rms_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @property
    def patch_dim(self) -> int:          # flattened Conv3d receptive field (HF modeling_qwen2_5_vl.py:99-122)
        return self.in_channels * self.temporal_patch * self.patch * self.patch

    @property
    def merge_unit(self) -> int:
        return self.spatial_merge * self.spatial_merge


@dataclass(frozen=True)
class QwenTextConfig:
    vocab: int = 152064
    hidden: int = 3584
    layers: int = 28
    heads: int = 28
    kv_heads: int = 4
    mlp: int = 18944
    rope_theta: float = 1000000.0
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    rms_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclass(frozen=True)
class Qwen25VLConfig:
    name: str = "qwen2.5-vl-7b"
    vision: QwenVisionConfig = field(default_factory=QwenVisionConfig)
    text: QwenTextConfig = field(default_factory=QwenTextConfig)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653


QWEN25_VL_7B = Qwen25VLConfig()
QWEN_TINY = Qwen25VLConfig(
    name="qwen-tiny",
    vision=QwenVisionConfig(depth=4, hidden=64, heads=2, mlp=96, window=56, fullatt_blocks=(1, 3), out_hidden=128),
    text=QwenTextConfig(vocab=512, hidden=128, layers=2, heads=4, kv_heads=2, mlp=256, mrope_section=(4, 6, 6)),
    image_token_id=4, video_token_id=5, vision_start_token_id=6, vision_end_token_id=7,
)
QWEN_SMALL = Qwen25VLConfig(
    name="qwen-small",
    vision=QwenVisionConfig(depth=4, hidden=320, heads=4, mlp=200, window=112, fullatt_blocks=(3,), out_hidden=256),   # 80-wide heads, mlp not a multiple of 32
    text=QwenTextConfig(vocab=1024, hidden=256, layers=3, heads=2, kv_heads=1, mlp=384, mrope_section=(16, 24, 24)),
    image_token_id=4, video_token_id=5, vision_start_token_id=6, vision_end_token_id=7,
)
QWEN_SMALL_C80 = Qwen25VLConfig(
    name="qwen-small-c80",          # the 7B tower's head geometry in small: 80-wide heads whose q | k | v ranges are whole 128-column blocks
    vision=QwenVisionConfig(depth=4, hidden=640, heads=8, mlp=712, window=112, fullatt_blocks=(1, 3), out_hidden=256),
    text=QwenTextConfig(vocab=1024, hidden=256, layers=2, heads=2, kv_heads=1, mlp=384, mrope_section=(16, 24, 24)),
    image_token_id=4, video_token_id=5, vision_start_token_id=6, vision_end_token_id=7,
)
QWEN_TINY_G7 = Qwen25VLConfig(
    name="qwen-tiny-g7",            # the 7B language model's grouped-query ratio in small: 7 query heads per key/value head (7 x 64-wide heads)
    vision=QwenVisionConfig(depth=4, hidden=64, heads=2, mlp=96, window=56, fullatt_blocks=(1, 3), out_hidden=448),
    text=QwenTextConfig(vocab=512, hidden=448, layers=2, heads=7, kv_heads=1, mlp=256, mrope_section=(8, 12, 12)),
    image_token_id=4, video_token_id=5, vision_start_token_id=6, vision_end_token_id=7,
)
_CONFIGS = {c.name: c for c in (QWEN25_VL_7B, QWEN_TINY, QWEN_SMALL, QWEN_SMALL_C80, QWEN_TINY_G7)}


def get_qwen_config(name: str) -> Qwen25VLConfig:
    return _CONFIGS[name]
