"""Image preprocessing of the CLIP-FlanT5 path, host side.

Restates what the reference does before the vision tower:
  1. ``expand2square`` with the CLIP mean colour when ``image_aspect_ratio == 'pad'``
     (/root/reference/t2v_metrics/models/vqascore_models/mm_utils.py:128-139,188,235);
  2. HF ``CLIPImageProcessor`` (HF models/clip/image_processing_clip.py:22-33, PIL backend
     HF image_processing_backends.py:521-660): bicubic resize of the shortest edge to 336, centre crop 336,
     x 1/255, normalise by OPENAI_CLIP_MEAN/STD, channels first.
Decode/resize stay on the CPU with PIL for bit-compatibility with the reference (row K11 of SURVEY.md §8 a-bis);
the output is handed to the device as bf16.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
from PIL import Image

# the integer half (pad, resize, crop -> uint8) lives in a torch-free module that also runs as a worker process (imgpool.py)
from ._imgprep import (OPENAI_CLIP_MEAN, OPENAI_CLIP_STD, _center_crop, _resize_shortest_edge, clip_preprocess_u8,  # noqa: F401
                       expand2square)


def clip_preprocess(img: Image.Image, image_size: int = 336, pad_to_square: bool = True) -> np.ndarray:
    """PIL RGB image -> float32 [3, image_size, image_size]."""
    arr = clip_preprocess_u8(img, image_size, pad_to_square).astype(np.float32) * np.float32(1.0 / 255.0)
    arr = (arr - np.asarray(OPENAI_CLIP_MEAN, dtype=np.float32)) / np.asarray(OPENAI_CLIP_STD, dtype=np.float32)
    return np.ascontiguousarray(arr.transpose(2, 0, 1))


def preprocess_batch(images: Sequence[Image.Image], image_size: int = 336, pad_to_square: bool = True) -> torch.Tensor:
    """-> float32 tensor [N,3,S,S] on the CPU (pinned if a GPU is present, for an async H2D copy)."""
    out = torch.empty(len(images), 3, image_size, image_size, dtype=torch.float32)
    for i, im in enumerate(images):
        out[i] = torch.from_numpy(clip_preprocess(im, image_size, pad_to_square))
    return out
