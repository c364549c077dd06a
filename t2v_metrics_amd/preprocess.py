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

from typing import Iterable, List, Sequence

import numpy as np
import torch
from PIL import Image

from .models.vqascore_models.mm_utils import expand2square

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _resize_shortest_edge(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return img
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    return img.resize((nw, nh), resample=Image.BICUBIC)


def _center_crop(arr: np.ndarray, size: int) -> np.ndarray:
    """arr [H,W,C]; crops (or zero-pads, as HF does) to size x size around the centre."""
    h, w = arr.shape[:2]
    top = (h - size) // 2
    left = (w - size) // 2
    if top >= 0 and left >= 0:
        return arr[top: top + size, left: left + size]
    out = np.zeros((size, size, arr.shape[2]), dtype=arr.dtype)
    nh, nw = max(size, h), max(size, w)
    padded = np.zeros((nh, nw, arr.shape[2]), dtype=arr.dtype)
    pt, pl = int(np.ceil((nh - h) / 2)), int(np.ceil((nw - w) / 2))
    padded[pt: pt + h, pl: pl + w] = arr
    top, left = (nh - size) // 2, (nw - size) // 2
    out[:] = padded[top: top + size, left: left + size]
    return out


def clip_preprocess_u8(img: Image.Image, image_size: int = 336, pad_to_square: bool = True) -> np.ndarray:
    """The integer part of the preprocessing (pad, PIL bicubic resize, centre crop) -> uint8 [image_size, image_size, 3];
    the float part (rescale + normalise + bf16) runs on the GPU (vqs_normalize_u8) with the same fp32 arithmetic."""
    img = img.convert("RGB")
    if pad_to_square:
        img = expand2square(img, tuple(int(x * 255) for x in OPENAI_CLIP_MEAN))
    img = _resize_shortest_edge(img, image_size)
    return np.ascontiguousarray(_center_crop(np.asarray(img), image_size))


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
