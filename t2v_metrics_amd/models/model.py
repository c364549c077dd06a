"""Plugin base class of the scoring API.

Interface contract (what a scorer plugin must provide) follows /root/reference/t2v_metrics/models/model.py:10-47:
a constructor taking ``(model_name, device, cache_dir)`` that ends by calling ``load_model()``, plus ``load_images`` and
``forward``; ``self.image_loader`` is the decode hook wrappers call per path.
"""
from abc import ABC, abstractmethod
from pathlib import Path
from typing import List

import numpy as np
import torch
from PIL import Image

from ..constants import HF_CACHE_DIR


def image_loader(image_path) -> Image.Image:
    """Decode one image path to an RGB PIL image.  ``.npy`` files hold OpenCV-style BGR arrays [H, W, 3] and are
    channel-flipped (reference model.py:10-14); every other suffix is handed to PIL."""
    if Path(str(image_path)).suffix.lower() == '.npy':
        bgr = np.load(image_path)
        return Image.fromarray(np.ascontiguousarray(bgr[..., ::-1]), 'RGB')
    with Image.open(image_path) as im:
        return im.convert("RGB")


class ScoreModel(ABC):
    """Base of every scorer plugin: stores the three constructor arguments, makes sure the cache directory exists and
    triggers ``load_model()``."""

    def __init__(self, model_name='clip-flant5-xxl', device='cuda', cache_dir=HF_CACHE_DIR):
        self.model_name, self.device, self.cache_dir = model_name, device, cache_dir
        Path(self.cache_dir).mkdir(parents=True, exist_ok=True)
        self.image_loader = image_loader
        self.load_model()

    @abstractmethod
    def load_model(self):
        """Build everything the plugin needs (weights, tokenizer, engine)."""

    @abstractmethod
    def load_images(self, image: List[str]) -> torch.Tensor:
        """Decode + preprocess the given paths; the result lives on ``self.device``."""

    @abstractmethod
    def forward(self, images: List[str], texts: List[str]) -> torch.Tensor:
        """One score per (image, text) pair, pairs given as two parallel lists."""
