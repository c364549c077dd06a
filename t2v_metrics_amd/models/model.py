"""Plugin base class of the scoring API.

Interface contract (what a scorer plugin must provide) follows /root/reference/t2v_metrics/models/model.py:10-47:
a constructor taking ``(model_name, device, cache_dir)`` that ends by calling ``load_model()``, plus ``load_images`` and
``forward``; ``self.image_loader`` is the decode hook wrappers call per path.
"""
from abc import ABC, abstractmethod
from pathlib import Path
from typing import List

import torch

from ..constants import HF_CACHE_DIR


from .._imgprep import image_loader  # noqa: E402,F401  (reference model.py:10-14; defined in the torch-free module the image workers run)


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
