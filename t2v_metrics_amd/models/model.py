"""Plugin base class of the scoring API (interface of /root/reference/t2v_metrics/models/model.py:10-47)."""
from abc import ABC, abstractmethod
from typing import List
import os

import numpy as np
import torch
from PIL import Image

from ..constants import HF_CACHE_DIR


def image_loader(image_path):
    """``.npy`` arrays are BGR (OpenCV order) and are flipped to RGB; everything else goes through PIL."""
    if str(image_path).split('.')[-1] == 'npy':
        return Image.fromarray(np.load(image_path)[:, :, [2, 1, 0]], 'RGB')
    return Image.open(image_path).convert("RGB")


class ScoreModel(ABC):
    def __init__(self, model_name='clip-flant5-xxl', device='cuda', cache_dir=HF_CACHE_DIR):
        self.model_name = model_name
        self.device = device
        self.cache_dir = cache_dir
        if not os.path.exists(self.cache_dir):
            os.makedirs(self.cache_dir)
        self.image_loader = image_loader
        self.load_model()

    @abstractmethod
    def load_model(self):
        """Load the model, tokenizer, etc."""

    @abstractmethod
    def load_images(self, image: List[str]) -> torch.Tensor:
        """Load the image(s) and return a preprocessed tensor on self.device."""

    @abstractmethod
    def forward(self, images: List[str], texts: List[str]) -> torch.Tensor:
        """n scores for n (image, text) pairs."""
