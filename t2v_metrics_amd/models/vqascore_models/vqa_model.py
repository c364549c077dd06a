"""VQAScore plugin interface (/root/reference/t2v_metrics/models/vqascore_models/vqa_model.py:7-18)."""
from abc import abstractmethod
from typing import List

import torch

from ..model import ScoreModel


class VQAScoreModel(ScoreModel):
    @abstractmethod
    def forward(self, images: List[str], texts: List[str], question_template: str,
                answer_template: str) -> torch.Tensor:
        """n scores for n (image, text) pairs.
        question_template / answer_template: strings with an optional {} replaced by the text."""
