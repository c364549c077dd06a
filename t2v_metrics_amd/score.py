"""Score base class: the M x N scoring call and the dataset loop.

Same public surface as /root/reference/t2v_metrics/score.py:18-156 (``Score(model, device, cache_dir, **kw)``,
``forward(images, texts, **kw) -> Tensor[M,N]``, ``batch_forward(dataset, batch_size, **kw) -> Tensor[n,n_vis,n_txt]``)
with the two hot loops rebuilt as real batches:

  * reference ``forward`` (score.py:104-106) calls the model once per image with the image path repeated N times,
    so every image is decoded, preprocessed and pushed through the ViT N times.  Here each distinct image is
    encoded once and all M*N pairs go through the T5 passes in engine-sized batches (``model.forward_grid``).
  * reference ``batch_forward`` (score.py:143-153) iterates DataLoader batches but still scores ONE pair per model
    call.  Here every DataLoader batch is flattened into one pair list (images deduplicated) and scored in one call.
  * with ``torch.distributed`` initialised, ``batch_forward`` shards the samples over ranks (contiguous blocks) and
    all-gathers the scores (t2v_metrics_amd/sharding.py) -- the reference has no multi-GPU path.

Video inputs: the reference falls back to extracting frames with ffmpeg/cv2 and concatenating them into one image
for image-only models (score.py:72-101).  That pre-processing is outside the hot path (SURVEY.md §2 "frame helpers
OUT OF SCOPE"); video paths raise NotImplementedError here instead of being silently mis-scored.
"""
from typing import List, Optional, TypedDict, Union

import torch
import torch.nn as nn

from .constants import HF_CACHE_DIR
from . import sharding


class ImageTextDict(TypedDict):
    images: List[str]
    texts: List[str]


_VIDEO_EXT = {'.mp4', '.avi', '.mov', '.mkv'}


class Score(nn.Module):
    def __init__(self, model: str, device: str = 'cuda', cache_dir: str = HF_CACHE_DIR, **kwargs):
        super().__init__()
        assert model in self.list_all_models()
        self.device = device
        self.model = self.prepare_scoremodel(model, device, cache_dir, **kwargs)
        self.model_name = model

    def prepare_scoremodel(self, model: str, device: str, cache_dir: str, **kwargs):
        raise NotImplementedError("Subclasses must implement prepare_scoremodel")

    def list_all_models(self) -> List[str]:
        raise NotImplementedError("Subclasses must implement list_all_models")

    def _out_device(self):
        dev = torch.device(self.device if str(self.device) != 'cuda' else 'cuda:0')
        return dev if (dev.type != 'cuda' or torch.cuda.is_available()) else torch.device('cpu')

    def forward(self, images: Optional[Union[str, List[str]]] = None, texts: Optional[Union[str, List[str]]] = None,
                num_frames: Optional[int] = 8, **kwargs) -> torch.Tensor:
        """m images x n texts -> fp32 tensor [m, n] on self.device; scores[i][j] = image i vs text j."""
        if isinstance(images, str):
            images = [images]
        if isinstance(texts, str):
            texts = [texts]
        if any(isinstance(img, str) and img[-4:].lower() in _VIDEO_EXT for img in images):
            raise NotImplementedError("video inputs need the reference's ffmpeg/cv2 frame-concat pre-processing "
                                      "(score.py:72-101), which is outside the MI355X hot path")
        if hasattr(self.model, 'forward_grid'):
            scores = self.model.forward_grid(images, texts, **kwargs)
        else:   # plain plugin interface: one call per image, as the reference does
            scores = torch.stack([self.model.forward([image] * len(texts), texts, **kwargs) for image in images])
        return scores.to(self._out_device(), torch.float32)

    def batch_forward(self, dataset: List[ImageTextDict], batch_size: int = 16, num_frames: int = 4,
                      **kwargs) -> torch.Tensor:
        """dataset[k] = {'images': [..n_vis..], 'texts': [..n_txt..]} -> fp32 [len(dataset), n_vis, n_txt]."""
        num_samples = len(dataset)
        if "videos" in dataset[0]:
            raise NotImplementedError("video datasets are outside the MI355X hot path")
        num_visuals = len(dataset[0]['images'])
        num_texts = len(dataset[0]['texts'])
        lo, hi = sharding.shard_range(num_samples)
        local = torch.zeros(hi - lo, num_visuals, num_texts)
        for start in range(lo, hi, batch_size):
            stop = min(hi, start + batch_size)
            images: List[str] = []
            texts: List[str] = []
            for k in range(start, stop):
                sample = dataset[k]
                assert len(sample['images']) == num_visuals, \
                    f"Number of image options in sample {k} is {len(sample['images'])}. Expected {num_visuals}."
                assert len(sample['texts']) == num_texts, \
                    f"Number of text options in sample {k} is {len(sample['texts'])}. Expected {num_texts} texts."
                for v in sample['images']:
                    for t in sample['texts']:
                        images.append(v)
                        texts.append(t)
            s = self.model.forward(images, texts, **kwargs)
            local[start - lo: stop - lo] = s.reshape(stop - start, num_visuals, num_texts).float().cpu()
        scores = sharding.gather_rows(local, num_samples)
        return scores.to(self._out_device())
