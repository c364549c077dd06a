"""Score base class: the M x N scoring call and the dataset loop.

Same public surface as /root/reference/t2v_metrics/score.py:18-156 (``Score(model, device, cache_dir, **kw)``,
``forward(images, texts, **kw) -> Tensor[M,N]``, ``batch_forward(dataset, batch_size, **kw) -> Tensor[n,n_vis,n_txt]``)
with the two hot loops rebuilt as real batches:

  * reference ``forward`` (score.py:104-106) calls the model once per image with the image path repeated N times,
    so every image is decoded, preprocessed and pushed through the ViT N times.  Here each distinct image is
    encoded once and all M*N pairs go through the T5 passes in engine-sized batches (``model.forward_grid``).
  * reference ``batch_forward`` (score.py:143-153) iterates DataLoader batches but still scores ONE pair per model
    call.  Here every DataLoader batch is flattened into one pair list (images deduplicated) and scored in one call.
  * multi-GPU (the reference has none) is OPT-IN: ``Score(..., distributed=True)`` (or ``shard=True`` on a call) with
    ``torch.distributed`` initialised makes ``forward`` shard the grid by IMAGE and ``batch_forward`` the samples over ranks (contiguous
    blocks) and all-gather the scores (t2v_metrics_amd/sharding.py).  These calls are then COLLECTIVES: every rank must make the same call
    with the same inputs -- checked (a digest of the inputs is all-gathered first; a mismatch raises on every rank).  The default keeps the
    reference's semantics: a call scores what it is given on the calling process, whatever process group the caller's training loop has.

Video inputs: as in the reference the decision is the model's ``video_mode`` (score.py:69-101): "direct" models get the
container paths untouched (Qwen2.5-VL reads frame arrays; container decode needs decord/ffmpeg, which this image does
not have, and says so itself), "concat" is the reference's ffmpeg/cv2 frame-concat pre-processing for image-only models
-- outside the hot path (SURVEY.md §2 "frame helpers OUT OF SCOPE"): NotImplementedError instead of a silent mis-score.
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
    def __init__(self, model: str, device: str = 'cuda', cache_dir: str = HF_CACHE_DIR, distributed: bool = False, **kwargs):
        """distributed (not in the reference; default False = its semantics): True = forward / batch_forward shard their work over the ranks of
        the initialised torch.distributed group and gather the scores -- collective calls, see the module docstring."""
        super().__init__()
        assert model in self.list_all_models()
        self.device = device
        self.distributed = bool(distributed)
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
        # video container paths: the gate is the MODEL's, as in the reference (score.py:69-101) -- a video-native model
        # (video_mode "direct", e.g. qwen2.5-vl-7b) receives the paths untouched and decides itself what it can read
        # (qwen2vl_model.py:135-158); "concat" = the ffmpeg/cv2 frame-concat pre-processing for image-only models, which is
        # outside the MI355X hot path; anything else prints the reference's message and returns None like it does.
        if any(isinstance(img, str) and img[-4:].lower() in _VIDEO_EXT for img in images):
            mode = getattr(self.model, "video_mode", None)
            if mode == "concat":
                raise NotImplementedError("video inputs for image-only models need the reference's ffmpeg/cv2 frame-concat "
                                          "pre-processing (score.py:72-98), which is outside the MI355X hot path; pass "
                                          "extracted frames or use a video-native model")
            elif mode != "direct":
                print("Invalid `video_mode` for the given model. Please check model's class attributes")
                return
        # Opt-in (distributed=True on the constructor, or shard=True here) with torch.distributed initialised (one process per GPU, every rank
        # making the SAME call): the grid is sharded BY IMAGE -- rank r encodes and scores the rows of its contiguous block of images, so each
        # image still goes through the vision tower once in the whole job, and one gather of fp32 rows hands every rank the [m, n] grid
        # (the reference's row loop, score.py:104-106, is the unit that is sharded).  The ranks' inputs are checked first.
        shard = bool(kwargs.pop("shard", self.distributed)) and sharding.world()[1] > 1
        if shard:
            sharding.assert_same_call("forward", [str(i) for i in images], list(texts))
        m = len(images)
        lo, hi = sharding.shard_range(m) if shard else (0, m)
        mine = images[lo:hi]
        if not mine:
            scores = torch.zeros(0, len(texts))
        elif hasattr(self.model, 'forward_grid'):
            scores = self.model.forward_grid(mine, texts, **kwargs)
        else:   # plain plugin interface: one call per image, as the reference does
            scores = torch.stack([self.model.forward([image] * len(texts), texts, **kwargs) for image in mine])
        if shard:                                 # every rank enters, also one that owns no image (m < world size)
            scores = sharding.gather_rows(scores.float().cpu().reshape(hi - lo, len(texts)), m)
        return scores.to(self._out_device(), torch.float32)

    def batch_forward(self, dataset: List[ImageTextDict], batch_size: int = 16, num_frames: int = 4,
                      **kwargs) -> torch.Tensor:
        """dataset[k] = {'images' | 'videos': [..n_vis..], 'texts': [..n_txt..]} -> fp32 [len(dataset), n_vis, n_txt]."""
        num_samples = len(dataset)
        # the reference keys a dataset's media under "videos" or "images" (score.py:124-128); a video-native model (video_mode
        # "direct") takes either -- the paths go to the model untouched, as in forward(); for image-only models a video dataset would
        # need the ffmpeg/cv2 frame-concat pre-processing, which is outside the MI355X hot path
        media_type = "videos" if "videos" in dataset[0] else "images"
        if media_type == "videos" and getattr(self.model, "video_mode", None) != "direct":
            raise NotImplementedError("video datasets need a video-native model (video_mode 'direct'); the frame-concat path for image-only "
                                      "models is outside the MI355X hot path")
        num_visuals = len(dataset[0][media_type])
        num_texts = len(dataset[0]['texts'])
        shard = bool(kwargs.pop("shard", self.distributed)) and sharding.world()[1] > 1
        if shard:      # a collective: the ranks must hold the same dataset (checked on its size and its first / last samples)
            sharding.assert_same_call("batch_forward", num_samples, num_visuals, num_texts, [str(v) for v in dataset[0][media_type]], list(dataset[0]['texts']),
                                      [str(v) for v in dataset[-1][media_type]], list(dataset[-1]['texts']))
        lo, hi = sharding.shard_range(num_samples) if shard else (0, num_samples)
        local = torch.zeros(hi - lo, num_visuals, num_texts)
        for start in range(lo, hi, batch_size):
            stop = min(hi, start + batch_size)
            images: List[str] = []
            texts: List[str] = []
            for k in range(start, stop):
                sample = dataset[k]
                assert len(sample[media_type]) == num_visuals, \
                    f"Number of visual (image/video) options in sample {k} is {len(sample[media_type])}. Expected {num_visuals} visuals."
                assert len(sample['texts']) == num_texts, \
                    f"Number of text options in sample {k} is {len(sample['texts'])}. Expected {num_texts} texts."
                for v in sample[media_type]:
                    for t in sample['texts']:
                        images.append(v)
                        texts.append(t)
            s = self.model.forward(images, texts, **kwargs)
            local[start - lo: stop - lo] = s.reshape(stop - start, num_visuals, num_texts).float().cpu()
        scores = sharding.gather_rows(local, num_samples) if shard else local
        return scores.to(self._out_device())
