"""CLIP-FlanT5 VQAScore model on the MI355X-native engine.

Host-side mirror of the v3.0 wrapper that the reference still documents
(/root/reference/V_3.0_README.md:209-214; the file itself, ``clip_t5_model.py``, is absent from the v3.1 tree --
SURVEY.md §0) behind the plugin interface that IS in the tree
(/root/reference/t2v_metrics/models/vqascore_models/vqa_model.py:7-18, .../models/model.py:16-47).

What runs where:
  host (this file)   prompt formatting (constants.py:5,8 + V_3.0_README.md:213-214), tokenisation with the image
                     sentinel (mm_utils.py:164-179), PIL decode + expand2square + CLIP preprocessing;
  device (engine)    everything from normalised pixels / token ids to the score -- hand-written gfx950 kernels
                     behind the C ABI (include/vqs.h).  No torch.nn forward, no HF forward, no CPU fallback.

Differences from the reference wrapper that do not change results:
  * identical image paths inside one call are decoded and encoded once (the reference re-encodes the same image
    for every text: score.py:105-106);
  * ``forward`` returns a CPU fp32 tensor like the reference; ``forward_grid`` scores an M x N grid in one pass.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ...config import get_config
from ...constants import (CONTEXT_LEN, DEFAULT_IMAGE_TOKEN, HF_CACHE_DIR, IGNORE_INDEX, IMAGE_TOKEN_INDEX, SYSTEM_MSG)
from .mm_utils import t5_tokenizer_image_token
from .vqa_model import VQAScoreModel

# {} is replaced by the caption (V_3.0_README.md:213-214)
default_question_template = 'Does this figure show "{}"? Please answer yes or no.'
default_answer_template = 'Yes'

CLIP_T5_MODELS = {
    'clip-flant5-xxl': {'config': 'clip-flant5-xxl', 'hf_repo': 'zhiqiulin/clip-flant5-xxl'},
    'clip-flant5-xl': {'config': 'clip-flant5-xl', 'hf_repo': 'zhiqiulin/clip-flant5-xl'},
}


def format_question(question: str) -> str:
    """v3.0 conversation style 't5_chat': system message, USER turn with the image placeholder, open ASSISTANT turn."""
    return SYSTEM_MSG + " USER: " + DEFAULT_IMAGE_TOKEN + "\n" + question + " ASSISTANT: "


def format_answer(answer: str) -> str:
    return answer


class CLIPT5Model(VQAScoreModel):
    """VQAScore with CLIP-FlanT5 (the reference's default model name is still 'clip-flant5-xxl':
    /root/reference/t2v_metrics/vqascore.py:11)."""
    video_mode = "concat"
    allows_image = True
    allows_video = False

    def __init__(self, model_name='clip-flant5-xxl', device='cuda', cache_dir=HF_CACHE_DIR, *, weights=None,
                 tokenizer=None, checkpoint: Optional[str] = None, config=None, max_pairs: int = 256,
                 max_images: int = 256, seed: int = 0, engine=None, num_workers: Optional[int] = None,
                 vision_tower: Optional[str] = None, image_workers: str = "process", engine_options: Optional[Dict[str, int]] = None):
        """
        engine_options: execution-form options of the HIP engine (include/vqs.h vqs_set_option), e.g. {'vit_fp16': 0, 'enc_fp16': 0} to run
                    the vision tower and the T5 encoder's attention side on bf16 operands (the reference's dtype) instead of IEEE fp16.
        weights:    None -> load ``checkpoint`` (a local HF directory of safetensors); 'seeded' -> seeded random
                    weights at the exact architecture (benchmarks, tests); or a dict name -> tensor.
        tokenizer:  any object with ``tokenizer(text).input_ids`` (HF protocol); None -> the slow T5 tokenizer of
                    the checkpoint directory (the reference uses ``AutoTokenizer(use_fast=False)``, mm_utils.py:198).
        config:     a ClipT5Config overriding the registry entry (tests use the tiny configurations).
        engine:     an already constructed engine-like object (tests inject a recording fake).
        num_workers: workers that decode + preprocess images; None -> min(32, cores this process may run on).
                    The reference does this serially inside forward(); at ~200-550 pairs/s the
                    GPU would otherwise wait on PNG/JPEG decode.
        image_workers: "process" (default) = worker processes (t2v_metrics_amd/imgpool.py: PIL decodes under the GIL, so
                    only processes scale the decode -- what a rank needs when it has 1/8 of the host's cores), "thread" = the
                    thread pool of rounds 1-3.  A custom ``self.image_loader`` always runs on threads (workers cannot see it).
        """
        assert config is not None or model_name in CLIP_T5_MODELS
        self._weights_arg, self._tokenizer_arg, self._checkpoint = weights, tokenizer, checkpoint
        self._cfg = config if config is not None else get_config(CLIP_T5_MODELS[model_name]['config'])
        self._seed, self._engine_arg = seed, engine
        self._engine_options = dict(engine_options or {})
        self._vision_tower_dir = vision_tower
        try:
            cores = len(os.sched_getaffinity(0))          # what this process may use (a rank's slice under taskset / the launcher)
        except (AttributeError, OSError):
            cores = os.cpu_count() or 1
        self.num_workers = min(32, cores) if num_workers is None else max(1, int(num_workers))
        if image_workers not in ("process", "thread"):
            raise ValueError("image_workers must be 'process' or 'thread'")
        self.image_workers = image_workers
        self._pool = None
        self._proc_pool = None
        self._staging = [None, None]              # two pinned uint8 staging buffers + the event of their last H2D copy
        self._staging_ev = [None, None]
        self._staging_next = 0
        self.max_pairs, self.max_images = int(max_pairs), int(max_images)
        self.context_len = CONTEXT_LEN
        self.image_aspect_ratio = 'pad'          # mm_utils.py:188,235
        super().__init__(model_name=model_name, device=device, cache_dir=cache_dir)

    # ------------------------------------------------------------------ loading
    def load_model(self):
        self.cfg = self._cfg
        self.tokenizer = self._tokenizer_arg if self._tokenizer_arg is not None else self._load_tokenizer()
        if self._engine_arg is not None:
            self.engine = self._engine_arg
            return
        from ...engine import VqsEngine   # raises if libvqs_hip.so is missing; no fallback
        from ...weights import load_checkpoint_weights, make_seeded_weights
        dev = torch.device(self.device if str(self.device) != 'cuda' else 'cuda:0')
        if isinstance(self._weights_arg, dict):
            weights = self._weights_arg
        elif self._weights_arg == 'seeded':
            weights = make_seeded_weights(self.cfg, seed=self._seed, device=dev)
        else:
            weights = load_checkpoint_weights(self.cfg, self._read_checkpoint(), dev, vision_state_dict=self._read_vision_tower())
        self.engine = VqsEngine(self.cfg, weights, device=dev, options=self._engine_options)

    def _checkpoint_dir(self) -> str:
        if self._checkpoint:
            return self._checkpoint
        repo = CLIP_T5_MODELS[self.model_name]['hf_repo']
        return os.path.join(self.cache_dir, repo.split('/')[-1])

    def _load_tokenizer(self):
        path = self._checkpoint_dir()
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"no tokenizer: {path} does not exist (no network here). Pass tokenizer=... or checkpoint=<local HF dir>.")
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(path, use_fast=False, model_max_length=self.context_len)

    def _read_checkpoint(self) -> Dict[str, torch.Tensor]:
        path = self._checkpoint_dir()
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"no checkpoint at {path} (no network here). Pass checkpoint=<local HF dir> or weights='seeded'.")
        from ...weights import read_checkpoint_dir
        return read_checkpoint_dir(path)

    def _read_vision_tower(self) -> Optional[Dict[str, torch.Tensor]]:
        """The CLIP tower's own directory, if the main checkpoint does not carry it (mm_utils.py:236-237): explicit
        ``vision_tower=<dir>``, else <cache_dir>/clip-vit-large-patch14-336 when it exists."""
        from ...weights import read_checkpoint_dir
        path = self._vision_tower_dir or os.path.join(self.cache_dir, "clip-vit-large-patch14-336")
        if os.path.isdir(path):
            return read_checkpoint_dir(path)
        if self._vision_tower_dir:
            raise FileNotFoundError(f"no vision tower checkpoint at {path}")
        return None

    # ------------------------------------------------------------------ host-side preparation
    def _executor(self):
        if self._pool is None and self.num_workers > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.num_workers, thread_name_prefix="vqs-img")
        return self._pool

    def _preprocess_one(self, path) -> torch.Tensor:
        from ...preprocess import clip_preprocess
        img = self.image_loader(path)
        return torch.from_numpy(clip_preprocess(img, self.cfg.vision.image, self.image_aspect_ratio == 'pad'))

    def _preprocess_one_u8(self, path) -> np.ndarray:
        from ...preprocess import clip_preprocess_u8
        return clip_preprocess_u8(self.image_loader(path), self.cfg.vision.image, self.image_aspect_ratio == 'pad')

    def _load_images_host(self, image: List[str]) -> torch.Tensor:
        """Decode + pad to square + CLIP-preprocess on the host (thread pool) -> fp32 [N,3,S,S]."""
        S = self.cfg.vision.image
        out = torch.empty(len(image), 3, S, S, dtype=torch.float32)
        pool = self._executor()
        results = pool.map(self._preprocess_one, image) if pool is not None else map(self._preprocess_one, image)
        for i, t in enumerate(results):
            out[i] = t
        return out

    def _staging_buffer(self, n: int):
        """One of two pinned uint8 staging buffers [>= n, S, S, 3], allocated once and reused (pinning 87 MB per call costs as much
        as the copy): a buffer is handed out again only after the H2D copy that last read it has completed (its event)."""
        S = self.cfg.vision.image
        k = self._staging_next
        self._staging_next ^= 1
        if self._staging_ev[k] is not None:
            self._staging_ev[k].synchronize()
        buf = self._staging[k]
        if buf is None or buf.shape[0] < n:
            buf = torch.empty(max(n, self.max_images), S, S, 3, dtype=torch.uint8)
            if torch.cuda.is_available():
                buf = buf.pin_memory()
            self._staging[k] = buf
        return k, buf[:n]

    def _use_process_pool(self) -> bool:
        from ..._imgprep import image_loader as default_loader
        return self.image_workers == "process" and self.num_workers > 1 and self.image_loader is default_loader

    def _load_images_host_u8(self, image: List[str]):
        """Decode + pad + PIL resize + crop on the host -> (staging slot, uint8 [N,S,S,3] pinned).  Worker processes by default
        (imgpool.py); threads for a custom image_loader or image_workers='thread'."""
        S = self.cfg.vision.image
        k, out = self._staging_buffer(len(image))
        arr = out.numpy()
        if len(image) == 1:
            # one image (the reference's per-pair loops, score.py:143-153): decode on the calling thread -- handing a single file to a worker process and
            # waiting for its reply costs more than it saves (a 512 x 512 PNG: 7-11 ms through the pool, tools/bench_call_latency.py); same bytes either way
            arr[0] = self._preprocess_one_u8(image[0])
            return k, out
        if self._use_process_pool():
            if self._proc_pool is None:
                from ...imgpool import ImageProcessPool
                self._proc_pool = ImageProcessPool(self.num_workers)
            np.copyto(arr, self._proc_pool.load_u8(list(image), S, self.image_aspect_ratio == 'pad'))
            return k, out
        pool = self._executor()

        def work(i_path):
            arr[i_path[0]] = self._preprocess_one_u8(i_path[1])       # the copy into the staging buffer happens in the worker

        if pool is not None:
            list(pool.map(work, enumerate(image)))
        else:
            for ip in enumerate(image):
                work(ip)
        return k, out

    def load_images(self, image: List[str]) -> torch.Tensor:
        """Decode + pad to square + CLIP-preprocess; returns bf16 [N,3,S,S] on the device.  With the HIP engine the
        per-pixel float work (rescale, normalise, bf16) runs on the GPU from a uint8 staging buffer (1/4 of the H2D
        bytes, no GIL-bound numpy arithmetic); test doubles without ``normalize_u8`` get the host float path."""
        if hasattr(self.engine, "normalize_u8") and str(self.device).startswith('cuda') and torch.cuda.is_available():
            from ...preprocess import OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
            k, host = self._load_images_host_u8(image)
            # This runs on the image thread ("vqs-img"), whose CURRENT device is 0 whatever the main thread set: the copy, the event that
            # guards the staging buffer's reuse and the kernel behind them must all sit on THIS model's device and its current stream
            # (an event recorded on device 0 while the DMA runs on cuda:k is complete at once -- the buffer would be overwritten under
            # the copy: ADVICE r4)
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device)
                u8 = host.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
            self._staging_ev[k] = ev
            return self.engine.normalize_u8(u8, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD)
        px = self._load_images_host(image)
        if str(self.device).startswith('cuda') and torch.cuda.is_available():
            return px.pin_memory().to(self.device, non_blocking=True).to(torch.bfloat16)
        return px.to(torch.bfloat16)

    def tokenize(self, questions: Sequence[str], answers: Sequence[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (input_ids int32 [n,L] right-padded with 0, one -200 each; labels int32 [n,T] padded with -100)."""
        q_ids = [t5_tokenizer_image_token(format_question(q), self.tokenizer, IMAGE_TOKEN_INDEX) for q in questions]
        a_ids = [t5_tokenizer_image_token(format_answer(a), self.tokenizer, IMAGE_TOKEN_INDEX) for a in answers]
        q_ids = [ids[: self.context_len] for ids in q_ids]
        a_ids = [ids[: self.context_len] for ids in a_ids]
        L = max(len(x) for x in q_ids)
        T = max(len(x) for x in a_ids)
        pad = self.cfg.t5.pad_id
        ids = torch.full((len(q_ids), L), pad, dtype=torch.int32)
        lab = torch.full((len(a_ids), T), IGNORE_INDEX, dtype=torch.int32)
        for i, x in enumerate(q_ids):
            if x.count(IMAGE_TOKEN_INDEX) != 1:
                raise ValueError("every question must contain exactly one <image> placeholder after templating")
            if pad in x:
                raise ValueError("tokenizer produced the pad id inside a prompt")
            ids[i, : len(x)] = torch.tensor(x, dtype=torch.int32)
        for i, x in enumerate(a_ids):
            lab[i, : len(x)] = torch.tensor(x, dtype=torch.int32)
        return ids, lab

    # ------------------------------------------------------------------ scoring
    @torch.no_grad()
    def score_pairs(self, images: Sequence[str], pair_image: Sequence[int], questions: Sequence[str],
                    answers: Sequence[str], return_logprobs: bool = False, _retry: bool = False):
        """Score pairs (images[pair_image[k]], questions[k], answers[k]).  Unique images are encoded once.  (_retry: the one re-score on bf16
        operands after an fp16 execution option produced a non-finite score -- see the end of this method.)"""
        n = len(questions)
        assert len(answers) == n and len(pair_image) == n
        if n == 0:           # empty M x N grids are legal in the reference (score.py:104 builds a [M, N] tensor of zeros)
            empty = torch.zeros(0, dtype=torch.float32)
            return (empty, torch.zeros(0, 0, dtype=torch.float32)) if return_logprobs else empty
        # Pipeline: the pool decodes/preprocesses image chunk i+1 while the GPU works on chunk i, and a batch of pairs is
        # scored as soon as all of its images are encoded -- so the (long) T5 scoring of early pairs overlaps the host work
        # for later images instead of waiting for every image first.  Launches are asynchronous; nothing here syncs.
        chunks = [list(images[s: s + self.max_images]) for s in range(0, len(images), self.max_images)]
        pool = self._executor()
        pending = pool.submit(self.load_images, chunks[0]) if (pool is not None and len(chunks) > 1) else None
        ids, lab = self.tokenize(questions, answers)
        idx = torch.as_tensor(list(pair_image), dtype=torch.int32)
        # Length bucketing: a batch is padded to ITS longest prompt and the engine computes over the padding, so pairs
        # are scored in order of prompt length (stable: equal lengths keep arrival order, i.e. image order) and every
        # batch's ids are cut to its own maximum; scores are scattered back to arrival order at the end.  With one
        # image chunk the whole set is sorted; with several, sorting is per image chunk so that the streaming overlap
        # (score a batch as soon as its images are encoded) survives.
        plen = (ids != self.cfg.t5.pad_id).sum(1)
        chunk_of_pair = idx.long() // self.max_images
        order = torch.argsort(chunk_of_pair * (int(plen.max()) + 1) + plen, stable=True)
        ids, lab, idx, plen = ids[order], lab[order], idx[order], plen[order]
        # pairs are scored in (sorted) order; prefix_max[k] = highest image index needed by pairs [0, k]
        prefix_max = torch.cummax(idx, 0).values if n > 0 else idx
        feats, n_enc, next_pair = None, 0, 0
        scores, lps = [], []
        for ci, chunk in enumerate(chunks):
            if pending is not None:
                px = pending.result()
                pending = pool.submit(self.load_images, chunks[ci + 1]) if ci + 1 < len(chunks) else None
            else:
                px = self.load_images(chunk)
            f = self.engine.encode_images(px)
            if feats is None:
                feats = f if len(chunks) == 1 else torch.empty((len(images),) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device)
            if len(chunks) > 1:
                feats[n_enc: n_enc + f.shape[0]] = f
            n_enc += f.shape[0]
            last = ci + 1 == len(chunks)
            while next_pair < n:
                e = min(n, next_pair + self.max_pairs)
                if int(prefix_max[e - 1]) >= n_enc or (e - next_pair < self.max_pairs and not last):
                    break
                keep = int(plen[next_pair:e].max())
                lp, sc = self.engine.score(feats, idx[next_pair:e], ids[next_pair:e, :keep].contiguous(), lab[next_pair:e])
                scores.append(sc)
                lps.append(lp)
                next_pair = e
        inv = torch.empty_like(order)
        inv[order] = torch.arange(n)
        sc = torch.cat(scores).float().cpu()[inv]
        if not bool(torch.isfinite(sc).all()):
            # Non-finite scores.  With an fp16 execution option on, an activation beyond 65 504 is stored as inf and reaches the score as NaN (the
            # library raises status bit 1 for the same condition, include/vqs.h "flags").  The options that are on by DEFAULT carry a bind-time
            # range proof (engine.fp16_range_proof) and cannot get here; one the caller insisted on can.  The reference returns a score for any
            # finite input (score.py:104-106), so: switch the fp16 options off, say so once, and score the batch again on bf16 operands.
            bad = int((~torch.isfinite(sc)).sum())
            on = [k for k in ("vit_fp16", "enc_fp16", "dec_fp16") if hasattr(self.engine, "get_option") and self.engine.get_option(k)]
            if on and not _retry:
                import warnings
                warnings.warn(f"t2v_metrics_amd: {bad} of {sc.numel()} scores were not finite with the fp16 execution options {on}: an activation left "
                              "the fp16 range.  Those options are now OFF for this scorer (bf16 operands, the reference's dtype) and the batch is re-scored.",
                              RuntimeWarning, stacklevel=2)
                for k in on:
                    self.engine.set_option(k, 0)
                return self.score_pairs(images, pair_image, questions, answers, return_logprobs, _retry=True)
            raise RuntimeError(f"{bad} of {sc.numel()} scores are not finite on bf16 operands: the checkpoint's weights or the inputs are not finite")
        if return_logprobs:
            return sc, torch.cat(lps).float().cpu()[inv]
        return sc

    def forward(self, images: List[str], texts: List[str], question_template: str = default_question_template,
                answer_template: str = default_answer_template, return_logprobs: bool = False) -> torch.Tensor:
        """n scores for n (image, text) pairs; score = exp(mean log P(answer tokens)) = the v3.0
        ``exp(-CrossEntropyLoss(mean))``."""
        assert len(images) == len(texts), "Number of images and texts must match"
        questions = [question_template.format(t) for t in texts]
        answers = [answer_template.format(t) for t in texts]
        uniq: Dict[str, int] = {}
        pair_image = [uniq.setdefault(str(p), len(uniq)) for p in images]
        return self.score_pairs(list(uniq.keys()), pair_image, questions, answers, return_logprobs)

    def forward_grid(self, images: List[str], texts: List[str], question_template: str = default_question_template,
                     answer_template: str = default_answer_template) -> torch.Tensor:
        """All M x N pairs in one batched pass -> fp32 [M, N] on the CPU (row i = image i)."""
        M, N = len(images), len(texts)
        questions = [question_template.format(t) for t in texts] * M
        answers = [answer_template.format(t) for t in texts] * M
        uniq: Dict[str, int] = {}
        img_ids = [uniq.setdefault(str(p), len(uniq)) for p in images]
        pair_image = [img_ids[i] for i in range(M) for _ in range(N)]
        return self.score_pairs(list(uniq.keys()), pair_image, questions, answers).reshape(M, N)

    @torch.no_grad()
    def generate_ids(self, images: List[str], texts: List[str], max_new_tokens: int = 16, _retry: bool = False) -> List[List[int]]:
        """Greedy decoding of one answer per (image, prompt) pair -> token ids, each cut after its first EOS.  (_retry: the one re-run on bf16
        operands after an fp16 execution option produced a non-finite logit -- status bit 1 of the pass, as in score_pairs.)"""
        assert len(images) == len(texts), "Number of images and texts must match"
        if not 1 <= max_new_tokens <= 512:
            raise ValueError("max_new_tokens must be in [1, 512]")
        uniq: Dict[str, int] = {}
        pair_image = [uniq.setdefault(str(p), len(uniq)) for p in images]
        paths = list(uniq.keys())
        feats = torch.cat([self.engine.encode_images(self.load_images(paths[s: s + self.max_images]))
                           for s in range(0, len(paths), self.max_images)], 0)
        ids, _ = self.tokenize(texts, [""] * len(texts))
        idx = torch.as_tensor(pair_image, dtype=torch.int32)
        eos = self.cfg.t5.eos_id
        out: List[List[int]] = []
        for s in range(0, len(texts), self.max_pairs):
            e = min(len(texts), s + self.max_pairs)
            toks = self.engine.generate(feats, idx[s:e], ids[s:e], max_new_tokens).cpu().tolist()
            if hasattr(self.engine, "stage") and hasattr(self.engine, "get_option") and int(self.engine.stage("flags")[0]) & 2:
                # a non-finite logit (argmax would have emitted token 0 silently): an fp16 tensor overflowed -- only possible with an option forced
                # against the bind-time range proof.  Same answer as score_pairs: fp16 options off, one warning, run again.
                on = [k for k in ("vit_fp16", "enc_fp16", "dec_fp16") if self.engine.get_option(k)]
                if on and not _retry:
                    import warnings
                    warnings.warn(f"t2v_metrics_amd: non-finite logits while generating with the fp16 execution options {on}; they are now OFF for this "
                                  "scorer (bf16 operands, the reference's dtype) and the call is run again.", RuntimeWarning, stacklevel=2)
                    for k in on:
                        self.engine.set_option(k, 0)
                    return self.generate_ids(images, texts, max_new_tokens, _retry=True)
                raise RuntimeError("non-finite logits while generating on bf16 operands: the checkpoint's weights or the inputs are not finite")
            for row in toks:
                out.append(row[: row.index(eos) + 1] if eos in row else row)
        return out

    def generate(self, images: List[str], texts: List[str], max_new_tokens: int = 16) -> List[str]:
        """The reference's ``model.generate(images=..., texts=...)`` (/root/reference/V_3.0_README.md:316-325): greedy
        decoding of a text answer per pair (HF GenerationMixin greedy search, decoder start = pad).  Runs on the HIP
        engine: encoder once, then one incremental decoder step per new token over a self-attention K/V cache."""
        return [self.tokenizer.decode(ids, skip_special_tokens=True) for ids in self.generate_ids(images, texts, max_new_tokens)]
