"""Qwen2.5-VL VQAScore wrapper on the MI355X engine -- the plugin-interface counterpart of the reference's
``Qwen2VLModel`` (/root/reference/t2v_metrics/models/vqascore_models/qwen2vl_model.py:93-301) for ``qwen2.5-vl-7b``.

Same recipe: prompt = chat template around one vision placeholder + ``question_template.format(text)``, greedy
generation of ``max_new_tokens`` tokens (default 1 = ONE prefill), score = geometric mean over the answer tokens of
softmax(processed_scores / temperature)[answer token] read from the LAST positions of the generated scores, with the
reference's trailing-special-token rule (:222-289).  "Processed" = what HF ``generate(output_scores=True)`` returns: the
logits after the generation_config's repetition penalty over prompt + generated ids (read from the checkpoint's
generation_config.json; 1.0 when absent, e.g. seeded weights).  What differs: samples are batched (the reference runs
batch 1, :190), the prefill runs on libvqs_hip (include/vqs_qwen.h; no KV cache: step k re-runs the prefill over the
prompt + k generated tokens), the softmax + gather run in the library's score head, frame resizing / patch flattening
are restated here instead of going through ``qwen_vl_utils`` + the HF processor (neither is installable offline).

Input support: ``.npy`` arrays ([H,W,3] image or [T,H,W,3] frames, as qwen2vl_model.py:145-155) and image files; video
container files need decord/ffmpeg (absent) -> NotImplementedError.  Any frame size smart_resize produces is accepted
(partial attention windows at the frame edge are handled by the engine's padded windowed layout).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from PIL import Image

from ...constants import HF_CACHE_DIR
from ...qwen import Qwen25VLConfig, get_qwen_config
from .vqa_model import VQAScoreModel

QWEN25_VL_MODELS = {
    'qwen2.5-vl-7b': {'config': 'qwen2.5-vl-7b', 'hf_repo': 'Qwen/Qwen2.5-VL-7B-Instruct', 'fps': 8.0},
}
default_question_template = 'Does this figure show "{}"? Please answer Yes or No.'     # qwen2vl_model.py:173
default_answer_template = 'Yes'
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
VIDEO_FILE_MAX_PIXELS = 360 * 420      # qwen2vl_model.py:141-144: ONLY for video container paths (not supported here)
# qwen_vl_utils (not installable here) resizes BEFORE the HF processor (the reference then passes do_resize=False,
# qwen2vl_model.py:208-216): its defaults are IMAGE_FACTOR = 28, MIN_PIXELS = 4*28*28, MAX_PIXELS = 16384*28*28
# [RECALLED from qwen_vl_utils/vision_process.py]; frames given as a list go through the same per-image routine with the
# message's max_pixels.
QVU_MIN_PIXELS = 4 * 28 * 28
QVU_MAX_PIXELS = 16384 * 28 * 28
IMAGE_PLACEHOLDER, VIDEO_PLACEHOLDER = "<|image_pad|>", "<|video_pad|>"


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    """HF models/qwen2_vl/image_processing_qwen2_vl.py:62-88."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar, w_bar = round(height / factor) * factor, round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar, w_bar = math.ceil(height * beta / factor) * factor, math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def patchify(frames: torch.Tensor, patch: int = 14, merge: int = 2, temporal: int = 2) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
    """frames fp32 [T, 3, H, W] (normalised) -> ([t*h*w, 3*temporal*patch*patch], (t, h, w)): the flat layout of HF
    Qwen2VLVideoProcessor.patchify (models/qwen2_vl/video_processing_qwen2_vl.py:236-274): the last frame is repeated
    to an even count, patches ordered block-major over merge x merge cells, features ordered [C, Tp, P, P]."""
    T, C, H, W = frames.shape
    if T % temporal:
        frames = torch.cat([frames, frames[-1:].expand(temporal - T % temporal, -1, -1, -1)], 0)
        T = frames.shape[0]
    t, h, w = T // temporal, H // patch, W // patch
    x = frames.reshape(t, temporal, C, h // merge, merge, patch, w // merge, merge, patch)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
    return x.reshape(t * h * w, C * temporal * patch * patch).contiguous(), (t, h, w)


HF_DEFAULT_TOP_K = 50      # transformers GenerationConfig default; a checkpoint's generation_config.json may override it


def _read_generation_json(checkpoint_dir: str) -> dict:
    import json
    path = os.path.join(checkpoint_dir, "generation_config.json")
    if not os.path.isfile(path):
        return {}
    with open(path) as f:
        return json.load(f)


def read_generation_config(checkpoint_dir: str):
    """(repetition_penalty, eos ids) from generation_config.json of the checkpoint directory: HF generate() applies the
    penalty in greedy mode too and stops at ANY of generation_config.eos_token_id (an int or a list -- [<|im_end|>,
    <|endoftext|>] for the Instruct checkpoints), not only at the tokenizer's eos.  (1.0, []) if the file is absent."""
    g = _read_generation_json(checkpoint_dir)
    eos = g.get("eos_token_id", [])
    eos = [int(eos)] if isinstance(eos, int) else [int(e) for e in (eos or [])]
    return float(g.get("repetition_penalty", 1.0)), eos


def read_top_k(checkpoint_dir: str) -> int:
    """generation_config.top_k of the checkpoint; HF's default 50 when the file or the key is absent.  The reference's sampling
    call (qwen2vl_model.py:541-546) passes do_sample / temperature / top_p only, so HF also applies THIS top-k filter
    (TopKLogitsWarper sits between the temperature and the top-p warpers, generation/utils.py _get_logits_processor)."""
    k = _read_generation_json(checkpoint_dir).get("top_k", HF_DEFAULT_TOP_K)
    return HF_DEFAULT_TOP_K if k is None else int(k)


def read_repetition_penalty(checkpoint_dir: str) -> float:
    return read_generation_config(checkpoint_dir)[0]


def chat_prompt(question: str, placeholder: str) -> str:
    """Qwen2.5-VL chat template, one user turn with one vision item followed by the text, generation prompt appended
    (what processor.apply_chat_template(messages, add_generation_prompt=True) renders, qwen2vl_model.py:196-200)."""
    return ("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n<|vision_start|>" + placeholder +
            "<|vision_end|>" + question + "<|im_end|>\n<|im_start|>assistant\n")


class _NonFiniteUnderFp16(Exception):
    """raised inside a pass when the fp16 forms' logits are not finite; caught by Qwen25VLModel._with_bf16_fallback"""


class Qwen25VLModel(VQAScoreModel):
    video_mode = "direct"
    allows_image = True

    def __init__(self, model_name='qwen2.5-vl-7b', device='cuda', cache_dir=HF_CACHE_DIR, checkpoint: Optional[str] = None,
                 weights=None, tokenizer=None, config: Optional[Qwen25VLConfig] = None, engine=None, max_batch: int = 32,
                 seed: int = 0, num_workers: Optional[int] = None):
        """weights: None -> ``checkpoint`` (local HF directory of safetensors); 'seeded' -> seeded random weights at the
        architecture; or a dict.  tokenizer: object with ``encode(text, add_special_tokens=False) -> List[int]`` that maps
        the chat-template special tokens (HF protocol); None -> the checkpoint's tokenizer."""
        assert config is not None or model_name in QWEN25_VL_MODELS, f"Model {model_name} not found in QWEN25_VL_MODELS"
        self._cfg = config if config is not None else get_qwen_config(QWEN25_VL_MODELS[model_name]['config'])
        self._weights_arg, self._tokenizer_arg, self._checkpoint, self._engine_arg = weights, tokenizer, checkpoint, engine
        self._seed, self.max_batch = seed, int(max_batch)
        self.num_workers = min(16, os.cpu_count() or 1) if num_workers is None else max(1, int(num_workers))
        super().__init__(model_name=model_name, device=device, cache_dir=cache_dir)

    # ------------------------------------------------------------------ loading
    def _checkpoint_dir(self) -> str:
        if self._checkpoint:
            return self._checkpoint
        return os.path.join(self.cache_dir, QWEN25_VL_MODELS[self.model_name]['hf_repo'].split('/')[-1])

    def load_model(self):
        self.cfg = self._cfg
        self.repetition_penalty = 1.0
        self._gen_eos_ids = []
        self.top_k = HF_DEFAULT_TOP_K
        if self._tokenizer_arg is not None:
            self.tokenizer = self._tokenizer_arg
        else:
            path = self._checkpoint_dir()
            if not os.path.isdir(path):
                raise FileNotFoundError(f"no tokenizer: {path} does not exist (no network here). Pass tokenizer=... or checkpoint=<local HF dir>.")
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(path)
        if self._engine_arg is not None:
            self.engine = self._engine_arg
            return
        from ...qwen.engine import QwenEngine          # raises if libvqs_hip.so is missing; no fallback
        from ...qwen.weights import make_seeded_qwen_weights
        dev = torch.device(self.device if str(self.device) != 'cuda' else 'cuda:0')
        if isinstance(self._weights_arg, dict):
            weights = self._weights_arg
        elif self._weights_arg == 'seeded':
            weights = make_seeded_qwen_weights(self.cfg, seed=self._seed)
        else:
            path = self._checkpoint_dir()
            if not os.path.isdir(path):
                raise FileNotFoundError(f"no checkpoint at {path} (no network here). Pass checkpoint=<local HF dir> or weights='seeded'.")
            from ...qwen.weights import load_qwen_checkpoint
            weights = load_qwen_checkpoint(path)           # accepts the published (legacy) and the in-memory key layout
            self.repetition_penalty, self._gen_eos_ids = read_generation_config(path)
            self.top_k = read_top_k(path)
        self.engine = QwenEngine(self.cfg, weights, device=dev)

    # ------------------------------------------------------------------ host-side preparation
    def load_images(self, paths: List[str], fps: float = None) -> List[Dict]:
        """-> [{'type': 'image'|'video', 'frames': uint8 [T,H,W,3]}] (qwen2vl_model.py:135-158)."""
        out = []
        for path in paths:
            low = str(path).lower()
            if low.endswith(('.mp4', '.avi', '.mov', '.mkv')):
                raise NotImplementedError("video container files need decord/ffmpeg, which this environment does not have; "
                                          "pass extracted frames as a [T,H,W,3] .npy array")
            if low.endswith('.npy'):
                arr = np.load(path)
                if arr.ndim == 3:
                    out.append({'type': 'image', 'frames': arr.astype('uint8')[None]})
                elif arr.ndim == 4:
                    out.append({'type': 'video', 'frames': arr.astype('uint8')})
                else:
                    raise ValueError(f"Unexpected shape for NumPy array in {path}")
            else:
                out.append({'type': 'image', 'frames': np.asarray(Image.open(path).convert('RGB'))[None]})
        return out

    def preprocess(self, item: Dict) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
        """Resize (smart_resize to multiples of 28), rescale, CLIP-normalise, flatten -> (patches fp32 [N, 1176], (t, h, w)).
        Frame lists from a 4-D .npy carry NO max_pixels in the reference's message (qwen2vl_model.py:151-153; the
        360*420 cap is set for container paths only, :141-144), so every frame goes through qwen_vl_utils' per-image
        routine with its default MAX_PIXELS, exactly like a still image."""
        v = self.cfg.vision
        frames = item['frames']
        H, W = frames.shape[1:3]
        factor = v.patch * v.spatial_merge
        rh, rw = smart_resize(H, W, factor=factor, min_pixels=QVU_MIN_PIXELS, max_pixels=QVU_MAX_PIXELS)
        if (rh, rw) != (H, W):
            frames = np.stack([np.asarray(Image.fromarray(f).resize((rw, rh), Image.BICUBIC)) for f in frames])
        x = torch.from_numpy(np.ascontiguousarray(frames)).permute(0, 3, 1, 2).to(torch.float32) * (1.0 / 255.0)
        mean = torch.tensor(OPENAI_CLIP_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(OPENAI_CLIP_STD).view(1, 3, 1, 1)
        return patchify((x - mean) / std, v.patch, v.spatial_merge, v.temporal_patch)

    def _prepare_media(self, images, fps=None):
        """Load and preprocess every DISTINCT medium once (same path string = same medium; the reference re-reads and re-encodes
        the file for every prompt, score.py:104-106 + qwen2vl_model.py:190).  -> (prepared [(patches, grid)] per distinct medium,
        kinds ['image'|'video'] per distinct medium, media_of [index into those] per sample)."""
        first: Dict[str, int] = {}
        media_of, uniq = [], []
        for p in images:
            key = p if isinstance(p, str) else None
            if key is not None and key in first:
                media_of.append(first[key])
                continue
            if key is not None:
                first[key] = len(uniq)
            media_of.append(len(uniq))
            uniq.append(p)
        items = self.load_images(uniq, fps)
        # decode/resize/patch-flatten on a thread pool (PIL and torch release the GIL in their inner loops)
        if len(items) > 1 and self.num_workers > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(self.num_workers, len(items))) as pool:
                prepared = list(pool.map(self.preprocess, items))
        else:
            prepared = [self.preprocess(it) for it in items]
        return prepared, [it['type'] for it in items], media_of

    def _media_windows(self, prepared, media_of):
        """Walk the samples grid by grid, medium by medium: yields (grid, sample indices, {medium: merged vision tokens
        [n_tok, out_hidden]}) for windows of at most max_batch DISTINCT media.  Every medium is in exactly one window, so it is
        encoded exactly once (one tower call per window) and its tokens live only while its own samples are scored -- the
        device holds max_batch media at a time however long the pair list is."""
        by_grid: Dict[Tuple[int, int, int], Dict[int, List[int]]] = {}
        for i, u in enumerate(media_of):
            by_grid.setdefault(prepared[u][1], {}).setdefault(u, []).append(i)
        for g, by_medium in by_grid.items():
            n_tok = g[0] * g[1] * g[2] // self.cfg.vision.merge_unit
            us = list(by_medium)
            for s in range(0, len(us), self.max_batch):
                chunk = us[s: s + self.max_batch]
                merged = self.engine.encode_vision(torch.cat([prepared[u][0] for u in chunk]), [g] * len(chunk))
                yield g, [i for u in chunk for i in by_medium[u]], {u: merged[k * n_tok: (k + 1) * n_tok] for k, u in enumerate(chunk)}

    def build_ids(self, question: str, kind: str, n_tokens: int) -> List[int]:
        ph = VIDEO_PLACEHOLDER if kind == 'video' else IMAGE_PLACEHOLDER
        ids = list(self.tokenizer.encode(chat_prompt(question, ph), add_special_tokens=False))
        pid = self.cfg.video_token_id if kind == 'video' else self.cfg.image_token_id
        if ids.count(pid) != 1:
            raise ValueError("the tokenizer must map the vision placeholder to exactly one token")
        k = ids.index(pid)
        # the HIP path splices every vision run through the video placeholder id (image = one-temporal-patch video)
        return ids[:k] + [self.cfg.video_token_id] * n_tokens + ids[k + 1:]

    # ------------------------------------------------------------------ scoring
    def _special_ids(self) -> List[int]:
        tok = self.tokenizer
        return [x for x in (getattr(tok, "eos_token_id", None), getattr(tok, "bos_token_id", None), getattr(tok, "pad_token_id", None))
                if x is not None]

    def _stop_ids(self) -> List[int]:
        """ids that end HF generation: generation_config.eos_token_id ([<|im_end|>, <|endoftext|>] for the Instruct
        checkpoints) when known, else the tokenizer's eos."""
        extra = getattr(self, "_gen_eos_ids", None)
        if extra:
            return list(extra)
        e = getattr(self.tokenizer, "eos_token_id", None)
        return [] if e is None else [e]

    def _processed_scores(self, logits: torch.Tensor, rows: List[List[int]]) -> torch.Tensor:
        """HF RepetitionPenaltyLogitsProcessor over prompt + generated ids (generation/logits_process.py): a seen token's
        score is divided by the penalty when positive and multiplied when negative.  Identity for penalty 1.0."""
        pen = float(getattr(self, "repetition_penalty", 1.0))
        if pen == 1.0:
            return logits
        out = logits.clone()
        for k, r in enumerate(rows):
            seen = torch.tensor(sorted(set(r)), dtype=torch.long, device=logits.device)
            v = out[k, seen]
            out[k, seen] = torch.where(v < 0, v * pen, v / pen)
        return out

    def _token_probs(self, scores: torch.Tensor, token_ids: torch.Tensor, temperature: float) -> torch.Tensor:
        """softmax(scores / temperature)[token] per row, on the device: the library's fp32 log-softmax + gather
        (vqs_score_head), not a [B, 152 064] softmax on the host."""
        if hasattr(self.engine, "lib"):
            from ... import engine as _eng
            x = scores if temperature == 1.0 else scores / temperature
            with torch.cuda.device(scores.device):          # the kernel launches on the current device's stream
                lp, _ = _eng.score_head(x.unsqueeze(1).contiguous(), token_ids.to(scores.device).reshape(-1, 1))
            return lp[:, 0].exp().float().cpu()
        p = torch.softmax(scores.float() / temperature, dim=-1)                 # engine doubles in the CPU tests
        return p[torch.arange(p.shape[0]), token_ids.to(p.device)].cpu()

    def _with_bf16_fallback(self, fn):
        """fn(), and once more on the bf16 forms (one warning, the option stays off) if the fp16 forms produced a non-finite logit."""
        try:
            return fn()
        except _NonFiniteUnderFp16:
            import warnings
            warnings.warn("t2v_metrics_amd: non-finite logits under the Qwen2.5-VL row's fp16 forms; option fp16 is now OFF for this scorer (bf16 "
                          "everywhere, the reference's dtype) and the call is run again.", RuntimeWarning, stacklevel=3)
            self.engine.set_option("fp16", 0)
            return fn()

    def _generate_scores(self, *args):
        return self._with_bf16_fallback(lambda: self._generate_scores_impl(*args))

    def _generate_scores_impl(self, images, texts, fps, question_template, answer_template, max_new_tokens):
        """Shared front half of forward / forward_with_trace: load, preprocess, batch by grid, greedy generation.
        -> per sample (processed score rows, generated ids, answer ids)."""
        assert len(images) == len(texts), "Number of images/videos and texts must match"
        if max_new_tokens < 1:
            raise ValueError("max_new_tokens must be >= 1")
        questions = [question_template.format(t) for t in texts]
        answers = [answer_template.format(t) for t in texts]
        prepared, kinds, media_of = self._prepare_media(images, fps)
        stops = self._stop_ids()
        out = [None] * len(images)
        # a sample's vision rows are its MEDIUM's (encoded once per call, however many prompts share it); samples are batched
        # by grid, at most max_batch at a time
        for g, idxs, merged_of in self._media_windows(prepared, media_of):
            for s in range(0, len(idxs), self.max_batch):
                chunk = idxs[s: s + self.max_batch]
                merged = torch.cat([merged_of[media_of[i]] for i in chunk])
                n_tok = g[0] * g[1] * g[2] // self.cfg.vision.merge_unit
                rows = [self.build_ids(questions[i], kinds[media_of[i]], n_tok) for i in chunk]
                a_ids = [list(self.tokenizer.encode(answers[i], add_special_tokens=False)) for i in chunk]
                if any(len(a) < 1 for a in a_ids):
                    raise ValueError("empty answer")
                # ---- greedy generation (HF generate, do_sample=False): ONE prefill that keeps the KV cache, then one cached
                # position per further token; a sample stops at its first stop id, exactly as batch-1 generate does in the
                # reference (:222-230) -- its later rows are computed and ignored (rows are independent)
                step_scores, gen = self._greedy(merged, rows, [g] * len(chunk), max_new_tokens, stops)
                for k, i in enumerate(chunk):
                    out[i] = (step_scores[k], gen[k], a_ids[k])
        return out

    @torch.no_grad()
    def forward(self, images: List[str], texts: List[str], fps=None, question_template: str = default_question_template,
                answer_template: str = default_answer_template, max_new_tokens: int = 1, temperature: float = 1.0) -> torch.Tensor:
        scores = torch.zeros(len(images), dtype=torch.float32)
        specials = self._special_ids()
        # ---- score the answer tokens from the LAST positions of the generated scores (:239-289)
        for i, (step_scores, gen, a_ids) in enumerate(self._generate_scores(images, texts, fps, question_template, answer_template,
                                                                               max_new_tokens)):
            n_ans, offset = len(a_ids), 0
            if gen[-1] in specials:
                n_ans = min(n_ans, len(step_scores) - 1)
                offset = 1
                if n_ans <= 0:
                    raise ValueError("No content tokens to score after removing special tokens")
            if len(step_scores) < n_ans:
                n_ans = len(step_scores)
            pos = [len(step_scores) - (n_ans - t + offset) for t in range(n_ans)]
            p = self._token_probs(torch.stack([step_scores[q] for q in pos]), torch.tensor(a_ids[:n_ans]), temperature)
            scores[i] = float(torch.prod(p.double()) ** (1.0 / n_ans))
        return scores

    @torch.no_grad()
    def forward_grid(self, images: List[str], texts: List[str], **kwargs) -> torch.Tensor:
        """M media x N texts -> [M, N] (Score.forward's grid, /root/reference/t2v_metrics/score.py:103-106) as ONE pair list:
        M tower passes, not M*N -- the reference calls forward([image] * N, texts) per image and re-encodes it N times."""
        flat_i = [im for im in images for _ in texts]
        flat_t = [t for _ in images for t in texts]
        return self.forward(flat_i, flat_t, **kwargs).reshape(len(images), len(texts))

    @torch.no_grad()
    def forward_with_trace(self, images: List[str], texts: List[str], fps=None, question_template: str = default_question_template,
                           answer_template: str = default_answer_template, max_new_tokens: int = 1, temperature: float = 1.0,
                           score_position: str = "end", debug: bool = False) -> Tuple[torch.Tensor, List[Dict]]:
        """forward() plus the per-sample trace of the reference's forward_with_trace (qwen2vl_model.py:303-493): `score_position`
        "end" scores the last n answer tokens (one earlier when the generation ends in a special token), "start" the first n; the
        trace holds the generated text, the scored positions and, per answer token, its probability and the five most likely
        alternatives of softmax(scores / temperature)."""
        assert score_position in ("start", "end"), f"score_position must be 'start' or 'end', got '{score_position}'"
        specials = self._special_ids()
        tok = self.tokenizer
        probs, traces = [], []
        for idx, (step_scores, gen, a_ids) in enumerate(self._generate_scores(images, texts, fps, question_template, answer_template,
                                                                                 max_new_tokens)):
            n_ans = len(a_ids)
            if score_position == "start":
                start, offset = 0, 0
            else:
                offset = 0
                if gen[-1] in specials:
                    n_ans = min(n_ans, len(step_scores) - 1)
                    offset = 1
                start = len(gen) - n_ans - offset
            start = max(start, 0)
            available = len(step_scores) - start
            if available < n_ans:
                print(f"  Warning: Only {available} tokens available at position, need {n_ans}, adjusting")
                n_ans = available
                a_ids = a_ids[:n_ans]
            if n_ans <= 0:
                raise ValueError("No tokens available to score at the specified position")
            scored_indices = list(range(start, start + n_ans))
            rows = torch.stack([step_scores[q] for q in scored_indices])
            p = self._token_probs(rows, torch.tensor(a_ids[:n_ans]), temperature)
            dist = torch.softmax(rows.float() / temperature, dim=-1)
            top_p, top_i = torch.topk(dist, 5, dim=-1)
            details = []
            for t in range(n_ans):
                details.append({'position': start + t, 'expected_token_id': a_ids[t], 'expected_token_text': tok.decode([a_ids[t]]),
                                'probability': float(p[t]),
                                'top_alternatives': [{'token_id': int(ti), 'token_text': tok.decode([int(ti)]), 'probability': float(tp)}
                                                     for tp, ti in zip(top_p[t].tolist(), top_i[t].tolist())]})
            prob = float(torch.prod(p.double()) ** (1.0 / n_ans))
            trace = {'generated_text': tok.decode(gen, skip_special_tokens=True), 'generated_length': len(gen),
                     'score_position': score_position, 'score_start_idx': start, 'scored_indices': scored_indices,
                     'scored_tokens_text': tok.decode(gen[start: start + n_ans], skip_special_tokens=True), 'probability': prob,
                     'token_details': details}
            if debug:
                print(f"Sample {idx + 1}/{len(images)}: {images[idx]} | {texts[idx]}\n  generated: {trace['generated_text']!r}; scoring positions "
                      f"{scored_indices}; probability {prob:.6f}")
            probs.append(prob)
            traces.append(trace)
        return torch.tensor(probs), traces

    @staticmethod
    def _warp(scores: torch.Tensor, temperature: float, top_p: float, top_k: int = 0) -> torch.Tensor:
        """The warper chain HF builds for generate(do_sample=True, temperature=, top_p=) (generation/utils.py
        _get_logits_processor, generation/logits_process.py): TemperatureLogitsWarper (scores / temperature), TopKLogitsWarper
        (generation_config.top_k, default 50: everything below the k-th largest score -> -inf; 0 / None = no filter), then
        TopPLogitsWarper (sorted ascending, the low tail whose cumulative probability is <= 1 - top_p -> -inf, the most likely
        token always stays)."""
        x = scores.float() / temperature
        if top_k and top_k > 0:
            k = min(int(top_k), x.shape[-1])
            x = x.masked_fill(x < torch.topk(x, k)[0][..., -1, None], float("-inf"))
        if top_p is not None and top_p < 1.0:
            srt, idx = x.sort(-1, descending=False)
            drop = torch.softmax(srt, -1).cumsum(-1) <= (1.0 - top_p)
            drop[..., -1:] = False
            x = x.masked_fill(torch.zeros_like(drop).scatter(-1, idx, drop), float("-inf"))
        return x

    def _greedy(self, merged, rows: List[List[int]], grids, max_new_tokens: int, stops: List[int], pick=None):
        """Shared generation loop: prompt rows (token id lists, one video run each) -> (per-sample list of processed score rows,
        per-sample generated ids).  `pick(proc) -> next ids` defaults to argmax (do_sample=False)."""
        n = len(rows)
        L = max(len(r) for r in rows)
        ids = torch.zeros(n, L, dtype=torch.long)
        mask = torch.zeros(n, L, dtype=torch.long)
        for k, r in enumerate(rows):
            ids[k, : len(r)] = torch.tensor(r)
            mask[k, : len(r)] = 1
        step_scores: List[List[torch.Tensor]] = [[] for _ in rows]
        gen: List[List[int]] = [[] for _ in rows]
        done = [False] * n
        state = None
        if max_new_tokens == 1:
            logits = self.engine.score_logits(merged, ids, mask, grids)
        else:
            logits, state = self.engine.prefill(merged, ids, mask, grids, max_new_tokens)
        # Backstop of the fp16 forms (include/vqs_qwen.h): their scales come from a bind-time PROOF, so a non-finite logit cannot be an fp16
        # overflow -- but a drop-in user gets a score for every finite input either way: _with_bf16_fallback re-runs the call on the bf16 forms
        if getattr(self.engine, "fp16_active", False) and not bool(torch.isfinite(logits).all()):
            raise _NonFiniteUnderFp16()
        for step in range(max_new_tokens):
            if step > 0:
                logits = self.engine.decode(state, torch.tensor([gen[k][-1] for k in range(n)], dtype=torch.long))
            proc = self._processed_scores(logits, [rows[k] + gen[k] for k in range(n)])
            nxt = (proc.argmax(-1) if pick is None else pick(proc)).cpu().tolist()
            for k in range(n):
                if done[k]:
                    continue
                step_scores[k].append(proc[k])
                gen[k].append(int(nxt[k]))
                done[k] = int(nxt[k]) in stops
            if all(done):
                break
        return step_scores, gen

    def generate(self, *args, **kw) -> List[str]:
        return self._with_bf16_fallback(lambda: self._generate_impl(*args, **kw))

    @torch.no_grad()
    def _generate_impl(self, images: List[str], texts: List[str], fps=None, max_new_tokens: int = 2048, temperature: float = 0.0,
                       do_sample: bool = None, top_p: float = 0.9) -> List[str]:
        """Free-form answers (qwen2vl_model.py:495-563): the text is the whole user turn; greedy unless temperature > 0, then
        HF's sampling chain for those kwargs: temperature, the generation config's top-k (default 50), nucleus top_p (`_warp`).
        Decoded with skip_special_tokens=True and stripped."""
        assert len(images) == len(texts), "Number of paths and texts must match"
        if do_sample is None:
            do_sample = temperature > 0
        prepared, kinds, media_of = self._prepare_media(images, fps)
        pick = None
        if do_sample and temperature > 0:
            def pick(proc):
                return torch.multinomial(torch.softmax(self._warp(proc, temperature, top_p, getattr(self, "top_k", HF_DEFAULT_TOP_K)), -1), 1)[:, 0]
        stops = self._stop_ids()
        out = [""] * len(images)
        for g, idxs, merged_of in self._media_windows(prepared, media_of):
            for s in range(0, len(idxs), self.max_batch):
                chunk = idxs[s: s + self.max_batch]
                merged = torch.cat([merged_of[media_of[i]] for i in chunk])
                n_tok = g[0] * g[1] * g[2] // self.cfg.vision.merge_unit
                rows = [self.build_ids(texts[i], kinds[media_of[i]], n_tok) for i in chunk]
                _, gen = self._greedy(merged, rows, [g] * len(chunk), max_new_tokens, stops, pick)
                for k, i in enumerate(chunk):
                    out[i] = self.tokenizer.decode(gen[k], skip_special_tokens=True).strip()
        return out
