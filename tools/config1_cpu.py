#!/usr/bin/env python3
"""BASELINE.json configs[0]: the reference CPU path, clip-flant5-xl, 4 images x 4 prompts (16 pairs), no GPU.

Images: /root/reference/images/0/{DALLE3.png,DeepFloyd.jpg,Midjourney.jpg,SDXL.jpg} (real decode + expand2square + 336-px
CLIP preprocessing, mm_utils.py:128-139); prompts: the four captions of /root/reference/V_3.0_README.md:122,123,178,179;
templates V_3.0_README.md:213-214.  The arithmetic is the reference's own (HF CLIPVisionModel + T5ForConditionalGeneration
cast to bf16, mm_utils.py:228; oracle/hf_reference.py); weights are seeded random at the exact XL architecture and token
ids come from a word-hash tokenizer (neither the checkpoint nor spiece.model exists offline), so SCORES are not
meaningful -- wall time, per-stage time and fp32-vs-bf16 |delta| are.  Two drivers:
  reference semantics  Score.forward as written (score.py:104-106): per image, model.forward([img]*N, texts) -- the
                       image is decoded, preprocessed and encoded N times per row;
  this repo's API      t2v_metrics_amd.VQAScore(device='cpu', engine=HFEngine) -> forward_grid: unique images once.
Where /root/reference does not exist (the GPU box), the four images are seeded stand-ins of the SAME file formats and sizes
(1024x1024 PNG, 256x256 JPEG, two 1024x1024 JPEGs) and the captions are the four README strings restated below, so that
`run(...)` is callable from bench.py's cpu_baseline leg ("timed on the host cores of the same box").
Output of the CLI: profiles/r3_config1_cpu_<model>_4x4.json"""
import json
import os
import sys
import time
import warnings
import zlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import t2v_metrics_amd as t2v  # noqa: E402
from oracle.hf_reference import HFEngine  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402

IMAGES = ["DALLE3.png", "DeepFloyd.jpg", "Midjourney.jpg", "SDXL.jpg"]
IMAGE_SIZES = [(1024, 1024), (256, 256), (1024, 1024), (1024, 1024)]      # of the reference's files, in that order
README_LINES = (178, 179, 122, 123)
CAPTIONS = ["The brown dog chases the black dog around the tree.",           # V_3.0_README.md:178,179,122,123
            "Two cats sit at the window, the blue one intently watching the rain, the red one curled up asleep.",
            "someone talks on the phone angrily while another person sits happily",
            "someone talks on the phone happily while another person sits angrily"]


class WordHashTokenizer:
    """tokenizer(text).input_ids protocol; one id per whitespace word + </s> (= 1).  Stand-in for SentencePiece."""

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, text):
        class R:
            pass
        r = R()
        r.input_ids = [3 + zlib.crc32(w.encode()) % (min(self.vocab, 32100) - 3) for w in text.split()] + [1]
        return r


def readme_captions():
    if not os.path.exists("/root/reference/V_3.0_README.md"):
        return list(CAPTIONS)
    lines = open("/root/reference/V_3.0_README.md").read().splitlines()
    caps = []
    import re
    for ln in README_LINES:
        caps.append(re.findall(r'"([^"]+)"', lines[ln - 1])[-1])      # the caption is the last double-quoted string of the line
    return caps


def image_paths():
    """the reference's four images where they exist; else seeded stand-ins of the same formats and sizes in a temp directory"""
    real = [os.path.join("/root/reference/images/0", f) for f in IMAGES]
    if all(os.path.exists(p) for p in real):
        return real, "the reference's images/0"
    import tempfile
    import numpy as np
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="vqs_config0_")
    out = []
    rng = np.random.RandomState(0)
    for name, (w_, h_) in zip(IMAGES, IMAGE_SIZES):
        small = rng.randint(0, 256, (h_ // 16, w_ // 16, 3), dtype=np.uint8)
        im = Image.fromarray(small).resize((w_, h_), Image.BICUBIC)
        p = os.path.join(tmp, name)
        im.save(p)
        out.append(p)
    return out, "seeded stand-ins of the reference's image formats and sizes (/root/reference is absent on this box)"


def run(model="clip-flant5-xl", reps=3, dtypes=(torch.bfloat16, torch.float32), weights=None, verbose=True, api_pass=True):
    cfg = get_config(model)
    paths, image_note = image_paths()
    texts = readme_captions()
    t0 = time.perf_counter()
    w = weights if weights is not None else make_seeded_weights(cfg, seed=0, device="cpu")
    t_w = time.perf_counter() - t0
    tok = WordHashTokenizer(cfg.t5.vocab)
    out = {"config": "BASELINE.json configs[0]: %s on CPU, 4 images x 4 prompts (16 pairs)" % model, "images": image_note, "texts": texts,
           "nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
           "cpu_model": [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0],
           "weights": f"seeded random, bf16, generated in {t_w:.0f} s", "tokenizer": "word-hash stand-in (no spiece.model offline)", "runs": {}}
    import transformers
    out["transformers"] = transformers.__version__
    scores = {}
    for dtype, tag in ((torch.bfloat16, "bf16 (reference as shipped)"), (torch.float32, "fp32 (same weights, fp32 arithmetic)")):
        if dtype not in dtypes or (dtype == torch.float32 and model.endswith("xxl")):
            continue
        eng = HFEngine(cfg, w, dtype)
        scorer = t2v.VQAScore(model="clip-flant5-xl", device="cpu", cache_dir="/tmp/none", config=cfg, tokenizer=tok, engine=eng)
        m = scorer.model
        # ---- reference semantics: Score.forward's row loop, N-fold image work per row
        walls, stages = [], []
        for r in range(reps + 1 if dtype == torch.bfloat16 else 1):
            eng.seconds = {"vision+projector": 0.0, "t5": 0.0}
            t_pre = 0.0
            t0 = time.perf_counter()
            rows = []
            for p in paths:
                t1 = time.perf_counter()
                px = m.load_images([p] * len(texts))                       # decode + expand2square + preprocess, N times
                t_pre += time.perf_counter() - t1
                q = [t2v.models.vqascore_models.clip_t5_model.default_question_template.format(t) for t in texts]
                a = [t2v.models.vqascore_models.clip_t5_model.default_answer_template.format(t) for t in texts]
                ids, lab = m.tokenize(q, a)
                feats = eng.encode_images(px)
                lp, sc = eng.score(feats, torch.arange(len(texts)), ids, lab)
                rows.append(sc)
            walls.append(time.perf_counter() - t0)
            stages.append({"decode+preprocess_s": t_pre, **{k + "_s": v for k, v in eng.seconds.items()}})
        grid_ref = torch.stack(rows)
        timed = sorted(walls[1:]) or walls
        med = timed[len(timed) // 2]
        run = {"reference_semantics": {"wall_s_all": walls, "wall_s_median_after_warmup": med, "pairs_per_s": 16 / med, "stages_last_rep": stages[-1]}}
        if not api_pass:
            out["runs"][tag] = run
            del eng, scorer
            continue
        # ---- this repo's API on the same engine: unique images once
        walls = []
        for r in range(reps + 1 if dtype == torch.bfloat16 else 1):
            eng.seconds = {"vision+projector": 0.0, "t5": 0.0}
            t0 = time.perf_counter()
            grid = scorer(images=paths, texts=texts)
            walls.append(time.perf_counter() - t0)
        timed = sorted(walls[1:]) or walls
        med = timed[len(timed) // 2]
        run["this_repo_api_forward_grid"] = {"wall_s_all": walls, "wall_s_median_after_warmup": med, "pairs_per_s": 16 / med,
                                             "stages_last_rep": {k + "_s": v for k, v in eng.seconds.items()}}
        assert tuple(grid.shape) == (4, 4) and bool(((grid >= 0) & (grid <= 1)).all())          # reference test.py:138-139
        run["max_rel_diff_grid_vs_row_loop"] = float(((grid.cpu() - grid_ref).abs() / grid_ref).max())
        scores[tag] = grid.cpu()
        out["runs"][tag] = run
        if verbose:
            print(tag, json.dumps(run)[:400], flush=True)
        del eng, scorer
    if len(scores) == 2:
        a, b = scores["bf16 (reference as shipped)"], scores["fp32 (same weights, fp32 arithmetic)"]
        out["bf16_vs_fp32_max_abs_dlog_score"] = float((a.log() - b.log()).abs().max())
    return out


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else "clip-flant5-xl"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = run(model, reps)
    dst = os.path.join(ROOT, "profiles", "r3_config1_cpu_%s_4x4.json" % model.replace("clip-flant5-", ""))
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
