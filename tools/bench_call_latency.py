#!/usr/bin/env python3
"""Latency of ONE call of the drop-in API -- the reference's per-pair loops (score.py:143-153, the CameraBench scripts): `VQAScore()(images=[path], texts=[text])`
from a PNG file on disk to the score on the host, next to the engine's share of it (pixels and ids resident).  Seeded weights, the stand-in whitespace
tokenizer of tools/bench_pipeline.py.  One JSON line per grid size."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import t2v_metrics_amd as t2v  # noqa: E402
from bench_pipeline import WordTokenizer  # noqa: E402


def median_ms(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3 * ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="clip-flant5-xxl")
    ap.add_argument("--grids", default="1x1,1x4,4x4")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cprofile", action="store_true", help="print the 35 most expensive functions (cumulative) of 20 1x1 calls")
    args = ap.parse_args()
    from t2v_metrics_amd.config import get_config
    from t2v_metrics_amd.models.vqascore_models.clip_t5_model import default_question_template
    cfg = get_config(args.model)
    tmp = tempfile.mkdtemp(prefix="vqs_call_")
    rng = np.random.RandomState(0)
    paths = []
    for i in range(8):
        base = rng.randint(0, 256, (args.size // 8, args.size // 8, 3), dtype=np.uint8)
        p = os.path.join(tmp, f"im{i}.png")
        Image.fromarray(base).resize((args.size, args.size), Image.BILINEAR).save(p)
        paths.append(p)
    texts = [f"a photo number {i} of someone doing something in place {i}" for i in range(8)]
    scorer = t2v.VQAScore(model=args.model, device="cuda", weights="seeded", tokenizer=WordTokenizer(cfg.t5.vocab))
    m = scorer.model
    if args.cprofile:
        import cProfile
        import pstats
        for _ in range(3):
            scorer(images=paths[:1], texts=texts[:1])
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            scorer(images=paths[:1], texts=texts[:1]).cpu()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
        return
    for grid in args.grids.split(","):
        ni, nt = (int(x) for x in grid.split("x"))
        ims, txs = paths[:ni], texts[:nt]
        for _ in range(3):
            sc = scorer(images=ims, texts=txs)
        call = median_ms(lambda: scorer(images=ims, texts=txs).cpu(), args.reps)
        load = median_ms(lambda: m.load_images(ims), args.reps)
        tok = median_ms(lambda: m.tokenize([default_question_template.format(t) for t in txs], ["Yes"] * nt), args.reps)
        px = m.load_images(ims)
        ids, lab = m.tokenize([default_question_template.format(t) for t in txs for _ in range(1)] * ni, ["Yes"] * (ni * nt))
        idx = torch.arange(ni, dtype=torch.int32).repeat_interleave(nt)
        eng = median_ms(lambda: m.engine.score(m.engine.encode_images(px), idx, ids, lab), args.reps)
        print(json.dumps({"model": args.model, "grid": grid, "pairs": ni * nt, "call_ms": call, "of_which": {"load_images_ms (PNG decode, resize, H2D, normalise)": load,
                          "tokenize_ms": tok, "engine_ms (encode_images + score, inputs resident)": eng}, "host_overhead_ms": call - eng,
                          "score_shape": list(sc.shape)}), flush=True)


if __name__ == "__main__":
    main()
