#!/usr/bin/env python3
"""Interleaved A/B of execution-option sets on ONE engine, one process, one box (round 6): the bench batch (clip-flant5-xxl,
B = 256) is scored under each option set in turn, `--rounds` times round-robin, `--steps` steps each; per set the median / min step time.
Box-to-box and minute-to-minute drift (+-1.5 %) cancels: every set sees every phase of the run.
  python tools/ab_options.py --sets "shipped;enc_fp16=2;enc_fp16=0" --rounds 6 --steps 3"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.engine import VqsEngine  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="clip-flant5-xxl")
    ap.add_argument("--sets", default="shipped;enc_fp16=2;enc_fp16=0")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    cfg = get_config(a.model)
    dev = torch.device("cuda:0")
    eng = VqsEngine(cfg, make_seeded_weights(cfg, seed=0, device=dev), device=dev)
    args = bench.parse_args(["--model", a.model, "--steps", "1", "--warmup", "1", "--cpu-pairs", "0", "--also", "none"]) if hasattr(bench, "parse_args") else None
    jobs, _ = bench.make_jobs(args, cfg, 0, 1, dev)
    pixels, img_index, ids, labels, _ = jobs[0]
    sets = []
    for sname in a.sets.split(";"):
        opts = {} if sname == "shipped" else {k: int(v) for k, v in (x.split("=") for x in sname.split(","))}
        sets.append((sname, opts))
    defaults = {k: eng.get_option(k) for k in ("vit_fp16", "proj_fp16", "enc_fp16", "dec_fp16")}

    def step():
        return eng.score(eng.encode_images(pixels), img_index, ids, labels)

    times = {n: [] for n, _ in sets}
    for n, o in sets:                                     # warm every form once
        for k, v in {**defaults, **o}.items():
            eng.set_option(k, v)
        step()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        for n, o in (sets if r % 2 == 0 else sets[::-1]):
            for k, v in {**defaults, **o}.items():
                eng.set_option(k, v)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            ev[0].record()
            for i in range(a.steps):
                step()
                ev[i + 1].record()
            torch.cuda.synchronize()
            times[n] += [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
    B = ids.shape[0]
    out = {"model": a.model, "batch": B, "rounds": a.rounds, "steps_per_round": a.steps, "sets": {}}
    for n, _ in sets:
        t = sorted(times[n])
        out["sets"][n] = {"median_ms": round(t[len(t) // 2], 2), "min_ms": round(t[0], 2), "max_ms": round(t[-1], 2), "pairs_per_s_median": round(1e3 * B / t[len(t) // 2], 2)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
