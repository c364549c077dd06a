#!/usr/bin/env python3
"""Option vit_fp16 on the benchmarked configuration (GPU box): throughput and |delta log P| against fp32 truth with the vision tower in
bf16 (default) and in IEEE fp16, same handle, same batch (bench.synth_batch, the bench's seed), one JSON record.

  python tools/gpu_vit_fp16_check.py [--model clip-flant5-xxl --batch 256 --steps 3 --pairs 16] > gpurun_out/vit_fp16_check.json

Test infrastructure: imports oracle/ (the truth) next to the engine.  Truth is bench.parity_sample's (the fp32 oracle evaluated in torch
fp32 on the device, 16 pairs)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = get_config(a.model)
    t0 = time.perf_counter()
    w = make_seeded_weights(cfg, seed=0, device=dev)
    pix, idx, ids, labels = bench.synth_batch(cfg, a.batch, 1234, dev)
    eng = VqsEngine(cfg, w, device=str(dev))
    out = {"model": cfg.name, "batch": a.batch, "steps": a.steps, "setup_s": round(time.perf_counter() - t0, 1), "modes": {}}
    job = (pix, idx, ids, labels, None)
    try:
        for mode, val in (("bf16 tower (default)", 0), ("fp16 tower (vit_fp16=1)", 1), ("bf16 tower again", 0)):
            eng.set_option("vit_fp16", val)
            lp, _ = eng.score(eng.encode_images(pix), idx, ids, labels)           # warm-up (workspace growth, first launches)
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t_enc = t_all = 0.0
            for _ in range(a.steps):
                e0.record()
                feats = eng.encode_images(pix)
                e1.record()
                lp, _ = eng.score(feats, idx, ids, labels)
                e2.record()
                torch.cuda.synchronize()
                t_enc += e0.elapsed_time(e1)
                t_all += e0.elapsed_time(e2)
            parity, _ = bench.parity_sample(cfg, w, eng, job, a.pairs)
            rec = {"ms_per_step": t_all / a.steps, "ms_tower_and_projector": t_enc / a.steps, "pairs_per_s": a.batch * a.steps / (t_all / 1e3),
                   "finite": bool(torch.isfinite(lp).all()), "dlogp_vs_fp32_truth": parity["gains"], "lp_head": [round(float(x), 6) for x in lp[:2].flatten()]}
            out["modes"][mode] = rec
            if val == 0 and "bf16 tower (default)" in out["modes"] and mode != "bf16 tower (default)":
                rec["bitwise_equal_to_first_bf16_run"] = bool(torch.equal(lp, first_lp))
            if mode == "bf16 tower (default)":
                first_lp = lp.clone()
            print(mode, json.dumps({k: rec[k] for k in ("ms_per_step", "ms_tower_and_projector", "pairs_per_s", "finite")}),
                  {g: (round(v["max"], 6), round(v["mean"], 6)) for g, v in parity["gains"].items()}, file=sys.stderr, flush=True)
    finally:
        eng.close()
    out["device_code_sha256_16"] = bench.device_code_hash()
    out["gemm_kernels_sha256_16"] = bench.gemm_kernels_hash()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
