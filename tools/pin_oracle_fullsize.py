#!/usr/bin/env python3
"""Pin the fp32 oracle to the HF modules AT THE REAL ARCHITECTURE (CPU only).  tests/test_oracle_golden.py pins
oracle/clip_t5_oracle.py to HF on tiny / small dimensions; this runs both at full clip-flant5-xl (or -xxl) size on the same seeded bf16
weights in fp32 -- HF CLIPVisionModel -> hidden_states[-2][:, 1:] -> mlp2x_gelu -> splice -> T5ForConditionalGeneration (oracle/hf_reference.py,
dtype float32) against Oracle -- on ragged pairs, and records the distances stage by stage.  Expectation: fp32 summation-order noise.

  python tools/pin_oracle_fullsize.py --model clip-flant5-xl --out profiles/r4_oracle_pinned_at_full_size_xl.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle.clip_t5_oracle import Oracle  # noqa: E402
from oracle.hf_reference import HFReference  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="clip-flant5-xl")
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--gain", type=float, default=4.0, help="lm_head gain of the seeded weights (a peaked head makes the logits test sharper)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import warnings
    warnings.filterwarnings("ignore")
    cfg = get_config(a.model)
    w = make_seeded_weights(cfg, seed=0, device="cpu", lm_head_gain=a.gain)
    pix, idx, ids, labels = bench.synth_batch(cfg, a.pairs, 77, "cpu", ragged=True)       # captions of different lengths: padding + masks
    idx, ids, labels = idx.long(), ids.long(), labels.long()
    keep = int((ids != 0).sum(1).max())
    ids = ids[:, :keep]
    t0 = time.time()
    o = Oracle(cfg, w).forward(pix.float(), idx, ids, labels, return_stages=True)
    t_o = time.time() - t0
    t0 = time.time()
    hf = HFReference(cfg, w, torch.float32)
    feats_hf = hf.encode_images(pix.float())
    out_hf = hf.score(feats_hf, idx, ids, labels)
    t_hf = time.time() - t0
    lens = (ids != 0).sum(1) - 1 + cfg.vision.n_patches
    valid = torch.arange(o["logits"].shape[1])[None] >= 0
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())
    rec = {"model": cfg.name, "pairs": a.pairs, "encoder_lengths": lens.tolist(), "lm_head_gain": a.gain, "threads": torch.get_num_threads(),
           "seconds": {"oracle_fp32": round(t_o, 1), "hf_fp32": round(t_hf, 1)},
           "projected_image_features_rel_to_absmax": rel(o["proj"], feats_hf.float()),
           "logits_max_abs_diff": float((o["logits"] - hf.last_logits).abs().max()), "logits_absmax": float(hf.last_logits.abs().max()),
           "logits_rel_to_absmax": rel(o["logits"], hf.last_logits),
           "label_logprobs_max_abs_diff": float((o["label_logprobs"] - out_hf[0]).abs().max()),
           "label_logprobs_hf": out_hf[0].tolist(), "scores_max_abs_diff": float((o["scores"] - out_hf[1]).abs().max()),
           "what": "oracle/clip_t5_oracle.py (fp32 restatement) vs the HF modules in fp32 (oracle/hf_reference.py) on the same seeded bf16 weights, ragged pairs of the bench generator"}
    print(json.dumps(rec, indent=1))
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    assert rec["logits_max_abs_diff"] <= 5e-4 * max(1.0, rec["logits_absmax"]) and rec["label_logprobs_max_abs_diff"] <= 5e-4, "oracle is not the HF arithmetic at this size"


if __name__ == "__main__":
    main()
