#!/usr/bin/env python3
"""Why |delta log P| <= 1e-3 cannot hold END TO END between ANY two bf16 evaluations of this path (CPU only; run here
or on the GPU box's host): bf16 rounding decisions are chaotic across GEMM stages.

 (1) the reference as shipped (HF modules cast to bf16, oracle/hf_reference.py) against ITSELF when only HF's attention
     implementation changes (sdpa, what transformers 5.x resolves to, vs eager, the 4.36-era path of the v3.0 release)
     or only the batch composition changes (a pair scored alone vs inside a batch);
 (2) the rounding-matched oracle (oracle/clip_t5_engine_rounding.py) against ITSELF when only the precision of the
     matrix-product accumulation changes (float64 vs float32: ~1e-7 relative, the size of a summation-order effect),
     with the fraction of differing bf16 elements of each encoder layer's norm output: it grows by a large factor per
     layer until it saturates, which is why a free-running comparison lands at the bf16 noise floor and the -m gpu
     tests check every launch stage-locked instead (tests/test_gpu_stage_locked.py).
Output is committed under profiles/ (r2_rounding_chaos.txt)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.clip_t5_engine_rounding import EngineRoundedOracle  # noqa: E402
from oracle.hf_reference import HFReference  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402


def main():
    import warnings
    warnings.filterwarnings("ignore")
    print(f"torch {torch.__version__}, threads {torch.get_num_threads()}")
    for fx in ("e2e_tiny_g1", "e2e_tiny_g4", "e2e_small_g1", "e2e_small_g4"):
        g = np.load(os.path.join(ROOT, "tests", "golden", fx + ".npz"))
        cfg = get_config(fx.split("_")[1])
        w = make_seeded_weights(cfg, seed=int(g["seed"]), device="cpu", lm_head_gain=float(g["gain"]))
        pix = torch.from_numpy(g["pixels"]).to(torch.bfloat16)
        idx, ids, labels = (torch.from_numpy(g[k]) for k in ("img_index", "ids", "labels"))
        truth = torch.from_numpy(g["logprobs_fp32"])
        print(f"\n== {fx}: {ids.shape[0]} pairs, log P range [{truth.min():.2f}, {truth.max():.2f}]")
        # (1) the reference against itself
        ref = {a: HFReference(cfg, w, torch.bfloat16, attn=a) for a in ("sdpa", "eager")}
        lp = {a: r.forward(pix, idx, ids, labels)["label_logprobs"] for a, r in ref.items()}
        print(f"  reference (HF bf16) vs fp32 truth:        sdpa {float((lp['sdpa'] - truth).abs().max()):.4f}   eager {float((lp['eager'] - truth).abs().max()):.4f}")
        print(f"  reference vs ITSELF, sdpa vs eager:       max |dlogP| = {float((lp['sdpa'] - lp['eager']).abs().max()):.4f}")
        alone = torch.stack([ref["sdpa"].forward(pix, idx[b:b + 1], ids[b:b + 1], labels[b:b + 1])["label_logprobs"][0] for b in range(ids.shape[0])])
        print(f"  reference vs ITSELF, pair alone vs batch: max |dlogP| = {float((alone - lp['sdpa']).abs().max()):.4f}")
        # (2) the rounding-matched oracle against itself
        recs, lps = {}, {}
        for name, acc in (("fp64", torch.float64), ("fp32", torch.float32)):
            o = EngineRoundedOracle(cfg, w, acc=acc)
            o.record = {}
            lps[name] = o.forward(pix.float(), idx, ids, labels)["label_logprobs"]
            recs[name] = o.record
        print(f"  rounding-matched oracle vs fp32 truth:    {float((lps['fp64'] - truth).abs().max()):.4f}")
        print(f"  rounding-matched oracle vs ITSELF, fp64 vs fp32 accumulation: max |dlogP| = {float((lps['fp64'] - lps['fp32']).abs().max()):.4f}")
        line = []
        for stack, n in (("vit", cfg.vision.layers_run), ("enc", cfg.t5.layers), ("dec", cfg.t5.dec_layers)):
            for i in range(n):
                a, b = recs["fp64"][f"{stack}.{i}.xn0"], recs["fp32"][f"{stack}.{i}.xn0"]
                line.append(f"{stack}.{i} {float((a != b).float().mean()):.1e}")
        print("  fraction of bf16 norm outputs that differ, layer by layer: " + "  ".join(line))


if __name__ == "__main__":
    main()
