#!/usr/bin/env python3
"""Lab: which property of the Qwen text gate|up GEMM (51712 x 37888 x 3584, 1.26 PFLOP/s) separates it from the T5-XXL wi GEMM
(155648 x 20480 x 4096, 1.48): one property changed at a time."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine
CASES = [("t5_wi", 155648, 20480, 4096, 5), ("qwen", 51712, 37888, 3584, 5), ("qwen_K4096", 51712, 37888, 4096, 5), ("qwen_N20480", 51712, 20480, 3584, 5),
         ("qwen_M155648_N20480", 155648, 20480, 3584, 5), ("qwen_plain", 51712, 37888, 3584, 0), ("qwen_K3072", 51712, 37888, 3072, 5),
         ("qwen_K3584_lda4096", 51712, 37888, 3584, 5)]
g = torch.Generator(device="cuda").manual_seed(0)
for tag, M, N, K, epi in CASES:
    ld = 4096 if tag.endswith("lda4096") else K
    A = torch.randn(M, ld, device="cuda", generator=g).to(torch.bfloat16)[:, :K]
    W = (torch.randn(N, ld, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)[:, :K]
    out = engine.gemm(A, W, epi, variant=3)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(6):
        engine.gemm(A, W, epi, out=out, variant=3)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 6
    print(json.dumps({"case": tag, "M": M, "N": N, "K": K, "ld": ld, "epi": epi, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del A, W, out
