#!/usr/bin/env python3
"""Lab: GEMM shapes of one XL step under the current VQS_L2_TOUCH mode (read once per process), variants auto / lock-step.
Prints one JSON line per (shape, variant).  HEADS epilogue shapes are timed with the plain bf16 epilogue."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine
engine.load_library()
mode = os.environ.get("VQS_L2_TOUCH", "0")
g = torch.Generator(device="cuda").manual_seed(0)
SHAPES = (("vit_qkv", 147712, 3072, 1024, 0), ("vit_out", 147712, 1024, 1024, 0), ("vit_fc1", 147712, 4096, 1024, 1),
          ("vit_fc2", 147712, 1024, 4096, 0), ("xl_qkv", 155648, 6144, 2048, 0), ("xl_o", 155648, 2048, 2048, 0),
          ("xl_wi", 155648, 10240, 2048, 5), ("xl_wo", 155648, 2048, 5120, 0))
for tag, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if tag.startswith("vit") else None
    for variant in (3, 7):
        out = engine.gemm(A, W, epi, bias=bias, variant=variant)
        for _ in range(3):
            engine.gemm(A, W, epi, bias=bias, out=out, variant=variant)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 12
        e0.record()
        for _ in range(reps):
            engine.gemm(A, W, epi, bias=bias, out=out, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"touch": mode, "tag": tag, "variant": variant, "ms": round(ms, 4),
                          "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del A, W, out
