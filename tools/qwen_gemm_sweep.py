#!/usr/bin/env python3
"""Lab: the Qwen2.5-VL-7B GEMM shapes (bench batch 64) under the library's tile order and alternatives (gm, ns); quad form.
Writes one JSON line per (shape, order) to stdout.  Tile order is bitwise-neutral (a permutation of the tile list)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine

B = int(os.environ.get("QB", "64"))
MT, MV = B * 808, B * 3072
SHAPES = [("txt_gate_up", MT, 37888, 3584, 5), ("txt_down", MT, 3584, 18944, 0), ("txt_qkv", MT, 4608, 3584, 0), ("txt_o", MT, 3584, 3584, 0),
          ("vis_gate_up", MV, 6848, 1280, 5), ("vis_down", MV, 1280, 3456, 0), ("vis_qkv", MV, 3840, 1280, 0), ("vis_proj", MV, 1280, 1280, 0)]
ORDERS = [None, (8, 1), (4, 1), (4, 2), (2, 2), (2, 4), (16, 1), (1, 4), (32, 1)]
only = os.environ.get("QSHAPES")
g = torch.Generator(device="cuda").manual_seed(0)
for tag, M, N, K, epi in SHAPES:
    if only and tag not in only.split(","):
        continue
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    out = None
    for order in ORDERS:
        try:
            out = engine.gemm(A, W, epi, out=out, variant=3, tile_order=order)
        except Exception as e:  # noqa
            print(json.dumps({"shape": tag, "order": order, "error": str(e)}))
            continue
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 6
        ev[0].record()
        for _ in range(reps):
            engine.gemm(A, W, epi, out=out, variant=3, tile_order=order)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        print(json.dumps({"shape": tag, "M": M, "N": N, "K": K, "epi": epi, "order": order, "ms": round(ms, 4),
                          "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del A, W, out
