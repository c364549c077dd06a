#!/usr/bin/env python3
"""Lab: row pitch of the A and W operands (elements) against GEMM rate, Qwen2.5-VL-7B shapes, library form.  Finding that started
it (tools/qwen_gemm_why.py): 51712 x 37888 x 3584 runs 9 % faster when both operands sit at a 4096-element pitch."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine
MT, MV = 64 * 808, 64 * 3072
CASES = []
for pa, pw in [(3584, 3584), (4096, 3584), (3584, 4096), (4096, 4096), (3648, 3648), (3840, 3840), (3584 + 32, 3584 + 32)]:
    CASES.append(("txt_gate_up", MT, 37888, 3584, 5, pa, pw))
for pa, pw in [(3584, 3584), (4096, 4096), (3648, 3648)]:
    CASES.append(("txt_qkv", MT, 4608, 3584, 0, pa, pw))
for pa, pw in [(18944, 18944), (18944 + 64, 18944 + 64), (20480, 20480)]:
    CASES.append(("txt_down", MT, 3584, 18944, 0, pa, pw))
for pa, pw in [(1280, 1280), (2048, 2048), (1344, 1344), (1536, 1536)]:
    CASES.append(("vis_gate_up", MV, 6848, 1280, 5, pa, pw))
for pa, pw in [(3456, 3456), (4096, 4096), (3520, 3520)]:
    CASES.append(("vis_down", MV, 1280, 3456, 0, pa, pw))
g = torch.Generator(device="cuda").manual_seed(0)
for tag, M, N, K, epi, pa, pw in CASES:
    A = torch.randn(M, pa, device="cuda", generator=g).to(torch.bfloat16)[:, :K]
    W = (torch.randn(N, pw, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)[:, :K]
    out = engine.gemm(A, W, epi, variant=3)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        engine.gemm(A, W, epi, out=out, variant=3)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    print(json.dumps({"case": tag, "M": M, "N": N, "K": K, "pitch_a": pa, "pitch_w": pw, "epi": epi, "ms": round(ms, 4),
                      "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}), flush=True)
    del A, W, out
