#!/usr/bin/env python3
"""Lab: is the GEMM plateau a power (DVFS) plateau?  Same kernel, same shapes, operands of different toggle activity."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(0)
for tag, M, N, K in (("sq8192", 8192, 8192, 8192), ("xl_qkv", 155648, 6144, 2048)):
    for kind in ("randn", "zeros", "ones", "uniform01"):
        if kind == "randn":
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        elif kind == "zeros":
            A = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16); W = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
        elif kind == "ones":
            A = torch.ones(M, K, device="cuda", dtype=torch.bfloat16); W = torch.ones(N, K, device="cuda", dtype=torch.bfloat16)
        else:
            A = torch.rand(M, K, device="cuda", generator=g).to(torch.bfloat16); W = torch.rand(N, K, device="cuda", generator=g).to(torch.bfloat16)
        for v in (3, 5):
            out = engine.gemm(A, W, 0, variant=v)
            ms = timeit(lambda: engine.gemm(A, W, 0, out=out, variant=v))
            print(json.dumps({"tag": tag, "kind": kind, "variant": v, "ms": round(ms, 4), "tflops": round(2 * M * N * K / ms / 1e9, 1)}), flush=True)
        if tag == "sq8192":
            Wt = W.t().contiguous()
            ms = timeit(lambda: torch.matmul(A, Wt))
            print(json.dumps({"tag": tag, "kind": kind, "variant": "torch", "ms": round(ms, 4), "tflops": round(2 * M * N * K / ms / 1e9, 1)}), flush=True)
