#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X: the GEMM shapes of the XL/XXL path, attention, norms.
Writes JSON lines to gpurun_out/microbench.jsonl.  Timing: HIP events, median of `reps` launches."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "microbench.jsonl")
VARIANTS = tuple(int(x) for x in os.environ.get("VQS_BENCH_VARIANTS", "0,2").split(","))


def timeit(fn, reps=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def emit(rec):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(json.dumps(rec), flush=True)


def main():
    dev = "cuda"
    quick = "--quick" in sys.argv
    gemm_only = "--gemm-only" in sys.argv
    if "--no-gemm" in sys.argv:
        shapes_skip = True
    else:
        shapes_skip = False
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = [  # (tag, M, N, K, epilogue)
        ("vit_qkv", 147712, 3072, 1024, 0), ("vit_out", 147712, 1024, 1024, 3), ("vit_fc1", 147712, 4096, 1024, 1),
        ("vit_fc2", 147712, 1024, 4096, 3), ("xl_qkv", 155648, 6144, 2048, 6), ("xl_o", 155648, 2048, 2048, 3),
        ("xl_wi", 155648, 10240, 2048, 5), ("xl_wo", 155648, 2048, 5120, 3), ("xl_dec_skinny", 512, 6144, 2048, 0),
        ("xl_lm_head", 512, 32128, 2048, 3), ("sq4096", 4096, 4096, 4096, 0), ("sq8192", 8192, 8192, 8192, 0),
        ("xxl_wi", 38912, 20480, 4096, 5), ("xxl_wo", 38912, 4096, 10240, 3),
    ]
    if quick:
        shapes = shapes[:2] + shapes[10:11]
    for tag, M, N, K, epi in ([] if shapes_skip else shapes):
        A = (torch.randn(M, K, device=dev, generator=g)).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        resid = torch.randn(M, N, device=dev, generator=g) if epi == 4 else None
        for variant in VARIANTS:
            kw = dict(S=608, H=N // 192) if epi == 6 else {}
            out = engine.gemm(A, W, epi, resid=resid, variant=variant, **kw)
            med, best = timeit(lambda: engine.gemm(A, W, epi, resid=resid, out=out, variant=variant, **kw))
            fl = 2.0 * M * N * K
            emit({"kernel": "gemm", "tag": tag, "M": M, "N": N, "K": K, "epi": epi, "variant": variant, "ms": med,
                  "tflops": fl / med / 1e9, "tflops_best": fl / best / 1e9})
        del A, W, resid, out
    if gemm_only:
        return
    # a torch (hipBLASLt) cross-check number for context on two shapes
    for M, N, K in (() if shapes_skip else ((4096, 4096, 4096), (155648, 2048, 2048))):
        A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        med, best = timeit(lambda: torch.matmul(A, W.t()))
        emit({"kernel": "torch.matmul(hipBLASLt)", "M": M, "N": N, "K": K, "ms": med, "tflops": 2.0 * M * N * K / med / 1e9})
        del A, W
    for tag, B, H, S, bias in (("vit", 256, 16, 577, False), ("t5xl", 256, 32, 608, True), ("t5xxl", 64, 64, 608, True)):
        q = torch.randn(B, H, S, 64, device=dev, generator=g).to(torch.bfloat16) * 0.5
        k = torch.randn(B, H, S, 64, device=dev, generator=g).to(torch.bfloat16) * 0.5
        v = torch.randn(B, H, S, 64, device=dev, generator=g).to(torch.bfloat16)
        table = torch.randn(H, 2 * S - 1, device=dev, generator=g) if bias else None
        kl = torch.full((B,), S, dtype=torch.int32, device=dev) if bias else None
        med, best = timeit(lambda: engine.attention(q, k, v, 1.0 if bias else 0.125, bias_table=table, key_len=kl))
        fl = 4.0 * B * H * S * S * 64
        emit({"kernel": "attention", "tag": tag, "B": B, "H": H, "S": S, "ms": med, "tflops": fl / med / 1e9})
        del q, k, v
    for M, D in ((155648, 2048), (147712, 1024), (38912, 4096)):
        x = torch.randn(M, D, device=dev, generator=g)
        w = torch.ones(D, device=dev, dtype=torch.bfloat16)
        med, best = timeit(lambda: engine.rmsnorm(x, w, 1e-6))
        emit({"kernel": "rmsnorm", "M": M, "D": D, "ms": med, "GBps": M * D * 6 / med / 1e6})
        med, best = timeit(lambda: engine.layernorm(x, w, w, 1e-5))
        emit({"kernel": "layernorm", "M": M, "D": D, "ms": med, "GBps": M * D * 6 / med / 1e6})
        # the in-situ form: fused residual add (bf16 delta) + norm; 12 algorithmic bytes per element
        d = (torch.randn(M, D, device=dev, generator=g) * 0.01).to(torch.bfloat16)
        med, best = timeit(lambda: engine.rmsnorm(x, w, 1e-6, delta=d))
        emit({"kernel": "rmsnorm+add", "M": M, "D": D, "ms": med, "GBps": M * D * 12 / med / 1e6})
        med, best = timeit(lambda: engine.layernorm(x, w, w, 1e-5, delta=d))
        emit({"kernel": "layernorm+add", "M": M, "D": D, "ms": med, "GBps": M * D * 12 / med / 1e6})
        del x, d


if __name__ == "__main__":
    main()
