#!/usr/bin/env python3
"""Lab: several builds of the ring GEMM form (variant 8 of the lab library) side by side in ONE process -- bitwise check against
variant 0 of the same build, then the rate on the path's shapes, interleaved over the builds, next to the product's 8-wave form
(variant 3) and hipBLASLt (torch.matmul).  Usage: lab_ring2.py name=path [name=path ...] [--no-check name] [--variant V] [--all]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402
from tools.lab_call import XL, XXL, VIT, emit, time_ms  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if "=" in a]
    nocheck = {sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--no-check"}
    variants = [int(sys.argv[i + 1]) for i, a in enumerate(sys.argv) if a == "--variant"] or [8]
    shapes = XXL + [VIT[1], VIT[0]] if "--all" not in sys.argv else XXL + XL + VIT
    tol = "--tol" in sys.argv
    libs = []
    for a in args:
        name, path = a.split("=", 1)
        libs.append((name, engine.load_library(os.path.join(ROOT, path))))

    def st():
        return torch.cuda.current_stream().cuda_stream

    def gemm(lib, A, W, out, bias, M, N, K, epi, S, H, v):
        rc = lib.vqs_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), None if bias is None else bias.data_ptr(), None, M, N, K, K, K,
                          out.shape[-1], epi, S, H, v, st())
        assert rc == 0, rc

    from tests.gpu_util import randn_bf16
    for name, lib in libs:
        if name in nocheck:
            continue
        bad = []
        stats = []
        for M, N, K, epi in [(33000, 2048, 128, 0), (33000 - 7, 2048 - 8, 128, 0), (20000, 4096, 1024, 1), (16384, 8192, 256, 5), (131072, 512, 2048, 0),
                             (256 * 300 + 1, 256, 320, 0), (70000, 512, 192, 2), (300, 264, 192, 0), (40, 64, 640, 0)]:
            A = randn_bf16(M, K, seed=81)
            W = randn_bf16(N, K, seed=82, scale=K ** -0.5)
            bias = randn_bf16(N, seed=83) if epi in (0, 1, 2) else None
            NO = N // 2 if epi == 5 else N
            ref = torch.empty(M, NO, dtype=torch.bfloat16, device="cuda")
            gemm(lib, A, W, ref, bias, M, N, K, epi, 0, 0, 0)
            for v in variants:
                first = None
                for rep in range(3):
                    out = torch.full((M, NO), float("nan"), dtype=torch.bfloat16, device="cuda")
                    gemm(lib, A, W, out, bias, M, N, K, epi, 0, 0, v)
                    if first is None:
                        first = out
                    elif not torch.equal(out, first):
                        bad.append({"variant": v, "shape": [M, N, K, epi], "rep": rep, "what": "not repeatable"})
                    if tol:
                        o, r = out.float(), ref.float()
                        d = (o - r).abs()
                        big = torch.maximum(o.abs(), r.abs())
                        ulp = torch.ldexp(torch.ones_like(big), torch.frexp(big).exponent - 8)
                        own = (d / ulp).max().item() if torch.isfinite(d).all() else float("inf")
                        frac = (d > 0).float().mean().item()
                        if rep == 0:
                            stats.append({"shape": [M, N, K, epi], "frac_diff": round(frac, 6), "max_own_ulps": own})
                        if not (own <= 1.0 and frac <= 0.02):
                            bad.append({"variant": v, "shape": [M, N, K, epi], "rep": rep, "max_own_ulps": own, "frac": frac,
                                        "nan": int(torch.isnan(o).sum())})
                    elif not torch.equal(out, ref):
                        d = (out.float() - ref.float()).abs()
                        bad.append({"variant": v, "shape": [M, N, K, epi], "rep": rep, "max_abs_diff": d.max().item(), "frac": (d > 0).float().mean().item()})
        emit({"part": "R2", "lib": name, "check_vs_variant0": ("<= 1 bf16 ulp" if tol else "bitwise"), "result": "ok" if not bad else "MISMATCH",
              "mismatches": bad[:6], "stats": stats})
    g = torch.Generator(device="cuda").manual_seed(0)
    for tag, M, N, K, epi, S, H, has_bias in shapes:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        res = {}
        for rnd in range(2):
            res.setdefault("8-wave v11", []).append(round(2.0 * M * N * K / time_ms(lambda: gemm(libs[0][1], A, W, out, None, M, N, K, 0, 0, 0, 11), 5) / 1e9, 1))
            for name, lib in libs:
                for v in variants:
                    ms = time_ms(lambda: gemm(lib, A, W, out, None, M, N, K, 0, 0, 0, v), 5)
                    res.setdefault("%s v%d" % (name, v), []).append(round(2.0 * M * N * K / ms / 1e9, 1))
            res.setdefault("torch_matmul", []).append(round(2.0 * M * N * K / time_ms(lambda: torch.matmul(A, W.t(), out=out), 5) / 1e9, 1))
        emit({"part": "R2", "shape": tag, "M": M, "N": N, "K": K, "tflops_plain_epilogue": res})
        del A, W, out
        torch.cuda.empty_cache()
    emit({"part": "done"})


if __name__ == "__main__":
    main()
