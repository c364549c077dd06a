#!/usr/bin/env python3
"""Lab (round 6, VERDICT r5 item 3: "diagnose `wo` with counters, not by elimination").  The XXL encoder's four GEMM shapes on bf16 operands,
quad form, library tile order, plus torch.matmul (hipBLASLt) on the same operands -- a few launches each, meant to run under
`rocprofv3 --kernel-trace --pmc ...` passes (tools/gpu_pmc.sh wo "python tools/lab_gemm_wo.py"): the L2's hit / miss counts and its
fabric-side read requests per launch put `wo` (K = 10 240, A = 3.2 GB) beside `wi` / `qkv` / `o` (K = 4 096).
part T (no profiler needed): the (gm, ns) tile-order sweep of `wo` with the Infinity-Cache working set of each order."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

SHAPES = [("xxl enc wo", 155648, 4096, 10240, 0), ("xxl enc o", 155648, 4096, 4096, 0), ("xxl enc qkv", 155648, 12288, 4096, 0), ("xxl enc wi", 155648, 20480, 4096, 5)]


def time_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]


def main():
    part = sys.argv[1] if len(sys.argv) > 1 else "P"
    g = torch.Generator(device="cuda").manual_seed(5)
    out = []
    for tag, M, N, K, epi in SHAPES:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        flops = 2.0 * M * N * K
        if part == "P":
            ms = time_ms(lambda: engine.gemm(A, W, epi, variant=3), 3)
            rec = {"shape": tag, "quad_ms": round(ms, 3), "quad_tflops": round(flops / ms / 1e9, 1)}
            if epi == 0:
                ms = time_ms(lambda: torch.matmul(A, W.t()), 3)
                rec.update(hipblaslt_ms=round(ms, 3), hipblaslt_tflops=round(flops / ms / 1e9, 1))
            out.append(rec)
        elif tag.endswith("wo"):
            rec = {"shape": tag, "by_order": {}}
            for gm in (1, 2, 4, 8, 16):
                for ns in (1, 2, 4):
                    ws_mb = (8.0 * gm * 256 * K * 2 + N / ns * K * 2) / 1e6          # vqs_kernels.h tile_order_working_set_mb
                    ms = time_ms(lambda: engine.gemm(A, W, epi, variant=3, tile_order=(gm, ns)), 5)
                    rec["by_order"][f"{gm}x{ns}"] = {"working_set_mb": round(ws_mb), "tflops": round(flops / ms / 1e9, 1)}
            out.append(rec)
        del A, W
        torch.cuda.empty_cache()
    for r in out:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
