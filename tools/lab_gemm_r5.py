#!/usr/bin/env python3
"""Lab (round 5): what is left between the quad GEMM and the vendor GEMM, measured on one box.
  part S  tile order (gm, ns) of the QUAD form on the XXL encoder's shapes, bf16 and fp16 operands (round 2's rule was found with the
          8-wave kernels); torch.matmul beside it
  part A  the T5 encoder's attention launch (B 256, H 64, S 608, bias + key mask) on bf16 vs fp16 tensors; the ViT's (H 16, S 577)
  part H  torch.matmul on the four shapes under the caller's rocprofv3 --kernel-trace: which hipBLASLt kernel runs each (see the trace)
Appends JSON lines to gpurun_out/lab_gemm_r5.jsonl."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "lab_gemm_r5.jsonl")
SHAPES = [("xxl enc wo", 155648, 4096, 10240, 0, 0, 0, 0), ("xxl enc o", 155648, 4096, 4096, 0, 0, 0, 2), ("xxl enc qkv", 155648, 12288, 4096, 6, 608, 64, 1),
          ("xxl enc wi", 155648, 20480, 4096, 5, 0, 0, 2)]
ORDERS = [(0, 0), (8, 1), (4, 1), (2, 1), (16, 1), (8, 2), (4, 2), (2, 2), (8, 4), (4, 4), (2, 4), (6, 1), (6, 2), (12, 1), (3, 2)]


def emit(rec):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(json.dumps(rec), flush=True)


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
    parts = sys.argv[1] if len(sys.argv) > 1 else "SAH"
    g = torch.Generator(device="cuda").manual_seed(5)
    for tag, M, N, K, epi, S, H, ft in SHAPES:
        dt = torch.float16 if ft else torch.bfloat16
        A = torch.randn(M, K, device="cuda", generator=g).to(dt)
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        flops = 2.0 * M * N * K
        if "S" in parts:
            rec = {"part": "S", "shape": tag, "M": M, "N": N, "K": K, "epilogue": epi, "ftype": ft, "tflops_by_order": {}}
            for gm, ns in ORDERS:
                fn = lambda: engine.gemm(A, W, epi, S=S, H=H, variant=3, tile_order=(gm, ns) if gm else None, ftype=ft)
                rec["tflops_by_order"][f"{gm}x{ns}"] = round(flops / time_ms(fn, 5) / 1e9, 1)
            if ft:                      # the same shape on bf16 operands, library order
                Ab, Wb = A.to(torch.bfloat16), W.to(torch.bfloat16)
                rec["bf16_operands_default_order"] = round(flops / time_ms(lambda: engine.gemm(Ab, Wb, epi, S=S, H=H, variant=3), 5) / 1e9, 1)
                del Ab, Wb
            if epi == 0:
                rec["torch_matmul"] = round(flops / time_ms(lambda: torch.matmul(A, W.t()), 5) / 1e9, 1)
            emit(rec)
        if "H" in parts and epi in (0, 5, 6):
            Ab, Wb = A.to(torch.bfloat16), W.to(torch.bfloat16)
            for _ in range(3):
                torch.matmul(Ab, Wb.t())
            torch.cuda.synchronize()
            emit({"part": "H", "shape": tag, "note": "3 x torch.matmul bf16 issued: see the kernel trace for the Tensile kernel's name"})
            del Ab, Wb
        del A, W
        torch.cuda.empty_cache()
    if "A" in parts:
        for tag, B, H, S, bias in (("t5-xxl encoder", 256, 64, 608, True), ("vit-l/14-336", 256, 16, 577, False)):
            q = (torch.randn(B, H, S, 64, device="cuda", generator=g) * (0.5 if bias else 1.0))
            k = (torch.randn(B, H, S, 64, device="cuda", generator=g) * (0.5 if bias else 1.0))
            v = torch.randn(B, H, S, 64, device="cuda", generator=g)
            table = torch.randn(H, 2 * S - 1, device="cuda", generator=g) if bias else None
            kl = torch.full((B,), S, dtype=torch.int32, device="cuda") if bias else None
            rec = {"part": "A", "shape": tag, "B": B, "H": H, "S": S, "ms": {}}
            for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16), ("bf16 again", torch.bfloat16), ("fp16 again", torch.float16)):
                qq, kk, vv = q.to(dt), k.to(dt), v.to(dt)
                rec["ms"][name] = round(time_ms(lambda: engine.attention(qq, kk, vv, 1.0 if bias else 0.125, bias_table=table, key_len=kl), 7), 4)
            rec["tflops_bf16"] = round(4.0 * B * H * S * S * 64 / rec["ms"]["bf16"] / 1e9, 1)
            emit(rec)


if __name__ == "__main__":
    main()
