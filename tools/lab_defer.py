#!/usr/bin/env python3
"""Lab (round 5): the quad GEMM with HALF of a full tile's plain epilogue deferred into the next tile's K loop
(csrc/gemm_quad_kernel.inc under VQS_QUAD_DEFER; library flavour lab_so/libvqs_<name>.so built by
`make -C t2v_metrics_amd/csrc variant NAME=<name> VFLAGS=-DVQS_QUAD_DEFER=1 LABDIR=../../lab_so`) against the product library,
one process, interleaved, on the path's plain-epilogue shapes.  Part C: bitwise equality on small shapes with partial tiles, a bias,
all three operand types; part T: timing + bitwise equality at full size.  Appends JSON lines to gpurun_out/lab_defer.jsonl."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "lab_defer.jsonl")


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


class Lib:
    def __init__(self, path=None):
        self.lib = engine.load_library(path)

    def gemm(self, *a, **k):
        saved = engine._lib
        engine._lib = self.lib
        try:
            return engine.gemm(*a, **k)
        finally:
            engine._lib = saved


def main():
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["defer"]
    epis = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    prod = Lib()
    variants = [(n, Lib(os.path.join(ROOT, "lab_so", f"libvqs_{n}.so"))) for n in names]
    g = torch.Generator(device="cuda").manual_seed(7)
    # ---- part C: same bits
    small = [(1000, 768, 512), (512, 512, 256), (2048, 1024, 1024), (256, 256, 320), (3000, 1280, 2048), (777, 520, 640), (4096, 2048, 256)]
    only_timing = "T" in (sys.argv[3] if len(sys.argv) > 3 else "")       # ablation flavours produce garbage by design: timing only
    for epi in [e for e in epis if e in (0, 1, 2, 5, 6) and not only_timing]:
        for M, N, K in small:
            for ft in (0, 1, 2):
                if epi == 5 and ft == 1:
                    continue
                for with_bias in (False, True):
                    dt = torch.float16 if ft else torch.bfloat16
                    A = torch.randn(M, K, device="cuda", generator=g).to(dt)
                    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
                    b = (torch.randn(N, device="cuda", generator=g)).to(torch.bfloat16) if with_bias else None
                    kw = dict(S=8 if epi == 6 else 0, H=N // 192 if epi == 6 else 0)
                    if epi == 6 and (N % 192 or M % 8 or (N // 3) % 128):
                        continue
                    ref = prod.gemm(A, W, epi, bias=b, variant=3, ftype=ft, **kw)
                    for name, v in variants:
                        for rep in range(3):
                            out = torch.full_like(ref, float("nan"))
                            v.gemm(A, W, epi, bias=b, variant=3, ftype=ft, out=out, **kw)
                            same = torch.equal(out.view(torch.int16), ref.view(torch.int16))
                            if not same:
                                bad = (out.view(torch.int16) != ref.view(torch.int16))
                                emit({"part": "C", "variant": name, "epi": epi, "M": M, "N": N, "K": K, "ft": ft, "bias": with_bias, "rep": rep, "same": False,
                                      "n_bad": int(bad.sum()), "first_bad": [int(x) for x in bad.nonzero()[0]], "nan": int(torch.isnan(out.float()).sum())})
                                break
                        else:
                            emit({"part": "C", "variant": name, "epi": epi, "M": M, "N": N, "K": K, "ft": ft, "bias": with_bias, "same": True})
    # ---- part D: where a multi-tile-per-workgroup launch differs (diagnosis)
    if "D" in (sys.argv[3] if len(sys.argv) > 3 else ""):
        for M, N, K in ((10240, 2048, 1024), (256 * 300, 256, 512), (256 * 520, 256, 256)):
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
            ref = prod.gemm(A, W, 0, variant=3)
            for name, v in variants:
                out = torch.full_like(ref, float("nan"))
                v.gemm(A, W, 0, variant=3, out=out)
                bad = out.view(torch.int16) != ref.view(torch.int16)
                nanm = torch.isnan(out.float())
                rows = bad.any(1).nonzero().flatten()
                cols = bad.any(0).nonzero().flatten()
                tiles = sorted(set((int(r) // 256, int(c) // 256) for r, c in bad.nonzero()[:: max(1, int(bad.sum()) // 4000)].tolist()))
                emit({"part": "D", "variant": name, "M": M, "N": N, "K": K, "n_bad": int(bad.sum()), "n_nan": int(nanm.sum()), "bad_and_nan": int((bad & nanm).sum()),
                      "row_mod128_hist": torch.bincount((rows % 128) // 16, minlength=8).tolist(), "col_mod128_hist": torch.bincount((cols % 128) // 16, minlength=8).tolist(),
                      "n_rows": int(rows.numel()), "n_cols": int(cols.numel()), "tiles_sample": tiles[:40], "n_tiles_bad": len(tiles),
                      "max_abs_diff": float((out.float() - ref.float()).nan_to_num(1e9).abs().max())})
        return
    # ---- part T: full size
    big = {0: [("xxl enc o", 155648, 4096, 4096, 2, False), ("xxl enc wo", 155648, 4096, 10240, 0, False), ("vit out_proj", 147712, 1024, 1024, 1, True),
               ("vit fc2", 147712, 1024, 4096, 1, True), ("xl enc o", 155648, 2048, 2048, 2, False)],
           5: [("xxl enc wi", 155648, 20480, 4096, 2, False), ("xl enc wi", 155648, 10240, 2048, 2, False)],
           6: [("xxl enc qkv", 155648, 12288, 4096, 1, False), ("vit qkv", 147712, 3072, 1024, 1, True)],
           1: [("vit fc1", 147712, 4096, 1024, 1, True)]}
    for epi in epis:
        for tag, M, N, K, ft, with_bias in big.get(epi, []):
            dt = torch.float16 if ft else torch.bfloat16
            A = torch.randn(M, K, device="cuda", generator=g).to(dt)
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
            b = (torch.randn(N, device="cuda", generator=g)).to(torch.bfloat16) if with_bias else None
            kw = dict(S=(608 if "enc" in tag else 577) if epi == 6 else 0, H=N // 192 if epi == 6 else 0)
            ref = prod.gemm(A, W, epi, bias=b, variant=3, ftype=ft, **kw)
            out = torch.empty_like(ref)
            flops = 2.0 * M * N * K
            rec = {"part": "T", "shape": tag, "epi": epi, "M": M, "N": N, "K": K, "ft": ft, "tflops": {}, "same": {}}
            order = [("product", prod)] + variants
            for rnd in range(3):
                for name, lib in order:
                    ms = time_ms(lambda: lib.gemm(A, W, epi, bias=b, variant=3, ftype=ft, out=out, **kw), 7)
                    rec["tflops"].setdefault(name, []).append(round(flops / ms / 1e9, 1))
            for name, lib in variants:
                out.fill_(float("nan"))
                lib.gemm(A, W, epi, bias=b, variant=3, ftype=ft, out=out, **kw)
                rec["same"][name] = bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            emit(rec)
            del A, W, ref, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
