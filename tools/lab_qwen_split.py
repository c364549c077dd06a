#!/usr/bin/env python3
"""Lab (round 6, VERDICT r5 item 6c: "give the Qwen shapes their own tile order").  The language model's gate|up GEMM of the Qwen2.5-VL-7B bench
batch (M = 51 712, N = 37 888, K = 3 584, pitch 4 096, fp16 operands) reads a 271 MB weight: more than the 256 MB Infinity Cache, and its
tile count (202 x 148) admits no column ranges (tile_of_slot needs equal per-XCD shares), so every order the library can pick sweeps all of W
per M-group.  Question: do column ranges pay here?  Answered without touching device code: the same product as 2 / 4 LAUNCHES over column
slices of W (a slice of the interleaved gate|up rows at a multiple of 256 is itself an interleaved gate|up weight; the result slice is written
in place through ldc), each slice's weight staying resident while the chip sweeps A.  Also: the (gm, 1) orders and the other shapes of the pass."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

PITCH = 4096
SHAPES = [("lm gate|up", 51712, 37888, 3584, 5, (1, 2, 4)), ("lm down", 51712, 3584, 18944, 0, (1, 2)), ("lm qkv", 51712, 4608, 3584, 0, (1, 2)),
          ("lm o", 51712, 3584, 3584, 0, (1,)), ("tower gate|up", 196608, 6848, 1280, 5, (1,)), ("tower down", 196608, 1280, 3456, 0, (1,)),
          ("tower qkv", 196608, 3840, 1280, 0, (1,)), ("tower proj", 196608, 1280, 1280, 0, (1,))]


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
    lib = engine.load_library()
    g = torch.Generator(device="cuda").manual_seed(11)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for tag, M, N, K, epi, splits in SHAPES:
        if only and only not in tag:
            continue
        pk = PITCH if K == 3584 else K                         # the product's x_pitch for the 3 584-wide stream; the other operands are dense
        A = torch.randn(M, pk, device="cuda", generator=g).to(torch.float16)
        W = (torch.randn(N, pk, device="cuda", generator=g) * K ** -0.5).to(torch.float16)
        NO = N // 2 if epi == 5 else N
        ft = 2 if epi == 5 else 1                              # the unscaled gated form writes bf16 (gemm_f16b_quad<5>: the T5 encoder's wi); same K loop as the product's gemm_f16s_quad<5>
        out = torch.empty(M, NO, dtype=torch.bfloat16 if ft == 2 else torch.float16, device="cuda")
        flops = 2.0 * M * N * K

        def launch(n_lo, n_cnt, gm, ns, dst):
            variant = 3 | (gm << 8) | (ns << 16) | (ft << 27)
            o_off = (n_lo // 2 if epi == 5 else n_lo) * 2
            rc = lib.vqs_gemm(A.data_ptr(), W.data_ptr() + n_lo * pk * 2, dst.data_ptr() + o_off, None, None, M, n_cnt, K, pk, pk, NO, epi, 0, 0,
                              variant, engine._stream_ptr())
            assert rc == 0, rc

        rec = {"shape": tag, "M": M, "N": N, "K": K, "pitch": pk, "weight_mb": round(N * K * 2 / 1e6), "tflops": {}}
        launch(0, N, 0, 0, out)
        torch.cuda.synchronize()
        ref = out.clone()
        for gm in (0, 1, 2, 4, 8):
            ms = time_ms(lambda: launch(0, N, gm, 1 if gm else 0, out), 5)
            rec["tflops"]["library order" if gm == 0 else f"{gm}x1"] = round(flops / ms / 1e9, 1)
        tiles_n = N // 256
        for parts in splits:
            if parts == 1 or tiles_n % parts:
                continue
            cnt = N // parts
            for gm in (0, 2, 4):
                out.zero_()

                def run():
                    for i in range(parts):
                        launch(i * cnt, cnt, gm, 1 if gm else 0, out)
                ms = time_ms(run, 5)
                rec["tflops"][f"{parts} launches over column slices, " + ("library order" if gm == 0 else f"{gm}x1")] = round(flops / ms / 1e9, 1)
                assert torch.equal(out, ref), "a column slice changed the result"
        rec["slices_bitwise_equal"] = True
        print(json.dumps(rec), flush=True)
        del A, W, out, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
