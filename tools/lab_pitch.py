#!/usr/bin/env python3
"""Lab (round 6): operand ROW PITCH of the quad GEMM.  Round 3 found the Qwen language model's K = 3 584 operands 10 % faster at a pitch of 4 096
elements (8 KiB) than dense (profiles/r3_call28_qwen_gemm_pitch.jsonl; 3 648 and 3 840 did not help: it is not about avoiding a power of two).
The T5 shapes were never asked the same question, and `wo` -- K = 10 240, a 20 KiB pitch, the one encoder shape that trails the others and whose
barriers cost 7-11 % where the K = 4 096 shapes pay 2-3 % (r6_call6/7) -- is the obvious candidate.  Same operands, same kernel, only lda / ldw
change (the results are bitwise equal; checked)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402

# (tag, M, N, K, epilogue, ftype, pitches)
SHAPES = [("xxl enc wo (bf16)", 155648, 4096, 10240, 0, 0, (10240, 10304, 11264, 12288, 14336, 16384)),
          ("xl enc wo (bf16)", 155648, 2048, 5120, 0, 0, (5120, 5184, 6144, 8192)),
          ("xl enc qkv (fp16)", 155648, 6144, 2048, 0, 1, (2048, 2112, 4096)),
          ("xl enc wi (fp16 -> bf16)", 155648, 10240, 2048, 5, 2, (2048, 4096)),
          ("vit fc1 (fp16)", 147712, 4096, 1024, 1, 1, (1024, 1088, 2048, 4096)),
          ("vit fc2 (fp16)", 147712, 1024, 4096, 0, 1, (4096, 4160, 8192)),
          ("vit qkv (fp16)", 147712, 3072, 1024, 0, 1, (1024, 2048, 4096)),
          ("xxl enc wi (fp16 -> bf16)", 155648, 20480, 4096, 5, 2, (4096, 4160, 8192))]


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
    g = torch.Generator(device="cuda").manual_seed(13)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for tag, M, N, K, epi, ft, pitches in SHAPES:
        if only and only not in tag:
            continue
        dt = torch.float16 if ft else torch.bfloat16
        A0 = torch.randn(M, K, device="cuda", generator=g).to(dt)
        W0 = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        NO = N // 2 if epi == 5 else N
        odt = torch.float16 if ft == 1 else torch.bfloat16
        flops = 2.0 * M * N * K
        rec = {"shape": tag, "M": M, "N": N, "K": K, "tflops_by_pitch(lda,ldw)": {}}
        ref = None
        combos = [(p, p) for p in pitches] + [(pitches[0], p) for p in pitches[1:2]] + [(p, pitches[0]) for p in pitches[1:2]]
        for pa, pw in combos:
            A = torch.zeros(M, pa, dtype=dt, device="cuda")
            A[:, :K] = A0
            W = torch.zeros(N, pw, dtype=dt, device="cuda")
            W[:, :K] = W0
            out = torch.empty(M, NO, dtype=odt, device="cuda")

            def launch():
                rc = lib.vqs_gemm(A.data_ptr(), W.data_ptr(), out.data_ptr(), None, None, M, N, K, pa, pw, NO, epi, 0, 0, 3 | (ft << 27), engine._stream_ptr())
                assert rc == 0, rc
            ms = time_ms(launch, 7)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), "the pitch changed the result"
            rec["tflops_by_pitch(lda,ldw)"][f"{pa},{pw}"] = round(flops / ms / 1e9, 1)
            del A, W, out
        rec["bitwise_equal"] = True
        print(json.dumps(rec), flush=True)
        del A0, W0, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
