#!/usr/bin/env python3
"""Lab: one GPU call that measures (A) the GEMM tile order (gm, ns) per shape in isolation, (C) the best orders in situ
(bench step, interleaved with the default order in ONE process, per-call-site GEMM table), (B) an A/B of two library builds
of the self-attention kernel (product vs build/lab/libvqs_<variant>.so) in isolation and (D) in situ.  Every record is
appended to gpurun_out/lab_call.jsonl as soon as it exists, so a cut-off call still leaves what it measured.

The tile order is a permutation of the tile list (bitwise-neutral, tests/test_gpu_kernels.py::test_gemm_tile_order_...);
part C also checks that the scores of the tuned run equal the default run's bit for bit.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd import engine  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402
import bench  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "lab_call.jsonl")
T0 = time.time()


def emit(rec):
    rec["t"] = round(time.time() - T0, 1)
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


# (tag, M, N, K, epilogue, S, H, has_bias)
XXL = [("xxl enc wi", 155648, 20480, 4096, 5, 0, 0, False), ("xxl enc wo", 155648, 4096, 10240, 0, 0, 0, False),
       ("xxl enc qkv", 155648, 12288, 4096, 6, 608, 64, False), ("xxl enc o", 155648, 4096, 4096, 0, 0, 0, False)]
XL = [("xl enc wi", 155648, 10240, 2048, 5, 0, 0, False), ("xl enc wo", 155648, 2048, 5120, 0, 0, 0, False),
      ("xl enc qkv", 155648, 6144, 2048, 6, 608, 32, False), ("xl enc o", 155648, 2048, 2048, 0, 0, 0, False)]
VIT = [("vit fc1", 147712, 4096, 1024, 1, 0, 0, True), ("vit fc2", 147712, 1024, 4096, 0, 0, 0, True),
       ("vit qkv", 147712, 3072, 1024, 6, 577, 16, True), ("vit out_proj", 147712, 1024, 1024, 0, 0, 0, True)]
ORDERS = [(8, 1), (4, 1), (2, 1), (16, 1), (8, 2), (4, 2), (2, 2), (16, 2), (8, 4), (4, 4)]


def sweep(shapes, reps, orders=ORDERS):
    best = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for tag, M, N, K, epi, S, H, has_bias in shapes:
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16) if has_bias else None
        out = engine.gemm(A, W, epi, bias=bias, S=S, H=H, variant=3)
        res = {}
        for rnd in range(2):                                   # two passes over the orders: drift shows as disagreement
            for o in orders:
                ms = time_ms(lambda: engine.gemm(A, W, epi, bias=bias, out=out, S=S, H=H, variant=3, tile_order=o), reps)
                res.setdefault("%dx%d" % o, []).append(round(2.0 * M * N * K / ms / 1e9, 1))
        base = min(res["8x1"])
        cand = max(res, key=lambda k: min(res[k]))
        gain = min(res[cand]) / max(res["8x1"]) - 1.0          # conservative: candidate's worst vs default's best
        emit({"part": "A", "shape": tag, "N": N, "K": K, "tflops": res, "best": cand, "gain_conservative": round(gain, 4)})
        best[(N, K)] = (cand, gain, base)
        del A, W, out
        torch.cuda.empty_cache()
    return best


def attention_ab(variant_lib, reps):
    lib2 = engine.load_library(variant_lib)
    lib1 = engine.load_library()
    g = torch.Generator(device="cuda").manual_seed(5)
    for tag, B, H, S, scale, with_bias in (("t5-xxl", 256, 64, 608, 1.0, True), ("t5-xl", 256, 32, 608, 1.0, True), ("vit", 256, 16, 577, 0.125, False)):
        q, k, v = [(torch.randn(B, H, S, 64, device="cuda", generator=g) * (1.0 if with_bias else 2.0)).to(torch.bfloat16) for _ in range(3)]
        bias = (torch.randn(H, 2 * S - 1, device="cuda", generator=g) * 2.0).contiguous() if with_bias else None
        klen = torch.randint(S - 40, S + 1, (B,), device="cuda", generator=g, dtype=torch.int32) if with_bias else None
        outs, ms = [], []
        for lib in (lib1, lib2):
            o = torch.empty(B * S, H * 64, dtype=torch.bfloat16, device="cuda")

            def call(lib=lib, o=o):
                rc = lib.vqs_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None if bias is None else bias.data_ptr(),
                                       None if klen is None else klen.data_ptr(), B, H, S, scale, torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            ms.append([time_ms(call, reps)])
            outs.append(o)
        for lib, o, m in ((lib1, outs[0], ms[0]), (lib2, outs[1], ms[1])):       # second, interleaved measurement
            def call(lib=lib, o=o):
                lib.vqs_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None if bias is None else bias.data_ptr(),
                                  None if klen is None else klen.data_ptr(), B, H, S, scale, torch.cuda.current_stream().cuda_stream)
            m.append(time_ms(call, reps))
        # fp32 reference on two samples
        nb = 2
        sc = torch.einsum("bhqd,bhkd->bhqk", q[:nb].float(), k[:nb].float()) * scale
        if bias is not None:
            idx = (torch.arange(S, device="cuda")[None, :] - torch.arange(S, device="cuda")[:, None]) + S - 1
            sc = sc + bias[:, idx][None]
            sc = sc.masked_fill(torch.arange(S, device="cuda")[None, None, None, :] >= klen[:nb, None, None, None], float("-inf"))
        ref = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, -1), v[:nb].float()).reshape(nb * S, H * 64)
        d = [(o[: nb * S].float() - ref).abs().max().item() for o in outs]
        dd = (outs[0].float() - outs[1].float()).abs()
        emit({"part": "B", "shape": tag, "ms_product": ms[0], "ms_variant": ms[1], "max_err_vs_fp32_product": d[0], "max_err_vs_fp32_variant": d[1],
              "max_abs_diff_between": dd.max().item(), "frac_elements_differing": (dd > 0).float().mean().item()})
        del q, k, v, outs
        torch.cuda.empty_cache()


def in_situ(model, configs, steps, rounds, variant_lib=None, tag="C"):
    """configs: [(name, "product" | "variant", {(N, K): gm | ns << 8 or 0})]; run interleaved, `rounds` times, `steps` steps each."""
    cfg = get_config(model)
    dev = torch.device("cuda", 0)
    weights = make_seeded_weights(cfg, seed=0, device=dev)
    px, ii, ids, lab = bench.synth_batch(cfg, 256, seed=1234, device=dev)
    engines = {"product": engine.VqsEngine(cfg, weights, device=dev)}
    if variant_lib:
        saved = engine._lib
        engine._lib = engine.load_library(variant_lib)
        engines["variant"] = engine.VqsEngine(cfg, weights, device=dev)
        engine._lib = saved
    keys = set()
    for _, _, o in configs:
        keys |= set(o)

    def apply(eng, orders):
        for k in keys:                                           # (N, K) = a tile order; a string = any other option name
            eng.set_option(k if isinstance(k, str) else "tile_order:%dx%d" % k, orders.get(k, 0))

    def run(eng, n):
        eng.profile(True)
        eng.profile_read(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            lp, sc = eng.score(eng.encode_images(px), ii, ids, lab)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        eng.profile(False)
        ng, gms, gfl = eng.profile_read(reset=True)
        rep = {l[:20].strip(): float(l.split("TFLOP/s")[1]) for l in eng.profile_report().splitlines() if "TFLOP/s" in l}
        return dt, gfl / gms / 1e9 if gms > 0 else 0.0, rep, lp, sc

    for e in engines.values():
        run(e, 1)                                                # warm-up, workspaces
    ref = None
    for r in range(rounds):
        for name, which, orders in configs:
            apply(engines[which], orders)
            dt, tf, rep, lp, sc = run(engines[which], steps)
            rec = {"part": tag, "model": model, "config": name, "round": r, "ms_per_step": round(dt * 1e3, 2), "pairs_per_s": round(256 / dt, 2),
                   "gemm_tflops": round(tf, 1), "sites": rep}
            if ref is None:
                ref = (lp.clone(), sc.clone())
            rec["bitwise_equal_to_first_config"] = bool(torch.equal(lp, ref[0]) and torch.equal(sc, ref[1]))
            if not rec["bitwise_equal_to_first_config"]:
                rec["max_abs_dlogp_vs_first_config"] = (lp - ref[0]).abs().max().item()
            emit(rec)
    for e in engines.values():
        e.close()
    del engines, weights
    torch.cuda.empty_cache()


FORCE_8x1_XXL = {(20480, 4096): 8 | 1 << 8, (4096, 10240): 8 | 1 << 8, (12288, 4096): 8 | 1 << 8, (4096, 4096): 8 | 1 << 8}
FORCE_8x1_XL = {(10240, 2048): 8 | 1 << 8, (2048, 5120): 8 | 1 << 8, (6144, 2048): 8 | 1 << 8, (2048, 2048): 8 | 1 << 8}
FINE = [(8, 1), (3, 1), (4, 1), (5, 1), (6, 1), (3, 2), (4, 2), (5, 2), (6, 2), (4, 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="H,N,F,X")
    ap.add_argument("--variant-lib", default=os.path.join(ROOT, "build", "lab", "libvqs_gemm_nt_store.so"))
    ap.add_argument("--attn-lib", default=os.path.join(ROOT, "build", "lab", "libvqs_attn_bias_acc.so"))
    a = ap.parse_args()
    parts = a.parts.split(",")
    emit({"part": "start", "device": torch.cuda.get_device_name(0), "parts": parts})
    if "A" in parts:                                             # call 25: isolated sweep of the orders
        sweep(XXL, reps=3)
        sweep(VIT, reps=5, orders=[(8, 1), (4, 1), (16, 1), (32, 1), (2, 1)])
    if "H" in parts:                                             # the library's choice by shape against the round-1/2 map, in situ
        in_situ("clip-flant5-xxl", [("forced 8x1", "product", FORCE_8x1_XXL), ("library choice", "product", {})], steps=3, rounds=3, tag="H")
    if "N" in parts and os.path.exists(a.variant_lib):           # a variant build against the product, both on the library's tile order
        in_situ("clip-flant5-xxl", [("product", "product", {}), ("variant " + os.path.basename(a.variant_lib), "variant", {})], steps=3, rounds=3,
                variant_lib=a.variant_lib, tag="N")
    if "F" in parts:
        sweep(XXL, reps=3, orders=FINE)
        sweep([s for s in XL if "wo" in s[0]], reps=5, orders=FINE)
    if "X" in parts:
        in_situ("clip-flant5-xl", [("forced 8x1", "product", FORCE_8x1_XL), ("library choice", "product", {})], steps=4, rounds=3, tag="HX")
        if os.path.exists(a.variant_lib):
            in_situ("clip-flant5-xl", [("product", "product", {}), ("variant " + os.path.basename(a.variant_lib), "variant", {})], steps=4, rounds=2,
                    variant_lib=a.variant_lib, tag="NX")
    if "S" in parts:                                             # non-temporal result stores, call site by call site, in situ
        def nt(*sites):
            return {"nt_store:%dx%d" % s: 1 for s in sites}
        wo, qkv, o, wi, fc2 = (4096, 10240), (12288, 4096), (4096, 4096), (20480, 4096), (1024, 4096)
        in_situ("clip-flant5-xxl", [("plain", "product", {}), ("nt wo", "product", nt(wo)), ("nt wo+qkv", "product", nt(wo, qkv)),
                                    ("nt wo+qkv+o", "product", nt(wo, qkv, o)), ("nt wo+qkv+o+wi", "product", nt(wo, qkv, o, wi)),
                                    ("nt wo+qkv+fc2", "product", nt(wo, qkv, fc2))], steps=3, rounds=3, tag="S")
        wo, qkv, o, wi = (2048, 5120), (6144, 2048), (2048, 2048), (10240, 2048)
        in_situ("clip-flant5-xl", [("plain", "product", {}), ("nt wo", "product", nt(wo)), ("nt wo+qkv", "product", nt(wo, qkv)),
                                   ("nt wo+qkv+o", "product", nt(wo, qkv, o)), ("nt wo+qkv+o+wi", "product", nt(wo, qkv, o, wi))],
                steps=4, rounds=3, tag="SX")
    if "T" in parts:                                             # yardstick: this library vs torch.matmul (hipBLASLt) on the path's shapes
        g = torch.Generator(device="cuda").manual_seed(0)
        for tag, M, N, K, epi, S, H, has_bias in XXL + XL + VIT:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
            out = engine.gemm(A, W, 0, variant=3)
            ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            res = {}
            for rnd in range(2):
                res.setdefault("vqs_plain_epilogue", []).append(round(2.0 * M * N * K / time_ms(lambda: engine.gemm(A, W, 0, out=out, variant=3), 5) / 1e9, 1))
                res.setdefault("torch_matmul", []).append(round(2.0 * M * N * K / time_ms(lambda: torch.matmul(A, W.t(), out=ref), 5) / 1e9, 1))
            emit({"part": "T", "shape": tag, "M": M, "N": N, "K": K, "tflops": res,
                  "max_abs_diff": (out.float() - ref.float()).abs().max().item()})
            del A, W, out, ref
            torch.cuda.empty_cache()
    if "L" in parts:                                             # A-panel L2 prefetch under the new tile order, site by site, in situ
        wo, qkv, o, wi = (4096, 10240), (12288, 4096), (4096, 4096), (20480, 4096)
        def opt(pairs):
            return {"l2_touch:%dx%d" % s: v for s, v in pairs}
        in_situ("clip-flant5-xxl", [("rule (wo on)", "product", {}), ("wo off", "product", opt([(wo, 2)])), ("wo on + o on", "product", opt([(o, 1)])),
                                    ("wo on + qkv on", "product", opt([(qkv, 1)])), ("wo on + wi on", "product", opt([(wi, 1)]))], steps=3, rounds=3, tag="L")
    if "AT" in parts:                                            # per-phase cycle split of the self-attention kernel (timing build)
        import ctypes
        path = os.path.join(ROOT, "build", "lab", "libvqs_attn_timing.so")
        lib = engine.load_library(path)
        lib.vqs_lab_set_attn_timing.argtypes = [ctypes.c_void_p]
        g = torch.Generator(device="cuda").manual_seed(5)
        names = ["wait+barrier+stage", "K reads + QK^T (+bias)", "mask / max / rescale", "exp + pack + V reads + PV", "prologue", "epilogue"]
        for tag, B, H, S, scale, with_bias in (("t5-xxl", 256, 64, 608, 1.0, True), ("t5-xl", 256, 32, 608, 1.0, True), ("vit", 256, 16, 577, 0.125, False)):
            q, k, v = [torch.randn(B, H, S, 64, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3)]
            bias = (torch.randn(H, 2 * S - 1, device="cuda", generator=g) * 2.0).contiguous() if with_bias else None
            o = torch.empty(B * S, H * 64, dtype=torch.bfloat16, device="cuda")
            buf = torch.zeros(8, dtype=torch.int64, device="cuda")
            assert lib.vqs_lab_set_attn_timing(buf.data_ptr()) == 0

            def call():
                assert lib.vqs_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None if bias is None else bias.data_ptr(), None,
                                         B, H, S, scale, torch.cuda.current_stream().cuda_stream) == 0
            call()
            torch.cuda.synchronize()
            buf.zero_()
            ms = time_ms(call, 3)                                 # 1 warm + 3 timed launches = 4 launches in the counters
            t = buf.cpu().tolist()
            waves, tiles = max(t[7], 1), max(t[6], 1)
            emit({"part": "AT", "shape": tag, "ms_per_call_with_probes": ms, "cycles_per_wave_tile": {n: round(t[i] / tiles, 1) for i, n in enumerate(names[:4])},
                  "cycles_per_wave": {n: round(t[4 + i] / waves, 1) for i, n in enumerate(names[4:])}, "wave_tiles": tiles, "waves": waves})
            assert lib.vqs_lab_set_attn_timing(None) == 0
    if "V" in parts:                                             # lock-step (launcher's rule) vs forced ping-pong schedule, ViT shapes
        g = torch.Generator(device="cuda").manual_seed(0)
        for tag, M, N, K, epi, S, H, has_bias in VIT + [("projector.0", 147456, 4096, 1024, 2, 0, 0, True)]:
            A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
            out = engine.gemm(A, W, epi, bias=bias, S=S, H=H, variant=3)
            res = {}
            for rnd in range(2):
                for v in (3, 5):
                    ms = time_ms(lambda: engine.gemm(A, W, epi, bias=bias, out=out, S=S, H=H, variant=v), 7)
                    res.setdefault("variant%d" % v, []).append(round(2.0 * M * N * K / ms / 1e9, 1))
            emit({"part": "V", "shape": tag, "N": N, "K": K, "tflops": res})
            del A, W, out
    if "BX" in parts and os.path.exists(a.attn_lib):             # attention bias-in-accumulator build, in situ at XL and XXL
        for model, steps in (("clip-flant5-xl", 4), ("clip-flant5-xxl", 3)):
            in_situ(model, [("product", "product", {}), ("variant " + os.path.basename(a.attn_lib), "variant", {})], steps=steps, rounds=3,
                    variant_lib=a.attn_lib, tag="BX")
    if "B" in parts and os.path.exists(a.attn_lib):
        attention_ab(a.attn_lib, reps=5)
    emit({"part": "done"})


if __name__ == "__main__":
    main()
