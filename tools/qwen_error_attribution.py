#!/usr/bin/env python3
"""Where the Qwen2.5-VL row's end-to-end |delta log P| comes from, by rounding class (the method of
tools/error_attribution.py for the CLIP-FlanT5 row).  TEST INFRASTRUCTURE: the oracle, evaluated in torch fp32 on the GPU.

oracle/qwen25vl_oracle.py::QwenOracle carries a hook at every point where the HIP engine stores a 16-bit tensor
(vqs_qwen.cpp; class names "<stack>.<what>": norm outputs, q|k|v, rotated q / k, probabilities, attention output, the two deltas per
layer, the gated product, the merger tensors, the final norm output).  One run = a set of classes rounded (bf16, or IEEE fp16, or
split-bf16 = 16 significant bits), everything else fp32; reported: distance of log P(answer token) and of the 5 most likely tokens'
log-probs from the all-fp32 run, per sample.  `lastrow` variants leave the LAST valid position of every sample unrounded in the
language model: what a precise (split-bf16 / fp32) re-evaluation of that one row over the bf16 KV cache would hold -- it is the
only row that feeds the head directly (the T5 decoder rows of the CLIP-FlanT5 path, profiles/r4_error_attribution.md).

With --engine the HIP engine scores the same samples and its distance from the same fp32 truth is printed beside the table.

  python tools/qwen_error_attribution.py --samples 8 --engine --out gpurun_out/qwen_attr
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.qwen25vl_oracle import QwenOracle  # noqa: E402
from t2v_metrics_amd.qwen import get_qwen_config  # noqa: E402
from t2v_metrics_amd.qwen.weights import make_seeded_qwen_weights  # noqa: E402


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def fp16(x):
    return x.to(torch.float16).to(torch.float32)


def split16(x):
    hi = bf16(x)
    return hi + bf16(x - hi)


def synth(cfg, B, grid, seed=1234):
    """tools/bench_qwen.py's generator: B videos of `grid`, 14 + 24 text tokens around the placeholder run."""
    n_patches = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(B * n_patches, cfg.vision.patch_dim, generator=g).to(torch.bfloat16)
    n_merged = n_patches // cfg.vision.merge_unit
    rows = []
    for _ in range(B):
        pre = torch.randint(10, min(cfg.text.vocab, 150000), (14,), generator=g)
        post = torch.randint(10, min(cfg.text.vocab, 150000), (24,), generator=g)
        rows.append(torch.cat([pre, torch.tensor([cfg.vision_start_token_id]), torch.full((n_merged,), cfg.video_token_id),
                               torch.tensor([cfg.vision_end_token_id]), post]))
    ids = torch.stack(rows)
    return px, ids, torch.ones_like(ids), [grid] * B


class Hook:
    """rounds the classes in `how` ({class or "stack.*": fn}); `lastrow_exact`: text-model tensors keep their last valid position unrounded."""

    def __init__(self, how, last=None, lastrow_exact=False):
        self.how, self.last, self.lastrow_exact = how, last, lastrow_exact

    def fn(self, cls):
        # most specific first: "vis.norm.7" -> "vis.norm" -> "vis.*"
        parts = cls.split(".")
        for n in range(len(parts), 1, -1):
            k = ".".join(parts[:n])
            if k in self.how:
                return self.how[k]
        return self.how.get(parts[0] + ".*")

    def __call__(self, cls, x, pos_dim):
        f = self.fn(cls)
        if f is None:
            return x
        y = f(x)
        if self.lastrow_exact and pos_dim is not None and cls.startswith("txt."):
            L = x.shape[pos_dim]
            shape = [1] * x.dim()
            shape[0], shape[pos_dim] = x.shape[0], L
            keep = (torch.arange(L, device=x.device)[None, :] == self.last[:, None]).reshape(shape)
            y = torch.where(keep, x, y)
        return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2.5-vl-7b")
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--chunk", type=int, default=4, help="samples per oracle evaluation")
    ap.add_argument("--grid", default="4,24,32")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--engine", action="store_true", help="also score the samples with the HIP engine")
    ap.add_argument("--only", default="")
    ap.add_argument("--set", default="r5", choices=["r5", "r6", "r6b", "r6c"], help="r5: the round-5 table; r6: what-ifs for the >= 11-bit forms (all with the last row exact)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    cfg = get_qwen_config(a.model)
    dev = torch.device(a.device)
    grid = tuple(int(x) for x in a.grid.split(","))
    n_patches = grid[0] * grid[1] * grid[2]
    w_cpu = make_seeded_qwen_weights(cfg, seed=0, device="cpu")
    px, ids, mask, grids = synth(cfg, a.samples, grid)
    yes_id = 9454 % cfg.text.vocab
    w = {k: v.to(dev) for k, v in w_cpu.items()} if dev.type != "cpu" else w_cpu
    last = (mask.long().sum(-1) - 1).to(dev)

    def evaluate(hook):
        outs = []
        o = QwenOracle(cfg, w, device=dev, hook=hook)
        with torch.device(dev):
            for s in range(0, a.samples, a.chunk):
                e = min(s + a.chunk, a.samples)
                if hook is not None:
                    hook.last = last[s:e]
                outs.append(o.forward(ids[s:e].to(dev), mask[s:e].to(dev), px[s * n_patches: e * n_patches].float().to(dev), grids[s:e]).float())
        return torch.log_softmax(torch.cat(outs), -1).cpu()

    t0 = time.time()
    ref = evaluate(None)
    top5 = ref.topk(5, -1).indices
    print(f"# {cfg.name}, {a.samples} samples of grid {grid} (L = {ids.shape[1]}), fp32 truth on {dev} in {time.time() - t0:.1f} s; "
          f"log P(answer) {[round(float(x), 3) for x in ref[:, yes_id]]}", flush=True)

    def stats(lp):
        d_yes = (lp[:, yes_id] - ref[:, yes_id]).abs()
        d_top = (lp.gather(-1, top5) - ref.gather(-1, top5)).abs().max(-1).values
        return {"answer_max": float(d_yes.max()), "answer_mean": float(d_yes.mean()), "top5_max": float(d_top.max()), "top5_mean": float(d_top.mean()),
                "answer_per_sample": [round(float(x), 6) for x in d_yes]}

    TXT_ALL = {"txt.*": bf16}
    VIS_ALL = {"vis.*": bf16}
    runs = [
        ("engine's storage points (all classes bf16)", {**TXT_ALL, **VIS_ALL}, False),
        ("language model only (tower exact)", TXT_ALL, False),
        ("tower only (language model exact)", VIS_ALL, False),
        ("all classes, LAST ROW of the language model exact", {**TXT_ALL, **VIS_ALL}, True),
        ("language model only, last row exact (tower exact)", TXT_ALL, True),
        ("last row exact + tower in fp16", {**TXT_ALL, "vis.*": fp16, "vis.in": bf16, "vis.merged": bf16}, True),
        ("last row exact + tower in fp16 + merged tokens fp16", {**TXT_ALL, "vis.*": fp16, "vis.in": bf16}, True),
        ("last row exact + tower split-bf16", {**TXT_ALL, "vis.*": split16, "vis.in": bf16, "vis.merged": bf16}, True),
        ("last row exact + tower exact + language-model norm/qkv/rope/p/attn in fp16", {"txt.*": bf16, "txt.norm": fp16, "txt.qkv": fp16, "txt.rope": fp16, "txt.p": fp16, "txt.attn": fp16}, True),
        ("last row exact + tower fp16 + merged fp16 + language-model norm/qkv/rope/p/attn in fp16",
         {"txt.*": bf16, "txt.norm": fp16, "txt.qkv": fp16, "txt.rope": fp16, "txt.p": fp16, "txt.attn": fp16, "vis.*": fp16, "vis.in": bf16}, True),
    ]
    for c in ("txt.norm", "txt.qkv", "txt.rope", "txt.p", "txt.attn", "txt.delta", "txt.act", "txt.out"):
        runs.append((f"class {c} alone", {c: bf16}, False))
    if a.set == "r6":
        # round 6: which 16-bit classes have to carry >= 11 bits for the row to sit under 1e-3 (all with the precise tail = last row exact).
        # exact = the class stays fp32 (what an fp32 / split result of that GEMM would hold); vis.in stays bf16 (the ABI's input dtype)
        exact = lambda x: x  # noqa: E731
        T16 = {"txt.*": fp16}
        V16 = {"vis.*": fp16, "vis.in": bf16}
        TA = {"txt.norm": fp16, "txt.qkv": fp16, "txt.rope": fp16, "txt.p": fp16, "txt.attn": fp16}
        VA = {"vis.norm": fp16, "vis.qkv": fp16, "vis.rope": fp16, "vis.p": fp16, "vis.attn": fp16, "vis.in": bf16}
        runs = [
            ("r5 ships: all bf16, last row exact", {**TXT_ALL, **VIS_ALL}, True),
            ("all fp16", {**T16, **V16}, True),
            ("all fp16, merged bf16", {**T16, **V16, "vis.merged": bf16}, True),
            ("all fp16, txt.delta bf16", {**T16, **V16, "txt.delta": bf16}, True),
            ("all fp16, txt.act bf16", {**T16, **V16, "txt.act": bf16}, True),
            ("all fp16, txt.delta + txt.act bf16", {**T16, **V16, "txt.delta": bf16, "txt.act": bf16}, True),
            ("all fp16, txt.delta exact", {**T16, **V16, "txt.delta": exact}, True),
            ("all fp16, vis.delta bf16", {**T16, **V16, "vis.delta": bf16}, True),
            ("all fp16, vis.act bf16", {**T16, **V16, "vis.act": bf16}, True),
            ("all fp16, vis.delta + vis.act bf16", {**T16, **V16, "vis.delta": bf16, "vis.act": bf16}, True),
            ("all fp16, vis.delta + vis.act + txt.delta + txt.act bf16", {**T16, **V16, "vis.delta": bf16, "vis.act": bf16, "txt.delta": bf16, "txt.act": bf16}, True),
            ("attention sides fp16 only (both stacks), rest bf16", {**TXT_ALL, **VIS_ALL, **TA, **VA}, True),
            ("txt all fp16, tower bf16", {**T16, **VIS_ALL}, True),
            ("txt all fp16, tower exact", {**T16}, True),
            ("tower all fp16, txt exact", {**V16}, True),
            ("tower all bf16, txt exact", {**VIS_ALL}, True),
            ("all fp16, last row NOT exact", {**T16, **V16}, False),
            ("all split-bf16 (16 bits)", {"txt.*": split16, "vis.*": split16, "vis.in": bf16}, True),
        ]
    if a.set == "r6b":
        # second pass: with every class in fp16 (6.9e-4 / 2.5e-4 over 32 samples), which classes are worth MORE than 11 bits
        exact = lambda x: x  # noqa: E731
        sc16 = lambda k: (lambda x: fp16(x * 2.0 ** -k) * 2.0 ** k)  # noqa: E731  (fp16 behind a power-of-two pre-scale: the range-safe delta form)
        A16 = {"txt.*": fp16, "vis.*": fp16, "vis.in": bf16}
        runs = [("all fp16", A16, True)]
        for c in ("vis.norm", "vis.qkv", "vis.rope", "vis.p", "vis.attn", "vis.delta", "vis.act", "vis.mid", "vis.merged",
                  "txt.norm", "txt.qkv", "txt.rope", "txt.p", "txt.attn", "txt.delta", "txt.act"):
            runs.append((f"all fp16, {c} exact", {**A16, c: exact}, True))
        runs += [
            ("all fp16, vis.delta + txt.delta exact", {**A16, "vis.delta": exact, "txt.delta": exact}, True),
            ("all fp16, vis.delta + vis.act exact", {**A16, "vis.delta": exact, "vis.act": exact}, True),
            ("all fp16, both deltas + both acts exact", {**A16, "vis.delta": exact, "txt.delta": exact, "vis.act": exact, "txt.act": exact}, True),
            ("all fp16, both deltas + both acts + vis.norm + txt.norm exact", {**A16, "vis.delta": exact, "txt.delta": exact, "vis.act": exact, "txt.act": exact, "vis.norm": exact, "txt.norm": exact}, True),
            ("all fp16, deltas as fp16 behind a 2^-6 pre-scale", {**A16, "vis.delta": sc16(6), "txt.delta": sc16(6)}, True),
            ("all fp16, deltas as fp16 behind a 2^-10 pre-scale", {**A16, "vis.delta": sc16(10), "txt.delta": sc16(10)}, True),
            ("all fp16, deltas + acts behind a 2^-8 pre-scale", {**A16, "vis.delta": sc16(8), "txt.delta": sc16(8), "vis.act": sc16(8), "txt.act": sc16(8)}, True),
            ("all fp16, deltas split-bf16", {**A16, "vis.delta": split16, "txt.delta": split16}, True),
        ]
    if a.set == "r6c":
        # third pass: WHICH tower norm outputs carry the fp16 forms' residual (vis.norm exact halves the mean: r6_call2)
        exact = lambda x: x  # noqa: E731
        A16 = {"txt.*": fp16, "vis.*": fp16, "vis.in": bf16}
        blocks = list(range(cfg.vision.depth))
        full = list(cfg.vision.fullatt_blocks)
        runs = [("all fp16", A16, True), ("all fp16, vis.norm exact", {**A16, "vis.norm": exact}, True),
                ("all fp16, merger norm exact", {**A16, "vis.norm.merger": exact}, True),
                ("all fp16, block norms exact (merger fp16)", {**A16, "vis.norm": exact, "vis.norm.merger": fp16}, True),
                ("all fp16, merger norm + mid + merged exact", {**A16, "vis.norm.merger": exact, "vis.mid": exact, "vis.merged": exact}, True),
                ("all fp16, norms of the full-attention blocks exact", {**A16, **{f"vis.norm.{i}": exact for i in full}}, True),
                ("all fp16, norms of blocks 0-7 exact", {**A16, **{f"vis.norm.{i}": exact for i in blocks[:8]}}, True),
                ("all fp16, norms of blocks 8-15 exact", {**A16, **{f"vis.norm.{i}": exact for i in blocks[8:16]}}, True),
                ("all fp16, norms of blocks 16-23 exact", {**A16, **{f"vis.norm.{i}": exact for i in blocks[16:24]}}, True),
                ("all fp16, norms of blocks 24-31 exact", {**A16, **{f"vis.norm.{i}": exact for i in blocks[24:]}}, True),
                ("all fp16, merger norm split-bf16", {**A16, "vis.norm.merger": split16}, True),
                ("all fp16, merger norm exact + both deltas exact", {**A16, "vis.norm.merger": exact, "vis.delta": exact, "txt.delta": exact}, True),
                ]
    if a.only:
        keep = set(a.only.split(";"))
        runs = [r for r in runs if r[0] in keep]
    results = {}
    for name, how, lastrow in runs:
        t0 = time.time()
        st = stats(evaluate(Hook(how, lastrow_exact=lastrow)))
        st["seconds"] = round(time.time() - t0, 1)
        results[name] = st
        print(f"{name:100s} answer: max {st['answer_max']:.2e} mean {st['answer_mean']:.2e}   top-5: max {st['top5_max']:.2e} mean {st['top5_mean']:.2e}  ({st['seconds']} s)", flush=True)
    if a.engine:
        from t2v_metrics_amd.qwen.engine import QwenEngine
        del w
        torch.cuda.empty_cache()
        eng = QwenEngine(cfg, w_cpu, device=dev)
        lp = torch.log_softmax(eng.score_logits(eng.encode_vision(px, grids), ids, mask, grids).float(), -1).cpu()
        st = stats(lp)
        results["HIP engine"] = st
        print(f"{'HIP engine (measured)':100s} answer: max {st['answer_max']:.2e} mean {st['answer_mean']:.2e}   top-5: max {st['top5_max']:.2e} mean {st['top5_mean']:.2e}", flush=True)
    results["_meta"] = {"model": cfg.name, "samples": a.samples, "grid": grid, "L": int(ids.shape[1]), "device": str(dev), "answer_id": yes_id,
                        "logp_answer_fp32": [round(float(x), 4) for x in ref[:, yes_id]]}
    if a.out:
        with open(a.out + ".json", "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
