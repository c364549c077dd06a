#!/usr/bin/env python3
"""Where the end-to-end |delta log P| of the HIP path comes from, by ROUNDING CLASS (CPU only).

The rounding-matched oracle (oracle/clip_t5_engine_rounding.py) has a round-to-bf16 at every point where the engine holds a
bf16 tensor; each such site carries a class name (`EngineRoundedOracle.CLASSES`, "<stack>.<what>").  This tool evaluates the
bench pairs with ONE class rounding and everything else in fp32, and tabulates the distance of the label log-probs from the
all-fp32 run -- the contribution of that class alone -- next to the stack-level groups, the engine's full set, and the
candidate set "everything the engine could hold in fp32 / split-bf16 at < 2 % of a step" (decoder side, lm_head input,
projector) switched off.  The lm_head is the last op and linear, so each run is read out under several head gains
(lm_head x gain: gain 4 = peaked head) at no extra cost.

Stage results are shared between runs: a decoder-side class reuses the unrounded vision + projector + encoder pass, an
encoder-side class the unrounded vision + projector pass.

  python tools/error_attribution.py --model clip-flant5-xl --pairs 16 --out profiles/r4_error_attribution
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

import bench  # noqa: E402  (synth_batch: the bench's own pair generator)
from oracle.clip_t5_engine_rounding import EngineRoundedOracle  # noqa: E402
from oracle.clip_t5_oracle import Oracle, shift_right  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402

ALL = EngineRoundedOracle.CLASSES
CHEAP = tuple(c for c in ALL if c.startswith(("dec.", "proj.")) or c == "enc.out")   # < 2.5 % of an XXL step in total


def stack_of(c):
    return c.split(".")[0]


class Runner:
    """Evaluates a class set, reusing upstream stage results that the set cannot have changed."""

    def __init__(self, cfg, w, batch, acc, gains):
        self.cfg, self.w, self.acc, self.gains = cfg, w, acc, gains
        self.pix, self.idx, self.ids, self.labels = self.batch = batch
        self.cache = {}

    def _oracle(self, classes, dec_precise, split=()):
        ship = "ship:vit_fp16" in split                                      # the engine's option itself (fp16 GEMM result of the projector, then bf16)
        enc16 = "attn" if "ship:enc_fp16_attn" in split else ("ship:enc_fp16" in split)   # the engine's option enc_fp16 (round 5) / its attention-sub-block-only what-if
        dec16 = "ship:dec_fp16" in split                                     # the engine's option dec_fp16 (round 5)
        half = tuple(c[5:] for c in split if c.startswith("half:"))          # "half:<class>" entries of a what-if's set: fp16 tensors
        split = tuple(c for c in split if not c.startswith(("half:", "ship:")))
        # vit_fp16=False: the table models every class as a bf16 rounding (the engine of rounds 1-3) and adds fp16 / split stages explicitly
        return EngineRoundedOracle(self.cfg, self.w, acc=self.acc, classes=classes, dec_precise=dec_precise, split_classes=split,
                                   half_classes=half, vit_fp16=ship, device=self.pix.device, enc_fp16=enc16,
                                   dec_fp16=dec16)

    def run(self, classes, dec_precise=False, split=()):
        classes = frozenset(classes)
        o = self._oracle(classes, dec_precise, split)
        with torch.no_grad():
            k_vit = frozenset(c for c in classes if stack_of(c) == "vit")
            k_proj = (k_vit, frozenset(c for c in classes if stack_of(c) == "proj"))
            k_enc = (k_proj, frozenset(c for c in classes if stack_of(c) == "enc"))
            sp = frozenset(split)                      # a what-if's split set changes what a class's rounding is
            if ("vit", k_vit, sp) not in self.cache:
                self.cache[("vit", k_vit, sp)] = o.vision_features(self.pix)
            if ("proj", k_proj, sp) not in self.cache:
                self.cache[("proj", k_proj, sp)] = o.projector(self.cache[("vit", k_vit, sp)])
            if ("enc", k_enc, sp) not in self.cache:
                emb, mask, _ = o.splice(self.cache[("proj", k_proj, sp)], self.idx, self.ids)
                self.cache[("enc", k_enc, sp)] = (o.t5_encoder(emb, mask), mask)
            enc, mask = self.cache[("enc", k_enc, sp)]
            dec = o.t5_decoder(shift_right(self.labels, self.cfg.t5.decoder_start_id, self.cfg.t5.pad_id), enc, mask)
            logits = o.lm_logits(dec)
            return {g: Oracle.label_logprobs(logits * g, self.labels) for g in self.gains}

    def drop(self, stack):
        for k in [k for k in self.cache if k[0] == stack and k[1] != self._empty_key(stack)]:
            del self.cache[k]

    @staticmethod
    def _empty_key(stack):
        e = frozenset()
        return {"vit": e, "proj": (e, e), "enc": ((e, e), e)}[stack]


def stats(lp, ref):
    d = (lp - ref).abs()
    return {"max": float(d.max()), "mean": float(d.mean()), "yes_max": float(d[:, 0].max()), "yes_mean": float(d[:, 0].mean()),
            "signed_mean": float((lp - ref).mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="clip-flant5-xl")
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--gains", default="1,4")
    ap.add_argument("--acc", default="float32", choices=["float32", "float64"])
    ap.add_argument("--only", default="", help="semicolon-separated run names (default: all runs)")
    ap.add_argument("--out", default="")
    ap.add_argument("--device", default="cpu", help="cuda: evaluate the same oracle code in torch on the GPU (XXL, 64 pairs: a minute per run "
                    "instead of hours; test infrastructure either way -- the engine is not involved)")
    ap.add_argument("--chunk", type=int, default=0, help="evaluate the pairs in chunks of this many (bounds the [B,H,S,S] score tensors)")
    a = ap.parse_args()
    import warnings
    warnings.filterwarnings("ignore")
    cfg = get_config(a.model)
    gains = [float(g) for g in a.gains.split(",")]
    t00 = time.time()
    dev = torch.device(a.device)
    w = make_seeded_weights(cfg, seed=0, device=dev)
    pix, idx, ids, labels = bench.synth_batch(cfg, a.pairs, a.seed, dev)
    batch = (pix.float(), idx.long(), ids.long(), labels.long())
    chunk = a.chunk if a.chunk > 0 else a.pairs
    Rs = [Runner(cfg, w, (batch[0][s: s + chunk], torch.arange(min(chunk, a.pairs - s), device=dev), batch[2][s: s + chunk], batch[3][s: s + chunk]),
                 getattr(torch, a.acc), gains) for s in range(0, a.pairs, chunk)]        # synth_batch: pair i uses image i

    class R:                                      # the chunks side by side: every pair is independent of the others in its batch
        @staticmethod
        def run(classes, precise=False, split=()):
            with torch.device(dev):
                outs = [r.run(classes, precise, split) for r in Rs]
            return {g: torch.cat([o[g] for o in outs]).cpu() for g in gains}

        @staticmethod
        def drop(stack):
            for r in Rs:
                r.drop(stack)

    # per-class and group runs model the engine of rounds 1-3 (every class a plain bf16 rounding)
    runs = [("none (all fp32)", ())]
    runs += [(c, (c,)) for c in ALL]
    runs += [("stack vit.*", tuple(c for c in ALL if stack_of(c) == "vit")),
             ("stack proj.*", tuple(c for c in ALL if stack_of(c) == "proj")),
             ("stack enc.*", tuple(c for c in ALL if stack_of(c) == "enc")),
             ("stack dec.*", tuple(c for c in ALL if stack_of(c) == "dec")),
             ("cheap set (dec.* + proj.* + enc.out)", CHEAP),
             ("engine minus cheap set", tuple(c for c in ALL if c not in CHEAP)),
             ("engine (all classes)", ALL),
             # round 4's engine: the same classes with the decoder's sensitive tensors split-bf16 / fp32 (EngineRoundedOracle.DEC_SPLIT)
             ("engine with the precise decoder (round 4)", ALL, True),
             ("stack dec.* with the precise decoder (round 4)", tuple(c for c in ALL if stack_of(c) == "dec"), True)]
    # what a further precise stage could buy at best: the round-4 engine with that stage's classes not rounded at all
    without = lambda *drop: tuple(c for c in ALL if not any(c == d or (d.endswith(".*") and c.startswith(d[:-1])) for d in drop))
    runs += [("what-if: precise decoder, proj.* exact", without("proj.*"), True),
             ("what-if: precise decoder, vit.* exact", without("vit.*"), True),
             ("what-if: precise decoder, vit.* proj.* exact", without("vit.*", "proj.*"), True),
             ("what-if: precise decoder, vit.norm vit.delta exact", without("vit.norm", "vit.delta"), True),
             ("what-if: precise decoder, vit.norm vit.delta vit.qkv exact", without("vit.norm", "vit.delta", "vit.qkv"), True),
             ("what-if: precise decoder, enc.* exact", without("enc.*"), True),
             ("what-if: precise decoder, vit.* proj.* enc.* exact", without("vit.*", "proj.*", "enc.*"), True)]
    # ... and what it would buy as an implementable design: those tensors split-bf16 (16 bits) instead of exact
    VIT_SPLIT = ("vit.norm", "vit.v", "vit.attn", "vit.act", "vit.delta", "vit.feat")          # q, k and the probabilities stay bf16
    runs += [("what-if: precise decoder, split vit (norm v attn act delta feat)", ALL, True, VIT_SPLIT),
             ("what-if: precise decoder, split vit + split proj", ALL, True, VIT_SPLIT + ("proj.mid", "proj.out")),
             ("what-if: precise decoder, split vit without v", ALL, True, tuple(c for c in VIT_SPLIT if c != "vit.v")),
             ("what-if: precise decoder, split vit + split proj + split enc.out", ALL, True, VIT_SPLIT + ("proj.mid", "proj.out", "enc.out"))]
    # ... or as an fp16 tower: the same MFMA rate and bytes as bf16, three more significant bits (CLIP was trained in fp16; T5 is not fp16-safe)
    half = lambda *stacks: tuple("half:" + c for c in ALL if stack_of(c) in stacks)
    # (the last of these, with the projector's output rounded to fp16 and then to the bf16 feature tensor, is what ships: option vit_fp16)
    runs += [("what-if: precise decoder, vit.* in fp16", ALL, True, half("vit")),
             ("what-if: precise decoder, vit.* proj.* in fp16", ALL, True, half("vit", "proj")),
             ("what-if: bf16 decoder, vit.* proj.* in fp16", ALL, False, half("vit", "proj")),
             ("engine as shipped at the end of round 4 (precise decoder + option vit_fp16)", ALL, True, ("ship:vit_fp16",))]
    # next candidate: the encoder's attention side in fp16 as well -- what HF's own fp16 T5 path holds in fp16 (it keeps only `wo` in fp32,
    # modeling_t5.py `_keep_in_fp32_modules`); the residual stream is fp32 here anyway
    enc_attn_side = ("half:enc.norm", "half:enc.qkv", "half:enc.p", "half:enc.attn", "half:enc.out")
    runs += [("what-if: as shipped + encoder norm / q k v / P / attention output / final norm in fp16", ALL, True, ("ship:vit_fp16",) + enc_attn_side),
             ("what-if: as shipped + every encoder class in fp16", ALL, True, ("ship:vit_fp16",) + half("enc"))]
    # round 5: what is left once the encoder's attention side is fp16 -- the feature tensor (proj.out: fp16 GEMM result rounded to bf16 for the
    # C ABI), the encoder's bf16 sub-layer outputs / FFN product, the decoder's floor
    runs += [("r5: shipped + enc attention side fp16 + feature tensor fp16", ALL, True, ("ship:vit_fp16",) + enc_attn_side + ("half:proj.out",)),
             ("r5: shipped + feature tensor fp16", ALL, True, ("ship:vit_fp16", "half:proj.out")),
             ("r5: shipped + enc attention side fp16 + feature fp16, enc.delta enc.act exact", without("enc.delta", "enc.act"), True,
              ("ship:vit_fp16",) + enc_attn_side + ("half:proj.out",)),
             ("r5: shipped + enc attention side fp16 + feature fp16 + enc.out split", ALL, True,
              ("ship:vit_fp16", "half:enc.norm", "half:enc.qkv", "half:enc.p", "half:enc.attn", "enc.out", "half:proj.out")),
             ("r5: decoder floor (vit proj enc exact)", without("vit.*", "proj.*", "enc.*"), True, ()),
             ("r5: the engine as shipped in round 5 (precise decoder + vit_fp16 + enc_fp16)", ALL, True, ("ship:vit_fp16", "ship:enc_fp16")),
             ("r5: as round 5 but only the attention sub-block in fp16 (FFN norm output and wi stay bf16)", ALL, True, ("ship:vit_fp16", "ship:enc_fp16_attn")),
             ("r5: round 5 + decoder cross score path (q, q.Wk, probabilities) and the encoder output in fp16", ALL, True,
              ("ship:vit_fp16", "ship:enc_fp16", "half:enc.out", "half:dec.cq", "half:dec.cqk", "half:dec.cprobs")),
             ("r5: the engine with options vit_fp16 + enc_fp16 + dec_fp16 (what ships at the end of round 5)", ALL, True,
              ("ship:vit_fp16", "ship:enc_fp16", "ship:dec_fp16"))]
    runs = [(r + (False, ()))[:4] if len(r) < 4 else r for r in runs]          # (name, classes, precise decoder, split set)
    if a.only:
        keep = set(a.only.split(";"))
        runs = [r for r in runs if r[0] in keep or r[0].startswith("none")]
    # order: decoder-only sets first (cheapest), vit-touching sets last; drop stage caches a later run cannot reuse
    rank = lambda cl: 2 if any(stack_of(c) in ("vit", "proj") for c in cl) else 1 if any(stack_of(c) == "enc" for c in cl) else 0
    runs.sort(key=lambda r: rank(r[1]))
    results, ref = {}, None
    print(f"# {cfg.name}, {a.pairs} pairs (bench.synth_batch seed {a.seed}), labels {labels[0].tolist()}, products in {a.acc}, "
          f"gains {gains}, {torch.get_num_threads()} threads", flush=True)
    for name, classes, precise, split in runs:
        t0 = time.time()
        lp = R.run(classes, precise, split)
        if ref is None:
            assert not classes
            ref = lp
            with torch.device(dev):
                truth = torch.cat([Oracle(cfg, w, device=dev).forward(*r_.batch)["label_logprobs"] for r_ in Rs]).cpu()
            results["_none_vs_fp32_oracle"] = float((lp[1.0] - truth).abs().max()) if 1.0 in lp else None
            print(f"# unrounded run vs oracle/clip_t5_oracle.py (tiled attention, reassociated cross-attention, fp32): "
                  f"{results['_none_vs_fp32_oracle']:.2e};  log P(yes) range [{float(truth[:, 0].min()):.2f}, {float(truth[:, 0].max()):.2f}]", flush=True)
        results[name] = {f"gain{g:g}": stats(lp[g], ref[g]) for g in gains}
        results[name]["seconds"] = round(time.time() - t0, 1)
        print(f"{name:42s} " + "  ".join(f"g{g:g}: max {results[name][f'gain{g:g}']['max']:.2e} mean {results[name][f'gain{g:g}']['mean']:.2e}" for g in gains)
              + f"   ({results[name]['seconds']} s)", flush=True)
        if rank(classes) < 2:
            pass
        else:
            R.drop("vit"); R.drop("proj"); R.drop("enc")
        if rank(classes) == 1:
            R.drop("enc")
    results["_meta"] = {"model": cfg.name, "pairs": a.pairs, "device": a.device + (" (" + torch.cuda.get_device_name(0) + ")" if dev.type == "cuda" else ""), "seed": a.seed, "gains": gains, "acc": a.acc,
                        "threads": torch.get_num_threads(), "seconds": round(time.time() - t00, 1),
                        "logp_yes_fp32": [round(float(x), 4) for x in ref[gains[0]][:, 0]]}
    if a.out:
        with open(a.out + ".json", "w") as f:
            json.dump(results, f, indent=1)
        print("wrote", a.out + ".json")


if __name__ == "__main__":
    main()
