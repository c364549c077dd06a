#!/usr/bin/env python3
"""Throughput of the Qwen2.5-VL-7B VQAScore row (BASELINE.json configs[4]; SURVEY.md §8d "Config 5"): B synthetic 8-frame
336x448 videos per step -> grid (4, 24, 32) = 3072 patches -> 768 merged tokens + 40 text tokens per sample, one prefill,
P("Yes") from the last-position logits.  Seeded random weights at the public 7B dims.  Prints one JSON line; with
--cpu-samples n also times the fp32 oracle on the host (kind "port") and reports |d log P|."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from t2v_metrics_amd.qwen import get_qwen_config
from t2v_metrics_amd.qwen.weights import make_seeded_qwen_weights


def flops_per_sample(cfg, n_patches, L):
    v, t = cfg.vision, cfg.text
    vis = n_patches * (2 * v.patch_dim * v.hidden + v.depth * (8 * v.hidden * v.hidden + 6 * v.hidden * v.mlp))
    # attention: windows of 64 patches, full blocks over a frame
    full = len(v.fullatt_blocks)
    vis += n_patches * 4 * v.hidden * (64 * (v.depth - full) + 768 * full)
    nm = n_patches // v.merge_unit
    vis += nm * 2 * (v.hidden * v.merge_unit) * (v.hidden * v.merge_unit + v.out_hidden)
    kv = t.kv_heads * t.head_dim
    txt = L * t.layers * (2 * t.hidden * (t.hidden + 2 * kv) + 2 * t.hidden * t.hidden + 6 * t.hidden * t.mlp) + t.layers * 2 * L * L * t.hidden
    txt += 2 * t.hidden * t.vocab
    return vis + txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen2.5-vl-7b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-samples", type=int, default=0)
    ap.add_argument("--decode-steps", type=int, default=0, help="also time N cached decode steps behind one prefill (KV cache)")
    ap.add_argument("--parity-samples", type=int, default=0, help="|delta log P(answer)| of the first N samples against the fp32 oracle evaluated "
                    "in torch fp32 on the device (oracle/qwen25vl_oracle.py; 8 samples: ~3 s)")
    ap.add_argument("--tail", type=int, default=1, help="option tail_precise (1 = the default: logits from the precise re-evaluation of the last position)")
    ap.add_argument("--fp16", type=int, default=1, help="option fp16 (1 = the default: the range-safe fp16 forms; 0 = bf16 everywhere, the reference's dtype)")
    ap.add_argument("--rope-fused", type=int, default=1, help="option rope_fused (1 = the default: q / k rotated inside the q|k|v epilogue under the fp16 forms)")
    args = ap.parse_args()
    from t2v_metrics_amd.qwen.engine import QwenEngine
    cfg = get_qwen_config(args.model)
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    w = make_seeded_qwen_weights(cfg, seed=0, device="cpu")
    eng = QwenEngine(cfg, w, device=dev, fp16=bool(args.fp16))
    eng.set_option("tail_precise", args.tail)
    eng.set_option("rope_fused", args.rope_fused)
    fp16_on = eng.fp16_active
    rb, rs = eng.range_report()
    t_init = time.perf_counter() - t0
    B = args.batch
    grid = (4, 24, 32)
    n_patches = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(1234)
    px = torch.randn(B * n_patches, cfg.vision.patch_dim, generator=g).to(torch.bfloat16).to(dev)
    n_merged = n_patches // cfg.vision.merge_unit
    yes_id = 9454 % cfg.text.vocab
    rows = []
    for b in range(B):
        pre = torch.randint(10, min(cfg.text.vocab, 150000), (14,), generator=g)
        post = torch.randint(10, min(cfg.text.vocab, 150000), (24,), generator=g)
        rows.append(torch.cat([pre, torch.tensor([cfg.vision_start_token_id]), torch.full((n_merged,), cfg.video_token_id),
                               torch.tensor([cfg.vision_end_token_id]), post]))
    ids = torch.stack(rows)
    mask = torch.ones_like(ids)
    grids = [grid] * B

    def step():
        merged = eng.encode_vision(px, grids)
        return eng.score_logits(merged, ids, mask, grids)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.profile(True)
    eng.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.profile(False)
    n_gemm, gemm_ms, gemm_flops, gemm_bytes = eng.profile_read(reset=True)
    L = ids.shape[1]
    fl = flops_per_sample(cfg, n_patches, L)
    out = {"metric": "videos scored/sec, " + cfg.name, "value": B * args.steps / dt, "unit": "samples/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "dtype": ("fp16 behind bind-time power-of-two scales (every 16-bit activation of tower, merger and prefill; fp16 weight copies) + split-bf16 / fp32 "
                     "precise tail; fp32 accumulation, statistics, softmax, residual stream" if fp16_on else "bf16 (16-bit activations and weights) + split-bf16 / fp32 precise tail"),
           "rope_in_qkv_epilogue": bool(eng.get_option("rope_fused")),
           "fp16_forms": {"active": fp16_on, "sites": int(len(rs)), "sites_behind_a_scale": int((rs < 1.0).sum()) if len(rs) else 0,
                          "largest_proven_bound": float(rb.max()) if len(rb) else None, "smallest_scale": float(rs.min()) if len(rs) else None},
           "data": "synthetic (seeded patches, token ids, weights)",
           "config": {"workload": f"{cfg.name}, batch={B} x 8-frame 336x448 video (3072 patches -> 768 vision tokens) + 40 text tokens", "L": L},
           "algorithmic_tflop_per_sample": fl / 1e12, "model_tflops": B * args.steps / dt * fl / 1e12,
           "model_frac_of_mfma_peak": B * args.steps / dt * fl / 1e12 / 2500.0, "init_s": t_init,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "tail_precise": args.tail}
    if n_gemm > 0 and gemm_ms > 0:
        ach = gemm_flops / (gemm_ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": ("vqs::gemm_f16s_quad" if fp16_on else "vqs::gemm_bf16_quad") + " + gemm_bf16_persistent / stream forms (every GEMM launch of the step; "
                                     "gate|up-interleaved shapes as executed; language-model heads padded to 128 lanes)", "bound": "mfma", "achieved": ach, "peak": 2500.0,
                           "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None, "launches": n_gemm,
                           "avg_launch_ms": gemm_ms / n_gemm, "algorithmic_bytes_per_launch": gemm_bytes / n_gemm,
                           "gemm_share_of_step_time": gemm_ms * 1e-3 / dt}
    if args.decode_steps > 0:
        # generation beyond the first token: one prefill that keeps the KV cache, then one cached position per step.  A decode step
        # streams every language-model weight once (2 B x 7.07e9 parameters) and the cached K / V rows: HBM-bound.
        n = args.decode_steps
        merged = eng.encode_vision(px, grids)
        lg, state = eng.prefill(merged, ids, mask, grids, n + 2)
        tok = lg.argmax(-1)
        eng.decode(state, tok)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(n):
            h0 = time.perf_counter()
            lg = eng.decode(state, tok)                     # returns when the step's launches are ENQUEUED
            host += time.perf_counter() - h0
            tok = lg.argmax(-1)
        torch.cuda.synchronize()
        dtd = (time.perf_counter() - t0) / n
        t_ = cfg.text
        wbytes = 2.0 * (t_.layers * (t_.hidden * (t_.heads + 2 * t_.kv_heads) * t_.head_dim + t_.hidden * t_.heads * t_.head_dim + 3 * t_.hidden * t_.mlp)
                        + t_.vocab * t_.hidden)
        kvbytes = 2.0 * 2 * t_.layers * B * t_.kv_heads * (L + n / 2) * 128
        out["decode"] = {"ms_per_step": 1e3 * dtd, "host_enqueue_ms_per_step": 1e3 * host / n, "tokens_per_s": B / dtd, "batch": B, "steps": n,
                         "hbm_bound": {"weight_bytes": wbytes, "kv_bytes": kvbytes, "achieved_GBps": (wbytes + kvbytes) / dtd / 1e9, "peak_GBps": 8000.0,
                                       "frac": (wbytes + kvbytes) / dtd / 8e12}}
    if args.cpu_samples > 0:
        from oracle.qwen25vl_oracle import QwenOracle
        n = args.cpu_samples
        o = QwenOracle(cfg, w)
        t0 = time.perf_counter()
        ref = o.forward(ids[:n], mask[:n], px[: n * n_patches].float().cpu(), grids[:n])
        dtc = time.perf_counter() - t0
        lp = torch.log_softmax(logits[:n].float().cpu(), -1)[:, yes_id]
        lr = torch.log_softmax(ref, -1)[:, yes_id]
        out["cpu_baseline"] = {"value": n / dtc, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"first {n} sample(s) of the batch, fp32 oracle ({dtc:.1f} s)",
                               "max_abs_dlogp_hip_vs_oracle": float((lp - lr).abs().max())}
    if args.parity_samples > 0:
        # BASELINE.md section 3's criterion for this row: HIP vs fp32 truth on the SAME inputs.  Truth = the oracle's code in torch fp32 on
        # the device (the host cores need minutes per 7B-size sample); test infrastructure, after the timed region.
        from oracle.qwen25vl_oracle import QwenOracle
        n = min(args.parity_samples, B)
        t0 = time.perf_counter()
        o = QwenOracle(cfg, eng._weights, device=dev)
        refs = []
        with torch.device(dev):
            for s0 in range(0, n, 4):
                e0 = min(s0 + 4, n)
                refs.append(o.forward(ids[s0:e0].to(dev), mask[s0:e0].to(dev), px[s0 * n_patches: e0 * n_patches].float(), grids[s0:e0]).float())
        torch.cuda.synchronize()
        t_truth = time.perf_counter() - t0
        lr = torch.log_softmax(torch.cat(refs), -1)
        lh = torch.log_softmax(logits[:n].float(), -1)
        d_ans = (lh[:, yes_id] - lr[:, yes_id]).abs()
        top5 = lr.topk(5, -1).indices
        d_top = (lh.gather(-1, top5) - lr.gather(-1, top5)).abs().max(-1).values
        BOUND = 1e-3
        out["parity"] = {"pairs": n, "bound": BOUND, "truth": "oracle/qwen25vl_oracle.py evaluated in torch fp32 on the device (%.1f s)" % t_truth,
                         "gains": {"1": {"max": float(d_ans.max()), "mean": float(d_ans.mean()), "yes_token_max": float(d_ans.max()),
                                         "pairs_over_bound": int((d_ans > BOUND).sum()), "per_pair": [round(float(x), 6) for x in d_ans],
                                         "top5_max": float(d_top.max()), "top5_mean": float(d_top.mean()),
                                         "logp_yes_range": [round(float(lr[:, yes_id].min()), 3), round(float(lr[:, yes_id].max()), 3)]}},
                         "status": ("within the 1e-3 bound" if float(d_ans.max()) <= BOUND else
                                    "ABOVE the 1e-3 bound" + ("" if fp16_on else ": option fp16 is off -- the bf16 forms hold every 16-bit tensor at 8 significant bits "
                                                              "(profiles/r5_qwen_error_attribution.md, r6_call1_*)"))}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
