#!/usr/bin/env python3
"""VQAScore throughput bench (driver contract: one JSON line on rank 0).

Workload = BASELINE.json configs[1]: clip-flant5-xl, bf16, batch of 256 (image, text) pairs per GPU per
step, synthetic 224x224 uint8 images (bicubic-resized to the tower's 336x336 and CLIP-normalised once,
outside the timed region, pixels resident in HBM as bf16) + 32-position prompts (8 prefix ids, the image
sentinel, 24 suffix ids => encoder length 576 + 32 = 608), labels [2163, 1] (T = 2).  Distinct image per
pair (no ViT reuse).  Weights: seeded random at the exact architecture (no checkpoint offline).

A "step" = one full scoring pass over the batch: ViT-L/14-336 (23 layers) + projector + T5 encoder (24) +
2-row teacher-forced decoder (24) + lm_head + log-softmax/score.  N > 1: one process per GPU (torchrun),
independent replicas over disjoint pair shards (weak scaling), one RCCL all_gather of the scores at the end
of the timed region.

Extra objects:
  roofline     -- dominant kernel = vqs::gemm_bf16_kernel: algorithmic GEMM FLOPs (2*M*N*K per launch) over
                  the summed per-launch durations measured with HIP events on the launch stream during the
                  timed region (vqs_profile_*), against the 2.5 PFLOP/s dense bf16 MFMA peak.
  cpu_baseline -- the CPU oracle (oracle/clip_t5_oracle.py, fp32, bf16-rounded weights; kind "port") timed
                  on the host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense; /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def synth_batch(cfg, batch, seed, device, ragged=False):
    """SURVEY.md §8d "Config 2" generator; ragged=True is the "Config 3" stand-in (caption of 8-40 tokens between the
    8-token prefix + sentinel and the 24-token template suffix, right-padded with id 0 to the batch maximum)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 224, 224, 3), generator=g, dtype=torch.uint8)
    x = img.to(device).permute(0, 3, 1, 2).float()
    x = torch.nn.functional.interpolate(x, size=(cfg.vision.image, cfg.vision.image), mode="bicubic", align_corners=False)
    x = x.clamp(0, 255) / 255.0
    mean = torch.tensor(CLIP_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=device).view(1, 3, 1, 1)
    pixels = ((x - mean) / std).to(torch.bfloat16).contiguous()
    vocab = cfg.t5.vocab
    hi = min(32100, vocab)
    pre = torch.randint(3, hi, (batch, 8), generator=g)
    pre[:, 7] = 1
    suf = torch.randint(3, hi, (batch, 24), generator=g)
    suf[:, 23] = 1
    if ragged:
        cap_len = torch.randint(8, 41, (batch,), generator=g)
        width = 8 + 1 + 40 + 24
        ids = torch.zeros(batch, width, dtype=torch.int64)
        for i in range(batch):
            n = int(cap_len[i])
            cap = torch.randint(3, hi, (n,), generator=g)
            row = torch.cat([pre[i], torch.tensor([-200]), cap, suf[i]])
            ids[i, : row.numel()] = row
        ids = ids[:, : int(8 + 1 + cap_len.max() + 24)].to(torch.int32)
    else:
        ids = torch.cat([pre, torch.full((batch, 1), -200), suf], dim=1).to(torch.int32)
    labels = torch.tensor([[min(2163, vocab - 1), 1]] * batch, dtype=torch.int32)
    img_index = torch.arange(batch, dtype=torch.int32)
    return pixels, img_index.to(device), ids.to(device), labels.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="clip-flant5-xl")
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU per step")
    ap.add_argument("--cpu-pairs", type=int, default=2, help="pairs in the CPU-oracle sample (0 = skip)")
    ap.add_argument("--ragged", action="store_true", help="SURVEY config-3 stand-in: variable-length prompts, padded + masked")
    args = ap.parse_args()

    if args.model.startswith("qwen"):
        # BASELINE.json configs[4] (Qwen2.5-VL-7B, 8-frame video): measured by tools/bench_qwen.py, one GPU
        if args.gpus != 1:
            raise SystemExit("the Qwen2.5-VL bench line is single-GPU (replicas need no collective: run one process per GPU)")
        sys.argv = [os.path.join(ROOT, "tools", "bench_qwen.py"), "--model", args.model, "--steps", str(args.steps), "--warmup",
                    str(args.warmup), "--batch", str(min(args.batch, 64)), "--cpu-samples", str(min(args.cpu_pairs, 1))]
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_qwen
        return bench_qwen.main()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # VQS_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks share
    # devices, collectives go through host memory); the driver's runs use the default: nccl = RCCL, one GPU per rank.
    backend = os.environ.get("VQS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=device)
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod

    from t2v_metrics_amd.engine import VqsEngine

    cfg = get_config(args.model)
    weights = make_seeded_weights(cfg, seed=0, device=device)
    eng = VqsEngine(cfg, weights, device=device)
    B = args.batch
    pixels, img_index, ids, labels = synth_batch(cfg, B, seed=1234 + rank, device=device, ragged=args.ragged)
    L, T = ids.shape[1], labels.shape[1]
    s_e = L - 1 + cfg.vision.n_patches

    def step():
        feats = eng.encode_images(pixels)
        return eng.score(feats, img_index, ids, labels)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.profile(True)
    eng.profile_read(reset=True)
    t0 = time.perf_counter()
    all_scores = []
    for _ in range(args.steps):
        lp, sc = step()
        all_scores.append(sc)
    local = torch.cat(all_scores) if all_scores else torch.zeros(0, device=device)
    if dist is not None:
        local = local.to(coll_device)
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)            # the path's only exchange: scores to every rank
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile(False)
    gemm_bytes = eng.profile_bytes()
    n_gemm, gemm_ms, gemm_flops = eng.profile_read(reset=True)
    if rank == 0 and os.environ.get("VQS_BENCH_REPORT"):
        print(eng.profile_report(), file=sys.stderr, flush=True)

    t_el = torch.tensor([elapsed], device=coll_device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())
    total_pairs = B * args.steps * world
    value = total_pairs / elapsed if elapsed > 0 and args.steps > 0 else 0.0

    # real (unpadded) encoder lengths: padding the engine computes over is not algorithmic work
    lens = ((ids != 0).sum(1) - 1 + cfg.vision.n_patches).tolist()
    flops_pair = sum(cfg.flops_pair(int(l), T) for l in lens) / len(lens)
    out = {
        "metric": "image-text pairs scored/sec (whole node), " + cfg.name,
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (seeded 224x224 uint8 images resized to 336, seeded token ids, seeded random weights)",
        "config": {"workload": f"{cfg.name} bf16, batch={B} synthetic 224x224 + "
                               + ("ragged 41-73-tok prompts (padded, masked)" if args.ragged else "32-tok prompts") + " per GPU per step",
                   "pairs_per_gpu_per_step": B, "encoder_len": s_e, "decoder_len": T,
                   "parallelism": f"replica x{world} (pairs sharded, RCCL all_gather of scores)"},
        # FLOPs of the REFERENCE algorithm (SURVEY.md §8d formula).  The engine's reassociated decoder
        # cross-attention executes 4*S_e*D*I*layers fewer FLOPs per pair than that (same function);
        # "roofline" below is computed from the FLOPs the GEMM kernel really executed.
        "algorithmic_tflop_per_pair": flops_pair / 1e12,
        "executed_gemm_tflop_per_pair": (gemm_flops / max(total_pairs // world, 1)) / 1e12 if n_gemm > 0 else None,
        "model_tflops_per_gpu": value * flops_pair / 1e12 / world,
        "model_frac_of_mfma_peak": value * flops_pair / 1e12 / world / PEAK_BF16_TFLOPS,
    }
    if n_gemm > 0 and gemm_ms > 0:
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": "vqs::gemm_bf16_persistent / gemm_bf16_pingpong (every GEMM launch of the step)",
                           "bound": "mfma", "achieved": achieved,
                           "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                           "traffic": None, "launches": n_gemm, "avg_launch_ms": gemm_ms / n_gemm,
                           "avg_launch_tflop": gemm_flops / n_gemm / 1e12,
                           "algorithmic_bytes_per_launch": gemm_bytes / n_gemm,
                           "gemm_share_of_step_time": gemm_ms * 1e-3 / elapsed}
        # HBM-side bytes per GEMM launch: PMC counters cannot be read from inside the process; they come from the
        # committed rocprofv3 --pmc passes over this same command (tools/gpu_pmc_bench.sh -> profiles/*gemm_traffic.json)
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic_%s_b%d.json" % (cfg.name.replace("clip-flant5-", ""), B))
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            out["roofline"]["traffic"] = tj["traffic_bytes_per_launch"]
            out["roofline"]["traffic_unit"] = "bytes per launch (HBM-side: 2 x FETCH_SIZE + WRITE_SIZE), from " + os.path.basename(tpath)

    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        out["cpu_baseline"] = cpu_baseline(cfg, weights, pixels, img_index, ids, labels, args.cpu_pairs, lp)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(cfg, weights, pixels, img_index, ids, labels, n_pairs, lp_gpu):
    """Oracle (fp32 restatement of the reference's HF forward) on the host cores, first n_pairs pairs of the batch."""
    from oracle.clip_t5_oracle import Oracle
    w_cpu = {k: v.cpu() for k, v in weights.items()}
    orc = Oracle(cfg, w_cpu)
    px = pixels[:n_pairs].float().cpu()
    idx = torch.arange(n_pairs)
    t0 = time.perf_counter()
    ref = orc.forward(px, idx, ids[:n_pairs].cpu().long(), labels[:n_pairs].cpu().long())
    dt = time.perf_counter() - t0
    dlp = (lp_gpu[:n_pairs].float().cpu() - ref["label_logprobs"]).abs().max().item()
    return {"value": n_pairs / dt, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {n_pairs} pairs of the same batch, fp32 oracle, one pass ({dt:.1f} s)",
            "host_cpus": os.cpu_count(), "max_abs_dlogp_hip_vs_oracle": dlp}


if __name__ == "__main__":
    main()
