#!/usr/bin/env python3
"""Small-batch latency of one scoring pass (encode_images + score), eager launches vs a captured HIP graph.

At B = 256 a pass is ~1.4 s of GPU work behind ~2 000 asynchronous launches: launch cost is invisible.  At B = 1..16 (the
reference's interactive use: a handful of images x texts through Score.forward) the ~2 000 launches of a pass are the
bound.  The library never allocates, never synchronises and reads nothing from the host inside vqs_encode_images /
vqs_score, so the whole pass is capturable with hipStreamBeginCapture -- here through torch.cuda.CUDAGraph (which is a
hipGraph on ROCm).  Prints one JSON line per batch size."""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch  # noqa: E402
from t2v_metrics_amd.config import get_config  # noqa: E402
from t2v_metrics_amd.engine import VqsEngine  # noqa: E402
from t2v_metrics_amd.weights import make_seeded_weights  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="clip-flant5-xxl")
    ap.add_argument("--batches", default="1,4,16,64")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true", help="eager timing and a digest of the scores only (A/B of library builds: VQS_LIB_PATH)")
    args = ap.parse_args()
    cfg = get_config(args.model)
    dev = torch.device("cuda:0")
    w = make_seeded_weights(cfg, seed=0, device=dev)
    eng = VqsEngine(cfg, w, device=dev)
    for B in [int(x) for x in args.batches.split(",")]:
        pixels, idx, ids, labels = synth_batch(cfg, B, seed=7, device=dev)

        def step():
            return eng.score(eng.encode_images(pixels), idx, ids, labels)

        for _ in range(3):
            lp_e, sc_e = step()
        eager = timed(step, args.reps)
        lp_e, sc_e = step()
        lp_e, sc_e = lp_e.clone(), sc_e.clone()
        digest = hashlib.sha256(lp_e.cpu().numpy().tobytes() + sc_e.cpu().numpy().tobytes()).hexdigest()[:16]
        if args.no_graph:
            print(json.dumps({"model": cfg.name, "batch": B, "eager_ms": 1e3 * eager, "pairs_per_s_eager": B / eager, "scores_sha256_16": digest,
                              "library": os.environ.get("VQS_LIB_PATH", "product")}), flush=True)
            continue
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm-up on the capture stream (workspaces are sized here)
            step()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            lp_g, sc_g = step()
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(lp_g, lp_e) and torch.equal(sc_g, sc_e))
        graph = timed(g.replay, args.reps)
        print(json.dumps({"model": cfg.name, "batch": B, "eager_ms": 1e3 * eager, "hip_graph_ms": 1e3 * graph,
                          "speedup": eager / graph, "pairs_per_s_eager": B / eager, "pairs_per_s_graph": B / graph,
                          "bitwise_equal": same, "scores_sha256_16": digest}), flush=True)
        del g


if __name__ == "__main__":
    main()
