#!/usr/bin/env python3
"""End-to-end throughput of the drop-in API INCLUDING the host input pipeline: PNG decode +
expand2square + PIL bicubic 336 + CLIP normalise on the thread pool, pinned staging + H2D, tokenisation, engine.
Seeded weights, a stand-in whitespace tokenizer (no SentencePiece model offline).  Prints one JSON line."""
import argparse, json, os, sys, tempfile, time, zlib
import numpy as np
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import t2v_metrics_amd as t2v


class WordTokenizer:
    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, text):
        class R:
            pass
        r = R()
        r.input_ids = [3 + zlib.crc32(w.encode()) % (self.vocab - 3) for w in text.split()] + [1]
        return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1024)
    ap.add_argument("--size", type=int, default=512, help="PNG edge length")
    ap.add_argument("--model", default="clip-flant5-xl")
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--image-workers", default="process", choices=["process", "thread"], help="decode workers: processes (default) or the GIL-sharing thread pool of rounds 1-3")
    ap.add_argument("--host-slice", type=int, default=0, help="N > 1: run on 1/N of the host (what a rank gets at N GPUs per node): the first "
                    "cores/N physical cores of NUMA node 0 with their SMT siblings (sched_setaffinity before anything starts)")
    args = ap.parse_args()
    if args.host_slice > 1:
        from t2v_metrics_amd.sharding import _parse_cpulist
        node0 = _parse_cpulist(open("/sys/devices/system/node/node0/cpulist").read()) if os.path.exists("/sys/devices/system/node/node0/cpulist") else sorted(os.sched_getaffinity(0))
        cores = {}
        for c in node0:
            try:
                sib = tuple(_parse_cpulist(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
            except OSError:
                sib = (c,)
            cores.setdefault(min(sib), sib)
        total_cores = (os.cpu_count() or len(node0)) // max(1, len(next(iter(cores.values()))))
        take = max(1, total_cores // args.host_slice)
        mine = sorted(c for k in sorted(cores)[:take] for c in cores[k])
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, len(mine) // 2))
    tmp = tempfile.mkdtemp(prefix="vqs_imgs_")
    rng = np.random.RandomState(0)
    base = rng.randint(0, 256, (args.size // 8, args.size // 8, 3), dtype=np.uint8)
    def make(i):                         # smooth-ish images (upsampled noise + per-image offset): realistic PNG sizes
        im = Image.fromarray(np.roll(base, i, axis=0)).resize((args.size, args.size + (i % 3) * 16), Image.BILINEAR)
        p = os.path.join(tmp, f"im{i:05d}.png")
        im.save(p)
        return p
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(32, len(os.sched_getaffinity(0)))) as ex:
        paths = list(ex.map(make, range(args.pairs)))
    texts = [f"a photo number {i} of someone doing something in place {i % 17}" for i in range(args.pairs)]
    from t2v_metrics_amd.config import get_config
    cfg = get_config(args.model)
    scorer = t2v.VQAScore(model=args.model, device="cuda", weights="seeded", tokenizer=WordTokenizer(cfg.t5.vocab),
                          num_workers=args.workers, image_workers=args.image_workers)
    m = scorer.model
    m.forward(paths[:256], texts[:256])          # warm-up (workspaces, pools, worker processes)
    torch.cuda.synchronize()
    # host half of the PRODUCT path alone, warm: decode + pad + resize + crop of 256 files into the pinned uint8 staging buffer
    host_256 = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        m._load_images_host_u8(paths[256:512] if len(paths) >= 512 else paths[:256])
        host_256 = min(host_256, time.perf_counter() - t0)
    best = 1e9
    for _ in range(args.reps):
        t0 = time.perf_counter()
        sc = m.forward(paths, texts)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    # where a step's time goes: the same call once more with the library's per-launch HIP events on (GEMM launches only) and the
    # prompt lengths the tokenizer produced (the engine computes over 575 + L positions per pair)
    diag = {}
    if hasattr(m.engine, "profile"):
        from t2v_metrics_amd.models.vqascore_models.clip_t5_model import default_question_template
        ids, _ = m.tokenize([default_question_template.format(t) for t in texts[:256]], ["Yes"] * 256)
        m.engine.profile(True)
        m.engine.profile_read(reset=True)
        t0 = time.perf_counter()
        m.forward(paths, texts)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        m.engine.profile(False)
        n_gemm, gemm_ms, gemm_flops = m.engine.profile_read(reset=True)
        # the engine alone on the SAME inputs (pixels resident, same prompt lengths): the rate the pipeline can at best deliver
        px = m.load_images(paths[:256])
        ids_d, lab_d = m.tokenize([default_question_template.format(t) for t in texts[:256]], ["Yes"] * 256)
        idx_d = torch.arange(256, dtype=torch.int32)
        for _ in range(1):
            m.engine.score(m.engine.encode_images(px), idx_d, ids_d, lab_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.engine.score(m.engine.encode_images(px), idx_d, ids_d, lab_d)
        torch.cuda.synchronize()
        eng_only = 3 * 256 / (time.perf_counter() - t0)
        diag = {"profiled_call_wall_s": wall, "engine_only_same_inputs_pairs_per_s": eng_only, "ratio_to_engine_only_same_inputs": (args.pairs / best) / eng_only, "gemm_ms_per_256_pairs": gemm_ms * 256 / args.pairs, "gemm_tflops": gemm_flops / max(gemm_ms, 1e-9) / 1e9,
                "prompt_ids_per_pair_first_batch": int(ids.shape[1]), "encoder_len_first_batch": int(ids.shape[1]) - 1 + cfg.vision.n_patches}
    print(json.dumps({"metric": "pairs/s through VQAScoreModel.forward incl. PNG decode, preprocessing, H2D, tokenisation", **diag,
                      "value": args.pairs / best, "pairs": args.pairs, "png_edge": args.size, "model": args.model,
                      "workers": m.num_workers, "image_workers": args.image_workers, "host_cpus": os.cpu_count(),
                      "host_threads_allowed": len(os.sched_getaffinity(0)),
                      "host_preprocess_256_images_s": host_256, "score_range": [float(sc.min()), float(sc.max())]}))


if __name__ == "__main__":
    main()
