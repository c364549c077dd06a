"""N > 1 with REAL engines before the driver's scaling run does it (-m gpu).  The pool hands out one MI355X per call, so two
ranks share device 0 and their collectives go through gloo (host memory): everything of the multi-rank path except the RCCL transport itself
runs for real -- two processes, two VqsEngine replicas with their own weights / packed fp16 copies / workspaces on the device, the launcher,
the rank-local core affinity, the sharding of pairs (bench.py) and of an M x N grid by image (Score.forward with distributed=True), the
gathers.  The RCCL transport with one rank: tests/test_gpu_rccl_single_rank.py; world size 2 with an engine double on CPUs:
tests/test_sharding_gloo.py, tests/test_bench_harness.py.  Reference unit being sharded: /root/reference/t2v_metrics/score.py:104-106."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)
    return env


def _bench(gpus, extra_env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--model", "clip-flant5-xl", "--batch", "32", "--pairs", "128",
                        "--warmup", "1", "--cpu-pairs", "0", "--also", "none"], capture_output=True, text=True, timeout=900,
                       env=_env(VQS_BENCH_BACKEND="gloo", **(extra_env or {})), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_two_real_engine_ranks_on_one_device_equal_one_rank():
    """`python bench.py --gpus 2` launches its own two ranks; 128 pairs in total (inputs a function of the global pair index): the gathered
    scores' checksum must equal the single-rank run's to the bit -- a pair's score does not depend on which rank, or in which batch, it was
    scored -- and every rank reports its throughput and its peak HBM."""
    two = _bench(2)
    one = _bench(1)
    assert two["ranks_seen"] == 2 and two["n_gpus"] == 2 and two["collective"].startswith("gloo") and one["ranks_seen"] == 1
    assert two["config"]["total_pairs"] == one["config"]["total_pairs"] == 128
    assert two["scores_checksum"] == one["scores_checksum"], (two["scores_checksum"], one["scores_checksum"])
    assert len(two["per_rank_pairs_per_s"]) == 2 and all(v > 0 for v in two["per_rank_pairs_per_s"])
    assert len(two["per_rank_peak_hbm_gb"]) == 2 and all(5.0 < v < 60.0 for v in two["per_rank_peak_hbm_gb"]), two["per_rank_peak_hbm_gb"]   # XL: 5.7 GB of weights + copies + workspaces
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "multirank_two_engines_one_device.json"), "w") as f:
        json.dump({"gpus2": two, "gpus1": one}, f)


GRID_SCRIPT = r"""
import os, sys, json, torch, numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
import t2v_metrics_amd as t2v
from t2v_metrics_amd.config import get_config
from t2v_metrics_amd.weights import make_seeded_weights
from tests.test_host_api import FakeTokenizer
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
if world > 1:
    dist.init_process_group("gloo")
torch.cuda.set_device(0)
cfg = get_config("small")
w = make_seeded_weights(cfg, seed=7, device="cpu")
s = t2v.VQAScore(model="clip-flant5-xl", device="cuda:0", config=cfg, weights=w, tokenizer=FakeTokenizer(cfg.t5.vocab), image_workers="thread",
                 distributed=world > 1)
imgs = sorted(os.path.join(%(tmp)r, f) for f in os.listdir(%(tmp)r) if f.endswith(".png"))
texts = ["a red square", "two blue circles on a table", "nothing at all", "a cat"]
grid = s(images=imgs, texts=texts).cpu()
seen = sum(int(c) for c in getattr(s.model, "_n_encoded", [])) if hasattr(s.model, "_n_encoded") else -1
ds = [{"images": [imgs[k %% len(imgs)]], "texts": [texts[k %% 4], texts[(k + 1) %% 4]]} for k in range(9)]
bf = s.batch_forward(ds, batch_size=4).cpu()
if rank == 0:
    np.save(%(out)r %% world, np.concatenate([grid.numpy().ravel(), bf.numpy().ravel()]))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("GRID_OK", rank, tuple(grid.shape), tuple(bf.shape))
"""


def test_vqascore_grid_and_dataset_sharded_over_two_real_engine_ranks(tmp_path):
    """t2v_metrics_amd.VQAScore(..., distributed=True) on two gloo ranks that share the device: Score.forward shards the 5 x 4 grid by image,
    batch_forward the 9 samples, each rank runs its block on its own HIP engine; rank 0's gathered result equals the single-process one bit for bit."""
    from PIL import Image
    rng = np.random.RandomState(3)
    for i in range(5):
        Image.fromarray(rng.randint(0, 256, (48 + 8 * i, 64, 3), dtype=np.uint8)).save(str(tmp_path / f"im{i}.png"))
    out = str(tmp_path / "scores_w%d.npy")
    script = GRID_SCRIPT % {"root": ROOT, "tmp": str(tmp_path), "out": out}
    path = tmp_path / "grid_worker.py"
    path.write_text(script)
    r1 = subprocess.run([sys.executable, str(path)], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r1.returncode == 0 and "GRID_OK 0 (5, 4) (9, 1, 2)" in r1.stdout, (r1.stdout[-1500:], r1.stderr[-3000:])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                         str(path)], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r2.returncode == 0 and r2.stdout.count("GRID_OK") == 2, (r2.stdout[-1500:], r2.stderr[-3000:])
    a, b = np.load(out % 1), np.load(out % 2)
    assert a.shape == b.shape == (5 * 4 + 9 * 2,) and np.array_equal(a, b), np.abs(a - b).max()
    assert np.isfinite(a).all() and (a > 0).all() and (a < 1).all()
