"""Multi-GPU sharding of independent (image, text) work: one process per GPU, no data-path collective, one
gather of fp32 scores at the end.  Backend "nccl" is RCCL on ROCm (xGMI); the payload is a few
KB per rank, so the collective is latency-bound and a single all_gather is used.  The same code runs over gloo
on CPUs (tests/test_sharding_gloo.py)."""
from typing import Tuple

import torch


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def world() -> Tuple[int, int]:
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def shard_range(n: int, rank: int = None, world_size: int = None) -> Tuple[int, int]:
    """Contiguous block of [0, n) owned by `rank`: sizes differ by at most one, earlier ranks get the extra."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Inverse of shard_range: every rank contributes its block of rows, every rank receives all n_total rows."""
    d = _dist()
    if d is None:
        assert local.shape[0] == n_total
        return local
    rank, ws = d.get_rank(), d.get_world_size()
    per = -(-n_total // ws)
    backend = d.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.zeros((per,) + tuple(local.shape[1:]), dtype=torch.float32, device=dev)
    buf[: local.shape[0]] = local.to(dev, torch.float32)
    out = [torch.empty_like(buf) for _ in range(ws)]
    d.all_gather(out, buf)
    rows = []
    for r in range(ws):
        lo, hi = shard_range(n_total, r, ws)
        rows.append(out[r][: hi - lo])
    return torch.cat(rows, 0).cpu()
