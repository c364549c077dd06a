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


def assert_same_call(*parts) -> None:
    """The sharded scoring calls are COLLECTIVES: every rank must make the same call with the same inputs.  All-gathers a 64-bit digest of
    `parts` (lengths, paths, texts ...) and raises ValueError ON EVERY RANK when they differ -- instead of a deadlock (ranks disagreeing on
    whether to enter) or a silently mixed result (equal shapes, different inputs).  No-op without an initialised process group."""
    d = _dist()
    if d is None:
        return
    import hashlib
    h = hashlib.blake2b(repr(parts).encode("utf-8", "surrogatepass"), digest_size=8).digest()
    mine = torch.tensor([int.from_bytes(h, "little", signed=True)], dtype=torch.int64)
    backend = d.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = mine.to(dev)
    out = [torch.empty_like(mine) for _ in range(d.get_world_size())]
    d.all_gather(out, mine)
    vals = [int(x.item()) for x in out]
    if any(v != vals[0] for v in vals):
        raise ValueError("distributed scoring call: the ranks do not agree on the inputs (digests %s). With distributed=True every rank must call "
                         "with the SAME images / texts / dataset; ranks that score different inputs must not shard (distributed=False or shard=False)"
                         % [hex(v & 0xffffffffffffffff) for v in vals])


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


# ----------------------------------------------------------------------------------------------------------------------
# Host-core affinity of a rank: the cores of the NUMA node its GPU hangs off (first touch of the pinned staging buffers, the
# image workers), shared evenly between the ranks of that node.  The reference has no multi-GPU path.
# ----------------------------------------------------------------------------------------------------------------------
def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def plan_rank_affinity(gpu_numa, node_cpus, allowed, local_rank: int):
    """Pure planning step.  gpu_numa[i] = NUMA node of local GPU i (-1 = unknown); node_cpus = {node: [cpu, ...]}; allowed = cores
    this process may use now.  Rank r (GPU r) gets an equal slice of its GPU's node, cut among the ranks on that node in rank
    order; unknown topology (or a node with fewer allowed cores than ranks) falls back to an equal split of `allowed` by rank."""
    allowed = sorted(allowed)
    n = len(gpu_numa)
    if n <= 1 or not 0 <= local_rank < n:
        return allowed
    node = gpu_numa[local_rank]
    cpus = [c for c in node_cpus.get(node, []) if c in set(allowed)] if node is not None and node >= 0 else []
    peers = [r for r in range(n) if gpu_numa[r] == node]
    if cpus and len(cpus) >= len(peers):
        per = len(cpus) // len(peers)
        k = peers.index(local_rank)
        return cpus[k * per:(k + 1) * per]
    per = len(allowed) // n
    return allowed[local_rank * per:(local_rank + 1) * per] if per > 0 else allowed


def pci_bdf(props) -> str:
    """PCI address "dddd:bb:dd.0" of a device from torch's device properties: `pci_domain_id` / `pci_bus_id` / `pci_device_id` are
    INTEGERS (the bus id alone was taken for the address, so the sysfs lookup never hit)."""
    return "%04x:%02x:%02x.0" % (int(getattr(props, "pci_domain_id", 0)), int(props.pci_bus_id), int(getattr(props, "pci_device_id", 0)))


def gpu_numa_nodes(n_gpus: int, sysfs: str = "/sys/bus/pci/devices"):
    """NUMA node of each visible GPU from sysfs (the device's PCI address -> <sysfs>/<bdf>/numa_node), -1 if unknown.
    (`rocm-smi --showtoponuma` prints the same numbers.)"""
    out = []
    for i in range(n_gpus):
        node = -1
        try:
            with open(f"{sysfs}/{pci_bdf(torch.cuda.get_device_properties(i))}/numa_node") as f:
                node = int(f.read().strip())
        except Exception:                      # noqa: BLE001 -- no sysfs entry / no such attribute: unknown
            node = -1
        out.append(node)
    return out


def set_rank_affinity(local_rank: int, local_world: int):
    """Bind this process to its rank's cores (see plan_rank_affinity); returns the core list, or None when nothing was changed."""
    import os
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return None
    if local_world <= 1 or len(allowed) < 2 * local_world:
        return None
    node_cpus = {}
    base = "/sys/devices/system/node"
    try:
        for d in os.listdir(base):
            if d.startswith("node") and d[4:].isdigit():
                with open(os.path.join(base, d, "cpulist")) as f:
                    cpus = _parse_cpulist(f.read())

                def core_key(c):               # SMT siblings next to each other: a rank gets whole cores, not other ranks' second threads
                    try:
                        with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as g:
                            return (min(_parse_cpulist(g.read())), c)
                    except OSError:
                        return (c, c)
                node_cpus[int(d[4:])] = sorted(cpus, key=core_key)
    except OSError:
        node_cpus = {}
    n_gpus = torch.cuda.device_count() if torch.cuda.is_available() else 0
    numa = gpu_numa_nodes(n_gpus) if n_gpus >= local_world else [-1] * local_world
    cores = plan_rank_affinity(numa[:local_world] if len(numa) >= local_world else [-1] * local_world, node_cpus, allowed, local_rank)
    if not cores:
        return None
    try:
        os.sched_setaffinity(0, cores)
    except OSError:
        return None
    return cores
