"""Multi-GPU driver logic: windows are independent units and shard across ranks with NO data-path
collective.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on
the GPU node, "gloo" in the CPU tests); the only exchanges are a MAX-reduce of the wall time and one
all-gather of a small per-rank timing record."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List


@dataclass
class Rank:
    rank: int
    world: int
    local_rank: int

    @staticmethod
    def from_env() -> "Rank":
        return Rank(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                    int(os.environ.get("LOCAL_RANK", "0")))


def shard_seeds(rank: int, world: int, windows_per_gpu: int, base_seed: int = 20240923) -> List[int]:
    """Weak scaling: rank r owns windows r*B .. r*B+B-1 (seed = base + global window index)."""
    assert 0 <= rank < world and windows_per_gpu > 0
    return [base_seed + rank * windows_per_gpu + i for i in range(windows_per_gpu)]


def shard_windows(n_total: int, rank: int, world: int) -> List[int]:
    """Strong-scaling variant (BASELINE configs[3]: 64 windows over N GPUs): window i -> rank i mod world."""
    return [i for i in range(n_total) if i % world == rank]


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (returns the module or None for 1 rank)."""
    r = Rank.from_env()
    # OKVIS_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a 1-GPU box)
    if r.world <= 1 and not os.environ.get("OKVIS_FORCE_DIST"):
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(r.local_rank)
    dist.init_process_group(backend=backend)
    return dist


def _device(dist):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", Rank.from_env().local_rank)
    return torch.device("cpu")


def max_over_ranks(dist, value: float) -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(dist, record: List[float]) -> List[List[float]]:
    """The one collective of the design: all-gather of a fixed-size per-rank record
    {rank, windows, iterations, seconds, final_cost_sum}."""
    if dist is None:
        return [list(map(float, record))]
    import torch
    t = torch.tensor(record, dtype=torch.float64, device=_device(dist))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]


def barrier(dist):
    if dist is not None:
        dist.barrier()
