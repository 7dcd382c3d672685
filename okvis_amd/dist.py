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
    """BASELINE configs[3] (64 windows over N GPUs): window i -> rank i mod world.  The rule lives in the C-ABI
    (okvis_ba_shard); this is its Python view (falls back to the same one-liner if the library is not built)."""
    try:
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        ids = (C.c_int32 * max(1, (n_total + world - 1) // world))()
        n = C.c_int32()
        _lib.check(L.okvis_ba_shard(n_total, rank, world, ids, C.byref(n)), "shard")
        return [int(ids[i]) for i in range(n.value)]
    except ImportError:
        return [i for i in range(n_total) if i % world == rank]


class WindowRecordC(__import__("ctypes").Structure):
    """okvis_ba_window_record (include/okvis_amd_ba.h): the fixed-size record of the one all-gather"""
    _fields_ = [("window_id", __import__("ctypes").c_uint32), ("iterations", __import__("ctypes").c_uint32),
                ("final_cost", __import__("ctypes").c_double), ("seconds", __import__("ctypes").c_double)]


def batch_run(windows, rank: int, world: int, device: int, num_iter: int, options=None):
    """okvis_ba_batch_run: shard the job's windows (i -> rank i mod world), run this rank's share on `device`, return its
    records [[window_id, iterations, final_cost, seconds], ...].  The caller all-gathers them (gather_records)."""
    import ctypes as C
    from . import _lib
    from .window import WindowC
    L = _lib.lib()
    arr = (WindowC * len(windows))()
    keep = []
    for i, w in enumerate(windows):
        w.validate()
        wc, k = w.as_c()
        arr[i] = wc
        keep.append(k)
    cap = max(1, (len(windows) + world - 1) // world)
    recs = (WindowRecordC * cap)()
    n = C.c_int32()
    _lib.check(L.okvis_ba_batch_run(device, rank, world, len(windows), arr, C.byref(options) if options is not None else None,
                                    num_iter, C.cast(recs, C.c_void_p), C.byref(n)), "batch_run")
    del keep
    return [[float(recs[i].window_id), float(recs[i].iterations), recs[i].final_cost, recs[i].seconds] for i in range(n.value)]


def batch_run_gathered(windows, rank: int, world: int, device: int, num_iter: int, id_file: str, options=None, timeout_s=60.0):
    """okvis_ba_batch_run_gathered: the whole multi-GPU job from C++ — shard, run this rank's share, all-gather the records over
    RCCL (loaded by the library with dlopen; `id_file` carries the ncclUniqueId between the ranks).  Returns the records of
    ALL windows in window order; Python / torch is only the process launcher."""
    import ctypes as C
    from . import _lib
    from .window import OptionsC, WindowC
    L = _lib.lib()
    L.okvis_ba_batch_run_gathered.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, C.POINTER(WindowC), C.POINTER(OptionsC),
                                              C.c_int, C.c_char_p, C.c_double, C.c_void_p]
    arr = (WindowC * len(windows))()
    keep = []
    for i, w in enumerate(windows):
        w.validate()
        wc, k = w.as_c()
        arr[i] = wc
        keep.append(k)
    recs = (WindowRecordC * len(windows))()
    _lib.check(L.okvis_ba_batch_run_gathered(device, rank, world, len(windows), arr, C.byref(options) if options is not None else None,
                                             num_iter, os.fsencode(id_file), float(timeout_s), C.cast(recs, C.c_void_p)),
               "batch_run_gathered")
    del keep
    return [[float(r.window_id), float(r.iterations), r.final_cost, r.seconds] for r in recs]


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (returns the module or None for 1 rank)."""
    r = Rank.from_env()
    # OKVIS_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL path on a 1-GPU box)
    if r.world <= 1 and not os.environ.get("OKVIS_FORCE_DIST"):
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = os.environ.get("OKVIS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_device(r))
    dist.init_process_group(backend=backend)
    return dist


def local_device(r: "Rank | None" = None) -> int:
    """HIP ordinal of this rank: LOCAL_RANK, or 0 for every rank with OKVIS_SHARE_GPU=1 (several processes on ONE GPU:
    the 2-process check that fits a 1-GPU lease; RCCL refuses two ranks on one device, so that run uses gloo)."""
    r = r or Rank.from_env()
    return 0 if os.environ.get("OKVIS_SHARE_GPU") else r.local_rank


def _device(dist):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", local_device())
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
