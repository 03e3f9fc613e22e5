"""Data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The path shards naturally: every rank trains on its own mini-batch with its own,
globally numbered, Gibbs chains and holds a full replica of the (small) model.  The only exchange
is ONE all-reduce per training step over one flat float32 buffer: the encoder/decoder gradients
and -- on the steps that train the GRBM -- behind them, in the same buffer, the GRBM
sufficient-statistic differences d/dh, d/dJ.  For the 1.4-27 MB messages of n = 64..1024 a single
large collective is the xGMI-friendly shape (the ring is per-link bound; fewer, larger
collectives amortise its latency).  BatchNorm batch statistics and the MMD estimator are per rank
(DDP semantics); BatchNorm RUNNING statistics (read only in eval mode) are rank 0's: they are
broadcast before a checkpoint is written or an eval-mode entry point runs
(``ModelWrapper.sync_buffers``).  Replicas start from rank 0's parameters (``sync_replicas``).
The reference has no distributed code at all.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


class DataParallel:
    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.owns_group = False
        self.force = os.environ.get("DVG_FORCE_DIST") == "1"  # exercise the collective path with a single rank
        if self.rank < 0 or self.rank >= self.world_size:
            raise RuntimeError(f"DataParallel: RANK={self.rank} outside WORLD_SIZE={self.world_size}")
        # one process per GPU of ONE node: a launcher that hands out more local ranks than the node has devices would
        # otherwise surface as an opaque HIP 'invalid device ordinal' (or, worse, as two ranks on one device inside RCCL)
        if torch.cuda.is_available() and device is None and self.local_rank >= torch.cuda.device_count():
            raise RuntimeError(f"DataParallel: LOCAL_RANK={self.local_rank} but this node exposes "
                               f"{torch.cuda.device_count()} GPU(s); launch one process per visible device")
        if (self.world_size > 1 or self.force) and not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
            self.owns_group = True
        self.device = device or (torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu"))

    @property
    def active(self) -> bool:
        return self.world_size > 1 or self.force

    def _host_staged(self, t: torch.Tensor) -> bool:
        # gloo group over device tensors (two test ranks sharing one GPU): the collective runs on a host copy
        return t.is_cuda and dist.get_backend() == "gloo"

    def all_reduce_sum(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of one flat buffer: the single collective of a step (the 1/world_size of the mean
        is the Adam kernel's ``grad_scale``, so no separate scaling pass runs)."""
        if self.active:
            if self._host_staged(flat):
                host = flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                flat.copy_(host)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def all_reduce_mean(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place mean over ranks (tests and tools; the training step uses :meth:`all_reduce_sum`)."""
        if self.active:
            self.all_reduce_sum(flat)
            flat.mul_(1.0 / self.world_size)
        return flat

    def broadcast_(self, flat: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size > 1:
            if self._host_staged(flat):
                host = flat.cpu()
                dist.broadcast(host, src=src)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src=src)
        return flat

    def barrier(self):
        if self.world_size > 1 or self.force:
            dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.world_size == 1 and not self.force:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def describe(self) -> dict:
        """What the process group actually looks like from this rank (bench.py prints it: the line then says what RCCL
        saw, not what the command line claimed)."""
        info = {"backend": None, "world_size_seen": 1, "rank": self.rank, "local_rank": self.local_rank,
                "env_world_size": self.world_size, "forced_single_rank_group": bool(self.force and self.world_size == 1)}
        if dist.is_available() and dist.is_initialized():
            info.update(backend=dist.get_backend(), world_size_seen=dist.get_world_size(), rank=dist.get_rank())
        if torch.cuda.is_available():
            idx = self.device.index if self.device.type == "cuda" and self.device.index is not None else torch.cuda.current_device()
            info.update(device=f"cuda:{idx}", device_name=torch.cuda.get_device_name(idx), devices_visible=torch.cuda.device_count())
        return info

    def gather_objects(self, obj):
        """Every rank's ``obj`` on every rank (small Python objects: the bench's per-rank records)."""
        if not (self.world_size > 1 or self.force) or not dist.is_initialized():
            return [obj]
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out

    def shutdown(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()
