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

    def shutdown(self):
        if self.owns_group and dist.is_initialized():
            dist.destroy_process_group()
