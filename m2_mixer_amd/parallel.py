"""Data parallelism: one process per GPU, parameters replicated, ONE gradient all-reduce per step.

The reference never calls torch.distributed itself: `pl.Trainer(accelerator='gpu', devices=-1)` (run.py:59-74)
makes Lightning wrap the module in DistributedDataParallel, i.e. bucketed all-reduce(SUM)/world of the
gradients over NCCL, plus a parameter broadcast from rank 0 at start.  The samples are
independent through the whole network and every loss is a batch mean, so that is the only exchange
step the path has.  Here the gradients already live in one flat fp32 buffer (engine.py), so the
exchange is a single RCCL all-reduce over xGMI -- one collective, no bucketing, no per-tensor hooks; the
1/world factor is folded into the Adam kernel's grad_scale.

Works with any backend: "nccl" (= RCCL on ROCm) on the GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world_size); a no-op single-process setup when WORLD_SIZE is unset/1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # M2M_DIST_BACKEND=gloo lets several ranks share ONE GPU (rehearsals on a single-GPU box; RCCL needs a device per rank)
            backend = os.environ.get("M2M_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradSync:
    """All-reduce (SUM) of the flat gradient buffer; returns the scale (1/world) the optimizer applies.

    compress='bf16' halves the bytes on the wire (16.7 MB instead of 33.4 MB for M2-Mixer-B): the
    gradient is rounded to bf16, summed in bf16 by RCCL and widened back.  Default is fp32 (exact DDP
    semantics).  widen=False leaves the sum in `reduced_bf16` for an optimizer that reads bf16 gradients
    itself (engine.optimizer_step(scale, grad_bf16): one 50 MB pass less per step); flat_grad then still
    holds this rank's local gradient."""

    def __init__(self, group=None, compress: Optional[str] = None, widen: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.compress = compress
        self.widen = widen or compress != "bf16"
        self._buf = None

    @property
    def reduced_bf16(self) -> Optional[torch.Tensor]:
        """The all-reduced gradient in bf16 when it was NOT widened back into flat_grad, else None."""
        return self._buf if (self.compress == "bf16" and not self.widen and self.world > 1) else None

    def __call__(self, flat_grad: torch.Tensor) -> float:
        if self.world == 1:
            return 1.0
        if self.compress == "bf16":
            if self._buf is None or self._buf.numel() != flat_grad.numel():
                self._buf = torch.empty_like(flat_grad, dtype=torch.bfloat16)
            self._buf.copy_(flat_grad)
            dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
            if self.widen:
                flat_grad.copy_(self._buf)
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world


class PipelinedGradSync:
    """The exchange in K contiguous chunks, pipelined with the optimizer (round 4).

    GradSync is `graph | all_reduce(33 MB) | graph`: nothing overlaps the exchange, and Adam + the operand re-pack (~62 us on
    M2-Mixer-B) wait for its last byte.  Here the flat gradient is cut at the engine's parameter segments (one per tower: the
    chunks the engine can update AND re-pack independently); start() enqueues one asynchronous all-reduce per chunk, in
    order, on a communication stream behind the backward; wait(k) makes the compute stream wait for chunk k only, so chunk
    k's Adam + re-pack run while chunks k + 1 .. K - 1 are still on the wire:

        comm stream     | AR 0 | AR 1 | AR 2 |
        compute stream         | Adam + pack 0 | Adam + pack 1 | Adam + pack 2 |      exposed: AR total + the LAST chunk's update

    Same arithmetic as GradSync (fp32 sum over ranks, 1/world folded into Adam): the chunks partition the buffer, every
    element is reduced exactly once.  Expected exposed time at 8 GPUs (UNMEASURED on hardware: one-GPU boxes): the last
    chunk's ~20 us of update instead of ~62 us.  fp32 exchange only (the bf16-compressed form keeps GradSync).
    world == 1 (tests): start / wait are no-ops, the engine still runs its chunked update path."""

    def __init__(self, group=None, comm_stream: Optional["torch.cuda.Stream"] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pipelined = True
        self._comm = comm_stream
        self._works = []

    def start(self, flat_grad: torch.Tensor, bounds) -> float:
        """Enqueue the chunks' all-reduces (bounds: [(lo, hi), ...] ascending, a partition of the buffer); returns 1 / world."""
        covered = 0
        for lo, hi in bounds:
            if lo != covered or hi <= lo:
                raise ValueError("PipelinedGradSync: the chunks must partition the flat gradient in ascending order")
            covered = hi
        if covered != flat_grad.numel():
            raise ValueError("PipelinedGradSync: the chunks do not cover the flat gradient")
        self._works = []
        if self.world == 1:
            return 1.0
        if flat_grad.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=flat_grad.device)
            self._comm.wait_stream(torch.cuda.current_stream(flat_grad.device))      # the backward wrote the gradient
            with torch.cuda.stream(self._comm):
                for lo, hi in bounds:
                    self._works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for lo, hi in bounds:
                self._works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return 1.0 / self.world

    def wait(self, k: int) -> None:
        """Chunk k is reduced (stream-ordered for RCCL: the CURRENT stream waits; host-blocking for gloo)."""
        if self._works:
            self._works[k].wait()

    def __call__(self, flat_grad: torch.Tensor) -> float:
        """The whole buffer as one chunk (drop-in for GradSync where a caller does not pipeline)."""
        scale = self.start(flat_grad, [(0, flat_grad.numel())])
        self.wait(0)
        return scale


def broadcast_parameters(flat_param: torch.Tensor, src: int = 0, group=None) -> None:
    """DDP's initial parameter broadcast from rank 0."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_param, src=src, group=group)


def shard_batch_seed(base_seed: int, rank: int) -> int:
    """Synthetic-data seed of a rank (default_rng(1234 + rank))."""
    return base_seed + rank


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (bench timing contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
