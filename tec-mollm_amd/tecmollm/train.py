"""One training step with the reference's semantics (train.py:57-112, :358-372):
forward -> HuberLoss(delta=1) -> backward -> (data-parallel mean of gradients) -> clip_grad_norm_(1.0)
-> AdamW(lr 1e-4, wd 1e-2) -> CosineAnnealingWarmRestarts(T_0=10, T_mult=2, eta_min=1e-7).

MI355X-first differences from the reference's DDP wrapper (train.py:353-354):
  * every trainable gradient lives in ONE flat fp32 buffer (3 081 996 values = 12.33 MB at the default
    config); `p.grad` are views into it, so autograd accumulates in place and the data-parallel exchange
    is a single RCCL all-reduce over xGMI per optimizer step -- not one per micro-batch (the reference
    has no `no_sync()` around its accumulation loop);
  * the 1/world mean, clip_grad_norm_(1.0), AdamW and zero_grad are ONE fused pass (two launches) over the
    flat parameter/gradient/moment buffers (tecmollm/optim.py -> tecm_adamw_clip_step), without a host sync;
    the cosine-warm-restart learning rate is a closed-form host float.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def flatten_grads(params: Iterable[torch.nn.Parameter], tail: int = 0) -> torch.Tensor:
    """Allocate one flat gradient buffer (plus `tail` control slots) and make every p.grad a view into it."""
    params = list(params)
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total + tail, device=params[0].device, dtype=params[0].dtype)
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].view_as(p)
        o += n
    return flat


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None) -> int:
    """What the DDP constructor does once (train.py:354): every parameter and buffer of rank `src` replaces the
    other ranks' copy, so the ranks start identical whatever their seeds were.  Returns the bytes broadcast."""
    n = 0
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src, group=group)
            n += t.numel() * t.element_size()
    return n


def allreduce_mean_(flat: torch.Tensor, world_size: int, group=None) -> torch.Tensor:
    """The path's only collective: mean of the flat gradient over the data-parallel ranks (train.py:354)."""
    if world_size > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world_size)
    return flat


def clip_flat_(flat: torch.Tensor, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ semantics on the flat buffer; returns the (device) total norm."""
    total = torch.linalg.vector_norm(flat)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    flat.mul_(coef)
    return total


class TrainStep:
    """One optimizer step of train.py:57-112.  optimizer="native" (default): FlatAdamW + CosineWarmRestarts --
    the mean over ranks, clip, AdamW and zero_grad are one fused pass over the flat buffers (HIP only, raises
    for CPU parameters).  optimizer="torch": torch.optim.AdamW + torch's scheduler on the same flat gradient
    views -- kept for the world_size-2 gloo plumbing tests, which run host logic on CPU stand-in modules."""

    def __init__(self, model: torch.nn.Module, lr: float = 1e-4, weight_decay: float = 1e-2, max_norm: float = 1.0,
                 accumulation_steps: int = 1, world_size: int = 1, group=None, fused_huber: bool = True,
                 optimizer: str = "native", broadcast_init: bool = True, check_divergence: bool = True,
                 time_collective: bool = False):
        if optimizer not in ("native", "torch"):
            raise ValueError("optimizer must be 'native' or 'torch'")
        self.model = model
        self.rank = dist.get_rank(group) if world_size > 1 else 0
        if world_size > 1 and dist.get_world_size(group) != world_size:
            raise ValueError(f"world_size={world_size} but the process group has {dist.get_world_size(group)} ranks")
        if world_size > 1 and broadcast_init:
            broadcast_parameters(model, 0, group)              # before the flat buffers copy the parameters
        # rank-divergence check: 2 fp32 slots per rank ride behind the gradient in the SAME all-reduce (rank r fills
        # its own pair with the checksum of its parameters, zeros elsewhere; adding zeros is exact, so after the sum
        # every rank holds every checksum) -- no extra collective, no host synchronisation
        self.check_divergence = bool(check_divergence) and world_size > 1
        tail = 2 * world_size if self.check_divergence else 0
        self.params: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        self.max_norm = max_norm
        self.accumulation_steps = accumulation_steps
        self.world_size = world_size
        self.group = group
        self.fused_huber = fused_huber
        self.native = optimizer == "native"
        if self.native:
            from .optim import CosineWarmRestarts, FlatAdamW
            self.optimizer = FlatAdamW(self.params, lr=lr, weight_decay=weight_decay, grad_tail=tail)
            self.flat_grad = self.optimizer.flat_grad
            self.flat_grad_ext = self.optimizer.flat_grad_ext
            self.scheduler = CosineWarmRestarts(lr, T_0=10, T_mult=2, eta_min=1e-7)
        else:
            self.flat_grad_ext = flatten_grads(self.params, tail)
            self.flat_grad = self.flat_grad_ext[:self.flat_grad_ext.numel() - tail]
            self.optimizer = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay,
                                               fused=self.params[0].is_cuda)
            self.scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(self.optimizer, T_0=10, T_mult=2,
                                                                                 eta_min=1e-7)
        self._micro = 0
        # time_collective: bracket the one collective of the step with events on the stream it is enqueued behind
        # (RCCL runs it on its own stream; the current stream waits for it, so the bracket covers it) -- bench.py
        # reports it as dist.allreduce_ms so that a scaling curve can separate communication from compute
        self.time_collective = bool(time_collective) and world_size > 1
        self._coll_events: list = []
        self._csum_ws = None                           # 256 doubles of scratch for tecm_checksum_tail (allocated on first use)
        self._coll_host_s: list = []
        self._graphs: dict = {}                        # step_graphed: (shapes, mode) -> recorded micro-batch
        self._seed_word: Optional[torch.Tensor] = None

    def collective_ms(self) -> List[float]:
        """Duration of every timed all-reduce so far (ms): device events for GPU tensors, host clock for CPU ones."""
        if self._coll_events:
            torch.cuda.synchronize()
            return [a.elapsed_time(b) for a, b in self._coll_events]
        return [1e3 * t for t in self._coll_host_s]

    def reset_collective_timing(self) -> None:
        self._coll_events.clear()
        self._coll_host_s.clear()

    def _loss(self, out: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if self.fused_huber:
            from .functions import HuberFn
            return HuberFn.apply(out, y, 1.0)
        return torch.nn.functional.huber_loss(out, y, delta=1.0)

    def current_lr(self) -> float:
        return self.scheduler.lr if self.native else self.optimizer.param_groups[0]["lr"]

    def step(self, x, time_features, edge_index, edge_weight, y) -> torch.Tensor:
        """One micro-batch; the optimizer fires every `accumulation_steps` calls.  Returns the (device) loss."""
        loss = self._micro_batch(x, time_features, edge_index, edge_weight, y)
        self._micro += 1                               # micro-batches since the last optimizer step (train.py:92: (i+1) % acc
        if self._micro >= self.accumulation_steps:     # counted per epoch there: `finish_accumulation` restarts the count)
            self.finish_accumulation()
        return loss

    def _micro_batch(self, x, time_features, edge_index, edge_weight, y) -> torch.Tensor:
        """Forward, loss, backward and the hand-over of the gradients into the flat buffer (everything `step_graphed` records)."""
        if self.native:
            self.optimizer.detach_grads()              # backward hands its gradient tensors over; one add for all of them
        out = self.model(x, time_features, edge_index, edge_weight)
        if self.native and self.fused_huber and out.dim() == 4 and out.shape[3] == 1 and out.shape == y.shape:
            # loss and d loss / d out (already divided by accumulation_steps, train.py:78) from ONE strided kernel pair, fed
            # to autograd as the output gradient: no contiguous copies of the permuted prediction / the target, no ones_like,
            # no divide, no multiply by the upstream scalar
            from . import ops
            loss, dout = ops.huber_fwd_bwd_strided(out.detach(), y, 1.0, 1.0 / self.accumulation_steps)
            out.backward(dout)
            loss = loss[0]
        else:
            loss = self._loss(out, y)
            (loss / self.accumulation_steps).backward()
        if self.native:
            self.optimizer.absorb_grads()
        return loss.detach()

    # ------------------------------------------------------------------ recorded step (small batches)
    def step_graphed(self, x, time_features, edge_index, edge_weight, y) -> torch.Tensor:
        """`step` with the micro-batch's ~200 launches (forward, Huber loss, backward, gradient hand-over) recorded ONCE as a
        hipGraph and replayed: at the reference's per-GPU batch of 2 (scripts/train_2gpu.sh:4-12) the step is short enough for
        the host's launch rate to show.  The first call with a new (shapes, mode) runs eagerly (it is also the
        warm-up that fills every host-side cache), the second records and replays, later ones copy the batch into the
        recorded input tensors and replay.  What changes from replay to replay lives on the device: the dropout masks draw
        from recorded seeds PLUS one device word (`TecmDrop::seed_dev`) that the graph's first node advances.  The data-
        parallel exchange, clip + AdamW and the scheduler stay outside the graph (two launches with host scalars).
        Returns the loss tensor of the recorded step -- the SAME tensor every call; read it before the next one."""
        if not (self.native and x.is_cuda):
            raise RuntimeError("step_graphed needs the native optimizer and CUDA tensors")
        from . import devcheck, ops
        from ._lib import check, lib, stream_ptr
        tf_uniform = time_features.dim() == 4 and time_features.stride(2) == 0 and time_features.shape[2] > 1
        key = (tuple(x.shape), tuple(time_features.shape), tf_uniform, tuple(y.shape), str(y.dtype), self.model.training,
               torch.is_autocast_enabled("cuda"), str(torch.get_autocast_dtype("cuda")), self.accumulation_steps,
               id(edge_index))
        rec = self._graphs.get(key)
        if rec is None:                                # first sight of this configuration: a plain step, and the warm-up
            self._graphs[key] = False
            return self.step(x, time_features, edge_index, edge_weight, y)
        if rec is False:
            if self._seed_word is None:
                self._seed_word = torch.zeros(1, device=x.device, dtype=torch.int64)
            sx = x.detach().clone()
            sy = y.detach().clone()
            if tf_uniform:
                tf_base = time_features[:, :, :1, :].detach().clone()
                stf = tf_base.expand_as(time_features)
            else:
                tf_base = stf = time_features.detach().clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            prev = ops.SEED_WORD
            ops.SEED_WORD = self._seed_word
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    # a fresh word for every replay: an odd 64-bit increment walks all 2^64 values
                    check(lib().tecm_seed_advance(self._seed_word.data_ptr(), 0x9E3779B97F4A7C15, stream_ptr()),
                          "tecm_seed_advance")
                    loss = self._micro_batch(sx, stf, edge_index, edge_weight, sy)
            finally:
                ops.SEED_WORD = prev
            rec = self._graphs[key] = (g, sx, tf_base, sy, loss, edge_index)
        else:
            g, sx, tf_base, sy, loss, _ = rec
            sx.copy_(x, non_blocking=True)
            sy.copy_(y, non_blocking=True)
            tf_base.copy_(time_features[:, :, :1, :] if tf_uniform else time_features, non_blocking=True)
        rec[0].replay()
        errs = devcheck.error_word(x.device)           # the spatial kernel's index check: read back as an eager step does
        errs.poll()
        errs.post()
        self._micro += 1
        if self._micro >= self.accumulation_steps:
            self.finish_accumulation()
        return rec[4]

    # ------------------------------------------------------------------ data-parallel exchange
    def _param_checksum(self) -> torch.Tensor:
        if self.native:
            return self.optimizer.flat_param.sum(dtype=torch.float64)
        return torch.stack([p.detach().sum(dtype=torch.float64) for p in self.params]).sum()

    def _allreduce_grads(self) -> None:
        """The path's ONE collective per optimizer step (train.py:354 semantics): SUM of the flat gradient, with the
        per-rank parameter checksums in its tail."""
        if self.world_size == 1:
            return
        if self.time_collective:
            import time
            if self.flat_grad.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._allreduce_grads_untimed()
                e1.record()
                self._coll_events.append((e0, e1))
            else:
                t0 = time.perf_counter()
                self._allreduce_grads_untimed()
                self._coll_host_s.append(time.perf_counter() - t0)
            return
        self._allreduce_grads_untimed()

    def _allreduce_grads_untimed(self) -> None:
        if not self.check_divergence:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            return
        n = self.flat_grad.numel()
        tail = self.flat_grad_ext[n:]
        if self.native and tail.is_cuda:
            # three launches on the step's stream instead of eleven torch-dispatched ones (f64 sum, casts, zero_, stack, slice
            # copy | compare, any, cast, multiply, bitwise_or_): csrc/trainstep.hip
            from . import devcheck
            from ._lib import check, lib, stream_ptr
            if self._csum_ws is None:
                self._csum_ws = torch.empty(256, device=tail.device, dtype=torch.float64)
            fp = self.optimizer.flat_param
            check(lib().tecm_checksum_tail(fp.data_ptr(), fp.numel(), tail.data_ptr(), self.world_size, self.rank,
                                           self._csum_ws.data_ptr(), stream_ptr()), "tecm_checksum_tail")
            dist.all_reduce(self.flat_grad_ext, op=dist.ReduceOp.SUM, group=self.group)
            errs = devcheck.error_word(tail.device)
            errs.poll()
            check(lib().tecm_checksum_verify(tail.data_ptr(), self.world_size, errs.ptr(), devcheck.RANKS_DIVERGED,
                                             stream_ptr()), "tecm_checksum_verify")
            errs.post()
            return
        c = self._param_checksum()
        hi = c.float()
        lo = (c - hi.double()).float()
        tail.zero_()
        tail[2 * self.rank:2 * self.rank + 2] = torch.stack([hi, lo])
        dist.all_reduce(self.flat_grad_ext, op=dist.ReduceOp.SUM, group=self.group)
        pairs = tail.view(self.world_size, 2)
        diverged = (pairs != pairs[0]).any()
        if tail.is_cuda:
            from . import devcheck
            errs = devcheck.error_word(tail.device)
            errs.poll()
            errs.word.bitwise_or_(diverged.to(torch.int32) * devcheck.RANKS_DIVERGED)
            errs.post()
        elif bool(diverged):
            raise RuntimeError(f"data-parallel ranks no longer hold identical parameters: checksums {pairs.tolist()}")

    def finish_accumulation(self) -> None:
        """The boundary of an accumulation cycle (also train.py:117-126 for a trailing partial cycle):
        all-reduce, clip, AdamW, zero_grad, scheduler.  A call with nothing accumulated is a no-op (the reference
        only flushes a trailing PARTIAL cycle)."""
        if self._micro == 0:
            return
        self._micro = 0
        if self.native:
            self._allreduce_grads()
            self.optimizer.step(lr=self.scheduler.lr, max_norm=self.max_norm, grad_scale=1.0 / self.world_size,
                                zero_grad=True)
        else:
            self._allreduce_grads()
            if self.world_size > 1:
                self.flat_grad.div_(self.world_size)
            clip_flat_(self.flat_grad, self.max_norm)
            self.optimizer.step()
            self.flat_grad_ext.zero_()
        self.scheduler.step()
