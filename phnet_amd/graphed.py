"""Whole-step hipGraph capture: forward + loss + backward + optimizer of one clip recorded once and replayed.

The per-clip step is ~20 000 small launches (15 serial stage iterations); replaying them from a hipGraph removes the
host from the loop.  Works because the training path is shape-static and sync-free (device-side label assignment,
fixed-capacity memory tokens).  With data parallelism the RCCL collectives - SyncBatchNorm exchanges (device-resident
counts: nothing is read on the host), gradient buckets overlapped with the trunk's backward - are captured in the same
graph.  Those collectives are raw `ncclAllReduce` calls on our own streams (phnet_amd/rccl.py), never torch.distributed
Work objects: a torch collective under capture pulls the process group's internal stream into the capture and the group's
watchdog thread dies (hipErrorCapturedEvent) on its next poll of any eager Work it has not retired yet - rccl.py has the
mechanism; phnet_amd.parallel refuses torch collectives under capture.

Capture caveat (PyTorch, not ours): autograd graphs of earlier EAGER steps on the default stream must be dead before a
step is captured - a loss / gradient tensor that is still referenced keeps AccumulateGrad nodes bound to the default
stream and the capture faults.  Build the graphed step first, or drop those references before building it."""
from typing import Callable, Optional

import torch

from .trunk import weights_changed


def data_parallel_step(model, arena, reducer, optimizer, frames, lanes, loss_divisor: float, stage_done=None):
    """One data-parallel training step, eagerly or under hipGraph capture:
    zero | trunk forward (staged, SyncBatchNorm exchanges inside) + lane head + loss | head backward | bucket 0 out |
    trunk backward, further buckets as its stages finish | wait | optimizer.  `loss_divisor` includes the world size: the
    buckets are SUM-reduced."""
    enc = model.backbone

    def done(part: str):
        i = arena.bucket_of_part.get(part)
        if i is not None:
            reducer.issue(i)
        if stage_done is not None:
            stage_done(part)
    arena.zero()
    enc.staged = True
    try:
        loss = model({"frame": frames, "lanes": lanes}) / loss_divisor
        loss.backward()
        enc.finish_backward(done)
    finally:
        enc.staged = False
    reducer.finish()
    optimizer.step()
    return loss.detach()


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, frames: torch.Tensor, lanes: torch.Tensor,
                 loss_divisor: Optional[float] = None, warmup: int = 3, arena=None, between: Optional[Callable[[], None]] = None,
                 reducer=None):
        """reducer (parallel.BucketReducer over `arena`): data-parallel step - staged trunk (phnet_amd/trunk.py), gradient
        buckets all-reduced while the trunk's backward still runs, SyncBatchNorm exchanges where the model has
        nn.SyncBatchNorm containers; the whole step, collectives included, is ONE hipGraph (needs a backend whose
        collectives can be stream-captured: RCCL; with gloo run data_parallel_step eagerly).  loss_divisor should then
        include the world size (the buckets are SUM-reduced).
        between (older form): a callable run eagerly between backward and the optimizer step; the step is recorded as two
        graphs (forward+backward | optimizer) around it."""
        self.model, self.optimizer, self.arena, self.between, self.reducer = model, optimizer, arena, between, reducer
        self.frames, self.lanes = frames.clone(), lanes.clone()
        self.div = float(loss_divisor if loss_divisor is not None else frames.shape[0])
        if reducer is not None:
            import torch.distributed as dist
            from . import parallel, rccl
            if parallel.active() and dist.get_backend() == "nccl" and rccl.installed() is None:
                rccl.install()                              # collective (every rank builds its step at the same point); eager
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        if arena is None:
            optimizer.zero_grad(set_to_none=True)
        if reducer is not None:
            if arena is None or not hasattr(arena, "bucket_of_part"):
                raise ValueError("GraphedTrainStep(reducer=...) needs the arena of FlatAdamW.for_model(backward_order=True)")
            # ONE graph for the whole data-parallel step, the RCCL collectives captured inside it (the asynchronous bucket
            # all-reduces become parallel branches of the graph next to the trunk's backward).  thread_local: other threads
            # of the process (torch.distributed's watchdogs polling their own, never-capturing streams) may call HIP meanwhile.
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = self._step()
        elif between is None:
            with torch.cuda.graph(self.graph):
                self.loss = self._step(zero=arena is not None)
        else:
            # data-parallel: the RCCL watchdog thread of torch.distributed polls events while we capture; thread-local
            # capture mode keeps its (unrelated) calls from invalidating the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = self._fwd_bwd(zero=arena is not None)
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), capture_error_mode="thread_local"):
                self.optimizer.step()
        torch.cuda.synchronize()

    def _fwd_bwd(self, zero: bool = True):
        if zero:
            if self.arena is not None:
                self.arena.zero()                       # one memset; .grad views stay alive
            else:
                self.optimizer.zero_grad(set_to_none=True)
        loss = self.model({"frame": self.frames, "lanes": self.lanes}) / self.div
        loss.backward()
        return loss.detach()

    def _step(self, zero: bool = True):
        if self.reducer is not None:
            return data_parallel_step(self.model, self.arena, self.reducer, self.optimizer, self.frames, self.lanes, self.div)
        loss = self._fwd_bwd(zero)
        if self.between is not None:
            self.between()
        self.optimizer.step()
        return loss

    def __call__(self, frames: torch.Tensor, lanes: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.frames.copy_(frames, non_blocking=True)
        if lanes is not None:
            self.lanes.copy_(lanes, non_blocking=True)
        if hasattr(self.optimizer, "sync_lr"):
            self.optimizer.sync_lr()                    # LR schedule -> the device scalar the captured AdamW launch reads
        weights_changed()                               # the replay updates weights / running statistics behind torch's version counters
        self.graph.replay()
        if self.graph_opt is not None:
            self.between()
            self.graph_opt.replay()
        return self.loss


class GraphedInference:
    """Eval forward of a fixed-length clip (backbone + lane head + fused decode/NMS for every frame) captured in one
    hipGraph; `__call__` copies the frames in, replays, and returns the device-resident (kept_rows, num, anchors).
    frames [T,3,H,W]: one clip (RouterOL.infer_device); frames [B,T,3,H,W]: B clips per replay with the lane head
    batched across the clips (RouterOL.infer_clips_device)."""

    def __init__(self, model: torch.nn.Module, frames: torch.Tensor, warmup: int = 2):
        self.model = model.eval()
        self.frames = frames.clone()
        infer = model.infer_clips_device if frames.dim() == 5 else model.infer_device
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                infer(self.frames)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = infer(self.frames)
        torch.cuda.synchronize()

    def __call__(self, frames: torch.Tensor):
        self.frames.copy_(frames, non_blocking=True)
        self.graph.replay()
        return self.out
