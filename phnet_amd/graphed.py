"""Whole-step hipGraph capture: forward + loss + backward + optimizer of one clip recorded once and replayed.

The per-clip step is ~20 000 small launches (15 serial stage iterations); replaying them from a hipGraph removes the
host from the loop.  Works because the training path is shape-static and sync-free (device-side label assignment,
fixed-capacity memory tokens).  Single-process only (the DDP path stays eager)."""
from typing import Callable, Optional

import torch


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, frames: torch.Tensor, lanes: torch.Tensor,
                 loss_divisor: Optional[float] = None, warmup: int = 3, arena=None):
        self.model, self.optimizer, self.arena = model, optimizer, arena
        self.frames, self.lanes = frames.clone(), lanes.clone()
        self.div = float(loss_divisor if loss_divisor is not None else frames.shape[0])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        if arena is None:
            optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._step(zero=arena is not None)
        torch.cuda.synchronize()

    def _step(self, zero: bool = True):
        if zero:
            if self.arena is not None:
                self.arena.zero()                       # one memset; .grad views stay alive
            else:
                self.optimizer.zero_grad(set_to_none=True)
        loss = self.model({"frame": self.frames, "lanes": self.lanes}) / self.div
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, frames: torch.Tensor, lanes: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.frames.copy_(frames, non_blocking=True)
        if lanes is not None:
            self.lanes.copy_(lanes, non_blocking=True)
        self.graph.replay()
        return self.loss
