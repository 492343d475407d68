"""Optimizer over the flat arenas: one HIP launch updates every parameter of the model.

torch.optim.AdamW semantics (reference: libs/utils/optimizer.py:33-35 builds optim.AdamW with weight decay on the
matrices and none on biases / normalisation parameters, set_weight_decay ibid.).  Needs a GradArena built with
flatten_params=True whose parameter list starts with the decayed parameters."""
from typing import Iterable, Sequence, Tuple

import torch

from . import hip_ops as K
from .arena import GradArena


def split_decay(params: Iterable[torch.nn.Parameter]) -> Tuple[list, list]:
    """(decayed, not decayed): 1-D parameters - biases, normalisation affine - carry no weight decay."""
    params = [p for p in params if p.requires_grad]
    return [p for p in params if p.dim() > 1], [p for p in params if p.dim() <= 1]


class FlatAdamW:
    def __init__(self, arena: GradArena, n_decay: int, lr: float = 1e-3, betas: Sequence[float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        if arena.flat_params is None:
            raise ValueError("FlatAdamW needs GradArena(..., flatten_params=True)")
        self.arena, self.n_decay = arena, int(n_decay)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(arena.flat_params)
        self.exp_avg_sq = torch.zeros_like(arena.flat_params)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=arena.flat_params.device)     # on the device: graph-capturable
        self.param_groups = [{"lr": self.lr, "weight_decay": self.weight_decay}]                   # for LR schedulers / logging

    @classmethod
    def for_model(cls, model: torch.nn.Module, **kw):
        """Builds the arena (decayed parameters first) and the optimizer; returns (optimizer, arena)."""
        decay, no_decay = split_decay(model.parameters())
        arena = GradArena(decay + no_decay, flatten_params=True)
        return cls(arena, sum(p.numel() for p in decay), **kw), arena

    @torch.no_grad()
    def step(self):
        self.step_count.add_(1)
        K.adamw_step(self.arena.flat_params, self.arena.flat, self.exp_avg, self.exp_avg_sq, self.n_decay, self.step_count,
                     float(self.param_groups[0]["lr"]), self.betas[0], self.betas[1], self.eps, self.weight_decay)

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero()
