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


class FlatAdamW(torch.optim.Optimizer):
    """A torch.optim.Optimizer (two parameter groups: decayed / not decayed), so the caller step of the reference keeps working
    around it: `CosineAnnealingLR(optimizer, ...)` stepped per iteration (trainOL.py:121-124,228) and
    `GradScaler.step(optimizer)` (trainOL.py:225-227).  The learning rate lives in DEVICE memory (`lr_dev`): `step()` pushes
    `param_groups[0]['lr']` there when it runs eagerly; a hipGraph-captured step reads whatever `sync_lr()` wrote before the
    replay (GraphedTrainStep does that), so a schedule is followed without re-capturing."""

    def __init__(self, arena: GradArena, n_decay: int, lr: float = 1e-3, betas: Sequence[float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        if arena.flat_params is None:
            raise ValueError("FlatAdamW needs GradArena(..., flatten_params=True)")
        self.arena, self.n_decay = arena, int(n_decay)
        self.betas, self.eps, self.weight_decay = (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(arena.flat_params)
        self.exp_avg_sq = torch.zeros_like(arena.flat_params)
        dev = arena.flat_params.device
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)     # on the device: graph-capturable
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        decayed, rest, seen = [], [], 0
        for p in arena.params:                                                  # arena order: decayed parameters first
            (decayed if seen < self.n_decay else rest).append(p)
            seen += p.numel()
        groups = [{"params": decayed, "weight_decay": self.weight_decay}]
        if rest:
            groups.append({"params": rest, "weight_decay": 0.0})
        super().__init__(groups, dict(lr=float(lr), betas=self.betas, eps=self.eps, weight_decay=self.weight_decay))

    @classmethod
    def for_model(cls, model: torch.nn.Module, **kw):
        """Builds the arena (decayed parameters first) and the optimizer; returns (optimizer, arena)."""
        decay, no_decay = split_decay(model.parameters())
        arena = GradArena(decay + no_decay, flatten_params=True)
        return cls(arena, sum(p.numel() for p in decay), **kw), arena

    def sync_lr(self):
        """Copies param_groups[0]['lr'] (what LR schedulers write) into the device scalar the kernel reads.  Not capturable:
        call it eagerly (before a graph replay)."""
        self.lr_dev.fill_(float(self.param_groups[0]["lr"]))

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise ValueError("FlatAdamW.step takes no closure")
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        self.step_count.add_(1)
        K.adamw_step(self.arena.flat_params, self.arena.flat, self.exp_avg, self.exp_avg_sq, self.n_decay, self.step_count,
                     float(self.param_groups[0]["lr"]), self.betas[0], self.betas[1], self.eps, self.weight_decay, lr_dev=self.lr_dev)

    def zero_grad(self, set_to_none: bool = False):
        """One memset of the gradient arena; the .grad views stay (set_to_none is ignored: the HIP kernels accumulate into them)."""
        self.arena.zero()

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_count.copy_(sd["step"])
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        self.sync_lr()
