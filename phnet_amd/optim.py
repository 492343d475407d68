"""Optimizer over the flat arenas: one HIP launch updates every parameter of the model.

torch.optim.AdamW semantics (reference: libs/utils/optimizer.py:33-35 builds optim.AdamW with weight decay on the
matrices and none on biases / normalisation parameters, set_weight_decay ibid.).  Needs a GradArena built with
flatten_params=True whose parameter list starts with the decayed parameters."""
from typing import Iterable, Sequence, Tuple

import torch

from . import hip_ops as K
from .arena import GradArena, _view_like
from .trunk import weights_changed


def no_decay(name: str, p: torch.Tensor) -> bool:
    """The reference's predicate (libs/utils/optimizer.py:47 set_weight_decay): no weight decay on 1-D parameters and on anything
    NAMED `*.bias` - that includes the two-dimensional LayerNorm([C,P]) biases of the routing gate (`detNet.router.pre_norm.*.bias`,
    `DWNets.*.bias`, shape (64,36)), which a `dim() > 1` test would decay."""
    return p.dim() <= 1 or name.endswith(".bias")


def split_decay(model: torch.nn.Module) -> Tuple[list, list]:
    """(decayed, not decayed) in named_parameters() order = the two parameter groups build_optimizer hands to optim.AdamW
    (libs/utils/optimizer.py:41-55)."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    return [p for n, p in named if not no_decay(n, p)], [p for n, p in named if no_decay(n, p)]


def model_part(param_name: str) -> int:
    """Index into phnet_amd.trunk.PARTS of the part of RouterOL a parameter belongs to (by its state_dict name)."""
    if param_name.startswith("module."):                      # DistributedDataParallel wrapper
        param_name = param_name[len("module."):]
    if not param_name.startswith("backbone."):
        return 0                                               # detNet.* (the lane head) and anything that is not the encoder
    if param_name.startswith("backbone.neck."):
        return 1
    for i, stage in enumerate(("layer4", "layer3", "layer2", "layer1")):
        if f".{stage}." in param_name:
            return 2 + i
    return 6                                                   # conv1 / bn1


class FlatAdamW(torch.optim.Optimizer):
    """A torch.optim.Optimizer (two parameter groups: decayed / not decayed), so the caller step of the reference keeps working
    around it: `CosineAnnealingLR(optimizer, ...)` stepped per iteration (trainOL.py:121-124,228) and
    `GradScaler.step(optimizer)` (trainOL.py:225-227).  The learning rate lives in DEVICE memory (`lr_dev`): `step()` pushes
    `param_groups[0]['lr']` there when it runs eagerly; a hipGraph-captured step reads whatever `sync_lr()` wrote before the
    replay (GraphedTrainStep does that), so a schedule is followed without re-capturing."""

    def __init__(self, arena: GradArena, n_decay: int, lr: float = 1e-3, betas: Sequence[float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, groups=None):
        """groups: (decayed parameters, the rest), each in the order the CHECKPOINT numbers them (torch numbers parameters group by
        group in the order of the lists given to the optimizer; the reference passes named_parameters() order).  The arena may lay
        the same parameters out in any order as long as the decayed ones come first; default: the arena's order."""
        if arena.flat_params is None:
            raise ValueError("FlatAdamW needs GradArena(..., flatten_params=True)")
        self.arena, self.n_decay = arena, int(n_decay)
        self.weight_decay = float(weight_decay)
        betas, eps = (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(arena.flat_params)
        self.exp_avg_sq = torch.zeros_like(arena.flat_params)
        dev = arena.flat_params.device
        self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)     # on the device: graph-capturable
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        decayed, rest, seen = [], [], 0
        for p in arena.params:                                                  # arena order: decayed parameters first
            (decayed if seen < self.n_decay else rest).append(p)
            seen += p.numel()
        if groups is not None:
            if {id(p) for p in groups[0]} != {id(p) for p in decayed} or {id(p) for p in groups[1]} != {id(p) for p in rest}:
                raise ValueError("FlatAdamW: `groups` must hold exactly the arena's decayed / undecayed parameters")
            decayed, rest = list(groups[0]), list(groups[1])
        groups = [{"params": decayed, "weight_decay": self.weight_decay}]
        if rest:
            groups.append({"params": rest, "weight_decay": 0.0})
        super().__init__(groups, dict(lr=float(lr), betas=betas, eps=eps, weight_decay=self.weight_decay))

    @classmethod
    def for_model(cls, model: torch.nn.Module, backward_order: bool = True, **kw):
        """Builds the arena (decayed parameters first) and the optimizer; returns (optimizer, arena).
        backward_order: inside the decayed section the parameters are laid out in the order in which the backward pass
        FINISHES their gradients (lane head, neck, layer4 ... layer1, stem: phnet_amd.trunk.PARTS), and
        `arena.bucket_bounds` / `arena.bucket_of_part` describe 4 contiguous buckets for parallel.BucketReducer: bucket i can
        be all-reduced as soon as the part named in bucket_of_part has reported `stage_done` (the 1-D parameters - biases,
        normalisation affine: 0.3 % of the elements - travel with the last bucket)."""
        model_order = split_decay(model)                      # what the checkpoint's numbering follows
        if not backward_order:
            arena = GradArena(model_order[0] + model_order[1], flatten_params=True)
            return cls(arena, sum(p.numel() for p in model_order[0]), **kw), arena
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        part = {id(p): model_part(n) for n, p in named}
        decay = sorted(model_order[0], key=lambda p: part[id(p)])          # stable: keeps module order inside a part
        rest = sorted(model_order[1], key=lambda p: part[id(p)])
        arena = GradArena(decay + rest, flatten_params=True)
        n_decay = sum(p.numel() for p in decay)
        ends, off = {}, 0
        for p in decay:
            off += p.numel()
            ends[part[id(p)]] = off

        def end_of(*parts):                                   # end offset of the last non-empty part among `parts`
            e = [ends[i] for i in parts if i in ends]
            return max(e) if e else None
        b1 = end_of(0) or 0                                    # lane head
        b2 = end_of(1, 2) or b1                                # neck + layer4
        b3 = end_of(3) or b2                                   # layer3
        arena.bucket_bounds = [0, b1, b2, b3, arena.flat.numel()]
        arena.bucket_of_part = {"head": 0, "layer4": 1, "layer3": 2, "stem": 3}
        return cls(arena, n_decay, groups=model_order, **kw), arena

    def sync_lr(self):
        """Copies param_groups[0]['lr'] (what LR schedulers write) into the device scalar the kernel reads.  Not capturable:
        call it eagerly (before a graph replay)."""
        self.lr_dev.fill_(float(self.param_groups[0]["lr"]))

    def _hyper(self):
        """(lr, betas, eps, weight_decay) as the param_groups hold them NOW (schedulers, load_state_dict and user code write
        there).  The kernel applies ONE decay value to the leading n_decay elements and none to the rest, and one lr / betas /
        eps to everything - the reference's grouping (libs/utils/optimizer.py:41-55); anything else is refused loudly."""
        g0 = self.param_groups[0]
        for g in self.param_groups[1:]:
            if float(g.get("weight_decay", 0.0)) != 0.0:
                raise ValueError("FlatAdamW: only the first parameter group may carry weight decay")
            if float(g["lr"]) != float(g0["lr"]) or tuple(g["betas"]) != tuple(g0["betas"]) or float(g["eps"]) != float(g0["eps"]):
                raise ValueError("FlatAdamW: all parameter groups must share lr / betas / eps")
        return float(g0["lr"]), (float(g0["betas"][0]), float(g0["betas"][1])), float(g0["eps"]), float(g0["weight_decay"])

    @torch.no_grad()
    def step(self, closure=None):
        """One launch over the flat arenas.  Difference to torch.optim.AdamW, documented: a parameter whose gradient is all
        zero (torch: `grad is None`, skipped entirely) still receives its weight decay here - the arena has no notion of
        'no gradient'; on the reference's path that only concerns the transformer on a one-frame clip."""
        if closure is not None:
            raise ValueError("FlatAdamW.step takes no closure")
        lr, betas, eps, wd = self._hyper()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        self.step_count.add_(1)
        weights_changed()                                       # raw-pointer update: invalidates folded eval weights (trunk._folded)
        K.adamw_step(self.arena.flat_params, self.arena.flat, self.exp_avg, self.exp_avg_sq, self.n_decay, self.step_count,
                     lr, betas[0], betas[1], eps, wd, lr_dev=self.lr_dev)

    def zero_grad(self, set_to_none: bool = False):
        """One memset of the gradient arena; the .grad views stay (set_to_none is ignored: the HIP kernels accumulate into them)."""
        self.arena.zero()

    # ---- checkpoints: torch.optim.AdamW's own layout, so the reference's resume path works both ways
    # (trainOL.py:128 `optimizer.load_state_dict(checkpoint['optimizer'])`, :182 `'optimizer': optimizer.state_dict()`) ----
    def _param_slices(self):
        """[(index in torch's numbering, offset, numel, parameter)]: torch numbers parameters group by group in the order of the
        groups' lists (for_model: named_parameters() order, as the reference's build_optimizer); where a parameter lives in the
        arena is looked up by identity, not by position."""
        return [(i, *self.arena.offsets[id(p)], p) for i, p in enumerate(q for g in self.param_groups for q in g["params"])]

    def state_dict(self):
        step = self.step_count.detach().to(torch.float32).reshape(()).cpu()
        state = {}
        for i, off, n, p in self._param_slices():
            state[i] = {"step": step.clone(),
                        "exp_avg": _view_like(self.exp_avg[off:off + n], p).clone(),          # logical shape; channels_last
                        "exp_avg_sq": _view_like(self.exp_avg_sq[off:off + n], p).clone()}   # parameters keep OHWI memory order
        groups, start = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d.setdefault("amsgrad", False)
            d["params"] = list(range(start, start + len(g["params"])))
            start += len(g["params"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:                                   # round-1 private layout (flat moments)
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_count.copy_(sd["step"])
            for g, saved in zip(self.param_groups, sd["param_groups"]):
                g.update({k: v for k, v in saved.items() if k != "params"})
            self._hyper(); self.sync_lr()
            return
        saved_groups = sd["param_groups"]
        if [len(g["params"]) for g in saved_groups] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("FlatAdamW.load_state_dict: parameter groups of the checkpoint do not match this model "
                             f"({[len(g['params']) for g in saved_groups]} vs {[len(g['params']) for g in self.param_groups]})")
        order = [i for g in saved_groups for i in g["params"]]   # checkpoint id of our parameter #k
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        steps = []
        for (k, off, n, p), cid in zip(self._param_slices(), order):
            st = sd["state"].get(cid, sd["state"].get(str(cid)))
            if st is None:                                       # torch keeps no state for parameters that never had a gradient
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdamW.load_state_dict: state {cid} has shape {tuple(st['exp_avg'].shape)}, parameter {tuple(p.shape)}")
            if st.get("max_exp_avg_sq") is not None and saved_groups[0].get("amsgrad"):
                raise ValueError("FlatAdamW: amsgrad checkpoints are not supported")
            _view_like(self.exp_avg[off:off + n], p).copy_(st["exp_avg"])
            _view_like(self.exp_avg_sq[off:off + n], p).copy_(st["exp_avg_sq"])
            steps.append(int(float(st["step"])))
        if steps and min(steps) != max(steps):
            raise ValueError("FlatAdamW keeps ONE step counter; the checkpoint's parameters are at different steps "
                             f"({min(steps)}..{max(steps)})")
        self.step_count.fill_(steps[0] if steps else 0)
        for g, saved in zip(self.param_groups, saved_groups):
            g.update({k: v for k, v in saved.items() if k != "params"})
        self._hyper(); self.sync_lr()
