"""GPU-box tool: which ATen operators one training step of the headline workload still issues (forward and backward), counted by
operator and argument shapes with a TorchDispatchMode - the launches that are not ours."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from phnet_amd.config import make_cfg
from phnet_amd.libs.models.Router4OL import RouterOL
from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
from phnet_amd.optim import FlatAdamW
from phnet_amd.synthetic import make_clip, make_targets

SKIP = ("aten.view", "aten.detach", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.expand", "aten.unsqueeze", "aten.squeeze",
        "aten.slice", "aten.select", "aten.split", "aten.as_strided", "aten.permute", "aten.alias", "aten.empty", "aten.reshape",
        "aten.unbind", "aten.is_", "aten.stride", "aten.size", "aten.sym_", "aten._local_scalar", "aten.lift_fresh", "aten.unfold")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:3]
            self.c[(name, shapes)] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(3407)
    cfg = make_cfg(img_h=320, img_w=800, arch="resnet34")
    model = RouterOL(cfg, Criterion4OL(cfg)).to(dev).train()
    opt, arena = FlatAdamW.for_model(model, lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-4)
    lanes = make_targets(320, 800, 5).to(dev)
    clip = make_clip(320, 800, 5, seed=3407).to(dev)

    def step():
        arena.zero()
        loss = model({"frame": clip, "lanes": lanes}) / 5
        loss.backward()
        opt.step()
    step(); step()
    torch.cuda.synchronize()
    with Log() as log:
        step()
    torch.cuda.synchronize()
    tot = sum(log.c.values())
    print(f"{tot} ATen calls that can launch a kernel in one step")
    for (name, shapes), n in log.c.most_common(70):
        print(f"{n:5d}  {name:40s} {shapes}")


if __name__ == "__main__":
    main()
