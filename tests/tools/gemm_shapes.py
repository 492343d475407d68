"""GPU-box tool: every GEMM-shaped launch of ONE training step of the headline workload, grouped by (kernel symbol, problem
shape), each group re-issued back to back in a hipGraph and timed with HIP events - which problem shapes the step's GEMM time
is made of, and which of them run far below the rate of their kernel (tile / split plan candidates).
usage: python tests/tools/gemm_shapes.py [--mma f32] [--top 40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from phnet_amd import hip_ops
from phnet_amd.config import make_cfg
from phnet_amd.libs.models.Router4OL import RouterOL
from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
from phnet_amd.optim import FlatAdamW
from phnet_amd.synthetic import make_clip, make_targets


def main():
    mma = sys.argv[sys.argv.index("--mma") + 1] if "--mma" in sys.argv else "bf16x3"
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    hip_ops.set_mma_mode(mma)
    torch.manual_seed(3407)
    dev = torch.device("cuda", 0)
    cfg = make_cfg(img_h=320, img_w=800, arch="resnet34")
    model = RouterOL(cfg, Criterion4OL(cfg)).to(dev).train()
    opt, arena = FlatAdamW.for_model(model, lr=5e-4, betas=(0.9, 0.999), weight_decay=5e-4)
    T = 5
    lanes = make_targets(320, 800, T).to(dev)
    clip = make_clip(320, 800, T, seed=3407).to(dev)

    def step():
        arena.zero()
        loss = model({"frame": clip, "lanes": lanes}) / T
        loss.backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    hip_ops.TIMER = []
    step()
    torch.cuda.synchronize()
    records, hip_ops.TIMER = hip_ops.TIMER, None
    groups = {}
    for rec in records:
        groups.setdefault((rec[0], rec[6]), []).append(rec)
    side = torch.cuda.Stream()
    rows = []
    for (sym, shape), recs in groups.items():
        reps = max(1, 12 // len(recs))                 # at least a dozen launches per replay: a one-kernel graph times the replay, not the kernel
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                for rec in recs:
                    rec[5]()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 / reps * 1e3
        fl = sum(r[2] for r in recs)
        rows.append((us, len(recs), sym, shape, recs[0][1], fl))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"{len(records)} GEMM launches per step, {tot / 1e3:.2f} ms when re-issued group by group ({mma})")
    print(f"{'ms/step':>8} {'n':>4} {'us each':>8} {'TF/s':>7} {'split':>5}  shape (kind, rows, cols, depth, taps)   kernel")
    for us, n, sym, shape, sp, fl in rows[:top]:
        print(f"{us / 1e3:8.3f} {n:4d} {us / n:8.1f} {fl / us / 1e6:7.1f} {sp:5d}  {str(shape):44s} {sym[:60]}")


if __name__ == "__main__":
    main()
