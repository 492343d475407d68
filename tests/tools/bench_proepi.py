"""GPU-box probe: what a trunk-sized launch costs outside its K loop (1x1 convolutions with 1 and 4 K steps on the layer-1 map)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from bench_conv import timeit
for ci, r in [(16, 1), (64, 1), (64, 3)]:
    x = torch.randn(5, 80, 200, ci, device="cuda"); w = torch.randn(64, r, r, ci, device="cuda") * 0.05
    t = timeit(lambda: K.conv2d_fwd(x, w, None, 1, r // 2))
    print(f"conv {r}x{r} {ci}->64 on 5x80x200 ({r*r*ci//16} K steps): {t:.1f} us", flush=True)
