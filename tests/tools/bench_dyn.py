"""GPU-box probe: device time of the per-anchor products of the dynamic head (forward / backward) on the clip-sized problem
(5 x 240 anchors of 36 points), matrix-pipe kernels (csrc/dyn_mfma.hip) or, with --generic, the LDS / FMA kernels (csrc/dynhead.hip)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib
from tests.tools.bench_conv import timeit


def main():
    if "--generic" in sys.argv:
        assert lib().phnet_tune_dyn_mfma(0) == 0
    if "--per-anchor" in sys.argv:                         # forward with one wavefront per anchor (all three row fragments)
        assert lib().phnet_tune_dyn_mfma(3) == 0
    N, P = 1200, 36
    for k, j in ((64, 128), (128, 64)):
        x = torch.randn(N, P, k, device="cuda")
        w = torch.randn(N, k, j, device="cuda") * k ** -0.5
        gamma, beta = torch.rand(j, device="cuda") + 0.5, torch.randn(j, device="cuda") * 0.1
        y, stats = K.dyn_bmm_ln_relu_fwd(x, w, gamma, beta, 1e-5, True)
        dy = torch.randn_like(y)
        tf = timeit(lambda: K.dyn_bmm_ln_relu_fwd(x, w, gamma, beta, 1e-5, True))
        dg, db = torch.zeros(j, device="cuda"), torch.zeros(j, device="cuda")
        tb = timeit(lambda: K.dyn_bmm_ln_relu_bwd(dy, x, w, y, stats, gamma, 1e-5, True, dg, db, True))
        print(f"dyn {N} anchors x [{P}x{k}]@[{k}x{j}]: forward {tf:.1f} us, backward (+ reduce) {tb:.1f} us", flush=True)


if __name__ == "__main__":
    main()
