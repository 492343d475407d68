"""How far are the hand-written fp32-MFMA GEMMs from the vendor library on the hot shapes?  (reference point only:
rocBLAS / hipBLASLt via torch.mm is NOT on the product path.)  Prints TF/s for fwd (x @ w.T), dgrad (dy @ w) and
wgrad (dy.T @ x) of every shape, ours vs torch.mm, graph-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from tests.tools.bench_conv import timeit

SHAPES = [("hyper 1024->8192", 240, 1024, 8192), ("hyper 4608->1024", 240, 4608, 1024), ("fold 64->8192", 240, 64, 8192),
          ("gate 2304->576", 240, 2304, 576), ("out 2304->64", 240, 2304, 64), ("tower 192 chain", 240, 192, 192),
          ("qkv 128->384", 240, 128, 384), ("ffn 128->256", 240, 128, 256),
          ("trunk-like L1 1x1", 80000, 576, 64), ("trunk-like L2", 20000, 1152, 128), ("trunk-like L3", 5000, 2304, 256),
          ("trunk-like L4", 1250, 4608, 512)]


def main():
    for name, m, k, n in SHAPES:
        x, w, dy = torch.randn(m, k, device="cuda"), torch.randn(n, k, device="cuda") * 0.05, torch.randn(m, n, device="cuda")
        wt = w.t().contiguous()
        fl = 2.0 * m * k * n
        row = [f"{name:22s} M={m:6d} K={k:5d} N={n:5d}"]
        for tag, ours, ref in (("fwd", lambda: K.linear_fwd(x, w, None), lambda: torch.mm(x, wt)),
                               ("dgrad", lambda: K.linear_dgrad(dy, w), lambda: torch.mm(dy, w)),
                               ("wgrad", lambda: K.linear_wgrad(dy, x), lambda: torch.mm(dy.t(), x))):
            a, b = timeit(ours), timeit(ref)
            row.append(f"{tag}: ours {fl / a / 1e6:6.1f} TF/s ({a:6.1f} us)  lib {fl / b / 1e6:6.1f} TF/s ({b:6.1f} us)")
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    main()
