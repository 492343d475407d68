"""GPU-box micro-benchmark of small latency-bound kernels (BatchNorm finalize, label assignment, memory tokens): device time per
launch from a hipGraph of back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib, check
from tests.tools.bench_conv import timeit

def main():
    dev = "cuda"
    for nblk, C, M in ((1250, 64, 80000), (2500, 64, 320000), (313, 128, 20000), (626, 128, 20000), (79, 256, 5000), (316, 256, 5000), (20, 512, 1250), (160, 512, 1250)):
        part = torch.randn(nblk, 2, C, device=dev).abs()
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        out = [torch.empty(C, device=dev) for _ in range(4)]
        def f():
            check(lib().phnet_bn_finalize_partials(part.data_ptr(), nblk, M, C, 1e-5, 0.1, g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                   out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), K._stream()), "fin")
        print(f"bn_finalize nblk {nblk:5d} C {C:4d}: {timeit(f):6.1f} us")
    pred = torch.randn(240, 42, device=dev) * 0.3 + 0.4
    from phnet_amd.synthetic import make_targets
    tgt = make_targets(320, 800, 1).to(dev)[0]
    print(f"lane_assign: {timeit(lambda: K.lane_assign(pred, tgt, 800, 320)):6.1f} us")
    feat = torch.randn(240, 128, device=dev)
    rows = torch.tensor([3, 17, 100, -1], device=dev)
    print(f"memory_tokens: {timeit(lambda: K.memory_tokens(feat, rows)):6.1f} us")
    x = torch.randn(240, 128, device=dev); w = torch.randn(128, 128, device=dev); bb = torch.randn(128, device=dev)
    print(f"linear 240x128x128: {timeit(lambda: K.linear_fwd(x, w, bb)):6.1f} us")
    e = torch.empty(1, device=dev)
    print(f"empty elementwise (x.add_): {timeit(lambda: e.add_(1.0)):6.1f} us")

if __name__ == "__main__":
    main()
