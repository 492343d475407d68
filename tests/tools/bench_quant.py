"""GPU-box probe: TF/s of the layer-1 3x3 convolution (forward / dgrad / wgrad) against the number of 64x64 tiles of the launch
(how much of the clip-sized layers' time is the ragged last round of workgroups)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from bench_conv import timeit

def main():
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for N, H, W in [(1, 64, 256), (2, 64, 256), (3, 64, 256), (4, 64, 256), (4, 72, 256), (5, 64, 250), (5, 64, 256), (6, 64, 256), (8, 64, 256), (10, 64, 256), (16, 64, 256)]:
        x = torch.randn(N, H, W, C, device="cuda"); w = torch.randn(C, 3, 3, C, device="cuda") * 0.05
        gy = torch.randn(N, H, W, C, device="cuda")
        fl = 2.0 * N * H * W * C * 9 * C
        tf = timeit(lambda: K.conv2d_fwd(x, w, None, 1, 1))
        td = timeit(lambda: K.conv2d_dgrad(gy, w, (H, W), 1, 1))
        tw = timeit(lambda: K.conv2d_wgrad(gy, x, w.shape, 1, 1))
        print(f"tiles {N*H*W//64:5d}: fwd {tf:6.1f} us {fl/tf/1e6:6.1f} TF/s | dgrad {td:6.1f} us {fl/td/1e6:6.1f} | wgrad {tw:6.1f} us {fl/tw/1e6:6.1f}", flush=True)

if __name__ == "__main__":
    main()
