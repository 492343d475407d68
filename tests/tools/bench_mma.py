"""GPU-box A/B of the two GEMM arithmetic modes (phnet_tune_mma): f32-input MFMA vs split-bf16 (3 bf16 MFMAs per product).
Prints device time (hipGraph replay) and the error against an fp64 reference for fwd / dgrad / wgrad on the hot shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib
from bench_conv import timeit, SHAPES

def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())

def main():
    shapes = SHAPES[:4] + SHAPES[5:7] + SHAPES[8:9]
    for name, N, Hi, Wi, Ci, Co, R, st, pad in shapes:
        x = torch.randn(N, Hi, Wi, Ci, device="cuda"); w = torch.randn(Co, R, R, Ci, device="cuda") * 0.05
        ho, wo = K.conv_out_hw(Hi, Wi, R, R, st, pad)
        gy = torch.randn(N, ho, wo, Co, device="cuda")
        fl = 2.0 * N * ho * wo * Co * R * R * Ci
        xd = x.double().permute(0, 3, 1, 2).requires_grad_(True); wd = w.double().permute(0, 3, 1, 2).requires_grad_(True)
        yd = F.conv2d(xd, wd, None, st, pad)
        yd.backward(gy.double().permute(0, 3, 1, 2))
        y_ref = yd.detach().permute(0, 2, 3, 1); dx_ref = xd.grad.permute(0, 2, 3, 1); dw_ref = wd.grad.permute(0, 2, 3, 1)
        print(f"== {name}: {fl/1e9:.2f} GF")
        for mode in ((0, 3) if '--fast' in sys.argv else (0, 1, 2, 3)):
            assert lib().phnet_tune_mma(mode) == 0
            y = K.conv2d_fwd(x, w, None, st, pad); dx = K.conv2d_dgrad(gy, w, (Hi, Wi), st, pad); dw = K.conv2d_wgrad(gy, x, w.shape, st, pad)
            dw = dw[0] if isinstance(dw, tuple) else dw
            tf = timeit(lambda: K.conv2d_fwd(x, w, None, st, pad))
            td = timeit(lambda: K.conv2d_dgrad(gy, w, (Hi, Wi), st, pad))
            tw = timeit(lambda: K.conv2d_wgrad(gy, x, w.shape, st, pad))
            print(f"   mma={mode}: fwd {tf:6.1f} us {fl/tf/1e6:6.1f} TF/s err {rel(y, y_ref):.1e} | dgrad {td:6.1f} us {fl/td/1e6:6.1f} TF/s "
                  f"err {rel(dx, dx_ref):.1e} | wgrad {tw:6.1f} us {fl/tw/1e6:6.1f} TF/s err {rel(dw, dw_ref):.1e}", flush=True)
        lib().phnet_tune_mma(0)

if __name__ == "__main__":
    main()
