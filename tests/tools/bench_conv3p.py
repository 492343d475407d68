"""GPU-box micro-benchmark + parity check of the packed-weight 3x3 kernel (csrc/conv3p.hip) against the three-taps kernel it
replaces (conv3x3s1_kernel) on the trunk shapes of a 5-frame 320x800 clip (and of 8 clips with --clips 8)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib
from tests.tools.bench_conv import timeit

CLIPS = int(sys.argv[sys.argv.index("--clips") + 1]) if "--clips" in sys.argv else 1
SHAPES = [("layer1", 5, 80, 200, 64, 64), ("layer2", 5, 40, 100, 128, 128), ("layer3", 5, 20, 50, 256, 256), ("layer4", 5, 10, 25, 512, 512),
          ("fpn P3", 5, 40, 100, 64, 64)]


def main():
    torch.manual_seed(0)
    if "--narrow" in sys.argv:
        assert lib().phnet_conv3p_tune(-1) == 0                # 64-column workgroup tiles only
    if "--target" in sys.argv:
        assert lib().phnet_conv3p_tune(int(sys.argv[sys.argv.index("--target") + 1])) == 0
    for name, n, h, w, ci, co in SHAPES:
        n *= CLIPS
        x = torch.randn(n, h, w, ci, device="cuda")
        wt = torch.randn(co, 3, 3, ci, device="cuda") * (9 * ci) ** -0.5
        dy = torch.randn(n, h, w, co, device="cuda")
        res = torch.randn(n, h, w, co, device="cuda")
        pf, pd = K.conv3p_pack(wt, False), K.conv3p_pack(wt, True)
        if "--pmc" in sys.argv:                                # counter passes: the new kernel only, a few launches per shape
            for _ in range(3):
                K.conv3p(x, pf, co)
            torch.cuda.synchronize()
            continue
        ref = K.conv2d_fwd(x, wt, None, 1, 1)
        got = K.conv3p(x, pf, co)
        ref64 = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
        scale = float(ref64.abs().max())
        e_new, e_old = float((got - ref64).abs().max()) / scale, float((ref - ref64).abs().max()) / scale
        refd = K.conv2d_dgrad(dy, wt, (h, w), 1, 1, addend=res)
        gotd = K.conv3p(dy, pd, ci, dgrad=True, addend=res)
        ed = float((gotd - refd).abs().max()) / float(refd.abs().max())
        # fused epilogue: addend + relu, and the statistics rows
        gotf = K.conv3p(x, pf, co, addend=res, relu=True)
        reff = K.conv2d_fwd(x, wt, None, 1, 1, relu=True, addend=res)
        ef = float((gotf - reff).abs().max()) / scale
        gs, (part, nblk) = K.conv3p(x, pf, co, stats=True)
        sums = part[:nblk * 2 * co * 4].view(torch.float32).view(nblk, 2, co).double().sum(0)
        es = float((sums[0] - ref64.sum((0, 1, 2))).abs().max() / ref64.abs().sum((0, 1, 2)).max())
        eq = float((sums[1] - (ref64 ** 2).sum((0, 1, 2))).abs().max() / (ref64 ** 2).sum((0, 1, 2)).max())
        t_old = timeit(lambda: K.conv2d_fwd(x, wt, None, 1, 1))
        t_new = timeit(lambda: K.conv3p(x, pf, co))
        t_oldd = timeit(lambda: K.conv2d_dgrad(dy, wt, (h, w), 1, 1))
        t_newd = timeit(lambda: K.conv3p(dy, pd, ci, dgrad=True))
        t_pack = timeit(lambda: K.conv3p_pack(wt, False, pf))
        gf = 2.0 * n * h * w * co * 9 * ci / 1e9
        m = n * h * w
        need = 8 * m * co * 4 if m * co < (1 << 23) else 0
        print(f"{name:8s} M={m:6d} C={ci:3d}: err vs fp64 new {e_new:.2e} old {e_old:.2e} | dgrad vs old {ed:.2e} | fused {ef:.2e} | stats {es:.1e} {eq:.1e} | "
              f"fwd old {t_old:6.1f} us ({gf / t_old * 1e3:5.1f} TF/s) new {t_new:6.1f} us ({gf / t_new * 1e3:5.1f} TF/s) splits {lib().phnet_conv3p_splits(m, ci, co, need)} | "
              f"dgrad old {t_oldd:6.1f} new {t_newd:6.1f} us | pack {t_pack:5.1f} us", flush=True)


if __name__ == "__main__":
    main()
