"""GPU-box micro-benchmark of the implicit-GEMM kernel: tile / split-K sweep on the hot shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib

TRUNK_ONLY = "--trunk" in sys.argv
CLIPS = int(sys.argv[sys.argv.index("--clips") + 1]) if "--clips" in sys.argv else 1    # frames of CLIPS clips in one trunk pass
SHAPES = [  # name, N,Hi,Wi,Ci,Co,R,stride,pad
    ("layer1 3x3", 5, 80, 200, 64, 64, 3, 1, 1),
    ("layer2 3x3", 5, 40, 100, 128, 128, 3, 1, 1),
    ("layer3 3x3", 5, 20, 50, 256, 256, 3, 1, 1),
    ("layer4 3x3", 5, 10, 25, 512, 512, 3, 1, 1),
    ("stem 7x7", 5, 320, 800, 4, 64, 7, 2, 3),
    ("hyper 1024->8192", 240, 1, 1, 1024, 8192, 1, 1, 0),
    ("hyper 4608->1024", 240, 1, 1, 4608, 1024, 1, 1, 0),
    ("hyper 64->1024", 240, 1, 1, 64, 1024, 1, 1, 0),
    ("gate 2304->576", 240, 1, 1, 2304, 576, 1, 1, 0),
    ("out 2304->384", 240, 1, 1, 2304, 384, 1, 1, 0),
    ("tower 64->64", 240, 1, 1, 64, 64, 1, 1, 0),
    ("tower 128->128", 240, 1, 1, 128, 128, 1, 1, 0),
    # the same layers with the 5 frames of a clip batched (stage-major training schedule): 1200 rows
    ("clipB 1024->8192", 1200, 1, 1, 1024, 8192, 1, 1, 0),
    ("clipB 4608->1024", 1200, 1, 1, 4608, 1024, 1, 1, 0),
    ("clipB 64->8192", 1200, 1, 1, 64, 8192, 1, 1, 0),
    ("clipB 2304->576", 1200, 1, 1, 2304, 576, 1, 1, 0),
    ("clipB 2304->64", 1200, 1, 1, 2304, 64, 1, 1, 0),
]

def timeit(fn, iters=20):
    """Device time per call: the calls are captured in a hipGraph so that host launch overhead does not count."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3  # us

def main():
    if "--mma" in sys.argv:
        assert lib().phnet_tune_mma(1) == 0
        print("split-bf16 arithmetic")
    if "--mode" in sys.argv:
        mode = int(sys.argv[sys.argv.index("--mode") + 1])
        assert lib().phnet_tune_mma(mode) == 0
        print("GEMM arithmetic mode", mode)
    if "--pf" in sys.argv:
        pf = int(sys.argv[sys.argv.index("--pf") + 1])
        assert lib().phnet_tune_force_k_tile(-100 - pf) == 0
        print("register prefetch depth", pf)
    if "--no-taps3" in sys.argv:
        assert lib().phnet_tune_force_k_tile(-5) == 0
        print("three-taps 3x3 forward / dgrad kernel off")
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for name, N, Hi, Wi, Ci, Co, R, st, pad in (SHAPES[:4] if TRUNK_ONLY else SHAPES):
        if only and only not in name:
            continue
        N *= CLIPS
        x = torch.randn(N, Hi, Wi, Ci, device="cuda"); w = torch.randn(Co, R, R, Ci, device="cuda") * 0.05
        ho, wo = K.conv_out_hw(Hi, Wi, R, R, st, pad)
        gy = torch.randn(N, ho, wo, Co, device="cuda")
        fl = 2.0 * N * ho * wo * Co * R * R * Ci
        res = []
        ktiles = [0]
        if "--ktile" in sys.argv:
            ktiles = [16, 32] if "--pf" in sys.argv else [16, 32, 64]
        cfgs = [(0, 0)] if "--quick" in sys.argv else [(0, 0), (64, 64), (128, 64), (64, 128), (128, 128)]
        if "--ktile" in sys.argv:
            cfgs = [(64, 64), (128, 64), (128, 128)]
        for kt_, (bm, bn) in [(k_, c_) for k_ in ktiles for c_ in cfgs]:
            lib().phnet_tune_force_k_tile(kt_)
            for sp in ([0] if bm == 0 else ([1, 2, 4] if "--ktile" in sys.argv else [1, 2, 4, 8, 16])):
                M = N * ho * wo
                if bm and sp > 1 and (M // bm + 1) * (Co // bn + 1) * sp > 4096: continue
                lib().phnet_tune_force_conv_tile(bm, bn, sp)
                try:
                    if "--packed" in sys.argv and R == 3 and st == 1 and K.conv3p_applies(M, Ci, Co):
                        # what the trunk schedule runs since round 3: forward / data gradient on packed weights (csrc/conv3p.hip)
                        pf_, pd_ = K.conv3p_pack(w, False), K.conv3p_pack(w, True)
                        tf = timeit(lambda: K.conv3p(x, pf_, Co))
                        td = timeit(lambda: K.conv3p(gy, pd_, Ci, dgrad=True))
                    else:
                        tf = timeit(lambda: K.conv2d_fwd(x, w, None, st, pad))
                        td = timeit(lambda: K.conv2d_dgrad(gy, w, (Hi, Wi), st, pad)) if Ci >= 64 else float("nan")
                except RuntimeError as e:
                    continue
                res.append((f'{bm:3d}x{bn:3d} kt{kt_:2d}', bn, sp, tf, td))
        lib().phnet_tune_force_conv_tile(0, 0, 0)
        tw = timeit(lambda: K.conv2d_wgrad(gy, x, w.shape, st, pad))
        print(f"== {name}: {fl/1e9:.2f} GF; wgrad {tw:.1f} us = {fl/tw/1e6:.1f} TF/s")
        if "--wgrad" in sys.argv:
            for bm128 in ((1, 0, 5, 4) if "--mode" in sys.argv else (1, 0)):                # bit 2: 32 pixels per K step (mode 3)
                for target in (512, 768, 1250, 2000, 3000):
                    lib().phnet_tune_wgrad(bm128, target)
                    tw = timeit(lambda: K.conv2d_wgrad(gy, x, w.shape, st, pad))
                    print(f"   wgrad bm128={bm128 & 1} bkw={32 if bm128 & 4 else 16} target={target:5d}: {tw:7.1f} us {fl/tw/1e6:6.1f} TF/s")
            lib().phnet_tune_wgrad(1, 768)
        if "--wgrad3" in sys.argv and R == 3 and st == 1:
            for flags, target in ((1, 256), (1, 512), (1 | 16, 256), (1 | 16, 512)):    # workgroup target of the three-taps 3x3 kernel; bit 4: 16-pixel steps
                lib().phnet_tune_wgrad(flags, -target)
                tw = timeit(lambda: K.conv2d_wgrad(gy, x, w.shape, st, pad))
                print(f"   wgrad three-taps {32 if flags & 16 else 16}-pixel steps target={target:5d}: {tw:7.1f} us {fl/tw/1e6:6.1f} TF/s")
            lib().phnet_tune_wgrad(1 | 8, 768)
            tw = timeit(lambda: K.conv2d_wgrad(gy, x, w.shape, st, pad))
            print(f"   wgrad generic kernel          : {tw:7.1f} us {fl/tw/1e6:6.1f} TF/s")
            lib().phnet_tune_wgrad(1, -256); lib().phnet_tune_wgrad(1, 768)
        lib().phnet_tune_force_k_tile(0)
        for bm, bn, sp, tf, td in res:
            print(f"   tile {bm} splits {sp:2d}: fwd {tf:7.1f} us {fl/tf/1e6:6.1f} TF/s | dgrad {td:7.1f} us {fl/td/1e6:6.1f} TF/s")

if __name__ == "__main__":
    main()
