"""GPU-box micro-benchmark: the 3x3 weight-gradient kernels on the trunk shapes of a clip (accumulating into dW as the step does)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K

CLIPS = int(sys.argv[sys.argv.index("--clips") + 1]) if "--clips" in sys.argv else 1
SHAPES = [("layer1", 5, 80, 200, 64), ("layer2", 5, 40, 100, 128), ("layer3", 5, 20, 50, 256), ("layer4", 5, 10, 25, 512)]
if "--tiny" in sys.argv:      # a handful of K steps: what a launch costs before and after its loop (prologue, partial sums, reduce)
    SHAPES = [("tiny1", 1, 8, 32, 64), ("tiny4", 1, 8, 32, 512), ("tiny4b", 1, 16, 64, 512)]


def timeit(fn, iters=20):
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
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


def main():
    flags = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1]
    for name, n, h, w, c in SHAPES:
        n *= CLIPS
        x = torch.randn(n, h, w, c, device="cuda")
        gy = torch.randn(n, h, w, c, device="cuda")
        dw = torch.zeros(c, 3, 3, c, device="cuda")
        fl = 2.0 * n * h * w * c * c * 9
        out = []
        for f in flags:
            K.tune_wgrad(f, 768)
            t = timeit(lambda: K.conv2d_wgrad(gy, x, dw.shape, 1, 1, dw=dw, accumulate=True))
            out.append(f"flags {f:2d}: {t:6.1f} us {fl / t / 1e6:6.1f} TF/s")
        K.tune_wgrad(1, 768)
        print(f"{name} ({n}x{h}x{w}x{c}): " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
