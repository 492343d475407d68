"""GPU-box probe: device time of the fused routing-gate stack (forward / backward) on the clip-sized problem (5 x 240 planes of 64 x 36)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from bench_conv import timeit

def main():
    from phnet_amd._lib import lib
    if "--generic" in sys.argv:
        assert lib().phnet_tune_gate_wave(0) == 0              # one workgroup per plane (csrc/gate.hip) instead of one wavefront
    B, N, C, P = 5, 240, 64, 36
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g, device="cuda") * scale       # noqa: E731
    pd = [1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P)]
    for _ in range(4):
        pd += [rn(N, 1, 3, 3, scale=0.4), 0.1 * rn(N), 1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P),
               rn(N, 1, 3, 3, scale=0.4), 0.1 * rn(N), 1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P)]
    x = rn(B * N, C, P)
    out, saved = K.gate_stack_fwd(x, pd, 1e-5, True, anchors=N)
    gd = rn(B * N, C, P)
    grads = [torch.zeros_like(t) for t in pd]
    tf = timeit(lambda: K.gate_stack_fwd(x, pd, 1e-5, True, anchors=N))
    tb = timeit(lambda: K.gate_stack_bwd(gd, x, out, pd, saved, grads, 1e-5, True, anchors=N))
    print(f"gate stack {B}x{N} planes of {C}x{P}: forward {tf:.1f} us, backward (+ reduces) {tb:.1f} us")

if __name__ == "__main__":
    main()
