"""GPU-box A/B: wave-per-plane gate kernels (gate_wave.hip) against the generic ones (gate.hip) on the same operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib

def run(wave, B, N, C, P, x, pd, gd):
    assert lib().phnet_tune_gate_wave(int(wave)) == 0
    out, saved = K.gate_stack_fwd(x, pd, 1e-5, True, anchors=N)
    grads = [torch.full_like(t, float("nan")) for t in pd]
    K.gate_stack_bwd(gd, x, out, pd, saved, grads, 1e-5, False, anchors=N)
    torch.cuda.synchronize()
    return out, grads

def main():
    for B, N in ((1, 240), (5, 240), (2, 3)):
        C, P = 64, 36
        g = torch.Generator(device="cuda").manual_seed(0)
        rn = lambda *s, scale=1.0: torch.randn(*s, generator=g, device="cuda") * scale       # noqa: E731
        pd = [1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P)]
        for _ in range(4):
            pd += [rn(N, 1, 3, 3, scale=0.4), 0.1 * rn(N), 1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P),
                   rn(N, 1, 3, 3, scale=0.4), 0.1 * rn(N), 1.0 + 0.2 * rn(C, P), 0.1 * rn(C, P)]
        x, gd = rn(B * N, C, P), rn(B * N, C, P)
        ow, gw = run(True, B, N, C, P, x, pd, gd)
        og, gg = run(False, B, N, C, P, x, pd, gd)
        print(f"B={B} N={N}: out max diff {float((ow - og).abs().max()):.3e}")
        for i, (a, b) in enumerate(zip(gw, gg)):
            d = float((a - b).abs().max()); s = float(b.abs().max())
            flag = "  <-----" if d > 1e-4 * max(1.0, s) else ""
            print(f"   param {i:2d} {tuple(a.shape)}: max diff {d:.3e} scale {s:.3e}{flag}")

if __name__ == "__main__":
    main()
