import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phnet_amd import hip_ops as K
from phnet_amd._lib import lib
from tests.test_kernels_gpu import _gate_params, _gate_reference, dev

B, N, C, P = 5, 240, 64, 36
torch.manual_seed(100 + N)
eps = 1e-5
x = torch.randn(B, N, C, P, dtype=torch.float64)
params = _gate_params(N, C, P, seed=N + C)
ref = _gate_reference(x, params, eps)
gout = torch.randn_like(ref)
ref.backward(gout)
xd = dev(x.float().reshape(B * N, C, P))
pd = [dev(t.detach().float()) for t in params]
gd = dev(gout.float().reshape(B * N, C, P))
for wave in (1, 0, 1):
    assert lib().phnet_tune_gate_wave(wave) == 0
    out, saved = K.gate_stack_fwd(xd, pd, eps, True, anchors=N)
    grads = [torch.full_like(t, float("nan")) for t in pd]
    K.gate_stack_bwd(gd, xd, out, pd, saved, grads, eps, False, anchors=N)
    torch.cuda.synchronize()
    e = [(float((g.cpu().double().view(p.shape) - p.grad).abs().max()) / max(1.0, float(p.grad.abs().max())), i) for i, (g, p) in enumerate(zip(grads, params))]
    print("wave" if wave else "generic", "out err", float((out.cpu().double().view(B, N, C, P) - ref).abs().max()), "worst params", sorted(e, reverse=True)[:4])
    d = (grads[0].cpu().double() - params[0].grad).abs()
    idx = int(d.argmax()); print("   param0 worst at (c,p) =", idx // P, idx % P, float(d.max()), " count > 1e-3:", int((d > 1e-3).sum()))

CP = C * P
parts = {}
for wave in (1, 0):
    assert lib().phnet_tune_gate_wave(wave) == 0
    out, saved = K.gate_stack_fwd(xd, pd, eps, True, anchors=N)
    grads = [torch.zeros_like(t) for t in pd]
    K.gate_stack_bwd(gd, xd, out, pd, saved, grads, eps, False, anchors=N)
    torch.cuda.synchronize()
    ws = K.workspace(1, xd.device, 3)
    parts[wave] = ws[:18 * B * N * CP * 4].view(torch.float32).view(B * N, 18, CP).clone()
d = (parts[1] - parts[0]).abs()
per_j = d.amax(dim=(0, 2))
print("per-LN-row max |wave - generic| of the per-plane partials:", [f"{v:.2e}" for v in per_j.tolist()])
dj0 = d[:, 0].amax(dim=1)
bad = torch.nonzero(dj0 > 1e-3).flatten().tolist()
print("planes with row-0 differences > 1e-3:", len(bad), bad[:20])
if bad:
    n = bad[0]
    print("plane", n, "row0 wave", parts[1][n, 0, :6].tolist(), "generic", parts[0][n, 0, :6].tolist())
    print("plane", n, "row1 wave", parts[1][n, 1, :6].tolist(), "generic", parts[0][n, 1, :6].tolist())
    xx = xd[n].flatten()
    print("plane", n, "x mean/var", float(xx.mean()), float(xx.var(unbiased=False)))
