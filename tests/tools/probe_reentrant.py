import torch
class Deferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pre, w):
        ctx.save_for_backward(x, w)
        return pre.view_as(pre)
    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xx = x.detach().requires_grad_()
            y = torch.tanh(xx @ w)          # w is a leaf parameter: its .grad accumulates in the inner backward
            torch.autograd.backward(y, dy)
        return xx.grad, None, None
dev = "cuda"
w = torch.randn(64, 64, device=dev, requires_grad=True)
w.grad = torch.zeros_like(w)
x0 = torch.randn(32, 64, device=dev)
v = torch.randn(64, 64, device=dev, requires_grad=True)
v.grad = torch.zeros_like(v)
def step():
    w.grad.zero_(); v.grad.zero_()
    x = x0 @ v
    with torch.no_grad():
        pre = torch.tanh(x @ w)
    y = Deferred.apply(x, pre, w)
    loss = (y * y).sum()
    loss.backward()
    return loss
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
ref_w, ref_v = w.grad.clone(), v.grad.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    l = step()
w.grad.zero_(); v.grad.zero_()
g.replay(); torch.cuda.synchronize()
print("graph ok", torch.allclose(w.grad, ref_w), torch.allclose(v.grad, ref_v), float(l))
# reference without deferral
w.grad.zero_(); v.grad.zero_()
x = x0 @ v; y = torch.tanh(x @ w); (y*y).sum().backward()
print("matches direct", torch.allclose(w.grad, ref_w, atol=1e-5), torch.allclose(v.grad, ref_v, atol=1e-5))
