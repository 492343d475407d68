"""Flat gradient arena: every parameter's .grad is a view into ONE contiguous fp32 buffer (375 MB for ResNet-34).

Why (MI355X-first): (1) the per-clip loop uses every head weight five times (once per frame); instead of letting
autograd materialise five gradients and add them with ~1300 tiny kernels per step, the HIP backward kernels accumulate
straight into the arena (their split-K / partial-sum reduce passes take an `accumulate` flag); (2) zeroing gradients is
one memset; (3) data-parallel averaging is a handful of large RCCL all-reduces over the flat buffer - xGMI links are
point-to-point and per-link bound, so few large collectives beat DDP's 25 MB buckets.
"""
from typing import Dict, Iterable, Optional

import torch

_DIRECT: Dict[int, "GradArena"] = {}


def _view_like(chunk: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """A view of the flat `chunk` with the shape AND strides of parameter `p` (dense, or channels_last 4-D)."""
    if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
        co, ci, r, s = p.shape
        return chunk.view(co, r, s, ci).permute(0, 3, 1, 2)
    return chunk.view(p.shape)


class GradArena:
    def __init__(self, params: Iterable[torch.nn.Parameter], flatten_params: bool = False):
        """flatten_params=True also moves the parameter VALUES into one flat buffer (`flat_params`, same element order as
        the gradients) so that an optimizer can update everything with one launch (phnet_amd.optim.FlatAdamW)."""
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        pad = (-total) % 4                                            # the flat optimizer kernel works on float4
        self.flat = torch.zeros(total + pad, dtype=torch.float32, device=dev)
        self.flat_params = torch.zeros(total + pad, dtype=torch.float32, device=dev) if flatten_params else None
        self.offsets = {}
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = _view_like(self.flat[off:off + n], p)
            if flatten_params:
                with torch.no_grad():
                    dst = _view_like(self.flat_params[off:off + n], p)
                    dst.copy_(p.data)
                    p.data = dst
            self.offsets[id(p)] = (off, n)
            _DIRECT[id(p)] = self
            off += n
        self.numel = total

    def zero(self):
        self.flat.zero_()

    def release(self):
        for p in self.params:
            _DIRECT.pop(id(p), None)


class _GradSink(torch.autograd.Function):
    """Identity over a weight that is DERIVED once per clip (folded Linear pairs, the concatenated / block-diagonal tower
    weights) and then used by every frame and stage of the clip.  The HIP weight-gradient kernels of those uses accumulate
    straight into `buf`; autograd sees no gradient from them and this node hands `buf` on once, instead of the engine
    summing 15 per-use gradients with 14 add kernels per derived tensor."""

    @staticmethod
    def forward(ctx, w, buf):
        ctx.set_materialize_grads(False)
        ctx.buf = buf
        return w.detach()

    @staticmethod
    def backward(ctx, g):
        return (ctx.buf if g is None else ctx.buf + g), None


class SinkPool:
    """Zero-initialised side buffers for the grad sinks of ONE clip, carved out of a few large zero-filled chunks (one fill
    launch per chunk instead of one per derived weight: 24 per clip).  A pool belongs to one forward pass - every clip gets
    fresh buffers, so several forward passes before one backward (trainOL.py:205-212) still keep their gradients apart."""
    CHUNK = 3 << 20                                     # floats; the derived weights of one clip need ~2.3 M

    def __init__(self):
        self.chunk, self.used = None, 0

    def take(self, like: torch.Tensor) -> torch.Tensor:
        n = like.numel()
        pad = (-n) % 4
        if self.chunk is None or self.chunk.device != like.device or self.used + n + pad > self.chunk.numel():
            self.chunk = torch.zeros(max(self.CHUNK, n + pad), dtype=torch.float32, device=like.device)
            self.used = 0
        buf = self.chunk[self.used:self.used + n].view(like.shape)
        self.used += n + pad
        return buf


def grad_sink(w: torch.Tensor, pool: Optional[SinkPool] = None) -> torch.Tensor:
    """Returns `w` as a tensor whose gradient is gathered in a zero-initialised side buffer (see _GradSink)."""
    if not (torch.is_grad_enabled() and w.requires_grad):
        return w
    w = w.contiguous()
    buf = pool.take(w) if (pool is not None and w.dtype == torch.float32) else torch.zeros_like(w)
    out = _GradSink.apply(w, buf)
    out._phnet_sink = buf
    return out


def direct_grad(p) -> Optional[torch.Tensor]:
    """The buffer to accumulate into (arena view of a parameter, or the side buffer of a grad_sink tensor), or None."""
    sink = getattr(p, "_phnet_sink", None)
    if sink is not None:
        return sink
    if isinstance(p, torch.nn.Parameter) and id(p) in _DIRECT and p.grad is not None:
        return p.grad
    return None
