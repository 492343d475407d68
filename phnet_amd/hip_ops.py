import os
"""Tensor-level wrappers over the C-ABI (include/phnet_hip.h): shape checks, output allocation, stream plumbing.

PyTorch is used here only for device memory and the current HIP stream.  No op in this file has a
fallback: a non-CUDA tensor or a missing library raises.
"""
from typing import Optional, Tuple

import torch

from ._lib import check, lib

_WS = {}

# Optional kernel timer (bench.py): when TIMER is a list, every conv/linear GEMM launch is bracketed by two events on
# the launch stream and recorded as (kernel symbol, algorithmic FLOPs, start event, end event).
TIMER = None


_MMA_MODE = 3        # arithmetic of the GEMM kernels (set_mma_mode; 3 = "bf16x3", the library default); here only used to NAME kernel symbols for bench.py


_TAPS3 = True           # mirrors csrc/conv.hip g_taps3 (kernel names of the bench's per-kernel accounting only)


def _gemm_symbol(m, co, k, ws_bytes, dgrad, ci_a, in_dil=1, taps3=False):
    import ctypes
    bm, bn, sp, kt = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    check(lib().phnet_conv2d_plan(m, co, k, ws_bytes, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp), ctypes.byref(kt)),
          "phnet_conv2d_plan")
    uni = bm.value == 64 and bn.value == 64 and ci_a % kt.value == 0      # uniform-tap variant (csrc/conv.hip)
    pf = 4 if _MMA_MODE == 3 else 1                                       # register prefetch ring; buffer loads (launch_conv)
    buf = uni and _MMA_MODE == 3 and in_dil == 1
    if taps3 and buf and kt.value == 16 and _TAPS3:                       # three-taps 3x3 / stride-1 kernel (launch_conv)
        return f"conv3x3s1_kernel<{'true' if dgrad else 'false'}>", sp.value
    return (f"conv_igemm_kernel<{bm.value}, {bn.value}, {'true' if dgrad else 'false'}, {kt.value}, {'true' if uni else 'false'}, "
            f"{_MMA_MODE}, {pf}, {'true' if buf else 'false'}>", sp.value)


def _timed_launch(sym_fn, flops, launch, shape=None, nbytes=None):
    """Runs `launch()`.  With the timer on, also records (symbol, split-K factor, FLOPs, start event, end event, launch):
    bench.py re-launches the recorded closures in isolation inside small hipGraphs to get device-side durations that
    are free of host launch gaps (the event pair around a live eager launch includes them for microsecond kernels)."""
    if TIMER is None:
        return launch()
    sym, splits = sym_fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = launch()
    e1.record()
    TIMER.append((sym, splits, flops, e0, e1, launch, shape, nbytes))     # shape: ("fwd" | "dgrad" | "wgrad" | "linbwd", GEMM rows, cols, depth) for tests/tools/gemm_shapes.py; nbytes: algorithmic HBM bytes
    return out


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype=torch.float32, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor; phnet_amd has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


_WS_RETIRED = []


def workspace(nbytes: int, device, slot: int = 0) -> torch.Tensor:
    """Grow-only scratch buffer per (device, slot).  Kernels on one stream serialise, so sharing is safe.
    A buffer that is outgrown is RETIRED, never freed: a captured hipGraph may hold its address (split-K partials,
    BatchNorm partials, LayerNorm scratch), and handing that memory back to the caching allocator would let a later replay
    scribble over tensors that reuse it.  (A few MB per growth step, a handful of steps per process.)"""
    key = (torch.device(device).index, slot)
    cur = _WS.get(key)
    if cur is None or cur.numel() < nbytes:
        if cur is not None:
            _WS_RETIRED.append(cur)
        cur = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = cur
    return cur


# ------------------------------------------------------------------------------------------------ NMS
def lane_nms(rows: torch.Tensor, scores: torch.Tensor, thresh: float, top_k: int,
             counts: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """rows [K,5+S] or [F,K,5+S]; returns (keep, num_to_keep, parent) shaped like the reference's outputs."""
    _req(rows, name="rows"); _req(scores, name="scores")
    batched = rows.dim() == 3
    f = rows.shape[0] if batched else 1
    k, prop = rows.shape[-2], rows.shape[-1]
    if scores.numel() != f * k:
        raise RuntimeError("scores must have one entry per row")
    if counts is not None:
        _req(counts, torch.int32, "counts")
    keep = torch.empty((f, k), dtype=torch.int64, device=rows.device)
    parent = torch.empty((f, k), dtype=torch.int64, device=rows.device)
    num = torch.empty((f,), dtype=torch.int64, device=rows.device)
    check(lib().phnet_lane_nms(_ptr(rows), _ptr(scores), _ptr(counts), f, k, prop - 5, float(thresh), int(top_k),
                               _ptr(keep), _ptr(num), _ptr(parent), _stream()), "phnet_lane_nms")
    if batched:
        return keep, num, parent
    return keep[0], num[0], parent[0]


# ------------------------------------------------------------------------------------------------ ROI pooling
def roi_pool_fwd(fmap: torch.Tensor, xs: torch.Tensor, ys: torch.Tensor, with_cp: bool = False):
    """fmap [B,h,w,64] NHWC, xs [B,N,P], ys [P] -> [B,N,P,64] (and the [B,N,64,P] copy for the gate)."""
    _req(fmap, name="fmap"); _req(xs, name="xs"); _req(ys, name="ys")
    b, h, w, c = fmap.shape
    _, n, p = xs.shape
    out = torch.empty((b, n, p, c), dtype=torch.float32, device=fmap.device)
    out_cp = torch.empty((b, n, c, p), dtype=torch.float32, device=fmap.device) if with_cp else None
    check(lib().phnet_roi_pool_fwd(_ptr(fmap), _ptr(xs), _ptr(ys), _ptr(out), _ptr(out_cp), b, n, p, h, w, c, _stream()),
          "phnet_roi_pool_fwd")
    return (out, out_cp) if with_cp else out


def roi_pool_bwd(dout, fmap, xs, ys, dmap: Optional[torch.Tensor], need_dxs: bool):
    _req(dout, name="dout")
    b, h, w, c = fmap.shape
    _, n, p = xs.shape
    dxs = torch.empty_like(xs) if need_dxs else None
    check(lib().phnet_roi_pool_bwd(_ptr(dout), _ptr(fmap), _ptr(xs), _ptr(ys), _ptr(dmap), _ptr(dxs),
                                   b, n, p, h, w, c, _stream()), "phnet_roi_pool_bwd")
    return dxs


# ------------------------------------------------------------------------------------------------ conv / linear
def conv_out_hw(hi, wi, r, s, stride, pad):
    return (hi + 2 * pad - r) // stride + 1, (wi + 2 * pad - s) // stride + 1


def conv2d_fwd(x, w, bias, stride: int, pad: int, relu: bool = False, out: Optional[torch.Tensor] = None,
               addend: Optional[torch.Tensor] = None, stats: bool = False):
    """x NHWC [N,Hi,Wi,Ci]; w OHWI [Co,R,S,Ci]; -> NHWC [N,Ho,Wo,Co].
    addend (shaped like the output): y = relu?(conv + bias + addend) in the kernel's epilogue.
    stats=True: also returns (partial, nblk) - per-channel (sum | sum of squares) partials of y written by the epilogue, for
    bn_fwd(..., partials=...): no separate statistics pass over y."""
    _req(x, name="x"); _req(w, name="w")
    n, hi, wi, ci = x.shape
    co, r, s, ci2 = w.shape
    assert ci == ci2, (x.shape, w.shape)
    ho, wo = conv_out_hw(hi, wi, r, s, stride, pad)
    if out is None:
        out = torch.empty((n, ho, wo, co), dtype=torch.float32, device=x.device)
    # split-K scratch: the plan (csrc/conv.hip plan_conv) sees the size THIS shape asks for, not whatever the shared buffer
    # has grown to - the same shape always runs the same plan, whatever ran before it in the process
    need = 8 * n * ho * wo * co * 4 if n * ho * wo * co < (1 << 23) else 0
    ws = workspace(need, x.device) if need else None
    m, k = n * ho * wo, r * s * ci
    if addend is not None or stats:
        part, nblk = None, 0
        if stats:
            nblk = int(lib().phnet_conv2d_stats_blocks(m, co, k, need))
            part = workspace(nblk * 2 * co * 4, x.device, 1)
        if addend is not None:
            _req(addend, name="addend")
            assert addend.shape == out.shape
        _timed_launch(lambda: _gemm_symbol(m, co, k, need, False, ci, taps3=(r == 3 and s == 3 and stride == 1 and pad == 1)), 2.0 * m * co * k,
                      lambda: check(lib().phnet_conv2d_fwd_fused(_ptr(x), _ptr(w), _ptr(bias), _ptr(addend), _ptr(out), _ptr(part), n, hi, wi,
                                                                 ci, co, r, s, stride, pad, int(relu), _ptr(ws), need, _stream()),
                                    "phnet_conv2d_fwd_fused"), shape=("fwd", m, co, k, r), nbytes=4.0 * (n * hi * wi * ci + m * co + co * k))
        return (out, (part, nblk)) if stats else out
    _timed_launch(lambda: _gemm_symbol(m, co, k, need, False, ci, taps3=(r == 3 and s == 3 and stride == 1 and pad == 1)), 2.0 * m * co * k,
                  lambda: check(lib().phnet_conv2d_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), n, hi, wi, ci, co, r, s, stride,
                                                       pad, int(relu), _ptr(ws), need, _stream()), "phnet_conv2d_fwd"),
                  shape=("fwd", m, co, k, r), nbytes=4.0 * (n * hi * wi * ci + m * co + co * k))
    return out


def conv3p_applies(m: int, ca: int, nn: int) -> bool:
    return bool(lib().phnet_conv3p_applies(int(m), int(ca), int(nn)))


def conv3p_pack(w: torch.Tensor, dgrad: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w OHWI [Co,3,3,Ci] f32 -> the packed operand of conv3p (three bf16 planes in MFMA fragment order, csrc/conv3p.hip)."""
    _req(w, name="w")
    co, r, s_, ci = w.shape
    assert r == 3 and s_ == 3, w.shape
    nbytes = int(lib().phnet_conv3p_packed_bytes(co, ci))
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    assert out.numel() >= nbytes and out.is_contiguous()
    check(lib().phnet_conv3p_pack(_ptr(w), _ptr(out), co, ci, int(dgrad), _stream()), "phnet_conv3p_pack")
    return out


class Conv3pPackPlan:
    """The packed images (forward + data gradient) of a list of 3x3 weights in ONE buffer, refreshed by ONE launch
    (phnet_conv3p_pack_jobs): what a training step runs once before its first convolution.  The job table holds raw pointers:
    it is rebuilt when a weight has moved (`matches`)."""

    def __init__(self, weights, with_dgrad: bool = True):
        """weights: contiguous OHWI tensors [Co,3,3,Ci]."""
        dev = weights[0].device
        self.ptrs = tuple(int(w.data_ptr()) for w in weights)
        sizes = [int(lib().phnet_conv3p_packed_bytes(w.shape[0], w.shape[3])) for w in weights]
        kinds = (False, True) if with_dgrad else (False,)
        self.buf = torch.empty(sum(sizes) * len(kinds), dtype=torch.uint8, device=dev)
        self.images, rows, off, first = [], [], 0, 0
        for w, nbytes in zip(weights, sizes):
            _req(w, name="w")
            co, _, _, ci = w.shape
            per = {}
            for dg in kinds:
                ca, nn = (co, ci) if dg else (ci, co)
                assert ca % 16 == 0 and nn % 32 == 0, w.shape
                img = self.buf[off:off + nbytes]
                rows.append([int(w.data_ptr()), int(img.data_ptr()), co | (ci << 32), int(dg), first])
                first += 9 * (ca // 16) * (nn // 32) * 64
                off += nbytes
                per[dg] = img
            self.images.append(per)
        self.total = first
        self.jobs = torch.tensor(rows, dtype=torch.int64).to(dev)

    def matches(self, weights) -> bool:
        return self.ptrs == tuple(int(w.data_ptr()) for w in weights)

    def refresh(self):
        check(lib().phnet_conv3p_pack_jobs(_ptr(self.jobs), self.jobs.shape[0], self.total, _stream()), "phnet_conv3p_pack_jobs")


def conv3p(x: torch.Tensor, packed: torch.Tensor, nn: int, dgrad: bool = False, bias=None, addend=None, relu: bool = False,
           stats: bool = False):
    """3x3 / stride 1 / pad 1 convolution (dgrad=False) or its data gradient (dgrad=True: x = dy) on packed weights.
    x NHWC [N,H,W,Ca] -> NHWC [N,H,W,nn]; stats as conv2d_fwd."""
    _req(x, name="x")
    n, h, w_, ca = x.shape
    out = torch.empty((n, h, w_, nn), dtype=torch.float32, device=x.device)
    m = n * h * w_
    need = 8 * m * nn * 4 if m * nn < (1 << 23) else 0
    ws = workspace(need, x.device) if need else None
    part, nblk = None, 0
    if stats:
        nblk = int(lib().phnet_conv3p_stats_blocks(m, ca, nn, need))
        part = workspace(nblk * 2 * nn * 4, x.device, 1)
    if addend is not None:
        _req(addend, name="addend")
        assert addend.shape == out.shape
    # the symbol rocprofv3 sees: <2> = the 128-column tile (csrc/conv3p.hip p3_plan: >= 128 output channels and >= 16384 pixels)
    _timed_launch(lambda: (f"conv3p_kernel<{2 if (nn % 128 == 0 and m >= 16384) else 1}>", int(lib().phnet_conv3p_splits(m, ca, nn, need))),
                  2.0 * m * nn * 9 * ca,
                  lambda: check(lib().phnet_conv3p_fwd(_ptr(x), _ptr(packed), _ptr(bias), _ptr(addend), _ptr(out), _ptr(part), n, h, w_, ca, nn,
                                                       int(relu), _ptr(ws), need, _stream()), "phnet_conv3p_fwd"),
                  shape=("dgrad" if dgrad else "fwd", m, nn, 9 * ca, 3),
                  nbytes=4.0 * m * (ca + nn) + 6.0 * 9 * ca * nn + (4.0 * m * nn if addend is not None else 0.0))
    return (out, (part, nblk)) if stats else out


def conv2d_dgrad(dy, w, in_hw: Tuple[int, int], stride: int, pad: int, addend: Optional[torch.Tensor] = None):
    _req(dy, name="dy"); _req(w, name="w")
    n = dy.shape[0]
    co, r, s, ci = w.shape
    hi, wi = in_hw
    dx = torch.empty((n, hi, wi, ci), dtype=torch.float32, device=dy.device)
    need = 8 * n * hi * wi * ci * 4 if n * hi * wi * ci < (1 << 23) else 0
    ws = workspace(need, dy.device) if need else None
    m, k = n * hi * wi, r * s * co
    _timed_launch(lambda: _gemm_symbol(m, ci, k, need, True, co, stride, taps3=(r == 3 and s == 3 and stride == 1 and pad == 1)), 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * co * r * s * ci,
                  lambda: check(lib().phnet_conv2d_dgrad(_ptr(dy), _ptr(w), _ptr(addend), _ptr(dx), n, hi, wi, ci, co, r, s, stride,
                                                         pad, _ptr(ws), need, _stream()), "phnet_conv2d_dgrad"),
                  shape=("dgrad", m, ci, k, r), nbytes=4.0 * (dy.numel() + m * ci + co * r * s * ci))
    return dx


CONV3P = os.environ.get("PHNET_CONV3P", "1") != "0"           # packed-weight 3x3 kernel for the trunk / FPN forward and data gradient (trunk.packed_path); bench A/B switch
_WGRAD3 = True          # mirrors csrc/conv.hip g_wgrad3 (kernel names of the bench's per-kernel accounting only)
_WGRAD3S = True         # mirrors g_wgrad3s (same purpose)
_WGRAD1S = True         # mirrors g_wgrad1s


def conv2d_wgrad(dy, x, w_shape, stride: int, pad: int, dw: Optional[torch.Tensor] = None, accumulate: bool = False,
                 dbias: Optional[torch.Tensor] = None):
    """dW (and, when `dbias` is given, the bias gradient = column sums of dy) in one launch (+ reduce when split).
    `accumulate` applies to both destinations."""
    _req(dy, name="dy"); _req(x, name="x")
    n, hi, wi, ci = x.shape
    co, r, s, _ = w_shape
    if dw is None:
        dw = torch.empty((co, r, s, ci), dtype=torch.float32, device=x.device)
        assert not accumulate
    need = lib().phnet_conv2d_wgrad_workspace(n, hi, wi, ci, co, r, s, stride, pad)
    ws = workspace(need, x.device) if need else None
    ho, wo = conv_out_hw(hi, wi, r, s, stride, pad)
    smallp = (r == 1 and s == 1 and stride == 1 and pad == 0 and n * ho * wo <= 256 and
              ((co + 63) // 64) * ((ci + 63) // 64) < 400)                                # few-rows Linear kernel (csrc/conv.hip)
    taps3 = (_MMA_MODE == 3 and r == 3 and s == 3 and stride == 1 and pad == 1 and ci % 64 == 0 and co % 64 == 0 and wi >= 16 and
             n * ho * wo >= 64 and _WGRAD3)                                              # three-taps kernel (csrc/conv.hip)
    _timed_launch(lambda: (f"linear_wgrad_smallp_kernel<64, 64, {1 if _MMA_MODE == 1 else 0}>" if smallp
                           else "wgrad1s_kernel" if (_WGRAD1S and _MMA_MODE == 3 and r == 1 and s == 1 and stride == 1 and pad == 0 and co % 128 == 0
                                                     and ci % 128 == 0 and n * ho * wo >= 256 and (co // 128) * (ci // 128) >= 64)
                           else "wgrad3s_kernel<2>" if taps3 and dbias is None and _WGRAD3S
                           else "conv_wgrad3x3_kernel<4, 16>" if taps3
                           else f"conv_wgrad_kernel<{128 if (co >= 128 and (co < 256 or n * ho * wo > 8192)) else 64}, 64, {_MMA_MODE}, 16, {4 if _MMA_MODE == 3 else 1}, {'true' if _MMA_MODE == 3 else 'false'}>", 0),
                  2.0 * n * ho * wo * co * r * s * ci,
                  lambda: check(lib().phnet_conv2d_wgrad(_ptr(dy), _ptr(x), _ptr(dw), _ptr(dbias), n, hi, wi, ci, co, r, s, stride,
                                                         pad, int(accumulate), _ptr(ws), need, _stream()), "phnet_conv2d_wgrad"),
                  shape=("wgrad", co, r * s * ci, n * ho * wo, r), nbytes=4.0 * (dy.numel() + x.numel() + co * r * s * ci))
    return dw


def linear_fwd(x2d, w, bias, relu: bool = False):
    """x [M,K], w [N,K] -> [M,N] (= conv 1x1 on an [M,1,1,K] image)."""
    m, k = x2d.shape
    return conv2d_fwd(x2d.view(m, 1, 1, k), w.view(w.shape[0], 1, 1, k), bias, 1, 0, relu).view(m, w.shape[0])


def linear_dgrad(dy2d, w):
    m, n = dy2d.shape
    k = w.shape[1]
    return conv2d_dgrad(dy2d.view(m, 1, 1, n), w.view(n, 1, 1, k), (1, 1), 1, 0).view(m, k)


def linear_wgrad(dy2d, x2d, dw: Optional[torch.Tensor] = None, accumulate: bool = False, dbias: Optional[torch.Tensor] = None):
    m, n = dy2d.shape
    k = x2d.shape[1]
    out = conv2d_wgrad(dy2d.view(m, 1, 1, n), x2d.view(m, 1, 1, k), (n, 1, 1, k), 1, 0,
                       None if dw is None else dw.view(n, 1, 1, k), accumulate, dbias)
    return out.view(n, k)


def linear_bwd_fusable(m: int, k: int, n: int) -> bool:
    return bool(lib().phnet_linear_bwd_fusable(m, k, n))


def linear_bwd(dy2d, x2d, w, dw: torch.Tensor, dbias: Optional[torch.Tensor], accumulate: bool, relu_y: Optional[torch.Tensor] = None):
    """dx = dy @ w and dw (+)= dy.T @ x (+ dbias) in one launch (few-rows Linear layers); returns dx.
    relu_y: saved output of a layer that ended in a ReLU - dy is masked by relu_y > 0 inside the kernel."""
    _req(dy2d, name="dy"); _req(x2d, name="x"); _req(w, name="w"); _req(dw, name="dw")
    m, n = dy2d.shape
    k = x2d.shape[1]
    dx = torch.empty((m, k), dtype=torch.float32, device=dy2d.device)
    sym = f"linear_bwd_fused_kernel<{'true' if n % 64 == 0 else 'false'}, {'true' if relu_y is not None else 'false'}, {1 if _MMA_MODE == 1 else 0}>"
    _timed_launch(lambda: (sym, 0), 4.0 * m * n * k,
                  lambda: check(lib().phnet_linear_bwd(_ptr(dy2d), _ptr(x2d), _ptr(w), _ptr(relu_y), _ptr(dx), _ptr(dw), _ptr(dbias), m, k, n,
                                                       int(accumulate), _stream()), "phnet_linear_bwd"), shape=("linbwd", m, n, k, 1))
    return dx


def nchw3_to_nhwc4(x):
    _req(x, name="frames")
    n, c, h, w = x.shape
    assert c == 3
    y = torch.empty((n, h, w, 4), dtype=torch.float32, device=x.device)
    check(lib().phnet_nchw3_to_nhwc4(_ptr(x), _ptr(y), n, h, w, _stream()), "phnet_nchw3_to_nhwc4")
    return y


def pad_channels(src2d, cd: int):
    _req(src2d, name="src")
    rows, cs = src2d.shape
    dst = torch.empty((rows, cd), dtype=torch.float32, device=src2d.device)
    check(lib().phnet_pad_channels(_ptr(src2d), _ptr(dst), rows, cs, cd, _stream()), "phnet_pad_channels")
    return dst


# ------------------------------------------------------------------------------------------------ BN / pool / FPN
def _partials(m, c, device, slot=1):
    nfloats = lib().phnet_channel_partials_size(m, c)
    return workspace(nfloats * 4, device, slot)


def bn_fwd(x, gamma, beta, running_mean, running_var, training: bool, eps: float, momentum: float,
           residual=None, relu: bool = True, partials=None):
    """x [..., C] NHWC conv output.  Returns (y, save_mean, save_invstd); stats tensors are None in eval.
    partials = (partial, nblk) from conv2d_fwd(..., stats=True): the batch statistics come from the convolution's epilogue."""
    _req(x, name="x")
    c = x.shape[-1]
    m = x.numel() // c
    dev = x.device
    scale = torch.empty(c, dtype=torch.float32, device=dev)
    shift = torch.empty(c, dtype=torch.float32, device=dev)
    sm = torch.empty(c, dtype=torch.float32, device=dev) if training else None
    si = torch.empty(c, dtype=torch.float32, device=dev) if training else None
    if training and partials is not None:
        check(lib().phnet_bn_finalize_partials(_ptr(partials[0]), int(partials[1]), m, c, eps, momentum, _ptr(gamma), _ptr(beta),
                                               _ptr(running_mean), _ptr(running_var), _ptr(sm), _ptr(si), _ptr(scale), _ptr(shift),
                                               _stream()), "phnet_bn_finalize_partials")
    else:
        part = _partials(m, c, dev) if training else None
        check(lib().phnet_bn_fwd_stats(_ptr(x), m, c, eps, momentum, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                       _ptr(running_var), _ptr(sm), _ptr(si), _ptr(scale), _ptr(shift), _ptr(part),
                                       int(training), _stream()), "phnet_bn_fwd_stats")
    y = torch.empty_like(x)
    check(lib().phnet_bn_apply(_ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y), m, c, int(relu), _stream()),
          "phnet_bn_apply")
    return y, sm, si


def bn_bwd(dy, x, y, save_mean, save_invstd, gamma, relu: bool, dres: Optional[torch.Tensor] = None,
           dres_accumulate: bool = False, dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
           param_accumulate: bool = False):
    """Returns (dx, dgamma, dbeta); writes/accumulates the residual-branch gradient into dres when given.
    dgamma/dbeta may be caller-provided destinations: overwritten, or added to with param_accumulate (gradient arena)."""
    _req(dy, name="dy")
    c = x.shape[-1]
    m = x.numel() // c
    dev = x.device
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=dev) if dgamma is None else dgamma
    dbeta = torch.empty(c, dtype=torch.float32, device=dev) if dbeta is None else dbeta
    c12 = torch.empty(2, c, dtype=torch.float32, device=dev)
    part = _partials(m, c, dev)
    check(lib().phnet_bn_bwd(_ptr(dy), _ptr(x), _ptr(y), _ptr(save_mean), _ptr(save_invstd), _ptr(gamma), _ptr(dx),
                             _ptr(dres), _ptr(dgamma), _ptr(dbeta), _ptr(part), _ptr(c12[0]), _ptr(c12[1]),
                             m, c, int(relu), int(dres_accumulate), int(param_accumulate), _stream()), "phnet_bn_bwd")
    return dx, dgamma, dbeta


def bn_fwd_sync(x, gamma, beta, running_mean, running_var, eps: float, momentum: float, residual=None, relu: bool = True,
                group=None, partials=None):
    """Training-mode BatchNorm whose statistics span the ranks of `group` (nn.SyncBatchNorm, trainOL.py:141), without any
    host round trip: local fp64 (sum x, sum x^2, count) -> ONE all-reduce of 2C+1 doubles -> statistics of the union batch,
    running statistics, scale / shift on the device.  Returns (y, mean, invstd, count) with `count` a device view (fp64[1])
    of the reduced element count, which the backward reads on the device too."""
    from . import parallel
    _req(x, name="x")
    c = x.shape[-1]
    m = x.numel() // c
    dev = x.device
    stats = torch.empty(4, c, dtype=torch.float32, device=dev)          # mean, invstd, scale, shift
    sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
    if partials is not None and partials[0] is not None:                # (sum, sum of squares) partials from the conv epilogue
        check(lib().phnet_bn_partials_to_sums(_ptr(partials[0]), int(partials[1]), m, c, _ptr(sums), _stream()), "phnet_bn_partials_to_sums")
    else:
        part = _partials(m, c, dev)
        check(lib().phnet_bn_local_sums(_ptr(x), m, c, _ptr(part), _ptr(sums), _stream()), "phnet_bn_local_sums")
    parallel.allreduce_sum_(sums, group)
    check(lib().phnet_bn_finalize_sums(_ptr(sums), c, eps, momentum, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                       _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _stream()), "phnet_bn_finalize_sums")
    y = torch.empty_like(x)
    check(lib().phnet_bn_apply(_ptr(x), _ptr(stats[2]), _ptr(stats[3]), _ptr(residual), _ptr(y), m, c, int(relu), _stream()),
          "phnet_bn_apply")
    return y, stats[0], stats[1], sums[2 * c:]


def bn_bwd_sync(dy, x, y, mean, invstd, gamma, relu: bool, count, dres: Optional[torch.Tensor] = None, group=None,
                dres_accumulate: bool = False, dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None,
                param_accumulate: bool = False):
    """SyncBatchNorm backward: local (sum g*xhat, sum g) -> ONE all-reduce of 2C floats -> dx with the global means (count is
    the device scalar of the forward).  dgamma / dbeta stay LOCAL sums (the gradient averaging across ranks treats them like
    every other parameter gradient, as with torch.nn.SyncBatchNorm under DDP); they are written / added to the given
    destinations."""
    from . import parallel
    _req(dy, name="dy")
    c = x.shape[-1]
    m = x.numel() // c
    dev = x.device
    sums = torch.empty(2, c, dtype=torch.float32, device=dev)
    scratch = torch.empty(2, c, dtype=torch.float32, device=dev)
    part = _partials(m, c, dev)
    check(lib().phnet_bn_bwd_reduce(_ptr(dy), _ptr(x), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(sums), _ptr(part),
                                    _ptr(scratch[0]), _ptr(scratch[1]), m, c, int(relu), _stream()), "phnet_bn_bwd_reduce")
    if dgamma is None:
        dgamma, dbeta = sums[0].clone(), sums[1].clone()
    elif param_accumulate:
        dgamma.add_(sums[0]); dbeta.add_(sums[1])
    else:
        dgamma.copy_(sums[0]); dbeta.copy_(sums[1])
    parallel.allreduce_sum_(sums, group)
    dx = torch.empty_like(x)
    check(lib().phnet_bn_bwd_apply_sums(_ptr(dy), _ptr(x), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(sums), _ptr(count),
                                        _ptr(scratch[0]), _ptr(scratch[1]), _ptr(dx), _ptr(dres), m, c, int(relu),
                                        int(dres_accumulate), _stream()), "phnet_bn_bwd_apply_sums")
    return dx, dgamma, dbeta


def maxpool_fwd(x):
    _req(x, name="x")
    n, hi, wi, c = x.shape
    ho, wo = (hi - 1) // 2 + 1, (wi - 1) // 2 + 1
    y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    arg = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    check(lib().phnet_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(arg), n, hi, wi, c, _stream()), "phnet_maxpool3x3s2_fwd")
    return y, arg


def maxpool_bwd(dy, arg, in_shape):
    n, hi, wi, c = in_shape
    dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    check(lib().phnet_maxpool3x3s2_bwd(_ptr(_req(dy)), _ptr(arg), _ptr(dx), n, hi, wi, c, _stream()), "phnet_maxpool3x3s2_bwd")
    return dx


def upsample_add_(fine, coarse):
    n, H, W, c = fine.shape
    _, h, w, _ = coarse.shape
    check(lib().phnet_upsample_add(_ptr(_req(fine)), _ptr(_req(coarse)), n, H, W, h, w, c, _stream()), "phnet_upsample_add")
    return fine


def upsample_add_bwd_(dfine, dcoarse):
    n, H, W, c = dfine.shape
    _, h, w, _ = dcoarse.shape
    check(lib().phnet_upsample_add_bwd(_ptr(_req(dfine)), _ptr(_req(dcoarse)), n, H, W, h, w, c, _stream()),
          "phnet_upsample_add_bwd")
    return dcoarse


def colsum(a2d, out: Optional[torch.Tensor] = None, accumulate: bool = False):
    _req(a2d, name="a")
    m, c = a2d.shape
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=a2d.device)
        accumulate = False
    ws = workspace(lib().phnet_colsum_workspace(m, c), a2d.device, 2)
    check(lib().phnet_colsum(_ptr(a2d), _ptr(out), m, c, int(accumulate), _ptr(ws), ws.numel(), _stream()), "phnet_colsum")
    return out


# ------------------------------------------------------------------------------------------------ LayerNorm / gate
def layernorm_fwd(x, w, b, eps: float = 1e-5, res=None, relu: bool = False, save_stats: bool = True):
    """LayerNorm over the trailing w.numel() elements.  Returns (y, mean, rstd)."""
    _req(x, name="x")
    L = w.numel()
    rows = x.numel() // L
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    check(lib().phnet_layernorm_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(res), _ptr(y), _ptr(mean), _ptr(rstd), rows, L, eps,
                                    int(relu), _stream()), "phnet_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, y, w, mean, rstd, relu: bool, need_dres: bool = False,
                  dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None, accumulate: bool = False):
    """Returns (dx, dres or None, dw, db).  dw/db may be caller-provided [L] destinations (accumulated into when asked)."""
    _req(dy, name="dy")
    L = w.numel()
    rows = x.numel() // L
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if need_dres else None
    if dw is None:
        dw, db, accumulate = torch.empty_like(w), torch.empty_like(w), False
    ws = workspace(lib().phnet_layernorm_bwd_workspace(rows, L), x.device, 2)
    check(lib().phnet_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(y), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dres),
                                    _ptr(dw), _ptr(db), rows, L, int(relu), int(accumulate), _ptr(ws), ws.numel(), _stream()),
          "phnet_layernorm_bwd")
    return dx, dres, dw, db


def dyn_bmm_ln_relu_fwd(x, w, gamma, beta, eps: float, save_stats: bool = True):
    """x [N,P,K], w [N,K,J] -> (y [N,P,J], stats [N,P,2] or None)."""
    _req(x, name="x"); _req(w, name="w")
    n, p, k = x.shape
    j = w.shape[2]
    if w.shape[0] != n or w.shape[1] != k or gamma.numel() != j:
        raise ValueError(f"dyn_bmm_ln_relu: x {tuple(x.shape)} / w {tuple(w.shape)} / gamma {tuple(gamma.shape)} mismatch")
    y = torch.empty((n, p, j), dtype=torch.float32, device=x.device)
    stats = torch.empty((n, p, 2), dtype=torch.float32, device=x.device) if save_stats else None
    check(lib().phnet_dyn_bmm_ln_relu_fwd(_ptr(x), _ptr(w), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), n, p, k, j, eps, _stream()),
          "phnet_dyn_bmm_ln_relu_fwd")
    return y, stats


def dyn_bmm_ln_relu_bwd(dy, x, w, y, stats, gamma, eps: float, need_dx: bool = True,
                        dgamma: Optional[torch.Tensor] = None, dbeta: Optional[torch.Tensor] = None, accumulate: bool = False):
    """Returns (dx or None, dw, dgamma, dbeta)."""
    _req(dy, name="dy")
    n, p, k = x.shape
    j = w.shape[2]
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty_like(w)
    if dgamma is None:
        dgamma, dbeta, accumulate = torch.empty_like(gamma), torch.empty_like(gamma), False
    ws = workspace(n * 2 * j * 4, x.device, 2)
    check(lib().phnet_dyn_bmm_ln_relu_bwd(_ptr(dy), _ptr(x), _ptr(w), _ptr(y), _ptr(stats), _ptr(gamma), _ptr(dx), _ptr(dw),
                                          _ptr(dgamma), _ptr(dbeta), n, p, k, j, eps, int(accumulate), _ptr(ws), ws.numel(), _stream()),
          "phnet_dyn_bmm_ln_relu_bwd")
    return dx, dw, dgamma, dbeta


def dwconv3x3(x, w, bias, flip: bool = False):
    """x [N,C,P] planes, w [N,3,3] (or [N,1,3,3]), bias [N] or None."""
    _req(x, name="x"); _req(w, name="w")
    n, c, p = x.shape
    y = torch.empty_like(x)
    check(lib().phnet_dwconv3x3(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, c, p, int(flip), _stream()), "phnet_dwconv3x3")
    return y


def dwconv3x3_wgrad(dy, x, dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None, accumulate: bool = False):
    _req(dy, name="dy")
    n, c, p = x.shape
    if dw is None:
        dw = torch.empty((n, 1, 3, 3), dtype=torch.float32, device=x.device)
        db = torch.empty((n,), dtype=torch.float32, device=x.device)
        accumulate = False
    check(lib().phnet_dwconv3x3_wgrad(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), n, c, p, int(accumulate), _stream()),
          "phnet_dwconv3x3_wgrad")
    return dw, db


def relu_bwd(dy, y):
    _req(dy, name="dy")
    dx = torch.empty_like(dy)
    check(lib().phnet_relu_bwd(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()), "phnet_relu_bwd")
    return dx


def lane_assign(pred, tgt, img_w: int, img_h: int, want_cost: bool = False):
    """pred [N,6+S], tgt [L,6+S] -> (rows_by_col [L] i64, rows_sorted [L] i64, n_valid [] i32[, cost [N,L]])."""
    _req(pred, name="pred"); _req(tgt, name="tgt")
    n, w = pred.shape
    L = tgt.shape[0]
    rows = torch.empty(L, dtype=torch.int64, device=pred.device)
    srt = torch.empty(L, dtype=torch.int64, device=pred.device)
    nv = torch.empty((), dtype=torch.int32, device=pred.device)
    cost = torch.empty((n, L), dtype=torch.float32, device=pred.device) if want_cost else None
    check(lib().phnet_lane_assign(_ptr(pred), _ptr(tgt), n, L, w - 6, float(img_w), float(img_h), _ptr(rows), _ptr(srt),
                                  _ptr(nv), _ptr(cost), _stream()), "phnet_lane_assign")
    return (rows, srt, nv, cost) if want_cost else (rows, srt, nv)


def lane_assign_tokens(pred, tgt, img_w: int, img_h: int, feat, out):
    """lane_assign(pred, tgt) + memory_tokens(feat, matched anchors, out=out) in one launch; out = (tokens [L+1,E] view, valid bool [L+1]).
    Returns rows_sorted (i64[L], -1 padded)."""
    _req(pred, name="pred"); _req(tgt, name="tgt"); _req(feat, name="feat")
    n, w = pred.shape
    L = tgt.shape[0]
    e = feat.shape[-1]
    tokens, valid = out
    _req(tokens, name="tokens out")
    if tokens.numel() != (L + 1) * e or valid.numel() != L + 1 or valid.dtype != torch.bool or not valid.is_contiguous() or feat.numel() != n * e:
        raise ValueError("lane_assign_tokens: out buffers do not match (tokens [L+1,E] f32, valid [L+1] bool, contiguous)")
    rows = torch.empty(L, dtype=torch.int64, device=pred.device)
    srt = torch.empty(L, dtype=torch.int64, device=pred.device)
    check(lib().phnet_lane_assign_tokens(_ptr(pred), _ptr(tgt), n, L, w - 6, float(img_w), float(img_h), _ptr(rows), _ptr(srt), _ptr(feat), e,
                                         _ptr(tokens), _ptr(valid), _stream()), "phnet_lane_assign_tokens")
    return srt


def lane_assign_one2many(pred, tgt, img_w: int, img_h: int):
    """pred [N,6+S], tgt [L<=4,6+S] (all label rows, valid flag in column 1) -> (rows i64[16], cols i64[16], n i32[]): the
    (anchor, label row) pairs of dynamic_assign.assignOne2Many in the reference's order, -1 padded; no host sync."""
    _req(pred, name="pred"); _req(tgt, name="tgt")
    n, w = pred.shape
    rows = torch.empty(16, dtype=torch.int64, device=pred.device)
    cols = torch.empty(16, dtype=torch.int64, device=pred.device)
    cnt = torch.empty((), dtype=torch.int32, device=pred.device)
    check(lib().phnet_lane_assign_one2many(_ptr(pred), _ptr(tgt), n, tgt.shape[0], w - 6, float(img_w), float(img_h), _ptr(rows), _ptr(cols),
                                           _ptr(cnt), _stream()), "phnet_lane_assign_one2many")
    return rows, cols, cnt


def frame_loss_variant(variant: int, preds, gates, tgt, img_w, img_h, cls_w, reg_w, iou_w):
    """The loss4OL (variant 1) / loss4OLV2 (variant 2) criterion of one frame in two launches (csrc/loss_variants.hip).
    preds: 6 x [N,6+S], gates: 3 x [N], tgt [L,6+S] -> (loss [1], dpred [6,N,6+S], dgate [3,N], pair_rows [6,16] i64,
    pair_cols [6,16] i64, rows_sorted [6,L] i64)."""
    import ctypes
    for t in list(preds) + list(gates) + [tgt]:
        _req(t, name="loss input")
    n, w = preds[0].shape
    L = tgt.shape[0]
    dev = tgt.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dpred = torch.empty((6, n, w), dtype=torch.float32, device=dev)
    dgate = torch.empty((3, n), dtype=torch.float32, device=dev)
    prow = torch.empty((6, 16), dtype=torch.int64, device=dev)
    pcol = torch.empty((6, 16), dtype=torch.int64, device=dev)
    srt = torch.full((6, L), -1, dtype=torch.int64, device=dev)
    scratch = torch.empty(6 * n + 192, dtype=torch.float32, device=dev)
    P = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in preds])
    G = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in gates])
    D = (ctypes.c_void_p * 6)(*[dpred[i].data_ptr() for i in range(6)])
    check(lib().phnet_frame_loss_variant(int(variant), P, G, _ptr(tgt), n, L, w - 6, float(img_w), float(img_h), float(cls_w), float(reg_w),
                                         float(iou_w), _ptr(loss), D, _ptr(dgate), _ptr(prow), _ptr(pcol), _ptr(srt), _ptr(scratch),
                                         _stream()), "phnet_frame_loss_variant")
    return loss, dpred, dgate, prow, pcol, srt


def frame_loss(preds, gates, tgt, img_w, img_h, cls_w, reg_w, iou_w, liou_hw, liou_h, liou_w):
    """preds: 6 x [N,6+S] (branch A stages 0..2, branch B stages 0..2), gates: 3 x [N], tgt [L,6+S].
    Returns (loss [1], dpred [6,N,6+S], dgate [3,N], rows_by_col [6,L] i64, rows_sorted [6,L] i64)."""
    import ctypes
    for t in list(preds) + list(gates) + [tgt]:
        _req(t, name="loss input")
    n, w = preds[0].shape
    L = tgt.shape[0]
    dev = tgt.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dpred = torch.empty((6, n, w), dtype=torch.float32, device=dev)
    dgate = torch.empty((3, n), dtype=torch.float32, device=dev)
    rows = torch.empty((6, L), dtype=torch.int64, device=dev)
    srt = torch.empty((6, L), dtype=torch.int64, device=dev)
    scratch = torch.empty(6 * n + 12, dtype=torch.float32, device=dev)
    P = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in preds])
    G = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in gates])
    D = (ctypes.c_void_p * 6)(*[dpred[i].data_ptr() for i in range(6)])
    check(lib().phnet_frame_loss(P, G, _ptr(tgt), n, L, w - 6, float(img_w), float(img_h), float(cls_w), float(reg_w),
                                 float(iou_w), float(liou_hw), float(liou_h), float(liou_w), _ptr(loss), D, _ptr(dgate),
                                 _ptr(rows), _ptr(srt), _ptr(scratch), _ptr(scratch[6 * n:]), _stream()), "phnet_frame_loss")
    return loss, dpred, dgate, rows, srt


def clip_loss(preds, gates, tgt, img_w, img_h, cls_w, reg_w, iou_w, liou_hw, liou_h, liou_w):
    """frame_loss for the T frames of a clip in the same two launches.  preds: T x 6 x [N,6+S], gates: T x 3 x [N], tgt [T,L,6+S].
    Returns (loss [T], dpred [T,6,N,6+S], dgate [T,3,N], rows_by_col [T,6,L] i64, rows_sorted [T,6,L] i64)."""
    import ctypes
    T = len(preds)
    flat_p = [t for fr in preds for t in fr]
    flat_g = [t for fr in gates for t in fr]
    for t in flat_p + flat_g + [tgt]:
        _req(t, name="loss input")
    if len(flat_p) != 6 * T or len(flat_g) != 3 * T or tgt.shape[0] != T:
        raise ValueError("clip_loss: 6 predictions and 3 gates per frame, one label block per frame")
    n, w = flat_p[0].shape
    L = tgt.shape[1]
    dev = tgt.device
    loss = torch.empty(T, dtype=torch.float32, device=dev)
    dpred = torch.empty((T, 6, n, w), dtype=torch.float32, device=dev)
    dgate = torch.empty((T, 3, n), dtype=torch.float32, device=dev)
    rows = torch.empty((T, 6, L), dtype=torch.int64, device=dev)
    srt = torch.empty((T, 6, L), dtype=torch.int64, device=dev)
    scratch = torch.empty(T * (6 * n + 12), dtype=torch.float32, device=dev)
    P = (ctypes.c_void_p * (6 * T))(*[t.data_ptr() for t in flat_p])
    G = (ctypes.c_void_p * (3 * T))(*[t.data_ptr() for t in flat_g])
    D = (ctypes.c_void_p * (6 * T))(*[dpred[f, i].data_ptr() for f in range(T) for i in range(6)])
    check(lib().phnet_clip_loss(P, G, _ptr(tgt), T, n, L, w - 6, float(img_w), float(img_h), float(cls_w), float(reg_w),
                                float(iou_w), float(liou_hw), float(liou_h), float(liou_w), _ptr(loss), D, _ptr(dgate),
                                _ptr(rows), _ptr(srt), _ptr(scratch), _ptr(scratch[T * 6 * n:]), _stream()), "phnet_clip_loss")
    return loss, dpred, dgate, rows, srt


def lane_update_fwd(priors, head, ys, img_w, img_h):
    """priors [N,6+S], head [N,HW] -> (preds, lines) [N,6+S]."""
    _req(priors, name="priors"); _req(head, name="head")
    n, w = priors.shape
    preds, lines = torch.empty_like(priors), torch.empty_like(priors)
    check(lib().phnet_lane_update_fwd(_ptr(priors), _ptr(head), _ptr(ys), _ptr(preds), _ptr(lines), n, w - 6, head.shape[1],
                                      float(img_w), float(img_h), _stream()), "phnet_lane_update_fwd")
    return preds, lines


def lane_update_bwd(dpreds, dlines, lines, head, ys, img_w, img_h, need_dpriors: bool):
    n, w = lines.shape
    dhead = torch.empty_like(head)
    dpriors = torch.empty_like(lines) if need_dpriors else None
    check(lib().phnet_lane_update_bwd(_ptr(dpreds), _ptr(dlines), _ptr(lines), _ptr(head), _ptr(ys), _ptr(dhead), _ptr(dpriors),
                                      n, w - 6, head.shape[1], float(img_w), float(img_h), _stream()), "phnet_lane_update_bwd")
    return dhead, dpriors


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def gate_stack_fwd(x, params, eps: float, save: bool, anchors: Optional[int] = None):
    """x [N,C,P] (N a multiple of `anchors`: planes of several frames); params: the 34 tensors in the C-ABI order.
    Returns (out, saved or None)."""
    _req(x, name="x")
    n, c, p = x.shape
    anchors = n if anchors is None else anchors
    out = torch.empty_like(x)
    saved = torch.empty(lib().phnet_gate_stack_saved_floats(n, c, p), dtype=torch.float32, device=x.device) if save else None
    check(lib().phnet_gate_stack_fwd(_ptr(x), _ptr_array(params), _ptr(out), _ptr(saved), n, anchors, c, p, eps, _stream()),
          "phnet_gate_stack_fwd")
    return out, saved


def gate_stack_bwd(gout, x, out, params, saved, grads, eps: float, accumulate: bool, anchors: Optional[int] = None):
    _req(gout, name="gout")
    n, c, p = x.shape
    anchors = n if anchors is None else anchors
    ws = workspace(lib().phnet_gate_stack_bwd_workspace(n, c, p), x.device, 3)
    check(lib().phnet_gate_stack_bwd(_ptr(gout), _ptr(x), _ptr(out), _ptr_array(params), _ptr(saved), _ptr_array(grads),
                                     n, anchors, c, p, eps, int(accumulate), _ptr(ws), ws.numel(), _stream()), "phnet_gate_stack_bwd")


def _rng_args(rng, width: int = 0):
    """rng = None | (state int64[1] device tensor, site id, drop probability[, item0, item_rows]) -> the three C arguments.
    Items (csrc/common.h): item0 = number of the first item of this launch, item_rows = rows per item when the launch covers a
    batch of items (0: the launch is one item); packed into the 64-bit call argument as site | item0 << 20 | item_elems << 32
    with item_elems = item_rows * width (width = elements per row of the tensor the mask is drawn for)."""
    if rng is None:
        return None, 0, 0.0
    state, call, p = rng[:3]
    item0, item_rows = (rng[3], rng[4]) if len(rng) > 3 else (0, 0)
    elems = int(item_rows) * int(width)
    assert 0 <= call < (1 << 20) and 0 <= item0 < (1 << 12) and 0 <= elems < (1 << 32), (call, item0, elems)
    return _ptr(state), int(call) | (int(item0) << 20) | (elems << 32), float(p)


def attention_fwd(q, k, v, heads: int, key_valid=None, keep=None, keep_scale: float = 1.0, rng=None, batch: int = 1):
    """q [Lq,E], k/v [Lk,E] (any row stride, unit column stride) -> (o [Lq,E], lse [H,Lq]).  Dropout of the attention
    weights: explicit `keep` u8[H,Lq,Lk] (+ keep_scale), or `rng` for the in-kernel counter-based mask."""
    e = q.shape[1]
    lq, lk = q.shape[0] // batch, k.shape[0] // batch          # `batch` clips: contiguous row blocks of q and of k / v
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1 and q.shape[0] == lq * batch and k.shape[0] == lk * batch
    out = torch.empty((lq * batch, e), dtype=torch.float32, device=q.device)
    lse = torch.empty((batch * heads, lq), dtype=torch.float32, device=q.device)
    check(lib().phnet_attention_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(key_valid), _ptr(keep), _ptr(out), _ptr(lse), batch, lq, lk, heads, e,
                                    q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(keep_scale), *_rng_args(rng), _stream()),
          "phnet_attention_fwd")
    return out, lse


def attention_bwd(q, k, v, o, dout, lse, heads: int, dq, dk, dv, key_valid=None, keep=None, keep_scale: float = 1.0, rng=None,
                  batch: int = 1):
    """Writes dq/dk/dv (views with arbitrary row stride)."""
    e = q.shape[1]
    lq, lk = q.shape[0] // batch, k.shape[0] // batch
    check(lib().phnet_attention_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(dout), _ptr(lse), _ptr(key_valid), _ptr(keep),
                                    _ptr(dq), _ptr(dk), _ptr(dv), batch, lq, lk, heads, e, q.stride(0), k.stride(0), v.stride(0),
                                    o.stride(0), dq.stride(0), dk.stride(0), dv.stride(0), float(keep_scale), *_rng_args(rng),
                                    _stream()), "phnet_attention_bwd")


def memory_tokens(feat, rows, out=None):
    """feat [N,E] / [N,1,E] with rows i64[L] (-1 padded) -> (tokens [L+1,1,E], valid bool[L+1]); or a batch of clips:
    feat [B,N,E] with rows i64[B,L] -> (tokens [B,L+1,E], valid bool[B,L+1]).  One launch.  out = (tokens, valid): write into
    caller-provided contiguous tensors (a slot of a ring buffer) instead of allocating."""
    _req(feat, name="feat"); _req(rows, torch.int64, "rows")
    batched = rows.dim() == 2
    b = rows.shape[0] if batched else 1
    l, e = rows.shape[-1], feat.shape[-1]
    n = feat.numel() // (b * e)
    if out is not None:
        tokens, valid = out
        _req(tokens, name="tokens out")
        if tokens.numel() != b * (l + 1) * e or valid.numel() != b * (l + 1) or valid.dtype != torch.bool or not valid.is_contiguous():
            raise ValueError("memory_tokens: out buffers do not match (tokens [..,L+1,E] f32, valid [..,L+1] bool, contiguous)")
    else:
        tokens = torch.empty((b, l + 1, e) if batched else (l + 1, 1, e), dtype=torch.float32, device=feat.device)
        valid = torch.empty((b, l + 1) if batched else (l + 1,), dtype=torch.bool, device=feat.device)
    check(lib().phnet_memory_tokens(_ptr(feat), _ptr(rows), _ptr(tokens), _ptr(valid), b, n, e, l, _stream()), "phnet_memory_tokens")
    return tokens, valid


def gate_tail_fwd(h, w, b):
    _req(h, name="h"); _req(w, name="w"); _req(b, name="b")
    n, k = h.shape
    if w.numel() != k or b.numel() != 1:
        raise ValueError(f"gate_tail: h {tuple(h.shape)} vs w {tuple(w.shape)} / b {tuple(b.shape)}")
    out = torch.empty((n,), dtype=torch.float32, device=h.device)
    check(lib().phnet_gate_tail_fwd(_ptr(h), _ptr(w), _ptr(b), _ptr(out), n, k, _stream()), "phnet_gate_tail_fwd")
    return out


def gate_tail_bwd(dout, out, h, w, need_dh: bool = True, dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None,
                  accumulate: bool = False):
    _req(dout, name="dout")
    n, k = h.shape
    dh = torch.empty_like(h) if need_dh else None
    if dw is None:
        dw = torch.empty((k,), dtype=torch.float32, device=h.device)
        db = torch.empty((1,), dtype=torch.float32, device=h.device)
        accumulate = False
    check(lib().phnet_gate_tail_bwd(_ptr(dout), _ptr(out), _ptr(h), _ptr(w), _ptr(dh), _ptr(dw), _ptr(db), n, k, int(accumulate),
                                    _stream()), "phnet_gate_tail_bwd")
    return dh, dw, db


def blend_priors(gate, a, b, idx):
    """gate [N] (or [1,N,1]), a/b [1,N,W], idx i64[P] -> (priors [1,N,W], on_map [1,N,P])."""
    _req(gate, name="gate"); _req(a, name="lines_a"); _req(b, name="lines_b"); _req(idx, torch.int64, "idx")
    w = a.shape[-1]
    n = a.numel() // w                                   # all leading dimensions are rows (frames / clips x anchors)
    p = idx.numel()
    priors = torch.empty_like(a)
    on_map = torch.empty(a.shape[:-1] + (p,), dtype=torch.float32, device=a.device)
    check(lib().phnet_blend_priors(_ptr(gate), _ptr(a), _ptr(b), _ptr(idx), _ptr(priors), _ptr(on_map), n, w, p, _stream()),
          "phnet_blend_priors")
    return priors, on_map


def adamw_step(p, g, m, v, n_decay: int, step, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, lr_dev=None):
    """In-place AdamW over flat fp32 buffers (all the same length, a multiple of 4); step = int64[1] device tensor.
    lr_dev (optional float32[1] device tensor) replaces `lr` (read on the device: capturable LR schedules)."""
    for t, name in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, name=name)
    _req(step, torch.int64, "step")
    if not (p.numel() == g.numel() == m.numel() == v.numel()):
        raise ValueError("adamw_step: p/g/m/v lengths differ")
    if lr_dev is not None:
        _req(lr_dev, name="lr_dev")
    check(lib().phnet_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(n_decay), _ptr(step), float(lr), _ptr(lr_dev),
                                 float(beta1), float(beta2), float(eps), float(weight_decay), _stream()), "phnet_adamw_step")


def dropout_add(x, res=None, rng=None):
    """res + dropout(x) (either part optional) in one launch; dropout_add(dy, None, rng) is the backward of the dropout."""
    _req(x, name="x")
    y = torch.empty_like(x)
    check(lib().phnet_dropout_add(_ptr(x), _ptr(res), _ptr(y), x.numel(), *_rng_args(rng, x.shape[-1]), _stream()), "phnet_dropout_add")
    return y


def dropout_add_ln_fwd(x, res, w, b, eps: float, rng=None, save_stats: bool = True):
    """t = res + dropout(x), h = LayerNorm(t)*w + b over the last w.numel() (<= 256) elements -> (t, h, mean, rstd)."""
    _req(x, name="x"); _req(res, name="res")
    L = w.numel()
    rows = x.numel() // L
    t, h = torch.empty_like(x), torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    check(lib().phnet_dropout_add_ln_fwd(_ptr(x), _ptr(res), _ptr(w), _ptr(b), _ptr(t), _ptr(h), _ptr(mean), _ptr(rstd), rows, L, eps,
                                         *_rng_args(rng, L), _stream()), "phnet_dropout_add_ln_fwd")
    return t, h, mean, rstd


def dropout_add_ln_bwd(dh, dt, t, w, mean, rstd, rng=None, dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None,
                       accumulate: bool = False):
    """Returns (dres, dx, dw, db); dt (gradient on the residual stream) may be None."""
    _req(dh, name="dh")
    L = w.numel()
    rows = t.numel() // L
    dres, dx = torch.empty_like(t), torch.empty_like(t)
    if dw is None:
        dw, db, accumulate = torch.empty_like(w), torch.empty_like(w), False
    ws = workspace(lib().phnet_layernorm_bwd_workspace(rows, L), t.device, 2)
    check(lib().phnet_dropout_add_ln_bwd(_ptr(dh), _ptr(dt), _ptr(t), _ptr(w), _ptr(mean), _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(dw),
                                         _ptr(db), rows, L, int(accumulate), *_rng_args(rng, L), _ptr(ws), ws.numel(), _stream()),
          "phnet_dropout_add_ln_bwd")
    return dres, dx, dw, db


def gelu_dropout_fwd(x, rng=None):
    _req(x, name="x")
    y = torch.empty_like(x)
    check(lib().phnet_gelu_dropout_fwd(_ptr(x), _ptr(y), x.numel(), *_rng_args(rng, x.shape[-1]), _stream()), "phnet_gelu_dropout_fwd")
    return y


def gelu_dropout_bwd(dy, x, rng=None):
    _req(dy, name="dy")
    dx = torch.empty_like(x)
    check(lib().phnet_gelu_dropout_bwd(_ptr(dy), _ptr(x), _ptr(dx), x.numel(), *_rng_args(rng, x.shape[-1]), _stream()), "phnet_gelu_dropout_bwd")
    return dx


def lane_decode(lines, conf_thresh: float, nms_thresh: float, top_k: int, img_w: int):
    """lines [N,6+S] or [F,N,6+S] -> dict(keep_mask u8, num i64, keep_c, anchors, anchors_sorted i64 [..,top_k],
    kept_rows [..,top_k,6+S]); everything stays on the device (no sync)."""
    _req(lines, name="lines")
    batched = lines.dim() == 3
    f = lines.shape[0] if batched else 1
    n, w = lines.shape[-2], lines.shape[-1]
    dev = lines.device
    out = dict(keep_mask=torch.empty((f, n), dtype=torch.uint8, device=dev), num=torch.empty((f,), dtype=torch.int64, device=dev),
               keep_c=torch.empty((f, top_k), dtype=torch.int64, device=dev), anchors=torch.empty((f, top_k), dtype=torch.int64, device=dev),
               anchors_sorted=torch.empty((f, top_k), dtype=torch.int64, device=dev),
               kept_rows=torch.empty((f, top_k, w), dtype=torch.float32, device=dev))
    check(lib().phnet_lane_decode(_ptr(lines), f, n, w - 6, float(conf_thresh), float(nms_thresh), int(top_k), float(img_w),
                                  _ptr(out["keep_mask"]), _ptr(out["num"]), _ptr(out["keep_c"]), _ptr(out["anchors"]),
                                  _ptr(out["anchors_sorted"]), _ptr(out["kept_rows"]), _stream()), "phnet_lane_decode")
    return out if batched else {k: v[0] for k, v in out.items()}


# ------------------------------------------------------------------------------------------------ Router4OLV2 family (inference)
def gate_v2_fwd(x_cp, w1, s1, t1, w2, s2, t2, wl, bl, out: Optional[torch.Tensor] = None):
    """x_cp [M,C,P] -> sigmoid(mean(Linear(flatten(conv-bn-relu x2)))) [M]  (csrc/v2head.hip)."""
    for t, nm in ((x_cp, "x"), (w1, "w1"), (s1, "s1"), (t1, "t1"), (w2, "w2"), (s2, "s2"), (t2, "t2"), (wl, "wl"), (bl, "bl")):
        _req(t, name=nm)
    m, c, p = x_cp.shape
    c1, c2 = w1.shape[0], w2.shape[0]
    if w1.numel() != c1 * c * 3 or w2.numel() != c2 * c1 or wl.numel() != p * c2 * p or bl.numel() != p:
        raise RuntimeError("gate_v2_fwd: parameter shapes do not match the feature map")
    if out is None:
        out = torch.empty((m,), dtype=torch.float32, device=x_cp.device)
    check(lib().phnet_gate_v2_fwd(_ptr(x_cp), _ptr(w1), _ptr(s1), _ptr(t1), _ptr(w2), _ptr(s2), _ptr(t2), _ptr(wl), _ptr(bl), _ptr(out),
                                  m, c, p, c1, c2, _stream()), "phnet_gate_v2_fwd")
    return out


def dyn_bmm_ln_relu_fwd_any(x, w, gamma, beta, eps: float):
    """x [N,P,K], w [N,K,J] -> relu(LayerNorm(x @ w)) [N,P,J] at run-time shapes (forward only)."""
    _req(x, name="x"); _req(w, name="w"); _req(gamma, name="gamma"); _req(beta, name="beta")
    n, p, k = x.shape
    j = w.shape[2]
    assert w.shape[0] == n and w.shape[1] == k and gamma.numel() == j
    y = torch.empty((n, p, j), dtype=torch.float32, device=x.device)
    check(lib().phnet_dyn_bmm_ln_relu_fwd_any(_ptr(x), _ptr(w), _ptr(gamma), _ptr(beta), _ptr(y), n, p, k, j, float(eps), _stream()),
          "phnet_dyn_bmm_ln_relu_fwd_any")
    return y


def route_lines(gates, a, b, hard: bool):
    """gates [S,M] (one row per refinement stage), a / b [M,W] -> [M,W]: hard selection or soft blend by the mean gate."""
    _req(gates, name="gates"); _req(a, name="a"); _req(b, name="b")
    s, m = gates.shape
    w = a.shape[-1]
    assert a.numel() == m * w and b.numel() == m * w
    out = torch.empty_like(a)
    check(lib().phnet_route_lines(_ptr(gates), _ptr(a), _ptr(b), _ptr(out), s, m, w, int(bool(hard)), _stream()), "phnet_route_lines")
    return out


def rowchain_fwd(x, resid=None, wa=None, ba=None, ln1=None, ffn=None, ln2=None, wg=None, bg=None, eps: float = 1e-5,
                 rng_a=None, rng_f=None, rng_3=None, want_t: bool = True, want_h: bool = False):
    """Row-local chain of a transformer layer in one launch (csrc/rowchain.hip), forward only.  x [R,E]; wa / ba: out-projection
    (then t = resid + dropout_a(.)); ln1 = (w, b); ffn = (W1, b1, W2, b2) with ln2 = (w, b); wg / bg: next projection.
    rng_*: DropoutStream.site tuples (None = no dropout at that site; all sites share one probability).  -> (t, h, y), each None
    when not produced."""
    _req(x, name="x")
    r, e = x.shape
    dev = x.device
    t = torch.empty((r, e), dtype=torch.float32, device=dev) if want_t else None
    h = torch.empty((r, e), dtype=torch.float32, device=dev) if want_h else None
    ng = wg.shape[0] if wg is not None else 0
    y = torch.empty((r, ng), dtype=torch.float32, device=dev) if wg is not None else None
    w1 = b1 = w2 = b2 = None
    ff = 0
    if ffn is not None:
        w1, b1, w2, b2 = ffn
        ff = w1.shape[0]
    sites = [s for s in (rng_a, rng_f, rng_3) if s is not None]
    state, p = (sites[0][0], sites[0][2]) if sites else (None, 0.0)

    def call(rng, width):
        return 0 if rng is None else _rng_args(rng, width)[1]
    for tns in (resid, wa, ba, w1, b1, w2, b2, wg, bg) + tuple(ln1) + (tuple(ln2) if ln2 is not None else ()):
        if tns is not None:
            _req(tns, name="rowchain operand")
    check(lib().phnet_rowchain_fwd(_ptr(x), _ptr(resid), _ptr(wa), _ptr(ba), _ptr(ln1[0]), _ptr(ln1[1]), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                                   _ptr(ln2[0]) if ln2 is not None else None, _ptr(ln2[1]) if ln2 is not None else None, _ptr(wg), _ptr(bg),
                                   _ptr(t), _ptr(h), _ptr(y), r, e, ff, ng, float(eps), _ptr(state), call(rng_a, e), call(rng_f, ff), call(rng_3, e),
                                   float(p), _stream()), "phnet_rowchain_fwd")
    return t, h, y


def tower_layout(n_towers: int, c: int, head_out):
    """-> (total floats, offsets[6] of w1 | b1 | w2 | b2 | wh | bh, HW) of the assembled tower weights (csrc/towers.hip)."""
    import ctypes
    ho = (ctypes.c_int32 * n_towers)(*[int(v) for v in head_out])
    offs = (ctypes.c_int64 * 6)()
    hw = ctypes.c_int32()
    total = int(lib().phnet_tower_layout(n_towers, c, ho, offs, ctypes.byref(hw)))
    if total == 0:
        raise RuntimeError("phnet_tower_layout: bad arguments")
    return total, list(offs), int(hw.value)


def assemble_towers(params, n_towers: int, c: int, head_out, dst):
    import ctypes
    ho = (ctypes.c_int32 * n_towers)(*[int(v) for v in head_out])
    check(lib().phnet_assemble_towers(_ptr_array([_req(p, name="tower parameter") for p in params]), n_towers, c, ho, _ptr(dst), _stream()),
          "phnet_assemble_towers")
    return dst


def scatter_tower_grads(src, grads, n_towers: int, c: int, head_out, accumulate: bool):
    import ctypes
    ho = (ctypes.c_int32 * n_towers)(*[int(v) for v in head_out])
    table = (ctypes.c_void_p * len(grads))(*[None if g is None else _req(g, name="tower gradient").data_ptr() for g in grads])
    check(lib().phnet_scatter_tower_grads(_ptr(src), table, n_towers, c, ho, int(accumulate), _stream()), "phnet_scatter_tower_grads")


def tower_chain_fwd(x, params, head_out, priors, ys, img_w, img_h):
    """x [R,C], params = 6*T tower tensors, priors [R,6+S] -> (preds, lines) [R,6+S]: towers + heads + lane prior update in one
    launch (csrc/rowchain.hip), forward only."""
    import ctypes
    _req(x, name="x"); _req(priors, name="priors")
    r, c = x.shape
    t = len(params) // 6
    ho = (ctypes.c_int32 * t)(*[int(v) for v in head_out])
    preds, lines = torch.empty_like(priors), torch.empty_like(priors)
    check(lib().phnet_tower_chain_fwd(_ptr(x), _ptr_array([_req(p, name="tower parameter") for p in params]), t, c, ho, _ptr(priors), _ptr(ys),
                                      _ptr(preds), _ptr(lines), r, priors.shape[1] - 6, float(img_w), float(img_h), _stream()),
          "phnet_tower_chain_fwd")
    return preds, lines


DEFAULT_MMA = "bf16x3"


def lane_raster(segs, n_lanes: int, height: int, width: int, lane_width: int):
    """Bit masks [n_lanes][height][ceil(width/32)] (int32 words) of the thick poly-lines: segs [S][5] int32 = (x0, y0, x1, y1, lane)
    (csrc/lane_iou.hip; the evaluator's cv::line canvases, evaluation/culane/src/lane_compare.cpp:17-49)."""
    _req(segs, torch.int32, "segs")
    assert segs.dim() == 2 and segs.shape[1] == 5
    masks = torch.zeros((n_lanes, height, (width + 31) // 32), dtype=torch.int32, device=segs.device)
    check(lib().phnet_lane_raster(_ptr(segs), segs.shape[0], _ptr(masks), n_lanes, height, width, lane_width, _stream()), "phnet_lane_raster")
    return masks


def lane_mask_stats(masks, pairs, width: int):
    """(area [n_lanes], inter [n_pairs]) int64: set bits of every mask and of masks[pairs[p, 0]] & masks[pairs[p, 1]]
    (lane_compare.cpp:51-55: cv::sum of the canvases and of their product)."""
    _req(masks, torch.int32, "masks"); _req(pairs, torch.int32, "pairs")
    n_lanes, height = masks.shape[0], masks.shape[1]
    assert masks.shape[2] == (width + 31) // 32 and pairs.dim() == 2 and pairs.shape[1] == 2
    area = torch.zeros(n_lanes, dtype=torch.int64, device=masks.device)
    inter = torch.zeros(pairs.shape[0], dtype=torch.int64, device=masks.device)
    check(lib().phnet_lane_mask_stats(_ptr(masks), n_lanes, height, width, _ptr(pairs), pairs.shape[0], _ptr(area), _ptr(inter),
                                      _stream()), "phnet_lane_mask_stats")
    return area, inter


def lane_mask_iou(segs, n_lanes: int, pairs, height: int, width: int, lane_width: int):
    """Areas of the drawn lanes and the pairwise intersections the IoU matrices need, two launches."""
    return lane_mask_stats(lane_raster(segs, n_lanes, height, width, lane_width), pairs, width)


def tune_k_tile(code: int) -> None:
    """Benchmark aid (process-global): phnet_tune_force_k_tile; -5 / -6 switch the three-taps 3x3 forward / dgrad kernel off / on."""
    global _TAPS3
    check(lib().phnet_tune_force_k_tile(code), "phnet_tune_force_k_tile")
    if code in (-5, -6):
        _TAPS3 = code == -6


def tune_wgrad(flags: int = 1, target: int = 768) -> None:
    """Benchmark aid (process-global): phnet_tune_wgrad - bit 3 of `flags` switches the three-taps 3x3 weight-gradient kernel off,
    bit 4 gives it 32-pixel steps, bit 5 switches its producer / consumer variant (csrc/wgrad3s.hip) off; a negative `target` is ITS workgroup target, a positive one the generic kernel's."""
    global _WGRAD3, _WGRAD3S, _WGRAD1S
    check(lib().phnet_tune_wgrad(flags, target), "phnet_tune_wgrad")
    _WGRAD3 = not (flags & 8)
    _WGRAD3S = not (flags & 32)
    _WGRAD1S = not (flags & 64)


def set_mma_mode(mode: str) -> None:
    """Arithmetic of the conv / linear GEMM kernels (process-global; set it before a step is captured in a hipGraph):
    "bf16x3" (default) - see below; "f32" = f32-input MFMA (v_mfma_f32_32x32x2_f32, the round-1 default); "split_bf16" = operands split into two bf16 terms in registers, 3 bf16 MFMAs per
    product, f32 accumulation (~4x the rounding noise of "f32"); "split3_bf16" = three bf16 terms (an exact split of the
    f32 value), 6 bf16 MFMAs per product, dropped terms <= 2^-24: the accuracy of "f32" (csrc/igemm.h); "bf16x3" = the same
    exact three-term arithmetic with the split done ONCE while a tile is staged into LDS (bf16 planes, transposed fragment
    reads): the fast form of "split3_bf16"."""
    global _MMA_MODE
    code = {"f32": 0, "split_bf16": 1, "split3_bf16": 2, "bf16x3": 3}[mode]
    check(lib().phnet_tune_mma(code), "phnet_tune_mma")
    _MMA_MODE = code
