"""Autograd nodes of the lane head, each one a thin shell around C-ABI kernels (phnet_amd/hip_ops.py).

Forward and backward of every node run hand-written HIP kernels; torch.autograd only orders the nodes.
"""
from typing import Optional

import torch

from . import hip_ops as K
from .arena import direct_grad


class _Linear(torch.autograd.Function):
    """y = relu?(x @ w.T + b) on the fp32-MFMA GEMM (replaces F.linear / addmm).
    Output widths that are not a multiple of 4 (the 1/2-wide score heads) are zero-padded to 4 for the kernel."""

    @staticmethod
    def forward(ctx, x, w, b, relu, rows=None):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w_full, b_full = w, b
        ctx.rows, ctx.full_n = rows, w.shape[0]
        if rows is not None:                       # a row block of a packed projection (MultiheadAttention.in_proj_*)
            w = w[rows[0]:rows[1]]
            b = None if b is None else b[rows[0]:rows[1]]
        n = w.shape[0]
        npad = (-n) % 4
        wc = w.contiguous() if npad == 0 else torch.cat([w, w.new_zeros(npad, w.shape[1])], 0)
        bc = None if b is None else (b.contiguous() if npad == 0 else torch.cat([b, b.new_zeros(npad)], 0))
        y = K.linear_fwd(x2, wc, bc, relu)
        ctx.save_for_backward(x2, wc, y if relu else None)
        ctx.relu, ctx.has_b, ctx.xshape, ctx.n = relu, b is not None, x.shape, n
        # arena-backed parameters: the backward kernels accumulate straight into .grad (phnet_amd/arena.py)
        ctx.w_direct = direct_grad(w_full) if npad == 0 else None
        ctx.b_direct = direct_grad(b_full) if (npad == 0 and b is not None) else None
        if rows is not None:
            ctx.w_direct = None if ctx.w_direct is None else ctx.w_direct[rows[0]:rows[1]]
            ctx.b_direct = None if ctx.b_direct is None else ctx.b_direct[rows[0]:rows[1]]
        out = y if npad == 0 else y[:, :n]
        return out.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        n, n4 = ctx.n, w.shape[0]
        g = dy.reshape(-1, n)
        if n4 != n:
            g = torch.cat([g, g.new_zeros(g.shape[0], n4 - n)], 1)
        g = g.contiguous()
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.w_direct is not None and
                (not want_b or ctx.b_direct is not None) and K.linear_bwd_fusable(g.shape[0], x2.shape[1], n4)):
            # few-rows layer with arena / sink destinations: ReLU mask, data, weight and bias gradient from ONE launch
            dx = K.linear_bwd(g, x2, w, ctx.w_direct, ctx.b_direct if want_b else None, accumulate=True,
                              relu_y=y if ctx.relu else None)
            return dx.view(ctx.xshape), None, None, None, None
        if ctx.relu:
            g = K.relu_bwd(g, y)
        dx = K.linear_dgrad(g, w).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            # weight and bias gradient come out of the same launch (the bias gradient is the column sum of g)
            if ctx.w_direct is not None and (not want_b or ctx.b_direct is not None):
                K.linear_wgrad(g, x2, dw=ctx.w_direct, accumulate=True, dbias=ctx.b_direct if want_b else None)
                want_b = False
            else:
                db_full = torch.empty(n4, dtype=torch.float32, device=g.device) if want_b else None
                dw = K.linear_wgrad(g, x2, dbias=db_full)[:n]
                if want_b:
                    db, want_b = db_full[:n], False
        if want_b:
            if ctx.b_direct is not None:
                K.colsum(g, out=ctx.b_direct, accumulate=True)
            else:
                db = K.colsum(g)[:n]
        if ctx.rows is not None:                   # no arena: hand autograd full-size gradients of the packed parameter
            r0, r1 = ctx.rows
            if dw is not None:
                full = dw.new_zeros(ctx.full_n, dw.shape[1]); full[r0:r1] = dw; dw = full
            if db is not None:
                full = db.new_zeros(ctx.full_n); full[r0:r1] = db; db = full
        return dx, dw, db, None, None


def linear(x, w, b=None, relu: bool = False, rows=None):
    """y = relu?(x @ w[rows].T + b[rows]); `rows=(r0, r1)` selects a row block of a packed weight without an autograd slice."""
    return _Linear.apply(x, w, b, relu, rows)


class _LayerNorm(torch.autograd.Function):
    """y = relu?(LN(x)*w + b (+res)) over the trailing w.numel() elements."""

    @staticmethod
    def forward(ctx, x, w, b, res, relu, eps):
        xc = x.contiguous()
        wf, bf = w.reshape(-1).contiguous(), b.reshape(-1).contiguous()
        rc = None if res is None else res.contiguous()
        y, mean, rstd = K.layernorm_fwd(xc, wf, bf, eps, rc, relu)
        ctx.save_for_backward(xc, wf, y if relu else None, mean, rstd)
        ctx.relu, ctx.has_res, ctx.wshape = relu, res is not None, w.shape
        ctx.w_direct, ctx.b_direct = direct_grad(w), direct_grad(b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, mean, rstd = ctx.saved_tensors
        if ctx.w_direct is not None and ctx.b_direct is not None:
            dx, dres, _, _ = K.layernorm_bwd(dy.contiguous(), x, y, w, mean, rstd, ctx.relu, ctx.has_res,
                                             dw=ctx.w_direct.view(-1), db=ctx.b_direct.view(-1), accumulate=True)
            return dx, None, None, dres, None, None
        dx, dres, dw, db = K.layernorm_bwd(dy.contiguous(), x, y, w, mean, rstd, ctx.relu, ctx.has_res)
        return dx, dw.view(ctx.wshape), db.view(ctx.wshape), dres, None, None


def layer_norm(x, w, b, res=None, relu: bool = False, eps: float = 1e-5):
    return _LayerNorm.apply(x, w, b, res, relu, eps)


class _DynBmmLnRelu(torch.autograd.Function):
    """relu(LayerNorm(x[n] @ w[n])) per anchor as one launch each way (dynamic_head.py:40-51)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, eps):
        xc, wc = x.contiguous(), w.contiguous()
        gc, bc = gamma.contiguous(), beta.contiguous()
        need = any(ctx.needs_input_grad)
        y, stats = K.dyn_bmm_ln_relu_fwd(xc, wc, gc, bc, eps, save_stats=need)
        if need:
            ctx.save_for_backward(xc, wc, y, stats, gc)
        ctx.eps = eps
        ctx.g_direct, ctx.b_direct = direct_grad(gamma), direct_grad(beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, stats, gamma = ctx.saved_tensors
        if ctx.g_direct is not None and ctx.b_direct is not None:
            dx, dw, _, _ = K.dyn_bmm_ln_relu_bwd(dy.contiguous(), x, w, y, stats, gamma, ctx.eps, ctx.needs_input_grad[0],
                                                 dgamma=ctx.g_direct.view(-1), dbeta=ctx.b_direct.view(-1), accumulate=True)
            return dx, dw, None, None, None
        dx, dw, dg, db = K.dyn_bmm_ln_relu_bwd(dy.contiguous(), x, w, y, stats, gamma, ctx.eps, ctx.needs_input_grad[0])
        return dx, dw, dg, db, None


def dyn_bmm_ln_relu(x, w, gamma, beta, eps: float = 1e-5):
    return _DynBmmLnRelu.apply(x, w, gamma, beta, eps)


class _DwConv(torch.autograd.Function):
    """Per-anchor depth-wise 3x3 over [N,C,P] planes (Conv2d(N,N,3,padding=1,groups=N) on [1,N,C,P])."""

    @staticmethod
    def forward(ctx, x, w, b):
        xc, wc = x.contiguous(), w.contiguous()
        ctx.save_for_backward(xc, wc)
        ctx.w_direct, ctx.b_direct = direct_grad(w), direct_grad(b)
        return K.dwconv3x3(xc, wc, b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        g = dy.contiguous()
        dx = K.dwconv3x3(g, w, None, flip=True) if ctx.needs_input_grad[0] else None
        if ctx.w_direct is not None and ctx.b_direct is not None:
            K.dwconv3x3_wgrad(g, x, dw=ctx.w_direct, db=ctx.b_direct, accumulate=True)
            return dx, None, None
        dw, db = K.dwconv3x3_wgrad(g, x)
        return dx, dw.view_as(w), db


def dwconv3x3(x, w, b):
    return _DwConv.apply(x, w, b)


class _RoiPool(torch.autograd.Function):
    """Lane-anchor ROI pooling on an NHWC map: (fmap [1,h,w,C], xs [1,N,P]) -> roi [1,N,P,C]; also returns the
    gate's [1,N,C,P] copy (no gradient: the reference feeds the gate a detached tensor, Router4OL.py:275)."""

    @staticmethod
    def forward(ctx, fmap, xs, ys):
        ctx.set_materialize_grads(False)
        fm, xc = fmap.contiguous(), xs.contiguous()
        roi, roi_cp = K.roi_pool_fwd(fm, xc, ys, with_cp=True)
        ctx.save_for_backward(fm, xc, ys)
        ctx.mark_non_differentiable(roi_cp)
        return roi, roi_cp

    @staticmethod
    def backward(ctx, droi, _unused):
        if droi is None:
            return None, None, None
        fm, xs, ys = ctx.saved_tensors
        dmap = torch.zeros_like(fm) if ctx.needs_input_grad[0] else None
        dxs = K.roi_pool_bwd(droi.contiguous(), fm, xs, ys, dmap, ctx.needs_input_grad[1])
        return dmap, dxs, None


def roi_pool(fmap, xs, ys):
    return _RoiPool.apply(fmap, xs, ys)


class _GateStack(torch.autograd.Function):
    """Gate depth-wise stack (pre_norm + 4 residual dw blocks) as one fused launch each way.  The input is the detached
    ROI copy (no input gradient, Router4OL.py:275); parameter gradients are accumulated straight into the gradient
    arena when every parameter is arena-backed, otherwise returned to autograd."""

    @staticmethod
    def forward(ctx, x, eps, anchors, *params):
        xc = x.contiguous()
        pc = [p.contiguous() for p in params]
        need = any(ctx.needs_input_grad[3:])          # (grad mode is off inside forward; ask autograd instead)
        out, saved = K.gate_stack_fwd(xc, pc, eps, need, anchors)
        ctx.eps, ctx.anchors = eps, anchors
        ctx.direct = [direct_grad(p) for p in params]
        ctx.pshapes = [p.shape for p in params]
        ctx.save_for_backward(xc, out, saved, *pc)
        return out

    @staticmethod
    def backward(ctx, gout):
        xc, out, saved, *pc = ctx.saved_tensors
        if all(d is not None for d in ctx.direct):
            K.gate_stack_bwd(gout.contiguous(), xc, out, pc, saved, ctx.direct, ctx.eps, True, ctx.anchors)
            return (None, None, None) + (None,) * len(pc)
        grads = [torch.empty_like(p) for p in pc]
        K.gate_stack_bwd(gout.contiguous(), xc, out, pc, saved, grads, ctx.eps, False, ctx.anchors)
        return (None, None, None) + tuple(g.view(s) for g, s in zip(grads, ctx.pshapes))


def gate_stack(x, params, eps: float = 1e-5, anchors=None):
    """x [N,C,P] planes; `anchors`: number of per-anchor filters when the planes cover several frames (N % anchors == 0)."""
    return _GateStack.apply(x, eps, anchors, *params)


class _LaneUpdate(torch.autograd.Function):
    """(priors [1,N,6+S], head [1,N,HW]) -> (preds, lines): prior update of both branches in one launch each way."""

    @staticmethod
    def forward(ctx, priors, head, ys, img_w, img_h):
        ctx.set_materialize_grads(False)           # `lines` usually gets no gradient: no zero tensors for it, please
        p2 = priors.reshape(-1, priors.shape[-1]).contiguous()
        h2 = head.reshape(-1, head.shape[-1]).contiguous()
        preds, lines = K.lane_update_fwd(p2, h2, ys, img_w, img_h)
        ctx.save_for_backward(lines, h2, ys)
        ctx.geom, ctx.pshape, ctx.hshape = (img_w, img_h), priors.shape, head.shape
        return preds.view(priors.shape), lines.view(priors.shape)

    @staticmethod
    def backward(ctx, dpreds, dlines):
        lines, h2, ys = ctx.saved_tensors
        dp = None if dpreds is None else dpreds.reshape(lines.shape).contiguous()
        dl = None if dlines is None else dlines.reshape(lines.shape).contiguous()
        if dp is None and dl is None:
            return None, None, None, None, None
        dhead, dpri = K.lane_update_bwd(dp, dl, lines, h2, ys, ctx.geom[0], ctx.geom[1], ctx.needs_input_grad[0])
        return (None if dpri is None else dpri.view(ctx.pshape)), dhead.view(ctx.hshape), None, None, None


def lane_update(priors, head, ys, img_w, img_h):
    return _LaneUpdate.apply(priors, head, ys, img_w, img_h)


class _GateTail(torch.autograd.Function):
    """sigmoid(relu(h @ w.T + b)) for the 1-wide last layer of the routing gate: one launch each way instead of a padded
    GEMM + sigmoid and their backward chain."""

    @staticmethod
    def forward(ctx, h, w, b):
        hc, wc, bc = h.contiguous(), w.contiguous().view(-1), b.contiguous()
        out = K.gate_tail_fwd(hc, wc, bc)
        ctx.save_for_backward(hc, wc, out)
        ctx.wshape = w.shape
        ctx.w_direct, ctx.b_direct = direct_grad(w), direct_grad(b)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, w, out = ctx.saved_tensors
        if ctx.w_direct is not None and ctx.b_direct is not None:
            dh, _, _ = K.gate_tail_bwd(dout.contiguous(), out, h, w, ctx.needs_input_grad[0],
                                       dw=ctx.w_direct.view(-1), db=ctx.b_direct, accumulate=True)
            return dh, None, None
        dh, dw, db = K.gate_tail_bwd(dout.contiguous(), out, h, w, ctx.needs_input_grad[0])
        return dh, dw.view(ctx.wshape), db


def gate_tail(h, w, b):
    return _GateTail.apply(h, w, b)


class DropoutStream:
    """Counter-based dropout masks (csrc/common.h): int64 device counters bumped once per training step, and a host-side id
    per dropout site of the step.  Kernels hash (counter, site, element) - no mask tensors, no RNG launches, and a
    replayed hipGraph still draws fresh masks because the counter bump is part of the captured step.  The counters form a
    ring of SLOTS (slot s only ever holds values = s mod SLOTS), so the masks of a forward stay reproducible for its
    backward even if up to SLOTS-1 further forwards run in between."""
    SLOTS = 8
    _state = {}
    _slot = {}
    _calls = 0

    @classmethod
    def _ring(cls, device) -> torch.Tensor:
        key = torch.device(device).index
        if key not in cls._state:
            base = (torch.initial_seed() & 0xFFFFFFFFFFFF) * cls.SLOTS
            cls._state[key] = torch.arange(cls.SLOTS, dtype=torch.int64, device=device) + base
            cls._slot[key] = 0
        return cls._state[key]

    @classmethod
    def begin_step(cls, device):
        ring, key = cls._ring(device), torch.device(device).index
        cls._slot[key] = (cls._slot[key] + 1) % cls.SLOTS
        ring[cls._slot[key]:cls._slot[key] + 1].add_(cls.SLOTS)
        cls._calls = 0

    _item = None          # (item0, item_rows) while inside items(): see csrc/common.h "Items"

    @classmethod
    def site(cls, device, p: float):
        """rng tuple (counter view, site id, p[, item0, item_rows]) for one dropout site, or None when p == 0."""
        if p <= 0.0:
            return None
        ring, key = cls._ring(device), torch.device(device).index
        cls._calls += 1
        rng = (ring[cls._slot[key]:cls._slot[key] + 1], cls._calls, float(p))
        return rng if cls._item is None else rng + cls._item

    @classmethod
    def items(cls, base: int, item0: int = 0, item_rows: int = 0):
        """Context: the dropout sites inside are numbered base+1, base+2, ... in program order and draw their masks per ITEM:
        a pass over ONE item (item0 = its number, item_rows = 0) and a pass over a batch of items (item0 = 0, item_rows = rows
        per item) that visit the same sites in the same order see the same masks.  This is what lets the (frame, stage) passes
        of branch B run one by one in the forward and as one batch in the backward (libs/models/Router4OL.py)."""
        return _ItemScope(cls, base, item0, item_rows)


class _ItemScope:
    def __init__(self, stream, base, item0, item_rows):
        self.stream, self.base, self.item = stream, int(base), (int(item0), int(item_rows))

    def __enter__(self):
        self.saved = (self.stream._calls, self.stream._item)
        self.stream._calls, self.stream._item = self.base, self.item
        return self

    def __exit__(self, *exc):
        self.stream._calls, self.stream._item = self.saved
        return False


class _DropoutAdd(torch.autograd.Function):
    """res + dropout(x) as one launch (transformer residuals); p = 0 is a plain residual add."""

    @staticmethod
    def forward(ctx, res, x, p):
        ctx.rng = DropoutStream.site(x.device, p)
        return K.dropout_add(x.contiguous(), res.contiguous(), ctx.rng)

    @staticmethod
    def backward(ctx, dy):
        dx = dy if ctx.rng is None else K.dropout_add(dy.contiguous(), None, ctx.rng)
        return dy, dx, None


def dropout_add(res, x, p: float = 0.0):
    return _DropoutAdd.apply(res, x, p)


class _DropoutAddLN(torch.autograd.Function):
    """(t, h) = (res + dropout(x), LayerNorm(t)): the residual update of a pre-norm decoder block together with the
    LayerNorm that opens the next block, one launch forward, one (+ the small finalize) backward - the backward also
    absorbs the fan-in add of the two consumers of t."""

    @staticmethod
    def forward(ctx, res, x, w, b, p, eps):
        ctx.set_materialize_grads(False)
        ctx.rng = DropoutStream.site(x.device, p)
        wf, bf = w.reshape(-1).contiguous(), b.reshape(-1).contiguous()
        need = any(ctx.needs_input_grad)
        t, h, mean, rstd = K.dropout_add_ln_fwd(x.contiguous(), res.contiguous(), wf, bf, eps, ctx.rng, save_stats=need)
        if need:
            ctx.save_for_backward(t, wf, mean, rstd)
        ctx.wshape = w.shape
        ctx.w_direct, ctx.b_direct = direct_grad(w), direct_grad(b)
        return t, h

    @staticmethod
    def backward(ctx, dt, dh):
        t, w, mean, rstd = ctx.saved_tensors
        if dh is None:                                            # only the residual stream was used downstream
            if dt is None:
                return None, None, None, None, None, None
            dx = dt if ctx.rng is None else K.dropout_add(dt.contiguous(), None, ctx.rng)
            return dt, dx, None, None, None, None
        dtc = None if dt is None else dt.contiguous()
        if ctx.w_direct is not None and ctx.b_direct is not None:
            dres, dx, _, _ = K.dropout_add_ln_bwd(dh.contiguous(), dtc, t, w, mean, rstd, ctx.rng,
                                                  dw=ctx.w_direct.view(-1), db=ctx.b_direct.view(-1), accumulate=True)
            return dres, dx, None, None, None, None
        dres, dx, dw, db = K.dropout_add_ln_bwd(dh.contiguous(), dtc, t, w, mean, rstd, ctx.rng)
        return dres, dx, dw.view(ctx.wshape), db.view(ctx.wshape), None, None


def dropout_add_ln(res, x, w, b, p: float = 0.0, eps: float = 1e-5):
    """Returns (res + dropout(x), LayerNorm(res + dropout(x)) * w + b)."""
    return _DropoutAddLN.apply(res, x, w, b, p, eps)


class _GeluDropout(torch.autograd.Function):
    """dropout(gelu(x)), exact (erf) GELU, one launch each way."""

    @staticmethod
    def forward(ctx, x, p):
        xc = x.contiguous()
        ctx.rng = DropoutStream.site(x.device, p)
        ctx.save_for_backward(xc)
        return K.gelu_dropout_fwd(xc, ctx.rng)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.gelu_dropout_bwd(dy.contiguous(), x, ctx.rng), None


def gelu_dropout(x, p: float = 0.0):
    return _GeluDropout.apply(x, p)


class _Attention(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v per head on the fused HIP kernels.  `packed` is either the [L,3E] output of the
    self-attention projection (q|k|v column blocks) or None; otherwise q [Lq,E] and kv [M,2E] (k|v) are separate.
    Gradients are written straight into packed buffers of the same layout."""

    @staticmethod
    def forward(ctx, heads, dropout_p, key_valid, q_in, kv_in, batch=1):
        e = q_in.shape[1] // 3 if kv_in is None else q_in.shape[1]
        q_in = q_in.contiguous()
        if kv_in is None:                      # self-attention, packed projection
            q, k, v = q_in[:, :e], q_in[:, e:2 * e], q_in[:, 2 * e:]
        else:
            kv_in = kv_in.contiguous()
            q, k, v = q_in, kv_in[:, :e], kv_in[:, e:]
        rng = DropoutStream.site(q.device, dropout_p)          # attention-weight dropout drawn inside the kernels
        kvu8 = None if key_valid is None else key_valid.contiguous().view(torch.uint8)      # bool is one byte: no conversion launch
        out, lse = K.attention_fwd(q, k, v, heads, kvu8, rng=rng, batch=batch)
        ctx.save_for_backward(q_in, kv_in, out, lse, kvu8)
        ctx.heads, ctx.rng, ctx.e, ctx.batch = heads, rng, e, batch
        return out

    @staticmethod
    def backward(ctx, dout):
        q_in, kv_in, out, lse, kvu8 = ctx.saved_tensors
        e = ctx.e
        dq_in = torch.empty_like(q_in)
        if kv_in is None:
            q, k, v = q_in[:, :e], q_in[:, e:2 * e], q_in[:, 2 * e:]
            dq, dk, dv = dq_in[:, :e], dq_in[:, e:2 * e], dq_in[:, 2 * e:]
            dkv_in = None
        else:
            dkv_in = torch.empty_like(kv_in)
            q, k, v = q_in, kv_in[:, :e], kv_in[:, e:]
            dq, dk, dv = dq_in, dkv_in[:, :e], dkv_in[:, e:]
        K.attention_bwd(q, k, v, out, dout.contiguous(), lse, ctx.heads, dq, dk, dv, kvu8, rng=ctx.rng, batch=ctx.batch)
        return None, None, None, dq_in, dkv_in, None


def attention_packed(qkv: torch.Tensor, heads: int, dropout_p: float = 0.0, batch: int = 1) -> torch.Tensor:
    """Self-attention on a packed [B*L,3E] projection (B clips = contiguous row blocks, attention inside each block)."""
    return _Attention.apply(heads, dropout_p, None, qkv, None, batch)


def attention_cross(q: torch.Tensor, kv: torch.Tensor, heads: int, dropout_p: float = 0.0,
                    key_valid: Optional[torch.Tensor] = None, batch: int = 1) -> torch.Tensor:
    """Cross-attention: q [B*Lq,E], kv [B*M,2E] packed (k|v), optional bool key mask [B*M] (per clip)."""
    return _Attention.apply(heads, dropout_p, key_valid, q, kv, batch)


class _AssembleTowers(torch.autograd.Function):
    """The T towers of a lane-head branch as the operands of ONE 3-GEMM chain (csrc/towers.hip): one launch writes the six assembled
    tensors (views of one buffer); the weight-gradient kernels of their uses accumulate straight into a zero-initialised twin of
    that buffer (arena.direct_grad finds it through `_phnet_sink`), and the backward scatters the twin into the parameters' own
    gradient buffers with one launch - instead of ~12 torch.cat / block_diag launches forward and ~40 slice / add_ launches
    backward per branch and clip."""

    @staticmethod
    def forward(ctx, c, head_out, pool, track, *params):
        ctx.set_materialize_grads(False)
        t = len(params) // 6
        total, offs, hw = K.tower_layout(t, c, head_out)
        dst = torch.empty(total, dtype=torch.float32, device=params[0].device)
        K.assemble_towers(params, t, c, head_out, dst)
        tc = t * c
        shapes = [(tc, c), (tc,), (tc, tc), (tc,), (hw, tc), (hw,)]
        ends = offs[1:] + [total]
        outs = tuple(dst[o:e].view(sh) for o, e, sh in zip(offs, ends, shapes))
        ctx.meta, ctx.params, ctx.buf = (t, c, list(head_out), offs, ends, shapes), params, None
        if track:
            ctx.buf = pool.take(dst) if pool is not None else torch.zeros_like(dst)
        return outs

    @staticmethod
    def backward(ctx, *gs):
        from .arena import direct_grad
        t, c, head_out, offs, ends, shapes = ctx.meta
        src = ctx.buf
        for g, o, e in zip(gs, offs, ends):                                    # gradients that came through autograd after all
            if g is not None:
                src = src.clone() if src is ctx.buf else src
                src[o:e] += g.reshape(-1)
        dests = [direct_grad(p) for p in ctx.params]
        need = ctx.needs_input_grad[4:]
        if all(d is not None for d, n in zip(dests, need) if n):
            K.scatter_tower_grads(src, [d if n else None for d, n in zip(dests, need)], t, c, head_out, accumulate=True)
            return (None, None, None, None) + (None,) * len(ctx.params)
        outs = [torch.empty_like(p, memory_format=torch.contiguous_format) if n else None for p, n in zip(ctx.params, need)]
        K.scatter_tower_grads(src, outs, t, c, head_out, accumulate=False)
        return (None, None, None, None) + tuple(outs)


def assemble_towers(c: int, head_out, pool, params):
    """-> (w1, b1, w2, b2, wh, bh) for PF.linear chains; each carries its gradient sink when autograd is on."""
    track = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    outs = _AssembleTowers.apply(c, tuple(int(v) for v in head_out), pool, track, *params)
    if track:
        node = outs[0].grad_fn                                                # the Function's ctx: holds the sink buffer
        _, _, _, offs, ends, shapes = node.meta
        for o_, a, e, sh in zip(outs, offs, ends, shapes):
            o_._phnet_sink = node.buf[a:e].view(sh)
    return outs
