"""Hand-scheduled forward/backward of the ResNet trunk + FPN neck on the HIP kernels (NHWC, fp32 MFMA).

The schedule below launches the C-ABI kernels directly and keeps its own tape.  Mirrors, as a schedule:
  libs/models/resnet.py:79-95 (BasicBlock), :293-307 (ResNet.forward)
  libs/models/fpn.py:109-163  (FPN.forward: drop layer1, 1x1 laterals, nearest top-down add, 3x3 outputs)
The nn.Conv2d / nn.BatchNorm2d objects handed in are parameter containers only (state_dict compatibility);
their own forward is never called.  Under nn.SyncBatchNorm containers (trainOL.py:141) the batch statistics are
all-reduced across ranks before normalisation (device-resident: hip_ops.bn_fwd_sync / bn_bwd_sync).

Two ways to drive it:
  * `EncoderFunction` - ONE autograd node for the whole encoder: PyTorch's autograd only sees
    (frames, parameters) -> (P3, P4, P5); this is what `RouterOL.forward` uses behind the reference's API
    (`loss.backward()` in the caller, DistributedDataParallel hooks, ...).
  * staged (`Encoder.staged = True`, used by phnet_amd.graphed.GraphedTrainStep for data-parallel steps): the forward
    runs outside autograd and hands out leaf tensors, the caller runs the lane head's backward and then calls
    `encoder_backward_staged` itself - from the main thread, so the schedule may stop between two layers for a collective
    (SyncBatchNorm exchange, a gradient bucket) and the step can still be recorded as a chain of hipGraphs.
`stage_done` callbacks fire when all parameter gradients of a part of the model are final ("head" at entry - the lane
head's backward is complete before the trunk's starts - then "neck", "layer4" ... "layer1", "stem"): the overlapped
gradient reduction (parallel.BucketReducer) hangs off them.
"""
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from . import hip_ops as K
from .arena import direct_grad

# called with a part name when the gradients of that part are final (set by GraphedTrainStep / bench.py for N > 1)
STAGE_DONE_HOOK: Optional[Callable[[str], None]] = None
PARTS = ("head", "neck", "layer4", "layer3", "layer2", "layer1", "stem")       # order in which gradients become final


def ohwi(w: torch.Tensor) -> torch.Tensor:
    """OIHW-logical conv weight -> contiguous [Co,R,S,Ci] view (free when the parameter is channels_last)."""
    return w.detach().permute(0, 2, 3, 1).contiguous()


def oihw_grad(dw_ohwi: torch.Tensor) -> torch.Tensor:
    return dw_ohwi.permute(0, 3, 1, 2)


def _sync_bn(bn) -> bool:
    from . import parallel
    return isinstance(bn, nn.SyncBatchNorm) and parallel.active(getattr(bn, "process_group", None))


class _ConvBN:
    """Tape record of conv -> BN(+residual)(+ReLU)."""
    __slots__ = ("conv", "bn", "stride", "pad", "x_in", "w", "c", "y", "sm", "si", "relu", "has_res", "sync_count", "packed")


def _is3x3s1(conv: nn.Conv2d) -> bool:
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.in_channels % 16 == 0 and conv.out_channels % 64 == 0 and conv.in_channels % 64 == 0)


def packed_path() -> bool:
    """The packed-weight 3x3 kernel (csrc/conv3p.hip) runs in the default GEMM arithmetic only."""
    return K._MMA_MODE == 3 and K.CONV3P


def _pack_plan(enc, convs):
    """Packed images (forward + data gradient) of every 3x3 / stride-1 weight of the encoder, refreshed by ONE launch per
    training step; {id(conv): {False: forward image, True: dgrad image}}."""
    weights = [ohwi(c.weight) for c in convs]
    plan = enc.__dict__.get("_phnet_pack_plan")
    if plan is None or not plan.matches(weights):
        if plan is not None:                                   # a captured hipGraph may still write the old images: retire, never free
            enc.__dict__.setdefault("_phnet_pack_retired", []).append(plan)
        plan = K.Conv3pPackPlan(weights)
        enc.__dict__["_phnet_pack_plan"] = plan
    plan.refresh()
    return {id(c): img for c, img in zip(convs, plan.images)}


def _conv3(x, w, packed, stats=False, bias=None, addend=None, relu=False):
    """3x3 / stride-1 forward: the packed-weight kernel when an image is at hand, else the generic entry point."""
    if packed is not None and K.conv3p_applies(x.shape[0] * x.shape[1] * x.shape[2], w.shape[3], w.shape[0]):
        return K.conv3p(x, packed[False], w.shape[0], bias=bias, addend=addend, relu=relu, stats=stats)
    return K.conv2d_fwd(x, w, bias, 1, 1, relu=relu, addend=addend, stats=stats)


def _conv_bn(tape: list, x, conv: nn.Conv2d, bn, training: bool, relu: bool, residual=None, w_override=None, packs=None):
    rec = _ConvBN()
    rec.conv, rec.bn = conv, bn
    rec.stride, rec.pad = conv.stride[0], conv.padding[0]
    rec.w = w_override if w_override is not None else ohwi(conv.weight)
    rec.x_in = x
    rec.packed = packs.get(id(conv)) if packs is not None else None
    if rec.packed is not None:                                 # (byte offsets of the packed kernel are 32-bit: huge batches take the generic path)
        m_ = x.shape[0] * x.shape[1] * x.shape[2]
        if not (K.conv3p_applies(m_, conv.in_channels, conv.out_channels) and K.conv3p_applies(m_, conv.out_channels, conv.in_channels)):
            rec.packed = None
    mom = bn.momentum if bn.momentum is not None else 0.1
    rec.sync_count = None
    rec.relu, rec.has_res = relu, residual is not None
    if not training and bn.running_mean is not None:
        # eval: scale / shift of the running statistics folded into the weights and a bias (cached until a parameter or a
        # statistic changes), residual add and ReLU in the GEMM epilogue - ONE launch per conv/BN/ReLU block, no
        # elementwise pass (resnet.py:79-95 in eval mode: y = relu(bn(conv(x)) + identity))
        wf, bf, pk = _folded(conv, bn, rec.w)
        if pk is not None and K.conv3p_applies(x.shape[0] * x.shape[1] * x.shape[2], conv.in_channels, conv.out_channels):
            rec.c = rec.y = K.conv3p(x, pk, wf.shape[0], bias=bf, addend=residual, relu=relu)
        else:
            rec.c = rec.y = K.conv2d_fwd(x, wf, bf, rec.stride, rec.pad, relu=relu, addend=residual)
        rec.sm = rec.si = None
    elif training and _sync_bn(bn):
        co = rec.w.shape[0]
        if rec.packed is not None and co <= 1024 and (co & (co - 1)) == 0:
            rec.c, partials = _conv3(x, rec.w, rec.packed, stats=True)
        elif co <= 1024 and (co & (co - 1)) == 0:                   # local statistics from the convolution's own epilogue, as below
            rec.c, partials = K.conv2d_fwd(x, rec.w, None, rec.stride, rec.pad, stats=True)
        else:
            rec.c, partials = K.conv2d_fwd(x, rec.w, None, rec.stride, rec.pad), None
        rec.y, rec.sm, rec.si, rec.sync_count = K.bn_fwd_sync(rec.c, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                              bn.running_var, bn.eps, mom, residual, relu,
                                                              getattr(bn, "process_group", None), partials=partials)
    else:
        # training: the batch statistics come out of the convolution's own epilogue (per-slab partial sums), no second pass
        co = rec.w.shape[0]
        fused_stats = training and co <= 1024 and (co & (co - 1)) == 0
        if fused_stats and rec.packed is not None:
            rec.c, partials = _conv3(x, rec.w, rec.packed, stats=True)
        elif fused_stats:
            rec.c, partials = K.conv2d_fwd(x, rec.w, None, rec.stride, rec.pad, stats=True)
        else:
            rec.c, partials = K.conv2d_fwd(x, rec.w, None, rec.stride, rec.pad), None
        rec.y, rec.sm, rec.si = K.bn_fwd(rec.c, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                         training, bn.eps, mom, residual, relu, partials=partials)
    tape.append(rec)
    return rec.y


# Bumped by everything that changes weights or running statistics BEHIND torch's version counters: the HIP kernels write
# through raw pointers (phnet_adamw_step, the BatchNorm running statistics of a training forward) and a hipGraph replay bumps
# nothing.  Part of the fold cache's key: an eval pass after training steps must not see the weights folded before them.
_WEIGHT_EPOCH = [0]


def weights_changed():
    _WEIGHT_EPOCH[0] += 1


def _folded(conv: nn.Conv2d, bn, w_ohwi: torch.Tensor):
    """(w * gamma / sqrt(var + eps) per output channel, beta - mean * gamma / sqrt(var + eps)) for an eval-mode conv -> BN pair.
    Cached ON the BatchNorm module (dies with the model), valid while no tensor version moved and no raw-pointer writer ran."""
    ver = (_WEIGHT_EPOCH[0], packed_path(), conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
           bn.running_var._version, conv.weight.data_ptr(), bn.running_mean.data_ptr(), w_ohwi.shape)
    hit = bn.__dict__.get("_phnet_fold")
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2], hit[3]
    with torch.no_grad():
        scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
        wf = (w_ohwi.reshape(w_ohwi.shape[0], -1) * scale[:, None]).reshape(w_ohwi.shape).contiguous()
        bf = (bn.bias.detach() - bn.running_mean * scale).contiguous()
        pk = K.conv3p_pack(wf, False) if (_is3x3s1(conv) and packed_path() and w_ohwi.shape[1] == 3) else None
    bn.__dict__["_phnet_fold"] = (ver, wf, bf, pk)
    return wf, bf, pk


class _Tape:
    """What the backward schedule needs from the forward."""
    __slots__ = ("enc", "tape", "argmax", "stem_shape", "feats", "lats", "lat_w", "out_w", "out_packs", "index", "nparams")


def encoder_fwd_schedule(enc, training: bool, frames: torch.Tensor):
    """(P3, P4, P5) NHWC and, in training, the tape for `encoder_bwd_schedule`."""
    model, neck = enc.backbone.model, enc.neck
    tape = []
    packs = None
    if training:
        weights_changed()                                      # running statistics are about to move (raw-pointer writes)
        if packed_path():
            # every 3x3 / stride-1 weight of the trunk and the neck split into bf16 planes in MFMA fragment order, forward and
            # data-gradient images, by ONE launch: the GEMM loops fetch them straight into registers (csrc/conv3p.hip)
            convs = [m for m in model.modules() if isinstance(m, nn.Conv2d) and _is3x3s1(m)]
            convs += [m.conv for m in neck.fpn_convs if _is3x3s1(m.conv)]
            if convs:
                packs = _pack_plan(enc, convs)
    x = K.nchw3_to_nhwc4(frames.contiguous())
    w_stem = K.pad_channels(ohwi(model.conv1.weight).view(-1, 3), 4).view(model.conv1.out_channels, 7, 7, 4)
    y = _conv_bn(tape, x, model.conv1, model.bn1, training, relu=True, w_override=w_stem)
    stem_shape = tuple(y.shape)
    y, argmax = K.maxpool_fwd(y)
    stages = []
    for name in ("layer1", "layer2", "layer3", "layer4"):
        for blk in getattr(model, name):
            h = _conv_bn(tape, y, blk.conv1, blk.bn1, training, relu=True, packs=packs)
            if blk.downsample is not None:
                idn = _conv_bn(tape, y, blk.downsample[0], blk.downsample[1], training, relu=False, packs=packs)
            else:
                idn = y
            y = _conv_bn(tape, h, blk.conv2, blk.bn2, training, relu=True, residual=idn, packs=packs)
        stages.append(y)
    feats = stages[-3:]                                        # fpn.py:113-115 drops layer1
    lat_w = [ohwi(m.conv.weight) for m in neck.lateral_convs]
    out_w = [ohwi(m.conv.weight) for m in neck.fpn_convs]
    lats = [K.conv2d_fwd(f, w, m.conv.bias.detach(), 1, 0) for f, w, m in zip(feats, lat_w, neck.lateral_convs)]
    for i in (2, 1):
        K.upsample_add_(lats[i - 1], lats[i])
    out_packs = [packs.get(id(m.conv)) if packs is not None else None for m in neck.fpn_convs]
    outs = [_conv3(l, w, pk, bias=m.conv.bias.detach()) for l, w, m, pk in zip(lats, out_w, neck.fpn_convs, out_packs)]
    ctx = None
    if training:
        counters = [r.bn.num_batches_tracked for r in tape if r.bn.num_batches_tracked is not None]
        if counters:
            torch._foreach_add_(counters, 1)               # one multi-tensor launch for the 36 BatchNorm step counters
        ctx = _Tape()
        ctx.enc, ctx.tape, ctx.argmax, ctx.stem_shape = enc, tape, argmax, stem_shape
        ctx.feats, ctx.lats, ctx.lat_w, ctx.out_w, ctx.out_packs = feats, lats, lat_w, out_w, out_packs
        ctx.index = {id(p): i for i, p in enumerate(enc.parameters())}
        ctx.nparams = len(ctx.index)
    return tuple(outs), ctx


def encoder_fwd_v2(enc, frames: torch.Tensor):
    """Inference schedule of the Router4OLV2 family's encoder (libs/models/Router4OLV2.py:20-30, libs/models/fpnV2.py:122-150):
    the trunk WITHOUT its last stage (`self.backbone(x)[:-1]`: layer4 is computed by the reference and thrown away - it is not
    run here), then the per-level-width FPN: 1x1 laterals (64/128/256 -> 16/32/64), top-down path with a 1x1 projection of the
    coarser level to the finer level's width BEFORE the nearest resize, 3x3 output convolutions.  BatchNorm in its folded
    eval form.  Returns the three levels fine -> coarse as NHWC [T,h,w,C_l]."""
    model, neck = enc.backbone.model, enc.neck
    tape = []
    x = K.nchw3_to_nhwc4(frames.contiguous())
    w_stem = K.pad_channels(ohwi(model.conv1.weight).view(-1, 3), 4).view(model.conv1.out_channels, 7, 7, 4)
    y = _conv_bn(tape, x, model.conv1, model.bn1, False, relu=True, w_override=w_stem)
    y, _ = K.maxpool_fwd(y)
    stages = []
    for name in ("layer1", "layer2", "layer3"):
        for blk in getattr(model, name):
            h = _conv_bn(tape, y, blk.conv1, blk.bn1, False, relu=True)
            idn = y if blk.downsample is None else _conv_bn(tape, y, blk.downsample[0], blk.downsample[1], False, relu=False)
            y = _conv_bn(tape, h, blk.conv2, blk.bn2, False, relu=True, residual=idn)
        stages.append(y)
    conv = lambda m, t, pad: K.conv2d_fwd(t, ohwi(m.conv.weight), m.conv.bias.detach(), 1, pad)      # noqa: E731
    lats = [conv(m, f, 0) for m, f in zip(neck.lateral_convs, stages)]
    for i in range(len(lats) - 1, 0, -1):
        K.upsample_add_(lats[i - 1], conv(neck.upsample_convs[i - 1], lats[i], 0))
    return tuple(conv(m, l, 1) for m, l in zip(neck.fpn_convs, lats))


def encoder_bwd_schedule(ctx: _Tape, d3, d4, d5, stage_done: Optional[Callable[[str], None]] = None) -> List:
    """Backward of the whole encoder; returns the per-parameter gradients that were NOT written straight into a gradient
    arena (list aligned with enc.parameters(), None where the arena took them)."""
    enc = ctx.enc
    model, neck = enc.backbone.model, enc.neck
    grads: List = [None] * ctx.nparams
    done = stage_done or (lambda name: None)
    done("head")                                       # the lane head's backward has run: its gradients are final

    def put(p, g):
        i = ctx.index[id(p)]
        grads[i] = g if grads[i] is None else grads[i] + g

    def dest_ohwi(p):
        d = direct_grad(p)                    # arena view with the parameter's channels_last strides
        return None if d is None else d.permute(0, 2, 3, 1)

    def conv_wgrad_into(p, dy, x, wshape, stride, pad, bias=None):
        """weight gradient (and the bias gradient of the FPN convs) from one launch; arena-backed destinations are
        accumulated in place, otherwise the gradients go back to the caller."""
        d = dest_ohwi(p)
        db = direct_grad(bias) if bias is not None else None
        if d is not None and d.is_contiguous() and (bias is None or db is not None):
            K.conv2d_wgrad(dy, x, wshape, stride, pad, dw=d, accumulate=True, dbias=db)
            return
        dbt = torch.empty(wshape[0], dtype=torch.float32, device=dy.device) if bias is not None else None
        put(p, oihw_grad(K.conv2d_wgrad(dy, x, wshape, stride, pad, dbias=dbt)))
        if bias is not None:
            put(bias, dbt)

    # ---- FPN ------------------------------------------------------------------------------------------
    douts = [d.contiguous() for d in (d3, d4, d5)]
    dl = []
    for i in range(3):
        m = neck.fpn_convs[i].conv
        conv_wgrad_into(m.weight, douts[i], ctx.lats[i], ctx.out_w[i].shape, 1, 1, bias=m.bias)
        if ctx.out_packs[i] is not None and K.conv3p_applies(douts[i].shape[0] * douts[i].shape[1] * douts[i].shape[2],
                                                             ctx.out_w[i].shape[0], ctx.out_w[i].shape[3]):
            dl.append(K.conv3p(douts[i], ctx.out_packs[i][True], ctx.out_w[i].shape[3], dgrad=True))
        else:
            dl.append(K.conv2d_dgrad(douts[i], ctx.out_w[i], tuple(ctx.lats[i].shape[1:3]), 1, 1))
    for i in (1, 2):
        K.upsample_add_bwd_(dl[i - 1], dl[i])
    dstage = []
    for i in range(3):
        m = neck.lateral_convs[i].conv
        conv_wgrad_into(m.weight, dl[i], ctx.feats[i], ctx.lat_w[i].shape, 1, 0, bias=m.bias)
        dstage.append(K.conv2d_dgrad(dl[i], ctx.lat_w[i], tuple(ctx.feats[i].shape[1:3]), 1, 0))
    done("neck")
    # ---- trunk, last block first ----------------------------------------------------------------------
    tape = ctx.tape
    pos = len(tape)

    def bn_back(rec, dy, dres=None, dres_acc=False):
        # arena destinations ACCUMULATE like every other backward kernel: the reference's caller runs several
        # forward/backward passes per optimizer step (trainOL.py:205-212, train_batch > 1); the arena is zeroed once per step
        dgd, dbd = direct_grad(rec.bn.weight), direct_grad(rec.bn.bias)
        direct = dgd is not None and dbd is not None
        kw = dict(dgamma=dgd, dbeta=dbd, param_accumulate=True) if direct else {}
        if rec.sync_count is not None:
            dx, dg, db = K.bn_bwd_sync(dy, rec.c, rec.y, rec.sm, rec.si, rec.bn.weight.detach(), rec.relu, rec.sync_count, dres,
                                       getattr(rec.bn, "process_group", None), dres_accumulate=dres_acc, **kw)
        else:
            dx, dg, db = K.bn_bwd(dy, rec.c, rec.y, rec.sm, rec.si, rec.bn.weight.detach(), rec.relu, dres, dres_acc, **kw)
        if not direct:
            put(rec.bn.weight, dg)
            put(rec.bn.bias, db)
        return dx

    def conv_back(rec, dc, need_dx=True, addend=None):
        if rec.conv is model.conv1:
            dw = K.conv2d_wgrad(dc, rec.x_in, rec.w.shape, rec.stride, rec.pad)
            dw = oihw_grad(K.pad_channels(dw.view(-1, 4), 3).view(dw.shape[0], 7, 7, 3))
            d = direct_grad(rec.conv.weight)
            if d is not None:
                d.add_(dw)                                                   # 9 408 elements: the stem's channel un-padding
            else:
                put(rec.conv.weight, dw)
        else:
            conv_wgrad_into(rec.conv.weight, dc, rec.x_in, rec.w.shape, rec.stride, rec.pad)
        if not need_dx:
            return None
        if rec.packed is not None:
            return K.conv3p(dc, rec.packed[True], rec.w.shape[3], dgrad=True, addend=addend)
        return K.conv2d_dgrad(dc, rec.w, tuple(rec.x_in.shape[1:3]), rec.stride, rec.pad, addend)

    dy = None
    stage_names = ("layer4", "layer3", "layer2", "layer1")
    for si, name in enumerate(stage_names):
        # gradient arriving at this stage's output: from the neck (stages 2..4) and from the next stage
        extra = dstage[2 - si] if si < 3 else None
        if dy is None:
            dy = extra
        elif extra is not None:
            dy = dy + extra
        for blk in reversed(list(getattr(model, name))):
            has_ds = blk.downsample is not None
            rec2 = tape[pos - 1]
            recd = tape[pos - 2] if has_ds else None
            rec1 = tape[pos - 3] if has_ds else tape[pos - 2]
            pos -= 3 if has_ds else 2
            dres = torch.empty_like(rec2.c)
            dc2 = bn_back(rec2, dy, dres)                     # dres = relu-masked gradient for the identity path
            dh = conv_back(rec2, dc2)
            dc1 = bn_back(rec1, dh)
            if has_ds:
                dcd = bn_back(recd, dres)
                dxd = conv_back(recd, dcd)
                dy = conv_back(rec1, dc1, addend=dxd)
            else:
                dy = conv_back(rec1, dc1, addend=dres)
        done(name)
    # ---- stem -------------------------------------------------------------------------------------------
    rec = tape[0]
    dpool = K.maxpool_bwd(dy, ctx.argmax, ctx.stem_shape)
    dc = bn_back(rec, dpool)
    conv_back(rec, dc, need_dx=False)
    done("stem")
    ctx.tape = ctx.feats = ctx.lats = None
    return grads


class EncoderFunction(torch.autograd.Function):
    """(frames NCHW, *encoder parameters) -> (P3, P4, P5) as NHWC tensors."""

    @staticmethod
    def forward(ctx, enc, training: bool, frames: torch.Tensor, *params):
        outs, tape = encoder_fwd_schedule(enc, training, frames)
        if training:
            ctx.tape = tape
        return outs

    @staticmethod
    def backward(ctx, d3, d4, d5):
        grads = encoder_bwd_schedule(ctx.tape, d3, d4, d5, STAGE_DONE_HOOK)
        ctx.tape = None
        return (None, None, None, *grads)


def encoder_forward(enc, frames: torch.Tensor):
    """enc: the Encoder container (backbone.model = ResNet container, neck = FPN container)."""
    params = [p for p in enc.parameters()]
    training = enc.training and torch.is_grad_enabled()
    if training and getattr(enc, "staged", False):
        return encoder_forward_staged(enc, frames)
    if enc.training and not training:
        # train-mode statistics without a tape (e.g. under no_grad): still the training arithmetic
        return EncoderFunction.apply(enc, True, frames, *[p.detach() for p in params])
    return EncoderFunction.apply(enc, enc.training, frames, *params)


def encoder_forward_staged(enc, frames: torch.Tensor):
    """Forward outside autograd: the outputs are LEAVES whose .grad the lane head's backward fills; the tape waits in
    `enc._staged` for `encoder_backward_staged`."""
    with torch.no_grad():
        outs, tape = encoder_fwd_schedule(enc, True, frames)
    leaves = tuple(o.detach().requires_grad_(True) for o in outs)
    enc._staged = (tape, leaves)
    return leaves


def encoder_backward_staged(enc, stage_done: Optional[Callable[[str], None]] = None):
    """Second half of a staged step: run after the lane head's `loss.backward()`, from the calling (main) thread."""
    tape, leaves = enc._staged
    enc._staged = None
    if any(l.grad is None for l in leaves):
        raise RuntimeError("encoder_backward_staged: the lane head's backward has not reached the feature maps")
    with torch.no_grad():
        grads = encoder_bwd_schedule(tape, leaves[0].grad, leaves[1].grad, leaves[2].grad, stage_done or STAGE_DONE_HOOK)
        for p, g in zip(enc.parameters(), grads):       # parameters outside an arena: plain .grad accumulation
            if g is not None:
                p.grad = g.clone() if p.grad is None else p.grad + g
