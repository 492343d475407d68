"""CPU oracle for the PHNet per-clip hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
file.  It is a from-scratch, *functional* fp32 restatement (plain torch CPU ops over a
flat ``{name: tensor}`` state dict that uses the reference's state_dict key names) of
the algorithm in the reference files cited next to every function
(paths relative to the reference checkout).  It is pinned against fixtures generated
by running the reference's own Python on CPU in the build container
(tests/golden/make_goldens.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

Nothing here is used, imported or timed by the product path in phnet_amd/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

BLOCKS_PER_LAYER = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}
LAYER_WIDTH = (64, 128, 256, 512)


@dataclass
class Geometry:
    """The cfg keys the hot path reads (options/options4OL.py; Router4OL.py:42-46,511-513)."""
    img_h: int = 320
    img_w: int = 800
    num_points: int = 36          # S: x offsets per lane
    num_priors: int = 240         # N: anchors
    sample_points: int = 36       # P: ROI samples per anchor (Router4OL.py:38)
    feat_channels: int = 64       # C
    refine_layers: int = 3
    max_lanes: int = 4
    save_freq_max: int = 8
    conf_threshold: float = 0.5
    nms_thres: float = 50.0
    cls_weight: float = 8.0
    reg_weight: float = 0.5
    iou_weight: float = 1.5
    arch: str = "resnet34"
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def n_strips(self) -> int:
        return self.num_points - 1


# --------------------------------------------------------------------------------------
# a1  ResNet trunk                               libs/models/resnet.py:79-95, 293-307
# --------------------------------------------------------------------------------------
def _bn(sd: State, key: str, x: Tensor, g: Geometry, training: bool, track: bool) -> Tensor:
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    if training and not track:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"],
                     training, g.bn_momentum, g.bn_eps)
    if training and track and (key + ".num_batches_tracked") in sd:
        sd[key + ".num_batches_tracked"] += 1
    return y


def resnet_trunk(sd: State, x: Tensor, g: Geometry, training: bool,
                 track_running_stats: bool = False, prefix: str = "backbone.backbone.model.") -> List[Tensor]:
    """7x7/2 stem + maxpool + four stages of basic blocks; returns the four stage outputs."""
    p = prefix
    x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, p + "bn1", x, g, training, track_running_stats))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for li, nblocks in enumerate(BLOCKS_PER_LAYER[g.arch]):
        for bi in range(nblocks):
            q = f"{p}layer{li + 1}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            y = F.conv2d(x, sd[q + "conv1.weight"], None, stride=stride, padding=1)
            y = F.relu(_bn(sd, q + "bn1", y, g, training, track_running_stats))
            y = F.conv2d(y, sd[q + "conv2.weight"], None, stride=1, padding=1)
            y = _bn(sd, q + "bn2", y, g, training, track_running_stats)
            if (q + "downsample.0.weight") in sd:
                idn = F.conv2d(x, sd[q + "downsample.0.weight"], None, stride=stride)
                idn = _bn(sd, q + "downsample.1", idn, g, training, track_running_stats)
            else:
                idn = x
            x = F.relu(y + idn)
        outs.append(x)
    return outs


# --------------------------------------------------------------------------------------
# a2  FPN                                        libs/models/fpn.py:109-163
# --------------------------------------------------------------------------------------
def fpn_neck(sd: State, feats: Sequence[Tensor], prefix: str = "backbone.neck.") -> Tuple[Tensor, ...]:
    """Drops the leading level (fpn.py:113-115), 1x1 laterals, nearest top-down add, 3x3 outputs.
    ConvModule here is a bare Conv2d with bias (norm_cfg=None, act_cfg=None: fpn.py:70-85)."""
    feats = list(feats)[-3:]
    lat = [F.conv2d(f, sd[f"{prefix}lateral_convs.{i}.conv.weight"], sd[f"{prefix}lateral_convs.{i}.conv.bias"])
           for i, f in enumerate(feats)]
    for i in (2, 1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
    return tuple(F.conv2d(lat[i], sd[f"{prefix}fpn_convs.{i}.conv.weight"],
                          sd[f"{prefix}fpn_convs.{i}.conv.bias"], padding=1) for i in range(3))


# --------------------------------------------------------------------------------------
# a3  anchors                                    libs/models/Router4OL.py:53-59, 152-211
# --------------------------------------------------------------------------------------
def sample_x_indexs(g: Geometry) -> Tensor:
    return (torch.linspace(0, 1, steps=g.sample_points, dtype=torch.float32) * g.n_strips).long()


def prior_feat_ys(g: Geometry) -> Tensor:
    return torch.flip(1 - sample_x_indexs(g).float() / g.n_strips, dims=[-1])


def prior_ys(g: Geometry) -> Tensor:
    return torch.linspace(1, 0, steps=g.num_points, dtype=torch.float32)


def initial_anchor_embeddings(g: Geometry) -> Tensor:
    """(start_y, start_x, theta) of the hand-placed anchors  (Router4OL.py:169-211)."""
    n = g.num_priors
    quarter, half = n // 4, n // 2
    side_step = 0.8 / (quarter // 2 - 1)
    bottom_step = 0.5 / (quarter // 2 + 1)
    e = torch.zeros(n, 3, dtype=torch.float32)
    for i in range(n):
        even = (i % 2 == 0)
        if i < quarter:                                   # left border
            e[i] = torch.tensor([(i // 2) * side_step, 0.0, 0.16 if even else 0.32])
        elif i < half:                                    # bottom, left half
            e[i] = torch.tensor([0.0, ((i - quarter) // 2 + 1) * bottom_step, 0.2 if even else 0.4])
        elif i < half + quarter:                          # bottom, right half
            e[i] = torch.tensor([0.0, ((i - half) // 2 + 1) * bottom_step + 0.5, 0.6 if even else 0.8])
        else:                                             # right border
            e[i] = torch.tensor([((i - half - quarter) // 2) * side_step, 1.0, 0.68 if even else 0.84])
    return e


def lane_xs_from_start(sy: Tensor, sx: Tensor, theta: Tensor, g: Geometry) -> Tensor:
    """x (normalised) at every prior_y of the straight line through (sx, sy) at angle theta.
    Router4OL.py:158-164 and :335-339 (same formula)."""
    ys = prior_ys(g).to(sy.dtype)
    return (sx * (g.img_w - 1) + ((1 - ys - sy) * g.img_h / torch.tan(theta * math.pi + 1e-5))) / (g.img_w - 1)


def priors_from_embeddings(emb: Tensor, g: Geometry) -> Tuple[Tensor, Tensor]:
    """[N,3] -> priors [N,6+S] (cols 0,1,5 zero) and priors_on_featmap [N,P].  Router4OL.py:152-167."""
    xs = lane_xs_from_start(emb[:, 0:1], emb[:, 1:2], emb[:, 2:3], g)
    zeros = emb.new_zeros(emb.shape[0], 1)
    priors = torch.cat([zeros, zeros, emb, zeros, xs], dim=1)
    return priors, priors[:, 6 + sample_x_indexs(g)]


# --------------------------------------------------------------------------------------
# a4  lane-anchor ROI pooling                    libs/models/Router4OL.py:132-150, 269-272
# --------------------------------------------------------------------------------------
def pool_anchor_features(fmap: Tensor, priors_on_featmap: Tensor, g: Geometry) -> Tensor:
    """fmap [1,C,h,w], priors_on_featmap [1,N,P] -> [1,N,C,P] (bilinear, align_corners, zero pad)."""
    xs = torch.flip(priors_on_featmap, dims=[2])                       # pairs x_k with y_k
    ys = prior_feat_ys(g).to(xs.dtype).view(1, 1, -1).expand_as(xs)
    grid = torch.stack([xs * 2.0 - 1.0, ys * 2.0 - 1.0], dim=-1)       # [1,N,P,2]
    return F.grid_sample(fmap, grid, mode="bilinear", padding_mode="zeros", align_corners=True).permute(0, 2, 1, 3)


# --------------------------------------------------------------------------------------
# a5  adaptive routing gate                      libs/models/Router.py:39-81
# --------------------------------------------------------------------------------------
def routing_gate(sd: State, stage: int, feat: Tensor, prefix: str = "detNet.router.") -> Tensor:
    """feat [1,N,C,P] (already detached by the caller) -> gate score [1,N,1] in [0.5,1)."""
    cp = list(feat.shape[2:])
    x = F.layer_norm(feat, cp, sd[f"{prefix}pre_norm.{stage}.weight"], sd[f"{prefix}pre_norm.{stage}.bias"])
    n = feat.shape[1]
    for b in range(4):
        q = f"{prefix}DWNets.{stage}.{b}."
        y = F.conv2d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=1, groups=n)
        y = F.relu(F.layer_norm(y, cp, sd[q + "1.weight"], sd[q + "1.bias"]))
        y = F.conv2d(y, sd[q + "3.weight"], sd[q + "3.bias"], padding=1, groups=n)
        y = F.layer_norm(y, cp, sd[q + "4.weight"], sd[q + "4.bias"])
        x = F.relu(y + x)
    q = f"{prefix}layers.{stage}."
    h = F.relu(F.linear(x.flatten(2), sd[q + "0.weight"], sd[q + "0.bias"]))
    h = F.relu(F.linear(h, sd[q + "2.weight"], sd[q + "2.bias"]))     # ReLU *before* the sigmoid
    return torch.sigmoid(h)


# --------------------------------------------------------------------------------------
# a6  dynamic per-anchor "convolution"           libs/models/utils/dynamic_head.py:31-59
# --------------------------------------------------------------------------------------
def _lin(sd: State, key: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def dynamic_head(sd: State, stage: int, pro_feat: Tensor, roi: Tensor, prefix: str = "detNet.DHead_series.") -> Tensor:
    """pro_feat [1,N,C], roi [1,N,P,C] -> [1,N,C]."""
    q = f"{prefix}{stage}."
    b, n, pnum, c = roi.shape
    roi = roi.reshape(b * n, pnum, c)
    pro = pro_feat.reshape(b * n, c)
    w1 = _lin(sd, q + "dynamic_layer_1.1", _lin(sd, q + "dynamic_layer_1.0", pro)).reshape(b * n, c, 2 * c)
    f = torch.bmm(roi, w1)
    f = F.relu(F.layer_norm(f, [2 * c], sd[q + "norm1.weight"], sd[q + "norm1.bias"]))
    w2 = _lin(sd, q + "dynamic_layer_2.1", _lin(sd, q + "dynamic_layer_2.0", f.detach().flatten(1))).reshape(b * n, 2 * c, c)
    f = torch.bmm(f, w2)
    f = F.relu(F.layer_norm(f, [c], sd[q + "norm2.weight"], sd[q + "norm2.bias"]))
    f = _lin(sd, q + "out_layer.1", _lin(sd, q + "out_layer.0", f.flatten(1)))
    f = F.layer_norm(f, [c], sd[q + "norm3.weight"], sd[q + "norm3.bias"])
    return f.view(b, n, c)


# --------------------------------------------------------------------------------------
# a7/a8  the two heterogeneous branches          libs/models/Router4OL.py:308-392
# --------------------------------------------------------------------------------------
def _tower(sd: State, key: str, x: Tensor) -> Tensor:
    """ModuleList [Linear, ReLU, Linear, ReLU] -> state-dict indices 0 and 2 (roi_gather.py:7-10)."""
    return F.relu(_lin(sd, key + ".2", F.relu(_lin(sd, key + ".0", x))))


def update_priors(priors: Tensor, cls_logits: Tensor, reg: Tensor, offsets: Tensor, g: Geometry) -> Tuple[Tensor, Tensor]:
    """Router4OL.py:328-345: start/theta accumulate through tanh, length is replaced, xs are
    re-derived from the new (sy,sx,theta); the returned pair is (with offsets, without offsets)."""
    syx_t = priors[..., 2:5] + torch.tanh(reg[..., :3])
    length = reg[..., 3:4]
    xs = lane_xs_from_start(syx_t[..., 0:1], syx_t[..., 1:2], syx_t[..., 2:3], g)
    lines = torch.cat([cls_logits, syx_t, length, xs], dim=-1)
    preds = torch.cat([cls_logits, syx_t, length, xs + offsets], dim=-1)
    return preds, lines


def branch_heads(sd: State, feat: Tensor, priors: Tensor, g: Geometry, suffix: str, prefix: str = "detNet.") -> Tuple[Tensor, Tensor]:
    cls = _lin(sd, f"{prefix}cls_layers{suffix}", _tower(sd, f"{prefix}cls_modules{suffix}", feat))
    reg = _lin(sd, f"{prefix}reg_layers{suffix}", _tower(sd, f"{prefix}reg_modules{suffix}", feat))
    off = _lin(sd, f"{prefix}iou_layers{suffix}", _tower(sd, f"{prefix}iou_modules{suffix}", feat))
    n = priors.shape[1]
    return update_priors(priors, cls.reshape(1, n, 2), reg.reshape(1, n, 4), off.reshape(1, n, g.num_points), g)


def _mha(sd: State, key: str, q_in: Tensor, kv_in: Tensor, nhead: int = 8) -> Tensor:
    """nn.MultiheadAttention, batch 1, no masks, dropout off.  q_in [L,E], kv_in [M,E] -> [L,E]."""
    e = q_in.shape[-1]
    w, bias = sd[key + ".in_proj_weight"], sd[key + ".in_proj_bias"]
    q = F.linear(q_in, w[:e], bias[:e])
    k = F.linear(kv_in, w[e:2 * e], bias[e:2 * e])
    v = F.linear(kv_in, w[2 * e:], bias[2 * e:])
    dh = e // nhead
    q = q.view(-1, nhead, dh).transpose(0, 1) * (1.0 / math.sqrt(dh))
    k = k.view(-1, nhead, dh).transpose(0, 1)
    v = v.view(-1, nhead, dh).transpose(0, 1)
    att = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    out = torch.bmm(att, v).transpose(0, 1).reshape(-1, e)
    return _lin(sd, key + ".out_proj", out)


def temporal_decoder(sd: State, tgt: Tensor, memory: Tensor, prefix: str = "detNet.transformer_Dec.") -> Tensor:
    """2 pre-norm decoder layers + final LayerNorm; tgt [N,E], memory [M,E].
    libs/models/utils/transformer.py:100-129, 275-298 (dropout disabled for parity)."""
    e = tgt.shape[-1]
    x = tgt
    for li in range(2):
        q = f"{prefix}layers.{li}."
        h = F.layer_norm(x, [e], sd[q + "norm1.weight"], sd[q + "norm1.bias"])
        x = x + _mha(sd, q + "self_attn", h, h)
        h = F.layer_norm(x, [e], sd[q + "norm2.weight"], sd[q + "norm2.bias"])
        x = x + _mha(sd, q + "multihead_attn", h, memory)
        h = F.layer_norm(x, [e], sd[q + "norm3.weight"], sd[q + "norm3.bias"])
        x = x + _lin(sd, q + "linear2", F.gelu(_lin(sd, q + "linear1", h)))
    return F.layer_norm(x, [e], sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])


# --------------------------------------------------------------------------------------
# a3-a9  one frame through the lane head         libs/models/Router4OL.py:253-306
# --------------------------------------------------------------------------------------
@dataclass
class FrameOutput:
    predictions_fir: List[Tensor]
    predictions_sec: List[Tensor]
    attn_feats: List[Tensor]          # per stage [N,128]: this frame's memory source
    gates: List[Tensor]               # per stage [1,N,1]
    stage_inputs: List[dict] = field(default_factory=list)   # per stage: priors/on_map/pro/mem fed to that stage (tests)
    locals_: List[Tensor] = field(default_factory=list)      # per stage dynamic-head output [1,N,C]


def lane_head_frame(sd: State, feats: Sequence[Tensor], memory: List[List[Tensor]], g: Geometry,
                    training: bool, prefix: str = "detNet.") -> FrameOutput:
    """feats = (P3,P4,P5) each [1,C,h,w]; memory = list over stored frames of per-stage [m,128] tokens."""
    levels = list(feats)[::-1]                                      # coarse -> fine
    if training:
        priors, on_map = priors_from_embeddings(sd[prefix + "prior_embeddings.weight"], g)
    else:
        priors, on_map = sd[prefix + "priors"], sd[prefix + "priors_on_featmap"]
    priors, on_map = priors.unsqueeze(0), on_map.unsqueeze(0)
    pro = sd[prefix + "pro_embedding.weight"].unsqueeze(0)
    pos = sd[prefix + "PositionEmbedding.embed.weight"]
    out = FrameOutput([], [], [], [])
    sxi = sample_x_indexs(g)
    for stage in range(g.refine_layers):
        out.stage_inputs.append(dict(priors=priors.detach(), on_map=on_map.detach(), pro=pro.detach(),
                                     mem=[fr[stage] for fr in memory]))
        pooled = pool_anchor_features(levels[stage], on_map, g)                 # [1,N,C,P]
        gate = routing_gate(sd, stage, pooled.detach(), prefix + "router.")
        local = dynamic_head(sd, stage, pro, pooled.transpose(2, 3), prefix + "DHead_series.")
        pro = local.detach()
        out.locals_.append(local.detach())
        pred_a, lines_a = branch_heads(sd, local, priors, g, "", prefix)
        attn = torch.cat([local[0], pos], dim=-1)                               # [N,128]
        mem = [fr[stage] for fr in memory]
        mem = torch.cat(mem, dim=0) if mem else None
        glob = temporal_decoder(sd, attn, mem, prefix + "transformer_Dec.") if (mem is not None and mem.shape[0]) else attn
        pred_b, lines_b = branch_heads(sd, glob.unsqueeze(0), priors, g, "_sec", prefix)
        out.predictions_fir.append(pred_a)
        out.predictions_sec.append(pred_b)
        out.attn_feats.append(attn)
        out.gates.append(gate)
        w = gate.detach()
        if stage != g.refine_layers - 1:
            priors = ((1 - w) * lines_a + w * lines_b).detach().clone()
            on_map = priors[..., 6 + sxi]
    return out


# --------------------------------------------------------------------------------------
# a11  criterion                                 libs/utils/loss4OLV3.py, dynamic_assign.py,
#                                                focal_loss.py, dynamic_assignV2.py
# --------------------------------------------------------------------------------------
def _pairwise_line_iou(pred_px: Tensor, tgt_px: Tensor, img_w: int, radius: float = 15.0) -> Tensor:
    """dynamic_assign.py:5-36 with aligned=False: [n,S],[m,S] -> [n,m]."""
    lo_p, hi_p = (pred_px - radius)[:, None, :], (pred_px + radius)[:, None, :]
    lo_t, hi_t = (tgt_px - radius)[None], (tgt_px + radius)[None]
    ovr = torch.min(hi_p, hi_t) - torch.max(lo_p, lo_t)
    uni = torch.max(hi_p, hi_t) - torch.min(lo_p, lo_t)
    bad = ((tgt_px < 0) | (tgt_px >= img_w))[None].expand_as(ovr)
    ovr = ovr.masked_fill(bad, 0.0)
    uni = uni.masked_fill(bad, 0.0)
    return ovr.sum(-1) / (uni.sum(-1) + 1e-9)


def assignment_cost(pred: Tensor, tgt: Tensor, g: Geometry) -> Tensor:
    """The matrix handed to the Hungarian solver (dynamic_assign.py:128-185). pred [N,6+S], tgt [m,6+S]."""
    pred, tgt = pred.detach().clone(), tgt.detach().clone()
    w, h = g.img_w, g.img_h
    pxs = pred[:, 6:] * (w - 1)
    txs = tgt[:, 6:]
    bad = (txs < 0) | (txs >= w)                                               # distance_cost :44-63
    d = (txs[None] - pxs[:, None]).abs().masked_fill(bad[None].expand(pxs.shape[0], -1, -1), 0.0)
    dist = d.sum(-1) / ((~bad).sum(1).float() + 1e-9)[None]
    dist = 1 - dist / (dist.max() + 1e-4)
    prob = pred[:, :2].sigmoid()                                               # focal_cost :66-80
    eps, alpha, gamma = 1e-12, 0.25, 2
    neg = -(1 - prob + eps).log() * (1 - alpha) * prob.pow(gamma)
    posc = -(prob + eps).log() * alpha * (1 - prob).pow(gamma)
    lab = tgt[:, 1].long()
    cls = posc[:, lab] - neg[:, lab]
    scale = torch.tensor([h - 1.0, w - 1.0])
    start = torch.cdist(pred[:, 2:4] * scale, tgt[:, 2:4] * scale, p=2)
    start = 1 - start / (start.max() + 1e-4)
    theta = torch.cdist(pred[:, 4:5], tgt[:, 4:5], p=1) * 180
    theta = 1 - theta / (theta.max() + 1e-4)
    cost = -(dist * start * theta) ** 2 * 3.0 + cls * 1.0
    return cost - _pairwise_line_iou(pxs, txs, w)


def hungarian(cost: Tensor) -> Tuple[Tensor, Tensor]:
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(cost.cpu().numpy(), maximize=False)
    return torch.as_tensor(r), torch.as_tensor(c)


def focal_vector(logits: Tensor, labels: Tensor, alpha=(0.1, 0.9), gamma: float = 2.0, eps: float = 1e-6) -> Tensor:
    """focal_loss.py:78-136 with the per-class alpha list of FocalLoss (:192-199); returns [N]."""
    p = F.softmax(logits, dim=1) + eps
    onehot = F.one_hot(labels, logits.shape[1]).to(logits.dtype) + 1e-6
    focal = -torch.tensor(alpha, dtype=logits.dtype) * torch.pow(1.0 - p, gamma) * torch.log(p)
    return (onehot * focal).sum(dim=1)


def lane_iou_loss(pred: Tensor, tgt: Tensor, half_width: float = 7.5 / 768, img_h: int = 400, img_w: int = 960) -> Tensor:
    """dynamic_assignV2.py:55-98 (CLRerNet LaneIoU) with the class defaults the reference instantiates."""
    dy = img_h / (pred.shape[1] - 1) * 2
    pd = (pred[:, 2:] - pred[:, :-2]).detach() * img_w
    pw = half_width * torch.sqrt(pd.pow(2) + dy ** 2) / dy
    pw = torch.cat([pw[:, :1], pw, pw[:, -1:]], dim=1)
    td = (tgt[:, 2:] - tgt[:, :-2]) * img_w
    td = torch.where(td.abs() > 1e4, torch.zeros_like(td), td)
    tw = half_width * torch.sqrt(td.pow(2) + dy ** 2) / dy
    tw = torch.cat([tw[:, :1], tw, tw[:, -1:]], dim=1)
    ovr = torch.min(pred + pw, tgt + tw) - torch.max(pred - pw, tgt - tw)
    uni = torch.max(pred + pw, tgt + tw) - torch.min(pred - pw, tgt - tw)
    bad = (tgt < 0) | (tgt >= 1.0)
    ovr = ovr.masked_fill(bad, 0.0)
    uni = uni.masked_fill(bad, 0.0)
    iou = ovr.sum(-1) / (uni.sum(-1) + 1e-9)
    return (1 - iou).mean()


def branch_loss(stage_preds: Sequence[Tensor], gt: Tensor, g: Geometry):
    """loss4OLV3.py:34-82 for one branch: ([matched rows per stage], cls[N], reg, iou)."""
    cls_sum, reg_sum, iou_sum = 0.0, 0.0, 0.0
    matched = []
    scale = torch.tensor([g.n_strips, g.img_w - 1.0, 180.0, g.n_strips], dtype=torch.float32)
    for preds in stage_preds:
        pred = preds[0]
        tgt = gt[0][gt[0][:, 1] == 1]
        labels = torch.zeros(pred.shape[0], dtype=torch.long)
        if tgt.shape[0] == 0:
            cls_sum = cls_sum + focal_vector(pred[:, :2], labels)
            matched.append(torch.zeros(0, dtype=torch.long))
            continue
        with torch.no_grad():
            rows, cols = hungarian(assignment_cost(pred, tgt, g))
        matched.append(rows)
        labels[rows] = 1
        cls_sum = cls_sum + focal_vector(pred[:, :2], labels)
        reg_sum = reg_sum + F.smooth_l1_loss(pred[rows, 2:6] * scale, tgt[cols, 2:6] * scale, reduction="none").mean()
        iou_sum = iou_sum + lane_iou_loss(pred[rows, 6:] * (g.img_w - 1) / g.img_w, tgt[cols, 6:] / g.img_w)
    k = 1 * g.refine_layers
    return matched, cls_sum / k, reg_sum / k, iou_sum / k


def frame_loss(out: FrameOutput, gt: Tensor, g: Geometry):
    """loss4OLV3.py:100-123.  gt [1,L,6+S].  Returns (matched rows of branch B per stage, scalar loss)."""
    _, cls_a, reg_a, iou_a = branch_loss(out.predictions_fir, gt, g)
    matched_b, cls_b, reg_b, iou_b = branch_loss(out.predictions_sec, gt, g)
    d = torch.stack(list(out.gates), dim=0).squeeze().mean(dim=0)             # [N]
    delta = torch.median(cls_a - cls_b).detach()
    cls = torch.sum((1 - d) * (cls_a - delta / 2) + d * (cls_b + delta / 2))
    total = (reg_a + reg_b) * g.reg_weight + (iou_a + iou_b) * g.iou_weight + cls * g.cls_weight
    return matched_b, total


# --------------------------------------------------------------------------------------
# a12/a13 decode                                 libs/models/Router4OL.py:394-479
# --------------------------------------------------------------------------------------
def nms_rows(pred: Tensor, g: Geometry) -> Tensor:
    """[K,6+S] predictions -> [K,5+S] NMS rows (theta dropped; pixel/strip units). Router4OL.py:454-458."""
    rows = torch.cat([pred[:, :4], pred[:, 5:]], dim=-1).detach().clone()
    rows[:, 3] = rows[:, 3] * (g.img_w - 1)
    rows[:, 4] = rows[:, 4] * g.n_strips
    rows[:, 5:] = rows[:, 5:] * (g.img_w - 1)
    return rows


def lane_points(row: Tensor, g: Geometry) -> Optional[np.ndarray]:
    """Router4OL.py:402-427 for one kept lane (length already rounded to strips): [n,2] float64 or None."""
    xs = row[6:].clone()
    n = g.n_strips
    start = min(max(0, int(round(row[2].item() * n))), n)
    end = min(start + int(round(row[5].item())) - 1, g.num_points - 1)
    inside = ((xs[:start] >= 0.) & (xs[:start] <= 1.)).numpy()
    mask = ~(inside[::-1].cumprod()[::-1].astype(bool))
    xs[end + 1:] = -2
    head = xs[:start]
    head[torch.from_numpy(mask.copy())] = -2
    ys = prior_ys(g)
    sel = xs >= 0
    lx, ly = xs[sel].flip(0).double(), ys[sel].flip(0).double()
    if lx.numel() <= 1:
        return None
    return torch.stack([lx, ly], dim=1).numpy()


def decode_frame(lines: Tensor, g: Geometry, nms_fn):
    """Router4OL.py:437-479 for one frame.  lines [N,6+S].  nms_fn(rows, scores, thr, top_k)
    -> (keep, num, parent).  Returns dict(keep_inds bool[N], keep int64[k], lanes=[(points, sx, sy, conf)])."""
    scores = torch.softmax(lines[:, :2], dim=1)[:, 1]
    keep_inds = scores >= g.conf_threshold
    cand = lines[keep_inds]
    if cand.shape[0] == 0:
        return dict(keep_inds=keep_inds, keep=torch.zeros(0, dtype=torch.long), lanes=[], kept_rows=cand)
    keep, num, _ = nms_fn(nms_rows(cand, g), scores[keep_inds].contiguous(), g.nms_thres, g.max_lanes)
    keep = keep[:int(num)]
    kept = cand[keep].clone()
    kept[:, 5] = torch.round(kept[:, 5] * g.n_strips)
    lanes = []
    for row in kept:
        pts = lane_points(row, g)
        if pts is not None:
            lanes.append((pts, float(row[3]), float(row[2]), float(row[1])))
    return dict(keep_inds=keep_inds, keep=keep, lanes=lanes, kept_rows=kept)


# --------------------------------------------------------------------------------------
# a10  clip loop + cross-frame memory            libs/models/Router4OL.py:515-584
# --------------------------------------------------------------------------------------
def memory_tokens(attn_feats: Sequence[Tensor], positives: Sequence[Tensor]) -> List[Tensor]:
    """Per stage: positive anchors' tokens in prior-index order + one mean token of the rest."""
    toks = []
    for feat, posi in zip(attn_feats, positives):
        mask = torch.zeros(feat.shape[0], dtype=torch.bool)
        mask[posi] = True
        toks.append(torch.cat([feat[mask], feat[~mask].mean(dim=0, keepdim=True)], dim=0).detach())
    return toks


def clip_forward(sd: State, frames: Tensor, lanes: Optional[Tensor], g: Geometry, training: bool,
                 nms_fn=None, track_running_stats: bool = False, collect: Optional[dict] = None, feats=None):
    """frames [T,3,H,W]; lanes [T,L,6+S] (train).  Train: summed loss over frames.  Eval: per-frame decode dicts.
    feats (optional): this clip's pyramid maps computed elsewhere - the slice of a trunk pass over the frames of several
    clips, i.e. data-parallel ranks with SyncBatchNorm (trainOL.py:141) seen from one rank."""
    if feats is None:
        feats = fpn_neck(sd, resnet_trunk(sd, frames, g, training, track_running_stats))
    memory: List[List[Tensor]] = []
    total = 0.0
    decoded = []
    for t in range(frames.shape[0]):
        out = lane_head_frame(sd, [f[t:t + 1] for f in feats], memory, g, training)
        if training:
            matched, loss_t = frame_loss(out, lanes[t:t + 1], g)
            total = total + loss_t
            positives = matched
        else:
            d = torch.stack(list(out.gates), dim=0).mean(dim=0)                 # [1,N,1]
            lines = out.predictions_sec[-1] * d + out.predictions_fir[-1] * (1 - d)
            dec = decode_frame(lines[0], g, nms_fn)
            dec["lines"] = lines[0]
            decoded.append(dec)
            pos_idx = torch.where(dec["keep_inds"])[0][dec["keep"]]
            positives = [pos_idx] * g.refine_layers
        if collect is not None:
            collect.setdefault("frames", []).append(out)
            collect.setdefault("positives", []).append(positives)
            if training:
                collect.setdefault("frame_loss", []).append(float(loss_t.detach()))
        with torch.no_grad():
            memory.append(memory_tokens(out.attn_feats, positives))
            if t >= g.save_freq_max:
                memory.pop(0)
    if collect is not None:
        collect["fpn"] = feats
    return total if training else decoded
