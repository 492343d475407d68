"""CPU oracle for the Router4OLV2 model family  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/ may import this file.  Functional fp32 restatement (torch CPU ops over a flat state dict with the reference's
key names) of what `testOLV3.py` runs: libs/models/Router4OLV2.py (RouterV2 :34-361, RouterOL :471-578), libs/models/fpnV2.py,
libs/models/Router.py:83-132 (AdaptiveRouter4LaneV2), libs/models/utils/dynamic_head.py:61-112 (DynamicConvV2),
libs/models/SeqFormer/position_encoding.py:61-86 (PositionalEncoding).  Evaluation path only: the training path of this family
cannot run as shipped (the model returns `predictions_lists`, Router4OLV2.py:283, while libs/utils/loss4OL.py:177 reads
`predictions_fir`).  Pinned against fixtures produced by the reference's own Python (tests/golden/make_goldens_v2.py ->
tests/golden/v2_*.npz; tests/test_oracle_v2.py).  Shared pieces (trunk, decoder layers, decode, Lane points) come from
oracle/phnet_cpu.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import phnet_cpu as O

Tensor = torch.Tensor
State = O.State


@dataclass
class GeometryV2:
    """cfg keys read by Router4OLV2.py:43-53,480-483 with the values of options/options4OLV3.py."""
    img_h: int = 320
    img_w: int = 800
    num_points: int = 72                                  # S
    num_priors: int = 240                                 # N
    sample_points: Tuple[int, ...] = (24, 48, 96)         # P per stage   (Router4OLV2.py:40)
    feat_channels: Tuple[int, ...] = (64, 32, 16)         # C per stage   (:36)  C*P = 1536 at every stage
    hidden: int = 256                                     # reg_hidden_dim (:37)
    refine_layers: int = 3
    max_lanes: int = 4
    save_freq: int = 1
    save_freq_max: int = 5
    conf_threshold: float = 0.5
    nms_thres: float = 50.0
    arch: str = "resnet18"
    neck_in: Tuple[int, ...] = (64, 128, 256)             # layer1..layer3 (the Encoder drops layer4, :28)
    neck_out: Tuple[int, ...] = (16, 32, 64)
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def n_strips(self) -> int:
        return self.num_points - 1


# ------------------------------------------------------------------------------------------------
# FPN V2                                          libs/models/fpnV2.py:70-100, 122-150
# ------------------------------------------------------------------------------------------------
def fpn_v2(sd: State, feats: Sequence[Tensor], prefix: str = "backbone.neck.") -> Tuple[Tensor, ...]:
    """Per-level widths: 1x1 laterals to out[i]; top-down: 1x1 `upsample_convs[i-1]` (out[i] -> out[i-1]) THEN nearest
    resize, added in place; 3x3 output convs.  ConvModule = bare Conv2d with bias (no norm, no activation)."""
    n = len(feats)
    lat = [F.conv2d(f, sd[f"{prefix}lateral_convs.{i}.conv.weight"], sd[f"{prefix}lateral_convs.{i}.conv.bias"])
           for i, f in enumerate(feats)]
    for i in range(n - 1, 0, -1):
        up = F.conv2d(lat[i], sd[f"{prefix}upsample_convs.{i - 1}.conv.weight"], sd[f"{prefix}upsample_convs.{i - 1}.conv.bias"])
        lat[i - 1] = lat[i - 1] + F.interpolate(up, size=lat[i - 1].shape[2:], mode="nearest")
    return tuple(F.conv2d(lat[i], sd[f"{prefix}fpn_convs.{i}.conv.weight"], sd[f"{prefix}fpn_convs.{i}.conv.bias"], padding=1)
                 for i in range(n))


def encoder_v2(sd: State, frames: Tensor, g: GeometryV2) -> Tuple[Tensor, ...]:
    """Router4OLV2.py:27-30: trunk outputs without the last level -> FPN V2 (eval: BatchNorm running statistics)."""
    g1 = O.Geometry(arch=g.arch, bn_eps=g.bn_eps, bn_momentum=g.bn_momentum)
    return fpn_v2(sd, O.resnet_trunk(sd, frames, g1, training=False)[:-1])


# ------------------------------------------------------------------------------------------------
# anchors and ROI pooling                         Router4OLV2.py:55-67, 143-178
# ------------------------------------------------------------------------------------------------
def sample_x_indexs(g: GeometryV2, stage: int) -> Tensor:
    return (torch.linspace(0, 1, steps=g.sample_points[stage], dtype=torch.float32) * g.n_strips).long()


def prior_feat_ys(g: GeometryV2, stage: int) -> Tensor:
    return torch.flip(1 - sample_x_indexs(g, stage).float() / g.n_strips, dims=[-1])


def _g1(g: GeometryV2) -> O.Geometry:
    return O.Geometry(img_h=g.img_h, img_w=g.img_w, num_points=g.num_points, num_priors=g.num_priors,
                      max_lanes=g.max_lanes, conf_threshold=g.conf_threshold, nms_thres=g.nms_thres)


def priors_from_embeddings(emb: Tensor, g: GeometryV2) -> Tuple[Tensor, Tensor]:
    """Router4OLV2.py:163-178: the V1 formula; the stage-0 sample columns."""
    pri, _ = O.priors_from_embeddings(emb, _g1(g))
    return pri, pri[:, 6 + sample_x_indexs(g, 0)]


def pool_anchor_features(fmap: Tensor, on_map: Tensor, g: GeometryV2, stage: int) -> Tensor:
    """fmap [1,C,h,w], on_map [1,N,P] -> [1,N,C,P] (Router4OLV2.py:143-161, 244-249)."""
    xs = torch.flip(on_map, dims=[2])
    ys = prior_feat_ys(g, stage).to(xs.dtype).view(1, 1, -1).expand_as(xs)
    grid = torch.stack([xs * 2.0 - 1.0, ys * 2.0 - 1.0], dim=-1)
    return F.grid_sample(fmap, grid, mode="bilinear", padding_mode="zeros", align_corners=True).permute(0, 2, 1, 3)


# ------------------------------------------------------------------------------------------------
# routing gate V2                                 libs/models/Router.py:83-132
# ------------------------------------------------------------------------------------------------
def routing_gate_v2(sd: State, stage: int, feat: Tensor, g: GeometryV2, prefix: str = "router.router.") -> Tensor:
    """feat [1,N,C,P] -> [1,N,1]: Conv1d(k3)+BN1d+ReLU, Conv1d(k1)+BN1d+ReLU (mmcv ConvModule: conv without bias, norm,
    activation), Flatten, Linear(96 -> P), mean over the P outputs, sigmoid.  Eval: running statistics."""
    b, n, c, p = feat.shape
    x = feat.reshape(b * n, c, p)
    for j, pad in ((0, 1), (1, 0)):
        q = f"{prefix}layers.{stage}.{j}."
        x = F.conv1d(x, sd[q + "conv.weight"], None, padding=pad)
        x = F.batch_norm(x, sd[q + "bn.running_mean"], sd[q + "bn.running_var"], sd[q + "bn.weight"], sd[q + "bn.bias"],
                         False, g.bn_momentum, g.bn_eps)
        x = F.relu(x)
    q = f"{prefix}layers.{stage}.3."
    s = F.linear(x.flatten(1), sd[q + "weight"], sd[q + "bias"]).reshape(b, n, -1)
    return torch.sigmoid(s.mean(dim=-1, keepdim=True))


# ------------------------------------------------------------------------------------------------
# DynamicConvV2                                   libs/models/utils/dynamic_head.py:61-112
# ------------------------------------------------------------------------------------------------
def dynamic_head_v2(sd: State, stage: int, pro_feat: Tensor, roi: Tensor, prefix: str = "router.DHead_series.") -> Tensor:
    """pro_feat [1,N,256], roi [1,N,P,C] -> [1,N,256]."""
    q = f"{prefix}{stage}."
    b, n, pnum, c = roi.shape
    roi = roi.reshape(b * n, pnum, c)
    pro = pro_feat.reshape(b * n, -1)
    lin = O._lin
    w1 = lin(sd, q + "dynamic_layer_1.1", lin(sd, q + "dynamic_layer_1.0", pro)).reshape(b * n, c, 2 * c)
    f = torch.bmm(roi, w1)
    f = F.relu(F.layer_norm(f, [2 * c], sd[q + "norm1.weight"], sd[q + "norm1.bias"]))
    w2 = lin(sd, q + "dynamic_layer_2.1", lin(sd, q + "dynamic_layer_2.0", f.detach().flatten(1))).reshape(b * n, 2 * c, c)
    f = torch.bmm(f, w2)
    f = F.relu(F.layer_norm(f, [c], sd[q + "norm2.weight"], sd[q + "norm2.bias"]))
    f = lin(sd, q + "out_layer.1", lin(sd, q + "out_layer.0", f.flatten(1)))
    f = F.layer_norm(f, [f.shape[-1]], sd[q + "norm3.weight"], sd[q + "norm3.bias"])
    return f.view(b, n, -1)


# ------------------------------------------------------------------------------------------------
# branches                                        Router4OLV2.py:288-361
# ------------------------------------------------------------------------------------------------
def branch_heads_v2(sd: State, feat: Tensor, priors: Tensor, g: GeometryV2, suffix: str, prefix: str = "router.") -> Tuple[Tensor, Tensor]:
    """Two towers (cls, reg); `reg_layers` emits (d start_y, d start_x, d theta, length, S offsets)."""
    cls = O._lin(sd, f"{prefix}cls_layers{suffix}", O._tower(sd, f"{prefix}cls_modules{suffix}", feat))
    reg = O._lin(sd, f"{prefix}reg_layers{suffix}", O._tower(sd, f"{prefix}reg_modules{suffix}", feat))
    n = priors.shape[1]
    reg = reg.reshape(1, n, 4 + g.num_points)
    return O.update_priors(priors, cls.reshape(1, n, 2), reg[..., :4], reg[..., 4:], _g1(g))


def positional_table(n_position: int, d_hid: int, temperature: float = 64.0) -> Tensor:
    """position_encoding.py:75-86 (normalize=False): interleaved sin / cos of position / T^(2*(i//2)/d)."""
    pos = torch.arange(n_position, dtype=torch.float32)
    dim_t = torch.arange(d_hid, dtype=torch.float32)
    dim_t = temperature ** (2 * (torch.div(dim_t, 2, rounding_mode="floor")) / d_hid)
    tab = pos[..., None] / dim_t
    tab[:, 0::2] = tab[:, 0::2].sin()
    tab[:, 1::2] = tab[:, 1::2].cos()
    return tab


@dataclass
class FrameOutputV2:
    predictions_fir: List[Tensor]
    predictions_sec: List[Tensor]
    attn_feats: List[Tensor]
    gates: List[Tensor]
    stage_inputs: List[dict] = field(default_factory=list)
    locals_: List[Tensor] = field(default_factory=list)


def lane_head_frame_v2(sd: State, feats: Sequence[Tensor], memory: Optional[List[List[Tensor]]], g: GeometryV2,
                       prefix: str = "router.") -> FrameOutputV2:
    """Router4OLV2.py:224-286, eval.  feats = FPN outputs fine -> coarse, each [1,C_l,h,w]; memory = None (the first
    `save_freq` frames: the decoder attends to the frame's own tokens, :320-325) or the list of stored frames."""
    levels = list(feats)[::-1]
    priors, on_map = sd[prefix + "priors"].unsqueeze(0), sd[prefix + "priors_on_featmap"].unsqueeze(0)
    pro = sd[prefix + "pro_embedding.weight"].unsqueeze(0)
    pos = sd[prefix + "PositionEmbedding.pos_table"]
    out = FrameOutputV2([], [], [], [])
    for stage in range(g.refine_layers):
        mem = None
        if memory is not None:
            mem = torch.cat([fr[stage] for fr in memory], dim=0)
        out.stage_inputs.append(dict(priors=priors.detach(), on_map=on_map.detach(), pro=pro.detach(), mem=mem))
        pooled = pool_anchor_features(levels[stage], on_map, g, stage)
        gate = routing_gate_v2(sd, stage, pooled.detach(), g, prefix + "router.")
        local = dynamic_head_v2(sd, stage, pro, pooled.transpose(2, 3), prefix + "DHead_series.")
        pro = local.detach()
        out.locals_.append(local.detach())
        pred_a, lines_a = branch_heads_v2(sd, local, priors, g, "", prefix)
        attn = local[0] + pos                                                   # content + sinusoidal table (:266-269)
        kv = mem if (mem is not None and mem.shape[0] != 0) else attn
        glob = O.temporal_decoder(sd, attn, kv, prefix + "transformer_Dec.")
        pred_b, lines_b = branch_heads_v2(sd, glob.unsqueeze(0), priors, g, "_sec", prefix)
        out.predictions_fir.append(pred_a)
        out.predictions_sec.append(pred_b)
        out.attn_feats.append(attn)
        out.gates.append(gate)
        w = gate.detach()
        if stage != g.refine_layers - 1:
            priors = ((1 - w) * lines_a + w * lines_b).detach().clone()
            on_map = priors[..., 6 + sample_x_indexs(g, stage + 1)]
    return out


def clip_forward_eval_v2(sd: State, frames: Tensor, g: GeometryV2, nms_fn, collect: Optional[dict] = None, feats=None):
    """Router4OLV2.py:485-558, eval: per frame hard routing `where(mean gate >= 0.5, branch B, branch A)` of the last
    stage (:508-511), decode + NMS, memory FIFO.  saveMemory4Test (:570-578) writes its positives into a temporary
    (`mask[keep_inds][keep] = True`), so the stored memory of a frame is ONE token per stage: the mean over all anchors."""
    if feats is None:
        feats = encoder_v2(sd, frames, g)
    memory: List[List[Tensor]] = []
    decoded = []
    g1 = _g1(g)
    for t in range(frames.shape[0]):
        out = lane_head_frame_v2(sd, [f[t:t + 1] for f in feats], None if t < g.save_freq else memory, g)
        d = torch.stack(list(out.gates), dim=0).mean(dim=0)                      # [1,N,1]
        lines = torch.where(d >= 0.5, out.predictions_sec[-1], out.predictions_fir[-1])
        dec = O.decode_frame(lines[0], g1, nms_fn)
        dec["lines"] = lines[0]
        decoded.append(dec)
        if collect is not None:
            collect.setdefault("frames", []).append(out)
        memory.append([a.mean(dim=0, keepdim=True).detach() for a in out.attn_feats])
        if t >= g.save_freq_max:
            memory.pop(0)
    if collect is not None:
        collect["fpn"] = feats
    return decoded
