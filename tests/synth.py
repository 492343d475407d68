"""Deterministic synthetic weights / clips / targets shared by the golden generator and the tests.

Everything is drawn from numpy PCG64 streams keyed by crc32(name), so the same tensors can be
rebuilt on any machine without shipping 90 M parameters.  The key/shape list is our own
statement of the reference's state_dict layout (SURVEY.md section 5 "checkpoint"); the golden
generator asserts it against the real reference and freezes it in tests/golden/state_keys.json.
"""
import math
import re
import zlib
from collections import OrderedDict

import numpy as np
import torch

from oracle import phnet_cpu as O
from oracle import phnet_cpu_v2 as O2


def _rng(name: str, salt: int = 0):
    return np.random.default_rng([zlib.crc32(name.encode()), salt])


def state_spec(g: O.Geometry) -> "OrderedDict[str, tuple]":
    """name -> shape, in the reference's registration order."""
    spec = OrderedDict()
    C, S, N, P = g.feat_channels, g.num_points, g.num_priors, g.sample_points

    def bn(prefix, c):
        spec[prefix + ".weight"] = (c,)
        spec[prefix + ".bias"] = (c,)
        spec[prefix + ".running_mean"] = (c,)
        spec[prefix + ".running_var"] = (c,)
        spec[prefix + ".num_batches_tracked"] = ()

    def lin(prefix, i, o):
        spec[prefix + ".weight"] = (o, i)
        spec[prefix + ".bias"] = (o,)

    def ln(prefix, *shape):
        spec[prefix + ".weight"] = tuple(shape)
        spec[prefix + ".bias"] = tuple(shape)

    p = "backbone.backbone.model."
    spec[p + "conv1.weight"] = (64, 3, 7, 7)
    bn(p + "bn1", 64)
    cin = 64
    for li, nb in enumerate(O.BLOCKS_PER_LAYER[g.arch]):
        w = O.LAYER_WIDTH[li]
        for bi in range(nb):
            q = f"{p}layer{li + 1}.{bi}."
            spec[q + "conv1.weight"] = (w, cin, 3, 3)
            bn(q + "bn1", w)
            spec[q + "conv2.weight"] = (w, w, 3, 3)
            bn(q + "bn2", w)
            if bi == 0 and li > 0:
                spec[q + "downsample.0.weight"] = (w, cin, 1, 1)
                bn(q + "downsample.1", w)
            cin = w
    p = "backbone.neck."
    for i, c in enumerate(O.LAYER_WIDTH[1:]):
        spec[f"{p}lateral_convs.{i}.conv.weight"] = (C, c, 1, 1)
        spec[f"{p}lateral_convs.{i}.conv.bias"] = (C,)
    for i in range(3):
        spec[f"{p}fpn_convs.{i}.conv.weight"] = (C, C, 3, 3)
        spec[f"{p}fpn_convs.{i}.conv.bias"] = (C,)
    d = "detNet."
    spec[d + "sample_x_indexs"] = (P,)
    spec[d + "prior_feat_ys"] = (P,)
    spec[d + "prior_ys"] = (S,)
    spec[d + "priors"] = (N, 6 + S)
    spec[d + "priors_on_featmap"] = (N, P)
    spec[d + "prior_embeddings.weight"] = (N, 3)
    for suffix, width in (("", C), ("_sec", 2 * C)):
        for kind in ("reg", "cls", "iou"):
            for idx in (0, 2):
                lin(f"{d}{kind}_modules{suffix}.{idx}", width, width)
        lin(f"{d}reg_layers{suffix}", width, 4)
        lin(f"{d}cls_layers{suffix}", width, 2)
        lin(f"{d}iou_layers{suffix}", width, S)
    E = 2 * C
    for li in range(2):
        q = f"{d}transformer_Dec.layers.{li}."
        for att in ("self_attn", "multihead_attn"):
            spec[q + att + ".in_proj_weight"] = (3 * E, E)
            spec[q + att + ".in_proj_bias"] = (3 * E,)
            lin(q + att + ".out_proj", E, E)
        lin(q + "linear1", E, 256)
        lin(q + "linear2", 256, E)
        for k in (1, 2, 3):
            ln(q + f"norm{k}", E)
    ln(d + "transformer_Dec.norm", E)
    spec[d + "PositionEmbedding.embed.weight"] = (N, C)
    for s in range(g.refine_layers):
        q = f"{d}DHead_series.{s}."
        lin(q + "dynamic_layer_1.0", C, C * 2 * C // 8)
        lin(q + "dynamic_layer_1.1", C * 2 * C // 8, C * 2 * C)
        lin(q + "dynamic_layer_2.0", 2 * C * P, C * 2 * C // 8)
        lin(q + "dynamic_layer_2.1", C * 2 * C // 8, C * 2 * C)
        ln(q + "norm1", 2 * C)
        ln(q + "norm2", C)
        lin(q + "out_layer.0", C * P, 6 * C)
        lin(q + "out_layer.1", 6 * C, C)
        ln(q + "norm3", C)
    spec[d + "pro_embedding.weight"] = (N, C)
    r = d + "router."
    for s in range(g.refine_layers):
        lin(f"{r}layers.{s}.0", C * P, C * P // 4)
        lin(f"{r}layers.{s}.2", C * P // 4, 1)
    for s in range(g.refine_layers):
        ln(f"{r}pre_norm.{s}", C, P)
    for s in range(g.refine_layers):
        for b in range(4):
            q = f"{r}DWNets.{s}.{b}."
            spec[q + "0.weight"] = (N, 1, 3, 3)
            spec[q + "0.bias"] = (N,)
            ln(q + "1", C, P)
            spec[q + "3.weight"] = (N, 1, 3, 3)
            spec[q + "3.bias"] = (N,)
            ln(q + "4", C, P)
    return spec


_NORM_RE = re.compile(r"(\.bn\d?|downsample\.1|\.norm\d?|pre_norm\.\d|DWNets\.\d\.\d\.[14])\.(weight|bias)$")


def synth_tensor(name: str, shape: tuple, g: O.Geometry) -> torch.Tensor:
    r = _rng(name)
    f32 = np.float32
    if name.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.long)
    if name.endswith("running_mean"):
        return torch.from_numpy(r.normal(0, 0.1, shape).astype(f32))
    if name.endswith("running_var"):
        return torch.from_numpy(r.uniform(0.5, 1.5, shape).astype(f32))
    m = _NORM_RE.search(name)
    if m:
        a = r.uniform(0.5, 1.5, shape) if m.group(2) == "weight" else r.normal(0, 0.1, shape)
        return torch.from_numpy(a.astype(f32))
    if name.endswith("prior_embeddings.weight"):
        e = O.initial_anchor_embeddings(g).numpy()
        return torch.from_numpy((e + r.normal(0, 0.004, e.shape)).astype(f32))
    if name.endswith("pro_embedding.weight"):
        return torch.from_numpy(r.normal(0, 1.0, shape).astype(f32))
    if name.endswith("PositionEmbedding.embed.weight"):
        return torch.from_numpy(r.uniform(0, 1.0, shape).astype(f32))
    if name.endswith(".bias") or name.endswith("in_proj_bias"):
        b = r.normal(0, 0.05, shape).astype(f32)
        if re.search(r"reg_layers(_sec)?\.bias$", name):
            b = (b * 0.2).astype(f32)
            b[3] += 0.6                     # lane length ~0.6 of the image so that NMS has work to do
        return torch.from_numpy(b)
    if len(shape) == 4:                     # conv weights
        fan_in = shape[1] * shape[2] * shape[3]
        gain = 1.0 if shape[1] == 1 else math.sqrt(2.0)
        return torch.from_numpy(r.normal(0, gain / math.sqrt(fan_in), shape).astype(f32))
    if len(shape) == 2:                     # linear weights
        std = 1.0 / math.sqrt(shape[1])
        if re.search(r"reg_layers(_sec)?\.weight$", name):
            std *= 0.1
        if re.search(r"iou_layers(_sec)?\.weight$", name):
            std *= 0.02
        return torch.from_numpy(r.normal(0, std, shape).astype(f32))
    raise KeyError(name)


def make_state(g: O.Geometry) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    for name, shape in state_spec(g).items():
        if name.split(".")[-1] in ("sample_x_indexs", "prior_feat_ys", "prior_ys", "priors", "priors_on_featmap"):
            continue
        sd[name] = synth_tensor(name, shape, g)
    pri, on_map = O.priors_from_embeddings(sd["detNet.prior_embeddings.weight"], g)
    sd["detNet.sample_x_indexs"] = O.sample_x_indexs(g)
    sd["detNet.prior_feat_ys"] = O.prior_feat_ys(g)
    sd["detNet.prior_ys"] = O.prior_ys(g)
    sd["detNet.priors"] = pri.clone()
    sd["detNet.priors_on_featmap"] = on_map.clone()
    return OrderedDict((k, sd[k]) for k in state_spec(g))


def calibrate_running_stats_(sd, frames: torch.Tensor, arch: str, bn_eps: float = 1e-5):
    """Replaces the trunk's BatchNorm running statistics (synthetic N(0, 0.1) / U(0.5, 1.5) values that no real checkpoint would
    hold: under them the eval-mode trunk does not normalise and its activations grow to 3.5e4 over 34 layers) by the batch
    statistics of `frames` - one training-mode pass of the oracle's trunk with momentum 1.  Eval activations are then O(1-10),
    as with trained weights, and comparisons against the oracle on this state need no allowance for the rounding of huge
    intermediate values.  In place; returns sd.  (Only for tests that compare against the ORACLE run on the same state - the
    fixtures produced by the reference were generated with the uncalibrated statistics and keep them.)"""
    with torch.no_grad():
        O.resnet_trunk(sd, frames, O.Geometry(arch=arch, bn_eps=bn_eps, bn_momentum=1.0), True, True)
    return sd


def make_clip(g: O.Geometry, T: int, seed: int = 3407) -> torch.Tensor:
    r = np.random.default_rng([seed, T, g.img_h, g.img_w])
    return torch.from_numpy(r.standard_normal((T, 3, g.img_h, g.img_w), dtype=np.float32))


def make_targets(g: O.Geometry, T: int, n_lanes: int = 3, max_lanes_in_label: int = 4, counts=None) -> torch.Tensor:
    """Straight synthetic lanes in the label layout of libs/dataset/openlane/transforms.py:264-297:
    [neg flag, pos flag, start_y, start_x/(W-1), theta, len/n_strips, xs in pixels (bottom->top), -1e5 invalid].
    counts (optional, length T): number of valid lanes per frame (ragged clips: 0 .. max_lanes_in_label)."""
    S, W, H = g.num_points, g.img_w, g.img_h
    strip = H / g.n_strips
    out = np.full((T, max_lanes_in_label, 6 + S), -1e5, dtype=np.float32)
    out[:, :, 0] = 1
    out[:, :, 1] = 0
    x0s, slopes = (0.2, 0.45, 0.7, 0.85), (4.0, 0.5, -4.0, -6.0)
    for t in range(T):
        for j in range(min(n_lanes if counts is None else int(counts[t]), max_lanes_in_label)):
            xs = x0s[j] * W + 5.0 * (t % 8) + slopes[j] * np.arange(S)          # (t % 8: long clips stay inside the tiny image)
            valid = (xs >= 0) & (xs < W)
            n = int(np.argmin(valid)) if not valid.all() else S
            n = min(n, S - 4 - j)
            xs = xs[:n]
            th = [math.atan(i * strip / (xs[i] - xs[0] + 1e-5)) / math.pi for i in range(1, n)]
            th = [v if v > 0 else 1 - abs(v) for v in th]
            out[t, j, 0], out[t, j, 1] = 0, 1
            out[t, j, 2] = 0.0
            out[t, j, 3] = xs[0] / (W - 1)
            out[t, j, 4] = sum(th) / len(th)
            out[t, j, 5] = n / g.n_strips
            out[t, j, 6:6 + n] = xs
    return torch.from_numpy(out)


# ------------------------------------------------------------------------------------------------ Router4OLV2 family
def state_spec_v2(g: "O2.GeometryV2") -> "OrderedDict[str, tuple]":
    """name -> shape of libs.models.Router4OLV2.RouterOL.state_dict(), in registration order (frozen against the
    reference in tests/golden/state_keys_v2.json by make_goldens_v2.py)."""
    spec = OrderedDict()
    g1 = O.Geometry(arch=g.arch)
    for k, v in state_spec(g1).items():                     # the trunk is the V1 trunk (all four layers are registered)
        if k.startswith("backbone.backbone."):
            spec[k] = v
    S, N, E = g.num_points, g.num_priors, g.hidden

    def bn(prefix, c):
        for leaf, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)), ("num_batches_tracked", ())):
            spec[f"{prefix}.{leaf}"] = shp

    def lin(prefix, i, o):
        spec[prefix + ".weight"] = (o, i)
        spec[prefix + ".bias"] = (o,)

    def ln(prefix, *shape):
        spec[prefix + ".weight"] = tuple(shape)
        spec[prefix + ".bias"] = tuple(shape)

    p = "backbone.neck."
    for i, (ci, co) in enumerate(zip(g.neck_in, g.neck_out)):
        spec[f"{p}lateral_convs.{i}.conv.weight"] = (co, ci, 1, 1)
        spec[f"{p}lateral_convs.{i}.conv.bias"] = (co,)
    for i, co in enumerate(g.neck_out):
        spec[f"{p}fpn_convs.{i}.conv.weight"] = (co, co, 3, 3)
        spec[f"{p}fpn_convs.{i}.conv.bias"] = (co,)
    for i in range(len(g.neck_out) - 1):
        spec[f"{p}upsample_convs.{i}.conv.weight"] = (g.neck_out[i], g.neck_out[i + 1], 1, 1)
        spec[f"{p}upsample_convs.{i}.conv.bias"] = (g.neck_out[i],)
    d = "router."
    for s, pnum in enumerate(g.sample_points):
        spec[f"{d}sample_x_indexs_{s}"] = (pnum,)
        spec[f"{d}prior_feat_ys_{s}"] = (pnum,)
    spec[d + "prior_ys"] = (S,)
    spec[d + "priors"] = (N, 6 + S)
    spec[d + "priors_on_featmap"] = (N, g.sample_points[0])
    spec[d + "prior_embeddings.weight"] = (N, 3)
    for suffix in ("", "_sec"):
        for kind in ("reg", "cls"):
            for idx in (0, 2):
                lin(f"{d}{kind}_modules{suffix}.{idx}", E, E)
        lin(f"{d}reg_layers{suffix}", E, S + 4)
        lin(f"{d}cls_layers{suffix}", E, 2)
    for li in range(2):
        q = f"{d}transformer_Dec.layers.{li}."
        for att in ("self_attn", "multihead_attn"):
            spec[q + att + ".in_proj_weight"] = (3 * E, E)
            spec[q + att + ".in_proj_bias"] = (3 * E,)
            lin(q + att + ".out_proj", E, E)
        lin(q + "linear1", E, 512)
        lin(q + "linear2", 512, E)
        for k in (1, 2, 3):
            ln(q + f"norm{k}", E)
    ln(d + "transformer_Dec.norm", E)
    spec[d + "PositionEmbedding.pos_table"] = (N, E)
    for s, (c, pnum) in enumerate(zip(g.feat_channels, g.sample_points)):
        q = f"{d}DHead_series.{s}."
        npar = c * 2 * c
        lin(q + "dynamic_layer_1.0", E, npar // 4)
        lin(q + "dynamic_layer_1.1", npar // 4, npar)
        lin(q + "dynamic_layer_2.0", 2 * c * pnum, npar // 4)
        lin(q + "dynamic_layer_2.1", npar // 4, npar)
        ln(q + "norm1", 2 * c)
        ln(q + "norm2", c)
        lin(q + "out_layer.0", c * pnum, 2 * E)
        lin(q + "out_layer.1", 2 * E, E)
        ln(q + "norm3", E)
    spec[d + "pro_embedding.weight"] = (N, E)
    r = d + "router."
    clast = g.feat_channels[-1]
    for s, (c, pnum) in enumerate(zip(g.feat_channels, g.sample_points)):
        spec[f"{r}layers.{s}.0.conv.weight"] = (c // 4, c, 3)
        bn(f"{r}layers.{s}.0.bn", c // 4)
        spec[f"{r}layers.{s}.1.conv.weight"] = (c // clast, c // 4, 1)
        bn(f"{r}layers.{s}.1.bn", c // clast)
        lin(f"{r}layers.{s}.3", c * pnum // clast, pnum)
    return spec


_V2_BUFFERS = re.compile(r"(sample_x_indexs_\d|prior_feat_ys_\d|prior_ys|priors|priors_on_featmap|pos_table)$")


def make_state_v2(g: "O2.GeometryV2") -> "OrderedDict[str, torch.Tensor]":
    spec = state_spec_v2(g)
    g1 = O.Geometry(img_h=g.img_h, img_w=g.img_w, num_points=g.num_points, num_priors=g.num_priors, arch=g.arch)
    sd = OrderedDict()
    for name, shape in spec.items():
        if _V2_BUFFERS.search(name):
            continue
        if len(shape) == 3:                                  # Conv1d weights of the gate
            r = _rng(name)
            sd[name] = torch.from_numpy(r.normal(0, math.sqrt(2.0 / (shape[1] * shape[2])), shape).astype(np.float32))
            continue
        sd[name] = synth_tensor(name, shape, g1)
    for suffix in ("", "_sec"):                              # rows 4.. of reg_layers are the per-row x offsets: keep them small
        sd[f"router.reg_layers{suffix}.weight"][4:] *= 0.2
        sd[f"router.reg_layers{suffix}.bias"][4:] *= 0.2
    for s in range(g.refine_layers):                         # gate logits of both signs (hard routing takes both branches)
        sd[f"router.router.layers.{s}.3.bias"] -= sd[f"router.router.layers.{s}.3.bias"].mean()
    pri, on_map = O2.priors_from_embeddings(sd["router.prior_embeddings.weight"], g)
    for s in range(g.refine_layers):
        sd[f"router.sample_x_indexs_{s}"] = O2.sample_x_indexs(g, s)
        sd[f"router.prior_feat_ys_{s}"] = O2.prior_feat_ys(g, s)
    sd["router.prior_ys"] = O.prior_ys(g1)
    sd["router.priors"] = pri.clone()
    sd["router.priors_on_featmap"] = on_map.clone()
    sd["router.PositionEmbedding.pos_table"] = O2.positional_table(g.num_priors, g.hidden)
    return OrderedDict((k, sd[k]) for k in spec)


def observe_criterion(crit, record):
    """record(output, gt_lane, diff, matched, loss) for every frame the criterion sees, whichever entry the schedule uses (one call
    per frame, or `clip_loss` for the frames of a clip at once).  Returns undo()."""
    fwd = crit.forward

    def hook(o, gt, diff=None):
        m, l = fwd(o, gt, diff)
        record(o, gt, diff, m, l)
        return m, l
    crit.forward = hook
    crit.frame_observer = record

    def undo():
        crit.forward = fwd
        crit.frame_observer = None
    return undo
