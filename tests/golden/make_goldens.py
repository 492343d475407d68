"""Golden-vector generator: runs the REFERENCE's own Python (read-only checkout at /root/reference)
on CPU with synthetic deterministic weights/inputs and freezes small input/output fixtures under
tests/golden/.  Run in the build container only:   python tests/golden/make_goldens.py

What is imported from the reference, unmodified: libs.models.Router4OL (RouterOL, DetNetV2, Encoder),
libs.models.resnet, libs.models.fpn, libs.models.Router, libs.models.utils.{dynamic_head,transformer,roi_gather},
libs.models.SeqFormer.position_encoding, libs.utils.{loss4OLV3,dynamic_assign,dynamic_assignV2,focal_loss,lane}.

What is stubbed because the package is not installed in the image:
  cv2 / torchvision / timm / imgaug   -> inert placeholders (only imported, never called on this path)
  mmcv.cnn.ConvModule                 -> Conv2d(bias=True) (+ optional ReLU): the only behaviour fpn.py:70-85 uses
  libs.utils.utility.mask_iou         -> inert placeholder (imported by loss4OLV3.py, unused)
  libs.ops.nms                        -> oracle/lane_nms.py (the reference extension is CUDA-only); NMS outputs
                                         in the fixtures therefore pin the *callers*, not the NMS arithmetic.
Deviations applied for determinism (documented in DESIGN.md):
  * dropout p=0 inside detNet.transformer_Dec (the reference trains with p=0.1);
  * detNet.prior_ys is restored to float32 after each decode (the reference silently upgrades the buffer
    to float64 on the first eval call, Router4OL.py:398-399).
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import scipy.interpolate  # noqa: F401  (must be imported before the np.bool alias below)
import scipy.optimize  # noqa: F401
import torch
import torch.nn as nn

from oracle import lane_nms as oracle_nms
from oracle import phnet_cpu as O
from tests import synth

REF = "/root/reference"


def install_shims():
    for n in ["cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
              "timm", "timm.models", "imgaug", "imgaug.augmenters"]:
        sys.modules[n] = MagicMock()
    tl = types.ModuleType("timm.models.layers")
    tl.trunc_normal_ = nn.init.trunc_normal_
    tl.DropPath = nn.Identity
    sys.modules["timm.models.layers"] = tl

    class ConvModule(nn.Module):
        def __init__(self, i, o, k, stride=1, padding=0, dilation=1, groups=1, bias="auto", conv_cfg=None,
                     norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **kw):
            super().__init__()
            assert norm_cfg is None
            self.conv = nn.Conv2d(i, o, k, stride=stride, padding=padding, dilation=dilation, groups=groups, bias=True)
            self.act = nn.ReLU(inplace) if act_cfg is not None else None

        def forward(self, x):
            x = self.conv(x)
            return self.act(x) if self.act is not None else x

    mm, mc = types.ModuleType("mmcv"), types.ModuleType("mmcv.cnn")
    mc.ConvModule = ConvModule
    mm.cnn = mc
    sys.modules["mmcv"], sys.modules["mmcv.cnn"] = mm, mc
    np.bool = np.bool_
    sys.path.insert(0, REF)
    import libs  # noqa: F401
    ops = types.ModuleType("libs.ops")
    ops.nms = oracle_nms.lane_nms
    sys.modules["libs.ops"] = ops
    import libs.utils  # noqa: F401
    util = types.ModuleType("libs.utils.utility")
    util.mask_iou = lambda *a, **k: None
    sys.modules["libs.utils.utility"] = util


class Cfg(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def haskey(self, k):
        return k in self


def ref_cfg(g: O.Geometry) -> Cfg:
    return Cfg(img_h=g.img_h, img_w=g.img_w, num_points=g.num_points, num_priors=g.num_priors,
               max_lanes=g.max_lanes, save_freq_max=g.save_freq_max,
               backbone=dict(resnet=g.arch, pretrained=False, replace_stride_with_dilation=[False, False, False], out_conv=False),
               neck=dict(in_channels=[128, 256, 512], out_channels=64, num_outs=3, attention=False),
               cls_weight=g.cls_weight, reg_weight=g.reg_weight, iou_weight=g.iou_weight,
               test_parameters=dict(conf_threshold=g.conf_threshold, nms_thres=g.nms_thres, nms_topk=g.max_lanes),
               dscfg=types.SimpleNamespace(crop_size=480, org_height=1280, org_width=1920))


def build_reference(g: O.Geometry):
    from libs.models.Router4OL import RouterOL
    from libs.utils.loss4OLV3 import Criterion4OL
    cfg = ref_cfg(g)
    torch.manual_seed(0)
    model = RouterOL(cfg=cfg, criterion=Criterion4OL(cfg=cfg))
    ref_keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    spec = synth.state_spec(g)
    assert list(ref_keys) == list(spec), "state_dict key order/names differ from tests/synth.state_spec"
    for k, shp in spec.items():
        assert tuple(ref_keys[k]) == tuple(shp), (k, ref_keys[k], shp)
    model.load_state_dict(synth.make_state(g), strict=True)
    for m in model.detNet.transformer_Dec.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0
    orig = model.detNet.predictions_to_pred

    def keep_fp32(*a, **k):
        try:
            return orig(*a, **k)
        finally:
            model.detNet.prior_ys = model.detNet.prior_ys.float()
    model.detNet.predictions_to_pred = keep_fp32
    return model, ref_keys


def grad_digest(model):
    names, norms, heads = [], [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        heads.append(p.grad.flatten()[:4].double().numpy().tolist() + [0.0] * max(0, 4 - p.grad.numel()))
    return names, np.array(norms), np.array([h[:4] for h in heads])


RAGGED_COUNTS = (0, 4, 1, 2)      # valid lanes per frame of the ragged case: none, the label's maximum, one, two


def run_train(model, g, T, tag, out, keep_fpn=False, keep_preds=True, counts=None):
    frames, lanes = synth.make_clip(g, T), synth.make_targets(g, T, counts=counts)
    model.train()
    model.zero_grad()
    rec = {"fir": [], "sec": [], "gate": [], "matched": []}
    det, crit = model.detNet, model.criterion
    det_fwd, crit_fwd = det.forward, crit.forward

    def det_hook(x, last_cuts=None):
        o, cut, diff = det_fwd(x, last_cuts)
        rec["fir"].append(torch.stack([p.detach()[0] for p in o["predictions_fir"]]))
        rec["sec"].append(torch.stack([p.detach()[0] for p in o["predictions_sec"]]))
        rec["gate"].append(torch.stack([d.detach()[0, :, 0] for d in diff]))
        return o, cut, diff

    def crit_hook(o, gt, diff=None):
        m, l = crit_fwd(o, gt, diff)
        rec["matched"].append([np.asarray(x, dtype=np.int64) for x in m])
        rec.setdefault("frame_loss", []).append(float(l.detach()))
        return m, l
    det.forward, crit.forward = det_hook, crit_hook
    fpn = {}
    if keep_fpn:
        h = model.backbone.register_forward_hook(lambda m, i, o: fpn.update({f"fpn{j}": t.detach() for j, t in enumerate(o)}))
    loss = model({"frame": frames, "lanes": lanes})
    loss.backward()
    if keep_fpn:
        h.remove()
    det.forward, crit.forward = det_fwd, crit_fwd
    names, norms, heads = grad_digest(model)
    out[f"{tag}_loss"] = np.float64(loss.item())
    out[f"{tag}_frame_loss"] = np.array(rec["frame_loss"])
    if keep_preds:
        out[f"{tag}_fir"] = torch.stack(rec["fir"]).numpy()          # [T,3,N,6+S]
        out[f"{tag}_sec"] = torch.stack(rec["sec"]).numpy()
    out[f"{tag}_gate"] = torch.stack(rec["gate"]).numpy()            # [T,3,N]
    mm = np.full((T, 3, g.max_lanes), -1, dtype=np.int64)
    for t, per in enumerate(rec["matched"]):
        for s, idx in enumerate(per):
            mm[t, s, :len(idx)] = idx
    out[f"{tag}_matched"] = mm
    out[f"{tag}_grad_norm"] = norms
    out[f"{tag}_grad_head"] = heads
    for k, v in fpn.items():
        out[f"{tag}_{k}"] = v.numpy()
    bn = model.backbone.backbone.model.bn1
    out[f"{tag}_bn1_running_mean"] = bn.running_mean.numpy().copy()
    out[f"{tag}_bn1_running_var"] = bn.running_var.numpy().copy()
    return names


def run_eval(model, g, T, tag, out):
    frames, lanes = synth.make_clip(g, T, seed=77), synth.make_targets(g, T)
    model.load_state_dict(synth.make_state(g), strict=True)      # undo the BN running-stat updates of run_train
    model.eval()
    rec = {"lines": [], "keep_inds": [], "keep": []}
    det = model.detNet
    gl = det.get_lanes

    def gl_hook(output, *a, **k):
        rec["lines"].append(output.detach()[0].clone())
        dec, ki, kp = gl(output, *a, **k)
        rec["keep_inds"].append(ki.numpy().copy())
        rec["keep"].append(np.asarray(kp, dtype=np.int64).copy())
        return dec, ki, kp
    det.get_lanes = gl_hook
    with torch.no_grad():
        res = model({"frame": frames, "lanes": lanes})
    det.get_lanes = gl
    out[f"{tag}_lines"] = torch.stack(rec["lines"]).numpy()
    out[f"{tag}_keep_inds"] = np.stack(rec["keep_inds"])
    kk = np.full((T, g.max_lanes), -1, dtype=np.int64)
    for t, k in enumerate(rec["keep"]):
        kk[t, :len(k)] = k
    out[f"{tag}_keep"] = kk
    npts = np.zeros((T, g.max_lanes), dtype=np.int64)
    pts = np.zeros((T, g.max_lanes, g.num_points, 2), dtype=np.float64)
    meta = np.zeros((T, g.max_lanes, 3), dtype=np.float64)
    for t, lanes_t in enumerate(res["lane_lines"]):
        for j, lane in enumerate(lanes_t):
            npts[t, j] = len(lane.points)
            pts[t, j, :len(lane.points)] = lane.points
            meta[t, j] = [float(lane.metadata["start_x"]), float(lane.metadata["start_y"]), float(lane.metadata["conf"])]
    out[f"{tag}_lane_npts"], out[f"{tag}_lane_pts"], out[f"{tag}_lane_meta"] = npts, pts, meta


QUAD_SEEDS = (3407, 3408, 3409, 3410)     # rank r holds clips 2r, 2r+1
PAIR_SEEDS = (3407, 3408)


def run_pair(model, g, T, out, seeds=None, tag="pair"):
    """Two clips as two data-parallel ranks with SyncBatchNorm (trainOL.py:141: convert_sync_batchnorm + DDP), emulated in
    one CPU process with the reference's own modules only: the trunk runs ONCE over the frames of both clips (= batch
    statistics over both ranks, and their backward), then each clip goes through RouterOL.forward with a stub in place of
    the trunk that hands out the clip's slice of those feature maps; the SUM of the two clip losses is back-propagated
    (DDP would average: a factor 1/2 on every gradient)."""
    seeds = PAIR_SEEDS if seeds is None else seeds
    n_clips = len(seeds)
    clips = [synth.make_clip(g, T, seed=s) for s in seeds]
    lanes = synth.make_targets(g, T)
    model.train()
    model.zero_grad()
    enc = model.backbone
    feats = enc(torch.cat(clips))

    class Slice(nn.Module):
        def __init__(self, lo):
            super().__init__()
            self.lo = lo

        def forward(self, x):
            return tuple(f[self.lo:self.lo + T] for f in feats)

    rec = {"matched": [], "frame_loss": []}
    crit = model.criterion
    crit_fwd = crit.forward

    def crit_hook(o, gt, diff=None):
        m, l = crit_fwd(o, gt, diff)
        rec["matched"].append([np.asarray(x, dtype=np.int64) for x in m])
        rec["frame_loss"].append(float(l.detach()))
        return m, l
    crit.forward = crit_hook
    losses = []
    for b, fr in enumerate(clips):
        model.backbone = Slice(b * T)
        losses.append(model({"frame": fr, "lanes": lanes}))
    model.backbone = enc
    crit.forward = crit_fwd
    total = sum(losses[1:], losses[0])
    total.backward()
    names, norms, heads = grad_digest(model)
    out[f"{tag}_loss"] = np.float64(total.item())
    out[f"{tag}_clip_loss"] = np.array([float(l.item()) for l in losses])
    out[f"{tag}_frame_loss"] = np.array(rec["frame_loss"]).reshape(n_clips, T)
    mm = np.full((n_clips, T, 3, g.max_lanes), -1, dtype=np.int64)
    for i, per in enumerate(rec["matched"]):
        for s_, idx in enumerate(per):
            mm[i // T, i % T, s_, :len(idx)] = idx
    out[f"{tag}_matched"] = mm
    out[f"{tag}_grad_norm"], out[f"{tag}_grad_head"] = norms, heads
    bn = model.backbone.backbone.model.bn1
    out[f"{tag}_bn1_running_mean"] = bn.running_mean.numpy().copy()
    out[f"{tag}_bn1_running_var"] = bn.running_var.numpy().copy()
    return names


def strided_digest(t: torch.Tensor):
    return t[..., ::4, ::5].numpy().copy(), t.double().sum(dim=(2, 3)).numpy()


def main():
    install_shims()
    torch.set_num_threads(8)
    keys = {}
    if "--only-pair" in sys.argv:
        # ---- tiny geometry, two clips = two data-parallel ranks with SyncBatchNorm (BASELINE.json configs[2] semantics) ----
        g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
        model, _ = build_reference(g)
        out = {}
        names = run_pair(model, g, 3, out)
        assert names == json.load(open(os.path.join(HERE, "grad_names_resnet18.json")))
        np.savez_compressed(os.path.join(HERE, "tiny_pair_syncbn_r18_64x160.npz"), **out)
        print("tiny pair", out["pair_loss"], out["pair_clip_loss"].tolist(), out["pair_matched"][:, 0].tolist())
        return
    if "--only-quad" in sys.argv:
        # ---- BASELINE.json configs[2] as ONE workload, scaled down: 2 ranks x 2 clips per rank = 4 clips whose BatchNorm
        # statistics are joint (SyncBatchNorm across the ranks AND the batch dimension inside a rank), summed loss ----------
        g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
        model, _ = build_reference(g)
        out = {}
        names = run_pair(model, g, 2, out, seeds=QUAD_SEEDS, tag="quad")
        assert names == json.load(open(os.path.join(HERE, "grad_names_resnet18.json")))
        np.savez_compressed(os.path.join(HERE, "tiny_quad_syncbn_r18_64x160.npz"), **out)
        print("tiny quad", out["quad_loss"], out["quad_clip_loss"].tolist(), out["quad_matched"][:, 0].tolist())
        return
    if "--only-config4" in sys.argv:
        # ---- BASELINE.json configs[3] geometry: 10-frame clip 3x384x960 (its DLA-34 backbone does not exist in the reference
        # - ResNet-34 instead): other feature-map sizes (12x30 .. 48x120) and a TRAINING clip longer than the memory depth ----
        g = O.Geometry(img_h=384, img_w=960, arch="resnet34")
        model, _ = build_reference(g)
        out = {}
        run_train(model, g, 10, "train", out, keep_preds=False)
        run_eval(model, g, 10, "eval", out)
        out["eval_lines"] = out["eval_lines"].astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "config4_r34_384x960.npz"), **out)
        print("config4", out["train_loss"], out["train_frame_loss"].tolist(), out["eval_keep"].tolist())
        return
    if "--only-long" in sys.argv:
        # ---- tiny geometry, 11-frame eval clip: three frames past the memory depth (save_freq_max = 8), so the FIFO of
        # saveMemory4Test (Router4OL.py:563-584) drops its oldest entries ---------------------------------------------
        g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
        model, _ = build_reference(g)
        out = {}
        run_eval(model, g, 11, "eval", out)
        out["eval_lines"] = out["eval_lines"].astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "tiny_long_eval_r18_64x160.npz"), **out)
        print("tiny long eval", out["eval_keep"].tolist())
        return
    if "--only-ragged" in sys.argv:
        # ---- tiny geometry, ragged targets: frames with 0 / 4 / 1 / 2 valid lanes (empty-target branch of
        # loss4OLV3.py:45-48, a full 4x4 assignment, memory made of the mean token only) -------------------------
        g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
        model, _ = build_reference(g)
        out = {}
        run_train(model, g, len(RAGGED_COUNTS), "train", out, counts=RAGGED_COUNTS)
        np.savez_compressed(os.path.join(HERE, "tiny_ragged_r18_64x160.npz"), **out)
        print("tiny ragged", out["train_loss"], out["train_frame_loss"].tolist(), out["train_matched"].tolist())
        return
    # ---- tiny geometry: every tensor kept ------------------------------------------------------------
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
    model, keys["resnet18"] = build_reference(g)
    out = {}
    names = run_train(model, g, 3, "train", out, keep_fpn=True)
    run_eval(model, g, 4, "eval", out)
    np.savez_compressed(os.path.join(HERE, "tiny_r18_64x160.npz"), **out)
    json.dump(names, open(os.path.join(HERE, "grad_names_resnet18.json"), "w"))
    print("tiny", out["train_loss"], out["eval_keep"].tolist())
    # ---- config 1: single 320x800 frame, ResNet-18, forward only ---------------------------------------
    g = O.Geometry(arch="resnet18")
    model, _ = build_reference(g)
    model.eval()
    with torch.no_grad():
        feats = model.backbone(synth.make_clip(g, 1))
    out = {}
    for j, f in enumerate(feats):
        out[f"fpn{j}_strided"], out[f"fpn{j}_chansum"] = strided_digest(f)
    run_eval(model, g, 1, "eval", out)
    np.savez_compressed(os.path.join(HERE, "config1_r18_320x800.npz"), **out)
    print("config1 done", out["eval_keep"].tolist())
    # ---- config 2: 5-frame 320x800 clip, ResNet-34, fwd+bwd and eval --------------------------------------
    g = O.Geometry(arch="resnet34")
    model, keys["resnet34"] = build_reference(g)
    out = {}
    names = run_train(model, g, 5, "train", out)
    run_eval(model, g, 5, "eval", out)
    for k in ("train_fir", "train_sec", "eval_lines"):
        out[k] = out[k].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "config2_r34_320x800.npz"), **out)
    json.dump(names, open(os.path.join(HERE, "grad_names_resnet34.json"), "w"))
    json.dump(keys, open(os.path.join(HERE, "state_keys.json"), "w"))
    print("config2", out["train_loss"], out["eval_keep"].tolist())


if __name__ == "__main__":
    main()
