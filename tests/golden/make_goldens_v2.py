"""Golden vectors for the Router4OLV2 model family: runs the REFERENCE's own Python
(/root/reference/libs/models/Router4OLV2.py = what testOLV3.py imports) on CPU, eval mode, with the deterministic synthetic
weights of tests/synth.py, and freezes tests/golden/v2_*.npz.  Build container only:  python tests/golden/make_goldens_v2.py

Imported from the reference unmodified: libs.models.Router4OLV2 (RouterOL, RouterV2, Encoder), libs.models.fpnV2,
libs.models.resnet, libs.models.Router (AdaptiveRouter4LaneV2), libs.models.utils.{dynamic_head,transformer,roi_gather},
libs.models.SeqFormer.position_encoding, libs.utils.lane.  Stubs for absent packages: those of make_goldens.py, with the
mmcv.cnn.ConvModule stand-in extended to what Router.py:93-106 asks of it (Conv1d without bias + BatchNorm1d named `bn`
+ ReLU: mmcv 1.2.5 semantics of conv_cfg=Conv1d / norm_cfg=BN1d / default act_cfg, bias='auto').

ONE repair of the reference, without which the model cannot be constructed at all: Router4OLV2.py:120-124 calls
`AdaptiveRouter4LaneV2(num_priors=..., features_channels=..., num_points=..., out_channels=1, reduction=4, stages=...)`
but Router.py:84 declares `__init__(self, features_channels, num_points, reduction, stages)` -> TypeError as shipped.
The constructor is wrapped to drop the two unknown keywords (`num_priors`, `out_channels`); nothing else is touched.
The training path stays unpinned: it cannot run as shipped (output key `predictions_lists` vs `predictions_fir` read by
libs/utils/loss4OL.py:177).  cfg = options/options4OLV3.py (72 points, neck 64/128/256 -> 16/32/64, save_freq 1,
save_freq_max 5, conf_threshold 0.5) at the frame sizes below.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.nn as nn

import make_goldens as MG
from oracle import phnet_cpu_v2 as O2
from tests import synth


def install_v2_shims():
    MG.install_shims()

    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                     conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **kw):
            super().__init__()
            one_d = conv_cfg is not None and conv_cfg.get("type") == "Conv1d"
            conv = nn.Conv1d if one_d else nn.Conv2d
            self.conv = conv(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                             groups=groups, bias=norm_cfg is None)
            self.has_norm = norm_cfg is not None
            if self.has_norm:
                assert norm_cfg["type"] == "BN1d" and one_d
                self.bn = nn.BatchNorm1d(out_channels)
            self.act = nn.ReLU(inplace) if act_cfg is not None else None

        def forward(self, x):
            x = self.conv(x)
            if self.has_norm:
                x = self.bn(x)
            return self.act(x) if self.act is not None else x

    sys.modules["mmcv.cnn"].ConvModule = ConvModule
    import libs.models.Router as R
    shipped = R.AdaptiveRouter4LaneV2.__init__

    def tolerant(self, num_priors=None, features_channels=None, num_points=None, out_channels=None, reduction=2, stages=3):
        shipped(self, features_channels=features_channels, num_points=num_points, reduction=reduction, stages=stages)
    R.AdaptiveRouter4LaneV2.__init__ = tolerant


def ref_cfg(g: O2.GeometryV2):
    return MG.Cfg(img_h=g.img_h, img_w=g.img_w, num_points=g.num_points, num_priors=g.num_priors, max_lanes=g.max_lanes,
                  save_freq=g.save_freq, save_freq_max=g.save_freq_max,
                  backbone=dict(resnet=g.arch, pretrained=False, replace_stride_with_dilation=[False, False, False], out_conv=False),
                  neck=dict(in_channels=list(g.neck_in), out_channels=list(g.neck_out), num_outs=3, start_level=0, end_level=-1,
                            attention=False),
                  test_parameters=dict(conf_threshold=g.conf_threshold, nms_thres=g.nms_thres, nms_topk=g.max_lanes),
                  dscfg=types.SimpleNamespace(crop_size=480, org_height=1280, org_width=1920))


def build_reference(g: O2.GeometryV2):
    from libs.models.Router4OLV2 import RouterOL
    torch.manual_seed(0)
    model = RouterOL(cfg=ref_cfg(g), criterion=None)
    ref_keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    spec = synth.state_spec_v2(g)
    assert list(ref_keys) == list(spec), "state_dict key order/names differ from tests/synth.state_spec_v2"
    assert all(tuple(ref_keys[k]) == tuple(spec[k]) for k in spec)
    # buffers the reference derives itself: our statements of them must be the same numbers
    own = model.state_dict()
    assert torch.equal(own["router.PositionEmbedding.pos_table"], O2.positional_table(g.num_priors, g.hidden))
    for s in range(g.refine_layers):
        assert torch.equal(own[f"router.sample_x_indexs_{s}"], O2.sample_x_indexs(g, s))
        assert torch.equal(own[f"router.prior_feat_ys_{s}"], O2.prior_feat_ys(g, s))
    model.load_state_dict(synth.make_state_v2(g), strict=True)
    orig = model.router.predictions_to_pred

    def keep_fp32(*a, **k):                                     # Router4OLV2.py:367-368 upgrades the buffer to float64 in place
        try:
            return orig(*a, **k)
        finally:
            model.router.prior_ys = model.router.prior_ys.float()
    model.router.predictions_to_pred = keep_fp32
    return model, ref_keys


def run_eval(model, g, T, out, seed=77, keep_stage_preds=True, keep_fpn=False):
    frames = synth.make_clip(g, T, seed=seed)
    model.eval()
    rec = {"fir": [], "sec": [], "gate": [], "attn": [], "lines": [], "keep_inds": [], "keep": [], "mem_rows": []}
    det = model.router
    fwd, gl = det.forward, det.get_lanes

    def fwd_hook(x, last_cuts=None):
        o, cut, diff = fwd(x, last_cuts)
        rec["fir"].append(torch.stack([p.detach()[0] for p in o["predictions_lists"]]))
        rec["sec"].append(torch.stack([p.detach()[0] for p in o["predictions_sec"]]))
        rec["gate"].append(torch.stack([d.detach()[0, :, 0] for d in diff]))
        rec["attn"].append(torch.stack([c.detach()[:, 0] for c in cut]))
        rec["mem_rows"].append(0 if last_cuts is None else sum(fr[0].shape[0] for fr in last_cuts))
        return o, cut, diff

    def gl_hook(output, *a, **k):
        rec["lines"].append(output.detach()[0].clone())
        dec, ki, kp = gl(output, *a, **k)
        rec["keep_inds"].append(ki.numpy().copy())
        rec["keep"].append(np.asarray(kp, dtype=np.int64).copy())
        return dec, ki, kp
    det.forward, det.get_lanes = fwd_hook, gl_hook
    fpn = {}
    if keep_fpn:
        h = model.backbone.register_forward_hook(lambda m, i, o: fpn.update({f"fpn{j}": t.detach() for j, t in enumerate(o)}))
    with torch.no_grad():
        res = model({"frame": frames, "lanes": torch.zeros(T, 4, 6 + g.num_points)})
    if keep_fpn:
        h.remove()
    det.forward, det.get_lanes = fwd, gl
    if keep_stage_preds:
        out["fir"] = torch.stack(rec["fir"]).numpy().astype(np.float32)          # [T,3,N,6+S]
        out["sec"] = torch.stack(rec["sec"]).numpy().astype(np.float32)
    out["gate"] = torch.stack(rec["gate"]).numpy()                               # [T,3,N]
    out["attn_mean"] = torch.stack(rec["attn"]).mean(dim=2).numpy()              # [T,3,256]: what the memory stores per frame
    out["mem_rows"] = np.array(rec["mem_rows"], dtype=np.int64)                  # memory tokens per stage seen by frame t
    out["lines"] = torch.stack(rec["lines"]).numpy().astype(np.float32)
    out["keep_inds"] = np.stack(rec["keep_inds"])
    kk = np.full((T, g.max_lanes), -1, dtype=np.int64)
    for t, k in enumerate(rec["keep"]):
        kk[t, :len(k)] = k
    out["keep"] = kk
    npts = np.zeros((T, g.max_lanes), dtype=np.int64)
    pts = np.zeros((T, g.max_lanes, g.num_points, 2), dtype=np.float64)
    meta = np.zeros((T, g.max_lanes, 3), dtype=np.float64)
    for t, lanes_t in enumerate(res["lane_lines"]):
        for j, lane in enumerate(lanes_t):
            npts[t, j] = len(lane.points)
            pts[t, j, :len(lane.points)] = lane.points
            meta[t, j] = [float(lane.metadata["start_x"]), float(lane.metadata["start_y"]), float(lane.metadata["conf"])]
    out["lane_npts"], out["lane_pts"], out["lane_meta"] = npts, pts, meta
    for k, v in fpn.items():
        out[k] = v.numpy()


def main():
    install_v2_shims()
    torch.set_num_threads(8)
    # ---- tiny geometry, 8 frames: three past the memory depth (save_freq_max = 5), every tensor kept -------------
    g = O2.GeometryV2(img_h=64, img_w=160)
    model, keys = build_reference(g)
    out = {}
    run_eval(model, g, 8, out, keep_fpn=True)
    np.savez_compressed(os.path.join(HERE, "v2_tiny_r18_64x160.npz"), **out)
    json.dump(keys, open(os.path.join(HERE, "state_keys_v2.json"), "w"))
    print("v2 tiny", out["keep"].tolist(), out["mem_rows"].tolist(),
          "hard-routed to branch B:", [(out["gate"][t].mean(0) >= 0.5).sum() for t in range(8)])
    # ---- the headline frame size (320x800, ResNet-18 as in options4OLV3.py), 6 frames ------------------------------
    g = O2.GeometryV2()
    model, _ = build_reference(g)
    out = {}
    run_eval(model, g, 6, out)
    np.savez_compressed(os.path.join(HERE, "v2_r18_320x800.npz"), **out)
    print("v2 320x800", out["keep"].tolist(), "hard-routed to branch B:", [(out["gate"][t].mean(0) >= 0.5).sum() for t in range(6)])


if __name__ == "__main__":
    main()
