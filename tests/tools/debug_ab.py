"""A/B on the GPU box: the tiny ragged training case with the packed-weight 3x3 kernel on and off - loss, FPN maps and
per-parameter gradients side by side (largest differences first), against the fp64 oracle values where given."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import phnet_cpu as O
from phnet_amd import hip_ops as K
from phnet_amd.config import make_cfg
from phnet_amd.libs.models.Router4OL import RouterOL
from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
from tests import synth


def run(flag, g, T, counts):
    K.CONV3P = flag
    cfg = make_cfg(img_h=g.img_h, img_w=g.img_w, arch=g.arch)
    model = RouterOL(cfg, Criterion4OL(cfg))
    model.load_state_dict(synth.make_state(g), strict=True)
    for m in model.detNet.transformer_Dec.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    model = model.cuda().train()
    frames, lanes = synth.make_clip(g, T).cuda(), synth.make_targets(g, T, counts=counts).cuda()
    feats = []
    h = model.backbone.register_forward_hook(lambda m, i, o: feats.extend(t.detach().clone() for t in o))
    from tests.test_model_gpu import _record_heads
    rec, undo = _record_heads(model)
    loss = model({"frame": frames, "lanes": lanes})
    loss.backward()
    h.remove(); undo()
    torch.cuda.synchronize()
    RECS.append(rec)
    return float(loss), feats, {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}


RECS = []


def main():
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
    la, fa, ga = run(True, g, 4, (0, 4, 1, 2))
    lb, fb, gb = run(False, g, 4, (0, 4, 1, 2))
    lc, fc, gc = run(False, g, 4, (0, 4, 1, 2))
    print("loss packed", la, "generic", lb, "generic again", lc)
    for i, (a, b, c) in enumerate(zip(fa, fb, fc)):
        print(f"FPN level {i}: max|packed - generic| {float((a - b).abs().max()):.3e}  generic run-to-run {float((b - c).abs().max()):.3e}  scale {float(b.abs().max()):.3e}")
    ra, rb = RECS[0], RECS[1]
    for t in range(4):
        for name in ("fir", "sec", "gate"):
            d = (ra[name][t].double() - rb[name][t].double()).abs()
            per_stage = d.flatten(1).max(dim=1).values.tolist()
            print(f"frame {t} {name}: max|packed - generic| per stage {['%.2e' % v for v in per_stage]}")
        ga_, gb_ = ra["gate"][t], rb["gate"][t]
        print(f"frame {t}: gates at the ReLU floor (0.5): packed {int((ga_ == 0.5).sum())} generic {int((gb_ == 0.5).sum())} differ {int(((ga_ == 0.5) != (gb_ == 0.5)).sum())}"
              f"  frame loss {ra['frame_loss'][t]:.6f} {rb['frame_loss'][t]:.6f}")
    rows = []
    for k in ga:
        rms = float(gb[k].norm()) / max(1.0, gb[k].numel() ** 0.5) + 1e-12
        rows.append((float((ga[k] - gb[k]).abs().max()) / rms, float((gb[k] - gc[k]).abs().max()) / rms, k))
    rows.sort(reverse=True)
    print("max |packed - generic| / RMS entry, generic run-to-run / RMS entry:")
    for r in rows[:25]:
        print("  %.4e  %.4e  %s" % r)
    k = "backbone.backbone.model.layer2.0.bn1.weight"
    print(k, ga[k][:4].tolist(), gb[k][:4].tolist())


if __name__ == "__main__":
    main()
