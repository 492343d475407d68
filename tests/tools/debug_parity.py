"""Debug helper (GPU box): per-stage error breakdown of the HIP model against the CPU oracle, same weights/inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import synth
from tests.test_model_gpu import _build, _record_heads
from oracle import phnet_cpu as O

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()), float(b.abs().max())

def main(tiny=True, training=True):
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18") if tiny else O.Geometry(arch="resnet34")
    T = 3 if tiny else 5
    model = _build(g)
    model.train(training)
    rec, undo = _record_heads(model)
    frames, lanes = synth.make_clip(g, T), synth.make_targets(g, T)
    sd = synth.make_state(g)
    col = {}
    with torch.no_grad():
        ref_feats = O.fpn_neck(sd, O.resnet_trunk(sd, frames, g, training))
        feats = model.backbone(frames.cuda())
    for j in range(3):
        print("fpn", j, rel(feats[j].permute(0, 3, 1, 2), ref_feats[j]))
    sd = synth.make_state(g)
    ctx = torch.enable_grad() if training else torch.no_grad()
    with ctx:
        if training:
            O.clip_forward(sd, frames, lanes, g, True, collect=col)
            model({"frame": frames.cuda(), "lanes": lanes.cuda()})
        else:
            from oracle import lane_nms as ON
            O.clip_forward(sd, frames, lanes, g, False, nms_fn=ON.lane_nms, collect=col)
            model({"frame": frames.cuda(), "lanes": lanes.cuda()})
    for t in range(T):
        fo = col["frames"][t]
        for s in range(3):
            a, b = rec["fir"][t][s], fo.predictions_fir[s][0]
            a2, b2 = rec["sec"][t][s], fo.predictions_sec[s][0]
            print(f"t{t} s{s} gate", rel(rec["gate"][t][s], fo.gates[s][0, :, 0]),
                  "fir[:6]", rel(a[:, :6], b[:, :6]), "fir xs", rel(a[:, 6:], b[:, 6:]),
                  "sec[:6]", rel(a2[:, :6], b2[:, :6]), "sec xs", rel(a2[:, 6:], b2[:, 6:]))
            e = (a[:, :6].double() - b[:, :6].detach().double()).abs()
            i = int(e.argmax()); print("     worst fir elem", i // 6, i % 6, float(a.flatten()[0]), "vals", a[i // 6, :6].tolist(), b[i // 6, :6].tolist())

if __name__ == "__main__":
    main(tiny="big" not in sys.argv, training="eval" not in sys.argv)
