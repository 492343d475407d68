"""Where does the teacher-forced stage check stand at the headline geometry?  Prints the worst elements per output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import phnet_cpu as O, lane_nms as ON
from tests import synth
from tests.test_model_gpu import _build

training = "--eval" not in sys.argv
g = O.Geometry(arch="resnet34")
T = 5
model = _build(g); model.train(training)
sd = synth.make_state(g)
frames, lanes = synth.make_clip(g, T), synth.make_targets(g, T)
col = {}
with torch.no_grad():
    O.clip_forward(sd, frames, lanes if training else None, g, training, nms_fn=ON.lane_nms, collect=col)
    feats = model.backbone(frames.cuda())
    for j in range(3):
        e = (feats[j].permute(0, 3, 1, 2).cpu() - col["fpn"][j]).abs()
        print(f"fpn{j} max err {float(e.max()):.3e} ref max {float(col['fpn'][j].abs().max()):.3f}")
    det = model.detNet
    for t in range(T):
        fo = col["frames"][t]
        levels = [f[t:t + 1] for f in feats][::-1]
        for s in range(3):
            si = fo.stage_inputs[s]
            mem = torch.cat(si["mem"], 0).unsqueeze(1).cuda() if si["mem"] else None
            r = det.stage_forward(levels[s], s, si["priors"].cuda(), si["on_map"].cuda().contiguous(), si["pro"].cuda(), mem)
            for name, got, ref in (("gate", r["gate"], fo.gates[s]), ("local", r["local"], fo.locals_[s]), ("attn", r["attn"][:, 0], fo.attn_feats[s]),
                                   ("A", r["pred_a"], fo.predictions_fir[s]), ("B", r["pred_b"], fo.predictions_sec[s])):
                a, b = got.detach().cpu().double().reshape(ref.shape), ref.double()
                if name in ("A", "B"):
                    head = (a[..., :6] - b[..., :6]).abs() / (1 + b[..., :6].abs())
                    xs = (a[..., 6:] - b[..., 6:]).abs() / (1 + b[..., 6:].abs().amax(-1, keepdim=True))
                    err = torch.cat([head, xs], -1)
                else:
                    err = (a - b).abs() / (1 + b.abs())
                m = float(err.max())
                if m > 3e-4:
                    idx = torch.nonzero(err == err.max())[0].tolist()
                    row = idx[-2] if len(idx) >= 2 else 0
                    extra = ""
                    if name in ("A", "B"):
                        extra = f" theta {float(b.reshape(-1, b.shape[-1])[row, 4]):.5f} sy {float(b.reshape(-1, b.shape[-1])[row, 2]):.4f} xs range {float(b.reshape(-1, b.shape[-1])[row, 6:].min()):.2f}..{float(b.reshape(-1, b.shape[-1])[row, 6:].max()):.2f} n_bad {int((err > 1e-3).sum())}"
                    print(f"t{t} s{s} {name}: max rel err {m:.3e} at {idx} got {float(a[tuple(idx)]):.6f} ref {float(b[tuple(idx)]):.6f}{extra}")
print("done")
