"""Golden vectors for the criterion variants: the reference's own `libs.utils.loss4OL.Criterion4OL`
(trainOLV2.py / trainOLV3.py) and `libs.utils.loss4OLV2.Criterion4OL` (one-to-many assignment, dynamic_assign.py:292-357) run on
CPU on the head outputs frozen in tests/golden/tiny_ragged_r18_64x160.npz (frames with 0 / 4 / 1 / 2 valid lanes) and
tiny_r18_64x160.npz (3 lanes).  Build container only:  python tests/golden/make_goldens_criteria.py
Stubs: those of make_goldens.py (nothing on this path touches them)."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_goldens as MG
from oracle import phnet_cpu as O
from tests import synth


def cases(g):
    a = dict(np.load(os.path.join(HERE, "tiny_ragged_r18_64x160.npz")))
    b = dict(np.load(os.path.join(HERE, "tiny_r18_64x160.npz")))
    la = synth.make_targets(g, 4, counts=MG.RAGGED_COUNTS)
    lb = synth.make_targets(g, 3)
    out = []
    for src, lanes, T in ((a, la, 4), (b, lb, 3)):
        for t in range(T):
            out.append((src["train_fir"][t], src["train_sec"][t], src["train_gate"][t], lanes[t:t + 1]))
    return out


def run(crit, fir, sec, gate, gt):
    f = [torch.from_numpy(fir[s]).unsqueeze(0).requires_grad_() for s in range(3)]
    s_ = [torch.from_numpy(sec[s]).unsqueeze(0).requires_grad_() for s in range(3)]
    d = [torch.from_numpy(gate[s]).view(1, -1, 1).requires_grad_() for s in range(3)]
    res = crit({"predictions_fir": f, "predictions_sec": s_}, gt.clone(), d)
    matched, loss = res[0], res[1]
    loss.backward()
    mm = np.full((3, 16), -1, dtype=np.int64)
    for i, m in enumerate(matched):
        m = np.asarray(m, dtype=np.int64)
        mm[i, :len(m)] = m
    return (float(loss), mm, torch.stack([x.grad[0] for x in f]).numpy(), torch.stack([x.grad[0] for x in s_]).numpy(),
            torch.stack([x.grad[0, :, 0] for x in d]).numpy())


def main():
    MG.install_shims()
    torch.set_num_threads(4)
    from libs.utils.loss4OL import Criterion4OL as V1
    from libs.utils.loss4OLV2 import Criterion4OL as V2
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
    cfg = MG.ref_cfg(g)
    out = {}
    for tag, cls in (("v1", V1), ("v2", V2)):
        crit = cls(cfg=cfg)
        rec = [run(crit, *c) for c in cases(g)]
        out[f"{tag}_loss"] = np.array([r[0] for r in rec])
        out[f"{tag}_matched"] = np.stack([r[1] for r in rec])
        out[f"{tag}_dfir"] = np.stack([r[2] for r in rec]).astype(np.float32)
        out[f"{tag}_dsec"] = np.stack([r[3] for r in rec]).astype(np.float32)
        out[f"{tag}_dgate"] = np.stack([r[4] for r in rec]).astype(np.float32)
        print(tag, out[f"{tag}_loss"].tolist(), out[f"{tag}_matched"][:, 2].tolist())
    np.savez_compressed(os.path.join(HERE, "criterion_variants_tiny.npz"), **out)


if __name__ == "__main__":
    main()
