"""Pins the criterion-variant oracle (oracle/criterion_variants_cpu.py) to fixtures produced by the reference's own
`libs.utils.loss4OL` / `libs.utils.loss4OLV2` criteria (tests/golden/make_goldens_criteria.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import criterion_variants_cpu as OC
from oracle import phnet_cpu as O
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RAGGED = (0, 4, 1, 2)


def cases(g):
    a = dict(np.load(os.path.join(GOLD, "tiny_ragged_r18_64x160.npz")))
    b = dict(np.load(os.path.join(GOLD, "tiny_r18_64x160.npz")))
    la, lb = synth.make_targets(g, 4, counts=RAGGED), synth.make_targets(g, 3)
    out = []
    for src, lanes, T in ((a, la, 4), (b, lb, 3)):
        for t in range(T):
            out.append((src["train_fir"][t], src["train_sec"][t], src["train_gate"][t], lanes[t:t + 1]))
    return out


@pytest.mark.parametrize("tag", ["v1", "v2"])
def test_criterion_variant_oracle_matches_reference(tag):
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
    gold = dict(np.load(os.path.join(GOLD, "criterion_variants_tiny.npz")))
    fn = OC.frame_loss_v1 if tag == "v1" else OC.frame_loss_v2
    for i, (fir, sec, gate, gt) in enumerate(cases(g)):
        f = [torch.from_numpy(fir[s]).unsqueeze(0).requires_grad_() for s in range(3)]
        s_ = [torch.from_numpy(sec[s]).unsqueeze(0).requires_grad_() for s in range(3)]
        d = [torch.from_numpy(gate[s]).view(1, -1, 1).requires_grad_() for s in range(3)]
        res = fn(f, s_, d, gt, g)
        matched, loss = res[0], res[1]
        loss.backward()
        assert abs(float(loss) - gold[f"{tag}_loss"][i]) <= 1e-5 * abs(gold[f"{tag}_loss"][i]), (i, float(loss))
        for st in range(3):
            assert matched[st].tolist() == [r for r in gold[f"{tag}_matched"][i, st].tolist() if r >= 0], (i, st)
        np.testing.assert_allclose(torch.stack([x.grad[0] for x in f]).numpy(), gold[f"{tag}_dfir"][i], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(torch.stack([x.grad[0] for x in s_]).numpy(), gold[f"{tag}_dsec"][i], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(torch.stack([x.grad[0, :, 0] for x in d]).numpy(), gold[f"{tag}_dgate"][i], atol=2e-5, rtol=1e-4)
        if tag == "v2":
            assert res[2].shape == (1, len(matched[-1]), 42)
    assert (gold["v2_matched"] >= 0).sum() > (gold["v1_matched"] >= 0).sum()          # the one-to-many rounds add anchors
