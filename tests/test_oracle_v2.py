"""Pins the Router4OLV2 CPU oracle (oracle/phnet_cpu_v2.py) to fixtures produced by the reference's own Python
(tests/golden/make_goldens_v2.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import lane_nms as ONMS
from oracle import phnet_cpu_v2 as O2
from tests import synth
from tests.test_oracle_golden import _lines_close

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def _check(gold, g, T, fpn=False):
    sd = synth.make_state_v2(g)
    col = {}
    with torch.no_grad():
        dec = O2.clip_forward_eval_v2(sd, synth.make_clip(g, T, seed=77), g, ONMS.lane_nms, collect=col)
    if fpn:
        for j in range(3):
            np.testing.assert_allclose(col["fpn"][j].numpy(), gold[f"fpn{j}"], atol=2e-5, rtol=1e-5)
    for t, d in enumerate(dec):
        fo = col["frames"][t]
        gate = torch.stack([x[0, :, 0] for x in fo.gates]).numpy()
        gerr = np.abs(gate - gold["gate"][t])
        assert (gerr <= 1e-4).mean() >= 0.99 and gerr.max() <= 5e-3, (t, float(gerr.max()))
        _lines_close(torch.stack([x[0] for x in fo.predictions_fir]).numpy(), gold["fir"][t], f"fir t={t}")
        _lines_close(torch.stack([x[0] for x in fo.predictions_sec]).numpy(), gold["sec"][t], f"sec t={t}")
        np.testing.assert_allclose(torch.stack([a.mean(dim=0) for a in fo.attn_feats]).numpy(), gold["attn_mean"][t], atol=2e-4)
        mem = fo.stage_inputs[0]["mem"]
        assert (0 if mem is None else mem.shape[0]) == int(gold["mem_rows"][t])        # FIFO depth seen by frame t
        _lines_close(d["lines"].numpy(), gold["lines"][t], f"lines t={t}")
        assert (d["keep_inds"].numpy() == gold["keep_inds"][t]).all()
        assert d["keep"].tolist() == [i for i in gold["keep"][t].tolist() if i >= 0]
        assert len(d["lanes"]) == int((gold["lane_npts"][t] > 0).sum())
        for j, (pts, sx, sy, conf) in enumerate(d["lanes"]):
            n = int(gold["lane_npts"][t, j])
            assert pts.shape == (n, 2)
            np.testing.assert_allclose(pts, gold["lane_pts"][t, j, :n], atol=1e-3)
            np.testing.assert_allclose([sx, sy, conf], gold["lane_meta"][t, j], atol=1e-3)


def test_v2_state_spec_matches_reference_state_dict():
    keys = json.load(open(os.path.join(GOLD, "state_keys_v2.json")))
    spec = synth.state_spec_v2(O2.GeometryV2(img_h=64, img_w=160))
    assert list(spec) == list(keys)
    assert all(list(spec[k]) == keys[k] for k in spec)


def test_v2_tiny_eight_frame_eval_matches_reference():
    """8 frames, three past the memory depth (save_freq_max = 5); frame 0 runs the self-attention fallback (no memory),
    every later frame attends to one mean token per stored frame (the saveMemory4Test quirk, Router4OLV2.py:570-578)."""
    gold = _load("v2_tiny_r18_64x160.npz")
    assert gold["mem_rows"].tolist() == [0, 1, 2, 3, 4, 5, 5, 5]
    hard = (gold["gate"].mean(axis=1) >= 0.5)
    assert hard.any() and (~hard).any()                         # the hard routing takes both branches in the fixture
    _check(gold, O2.GeometryV2(img_h=64, img_w=160), 8, fpn=True)


def test_v2_320x800_eval_matches_reference():
    _check(_load("v2_r18_320x800.npz"), O2.GeometryV2(), 6)
