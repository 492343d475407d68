"""Parity of the HIP-backed Router4OLV2 family (phnet_amd.libs.models.Router4OLV2: what testOLV3.py imports) on a real MI355X:
 (a) its own kernels against fp64 statements (per-level ROI pooling, gate, run-time-shape per-anchor products, 8 x 32 attention,
     hard routing),
 (b) every (frame, stage) teacher-forced with the CPU oracle's stage inputs, activations within 1e-3 * (1 + |ref|),
 (c) end to end against the fixtures produced by the reference's own Python (tests/golden/v2_*.npz): keep masks / kept
     indices exact, lane points 1e-3.
Inference only - the reference's training path of this family cannot run as shipped (tests/golden/make_goldens_v2.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import lane_nms as ONMS
from oracle import phnet_cpu as O
from oracle import phnet_cpu_v2 as O2
from tests import synth
from tests.test_model_gpu import ACT_TOL, _close, _close_lines

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    return dict(np.load(os.path.join(GOLD, name)))


def _build(g: O2.GeometryV2):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from phnet_amd.config import make_cfg_v2
    from phnet_amd.libs.models.Router4OLV2 import RouterOL
    model = RouterOL(make_cfg_v2(img_h=g.img_h, img_w=g.img_w, arch=g.arch))
    model.load_state_dict(synth.make_state_v2(g), strict=True)
    return model.cuda().eval()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("c,p,h,w", [(64, 24, 4, 10), (32, 48, 8, 20), (16, 96, 16, 40), (16, 96, 80, 200)])
def test_roi_pool_per_level_widths_vs_grid_sample(c, p, h, w):
    from phnet_amd import hip_ops as K
    r = np.random.default_rng(c * 1000 + p)
    fmap = torch.from_numpy(r.standard_normal((2, c, h, w)).astype(np.float32))
    xs = torch.from_numpy(r.uniform(-0.3, 1.3, (2, 7, p)).astype(np.float32))          # anchors leave the map on both sides
    ys = torch.flip(1 - (torch.linspace(0, 1, p) * 71).long().float() / 71, dims=[-1])
    grid = torch.stack([torch.flip(xs, dims=[2]) * 2 - 1, ys.view(1, 1, -1).expand_as(xs) * 2 - 1], dim=-1)
    ref = F.grid_sample(fmap.double(), grid.double(), mode="bilinear", padding_mode="zeros", align_corners=True).permute(0, 2, 3, 1)
    out, out_cp = K.roi_pool_fwd(_nhwc(fmap), xs.cuda(), ys.cuda(), with_cp=True)
    _close(out, ref, 2e-5, "roi")
    _close(out_cp, ref.permute(0, 1, 3, 2), 2e-5, "roi_cp")


@pytest.mark.parametrize("stage", [0, 1, 2])
def test_gate_v2_vs_oracle_statement(stage):
    model = _build(O2.GeometryV2(img_h=64, img_w=160))
    g = O2.GeometryV2(img_h=64, img_w=160)
    sd = synth.make_state_v2(g)
    c, p = g.feat_channels[stage], g.sample_points[stage]
    x = torch.from_numpy(np.random.default_rng(stage).standard_normal((2, 240, c, p)).astype(np.float32))
    ref = O2.routing_gate_v2({k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, stage, x.double(), g)
    out = model.router.router(x.cuda(), stage)
    _close(out, ref, 2e-5, f"gate stage {stage}")
    assert float(out.min()) < 0.5 < float(out.max())


@pytest.mark.parametrize("p,k,j", [(24, 64, 128), (24, 128, 64), (48, 32, 64), (48, 64, 32), (96, 16, 32), (96, 32, 16)])
def test_dyn_product_layernorm_relu_any_shape(p, k, j):
    from phnet_amd import hip_ops as K
    r = np.random.default_rng(p + k + j)
    x = torch.from_numpy(r.standard_normal((9, p, k)).astype(np.float32))
    w = torch.from_numpy((r.standard_normal((9, k, j)) / np.sqrt(k)).astype(np.float32))
    gam = torch.from_numpy(r.uniform(0.5, 1.5, j).astype(np.float32))
    bet = torch.from_numpy(r.normal(0, 0.1, j).astype(np.float32))
    ref = F.relu(F.layer_norm(torch.bmm(x.double(), w.double()), [j], gam.double(), bet.double(), 1e-5))
    out = K.dyn_bmm_ln_relu_fwd_any(x.cuda(), w.cuda(), gam.cuda(), bet.cuda(), 1e-5)
    _close(out, ref, 2e-5, "dyn")


@pytest.mark.parametrize("lq,lk,masked", [(240, 240, False), (240, 10, True), (240, 1, False), (17, 25, True)])
def test_attention_head_width_32(lq, lk, masked):
    from phnet_amd import hip_ops as K
    r = np.random.default_rng(lq + lk)
    H, D = 8, 32
    q = torch.from_numpy(r.standard_normal((lq, H * D)).astype(np.float32))
    kv = torch.from_numpy(r.standard_normal((lk, 2 * H * D)).astype(np.float32))
    valid = torch.ones(lk, dtype=torch.bool)
    if masked:
        valid[1::3] = False
    qd = q.double().view(lq, H, D).transpose(0, 1) / np.sqrt(D)
    kd = kv[:, :H * D].double().view(lk, H, D).transpose(0, 1)
    vd = kv[:, H * D:].double().view(lk, H, D).transpose(0, 1)
    s = torch.bmm(qd, kd.transpose(1, 2)).masked_fill(~valid.view(1, 1, -1), float("-inf"))
    ref = torch.bmm(torch.softmax(s, dim=-1), vd).transpose(0, 1).reshape(lq, H * D)
    kvc = kv.cuda()
    out, lse = K.attention_fwd(q.cuda(), kvc[:, :H * D], kvc[:, H * D:], H, valid.cuda().view(torch.uint8) if masked else None)
    _close(out, ref, 2e-5, "attention 8x32")
    _close(lse.view(H, lq), torch.logsumexp(s, dim=-1), 2e-5, "lse")


def test_route_lines_hard_and_soft():
    from phnet_amd import hip_ops as K
    r = np.random.default_rng(5)
    gates = torch.from_numpy(r.uniform(0.2, 0.8, (3, 240)).astype(np.float32))
    gates[:, 7] = 0.5                                                                   # mean exactly 0.5 -> branch B (>=)
    a = torch.from_numpy(r.standard_normal((240, 78)).astype(np.float32))
    b = torch.from_numpy(r.standard_normal((240, 78)).astype(np.float32))
    d = torch.stack(list(gates), dim=0).mean(dim=0).unsqueeze(1)
    hard = K.route_lines(gates.cuda(), a.cuda(), b.cuda(), True).cpu()
    assert torch.equal(hard, torch.where(d >= 0.5, b, a))
    assert torch.equal(hard[7], b[7])
    soft = K.route_lines(gates.cuda(), a.cuda(), b.cuda(), False).cpu()
    _close(soft, b.double() * d.double() + a.double() * (1 - d.double()), 2e-6, "soft")


# ------------------------------------------------------------------------------------------------ model
def test_v2_module_tree_has_the_reference_state_dict_layout():
    import json
    model = _build(O2.GeometryV2(img_h=64, img_w=160))
    keys = json.load(open(os.path.join(GOLD, "state_keys_v2.json")))
    sd = model.state_dict()
    assert list(sd) == list(keys)
    assert all(list(sd[k].shape) == keys[k] for k in keys)


def test_v2_training_mode_is_refused_like_the_reference_fails():
    model = _build(O2.GeometryV2(img_h=64, img_w=160))
    model.train()
    with pytest.raises(NotImplementedError):
        model({"frame": torch.zeros(2, 3, 64, 160).cuda(), "lanes": torch.zeros(2, 4, 78).cuda()})


@pytest.mark.parametrize("size", ["tiny", "320x800"])
def test_v2_encoder_maps_vs_oracle(size):
    """Trunk (without its last stage) + per-level-width FPN against the oracle at the strict bound, on running statistics that
    match the weights (tests/synth.py calibrate_running_stats_: eval activations O(1-10) as with a trained checkpoint)."""
    g = O2.GeometryV2(img_h=64, img_w=160) if size == "tiny" else O2.GeometryV2()
    model = _build(g)
    frames = synth.make_clip(g, 2, seed=77)
    sd = synth.calibrate_running_stats_(synth.make_state_v2(g), frames, g.arch, g.bn_eps)
    model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        ref = O2.encoder_v2(sd, frames, g)
        got = model.backbone(frames.cuda())
    for j in range(3):
        assert got[j].shape[-1] == g.neck_out[j]
        _close(got[j].permute(0, 3, 1, 2), ref[j], ACT_TOL, f"fpn level {j}")


@pytest.mark.parametrize("size,T", [("tiny", 7), ("320x800", 3)])
def test_v2_every_stage_teacher_forced_vs_oracle(size, T):
    """Every (frame, stage) of the head fed with the ORACLE's stage inputs (priors, sample positions, proposal features,
    memory) and the oracle's pyramid maps: gate, dynamic-head output, both branches within ACT_TOL - strict, per stage."""
    from phnet_amd import hip_ops as K
    g = O2.GeometryV2(img_h=64, img_w=160) if size == "tiny" else O2.GeometryV2()
    model = _build(g)
    det = model.router
    sd = synth.make_state_v2(g)
    col = {}
    with torch.no_grad():
        O2.clip_forward_eval_v2(sd, synth.make_clip(g, T, seed=77), g, ONMS.lane_nms, collect=col)
        levels = [_nhwc(f) for f in col["fpn"]][::-1]
        for t in range(T):
            fo = col["frames"][t]
            for s in range(g.refine_layers):
                si = fo.stage_inputs[s]
                fr = det.stage_front(levels[s][t:t + 1], s, si["priors"].cuda(), si["on_map"].cuda(), si["pro"].cuda())
                _close(fr["gate"], fo.gates[s], ACT_TOL, f"gate t={t} s={s}")
                _close(fr["local"], fo.locals_[s], ACT_TOL, f"local t={t} s={s}")
                _close_lines(fr["pred_a"], fo.predictions_fir[s], f"branch A t={t} s={s}")
                attn = (fr["local"][0] + det.PositionEmbedding.pos_table).unsqueeze(1)
                _close(attn[:, 0], fo.attn_feats[s], ACT_TOL, f"tokens t={t} s={s}")
                mem = si["mem"]
                pred_b, _ = det.forward_second(None if mem is None else mem.cuda().unsqueeze(1), attn, s, si["priors"].cuda())
                _close_lines(pred_b, fo.predictions_sec[s], f"branch B t={t} s={s}")
                assert (mem is None) == (t < g.save_freq)                       # frame 0: the self-attention fallback


def _end_to_end(gold, g, T):
    model = _build(g)
    frames = synth.make_clip(g, T, seed=77).cuda()
    with torch.no_grad():
        rows, nums, anchors, aux = model.infer_device(frames)
        res = model.lanes_from_device(rows, nums)
    torch.cuda.synchronize()
    gates = aux["gates"].cpu().numpy()                                           # [T,3,N]
    for t in range(T):
        d_ref = gold["gate"][t].mean(axis=0)
        clear = np.abs(d_ref - 0.5) > 1e-4                                        # anchors whose hard routing is not a coin flip
        assert clear.mean() > 0.95
        lines = aux["frames"][t]["lines"].cpu()
        _close_lines(lines[torch.from_numpy(clear)], gold["lines"][t][clear], f"lines t={t}", cascade=True)
        assert np.abs(gates[t] - gold["gate"][t]).max() <= 5e-2
        assert (aux["frames"][t]["keep_mask"].cpu().numpy().astype(bool) == gold["keep_inds"][t]).all(), t
        n = int(nums[t])
        assert aux["frames"][t]["keep_c"].cpu().numpy()[:n].tolist() == [i for i in gold["keep"][t].tolist() if i >= 0], t
        lanes_t = res["lane_lines"][t]
        assert len(lanes_t) == int((gold["lane_npts"][t] > 0).sum())
        for j, lane in enumerate(lanes_t):
            k = int(gold["lane_npts"][t, j])
            assert lane.points.shape == (k, 2)
            np.testing.assert_allclose(lane.points, gold["lane_pts"][t, j, :k], atol=1e-3)
    return model, frames, res


def test_v2_tiny_eight_frames_vs_reference_fixture():
    """8 frames (three past the memory depth): frame 0 through the self-attention fallback, memory = one mean token per stored
    frame (the saveMemory4Test quirk), hard routing taking both branches; keep masks and kept indices exact."""
    _end_to_end(_gold("v2_tiny_r18_64x160.npz"), O2.GeometryV2(img_h=64, img_w=160), 8)


def test_v2_320x800_vs_reference_fixture_and_module_api():
    g = O2.GeometryV2()
    model, frames, res = _end_to_end(_gold("v2_r18_320x800.npz"), g, 6)
    out = model({"frame": frames, "lanes": torch.zeros(6, 4, 78).cuda()})         # the reference's call (testOLV3.py)
    assert [len(x) for x in out["lane_lines"]] == [len(x) for x in res["lane_lines"]]
    # frame-by-frame stage 0 instead of the batched one: same result
    model.batch_stage0 = False
    with torch.no_grad():
        rows2, nums2, _, _ = model.infer_device(frames)
        model.batch_stage0 = True
        rows1, nums1, _, _ = model.infer_device(frames)
    assert torch.equal(nums1, nums2)
    _close(rows2, rows1, 2e-3, "stage-0 batching")        # other GEMM row counts -> other tile plans: fp32 re-association through the cascade


def test_v2_intended_memory_mode_differs_only_after_frame_zero():
    """faithful_memory = False stores the kept lanes' tokens as well (what Router4OLV2.py:570-578 evidently meant): frame 0 is
    unaffected (no memory yet), later frames may differ."""
    g = O2.GeometryV2(img_h=64, img_w=160)
    model = _build(g)
    frames = synth.make_clip(g, 4, seed=77).cuda()
    with torch.no_grad():
        r1, n1, _, _ = model.infer_device(frames)
        model.faithful_memory = False
        r2, n2, _, _ = model.infer_device(frames)
    assert torch.equal(r1[0], r2[0]) and int(n1[0]) == int(n2[0])
    assert bool(torch.isfinite(r2).all())
