"""CPU-side checks of the product: the C-ABI library exports what include/phnet_hip.h declares, the module tree has
the reference's state_dict layout, host-side decode logic matches the oracle, and nothing silently falls back to CPU."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import phnet_cpu as O
from phnet_amd import _lib
from phnet_amd.config import make_cfg
from tests import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def built():
    from phnet_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    decl = _lib.declared_functions()
    assert len(decl) >= 25
    handle = ctypes.CDLL(built)
    for name, _, _ in decl:
        assert hasattr(handle, name), name
    assert handle.phnet_abi_version() == 1
    exported = set(os.popen(f"nm -D --defined-only {built}").read().split())
    assert {n for n, _, _ in decl} <= exported
    assert not [s for s in exported if s.startswith("phnet_") and s not in {n for n, _, _ in decl}], "undeclared export"


def test_argument_validation_needs_no_gpu(built):
    lib = _lib.lib()
    assert lib.phnet_lane_nms(None, None, None, 1, 10, 36, 50.0, 4, None, None, None, None) == -1     # null outputs
    assert lib.phnet_lane_nms(None, None, None, 0, 10, 36, 50.0, 4, None, None, None, None) == 0      # zero frames: no-op
    assert lib.phnet_conv2d_fwd(None, None, None, None, 1, 8, 8, 3, 64, 3, 3, 1, 1, 0, None, 0, None) == -1   # Ci % 4
    assert lib.phnet_roi_pool_fwd(None, None, None, None, None, 1, 240, 36, 10, 25, 32, None) == -1   # C != 64
    bm, bn, sp, kt = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.phnet_conv2d_plan(80000, 64, 576, 0, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp), ctypes.byref(kt)) == 0
    assert (bm.value, bn.value, sp.value, kt.value) == (64, 64, 1, 16)
    assert lib.phnet_conv2d_plan(240, 64, 128, 0, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(sp), ctypes.byref(kt)) == 0
    assert kt.value == 64


@pytest.mark.parametrize("arch", ["resnet18", "resnet34"])
def test_module_tree_has_the_reference_state_dict_layout(arch):
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.utils.loss4OLV3 import Criterion4OL
    cfg = make_cfg(arch=arch)
    model = RouterOL(cfg, Criterion4OL(cfg))
    keys = json.load(open(os.path.join(GOLD, "state_keys.json")))[arch]          # frozen from the reference's state_dict()
    sd = model.state_dict()
    assert list(sd) == list(keys)
    assert all(list(sd[k].shape) == keys[k] for k in keys)
    g = O.Geometry(arch=arch)
    model.load_state_dict(synth.make_state(g), strict=True)                     # reference-shaped checkpoint loads strictly
    assert model.backbone.backbone.model.layer1[0].conv1.weight.is_contiguous(memory_format=torch.channels_last)
    # anchors and buffers equal the oracle's statement of Router4OL.py:169-211
    fresh = RouterOL(cfg, None)
    assert torch.equal(fresh.detNet.prior_embeddings.weight.detach(), O.initial_anchor_embeddings(g))
    pri, on_map = O.priors_from_embeddings(O.initial_anchor_embeddings(g), g)
    assert torch.allclose(fresh.detNet.priors, pri, atol=1e-6) and torch.allclose(fresh.detNet.priors_on_featmap, on_map, atol=1e-6)


def test_no_cpu_fallback():
    from phnet_amd import hip_ops
    from phnet_amd.libs.models.Router4OL import RouterOL
    from phnet_amd.libs.ops import nms
    with pytest.raises(RuntimeError):
        hip_ops.conv2d_fwd(torch.zeros(1, 4, 4, 4), torch.zeros(4, 1, 1, 4), None, 1, 0)
    with pytest.raises(RuntimeError):
        nms(torch.zeros(3, 41), torch.zeros(3), overlap=50.0, top_k=4)
    cfg = make_cfg(img_h=64, img_w=160, arch="resnet18")
    with pytest.raises(RuntimeError):
        RouterOL(cfg, None)({"frame": torch.zeros(1, 3, 64, 160), "lanes": torch.zeros(1, 4, 42)})
    import phnet_amd
    src = open(os.path.join(os.path.dirname(phnet_amd.__file__), "hip_ops.py")).read()
    assert "oracle" not in src


def test_host_side_lane_decode_matches_oracle():
    from phnet_amd.libs.models.Router4OL import RouterOL
    g = O.Geometry(img_h=64, img_w=160, arch="resnet18")
    det = RouterOL(make_cfg(img_h=64, img_w=160, arch="resnet18"), None).detNet
    r = np.random.default_rng(0)
    rows = torch.zeros(6, 42)
    rows[:, 2] = torch.tensor([0.0, 0.1, 0.3, 0.0, 0.5, 0.97])
    rows[:, 5] = torch.tensor([30.0, 20.0, 10.0, 1.0, 40.0, 3.0])
    rows[:, 6:] = torch.from_numpy(r.uniform(-0.2, 1.2, (6, 36)).astype(np.float32))
    lanes = det.predictions_to_pred(rows.clone())
    ref = [O.lane_points(row.clone(), g) for row in rows]
    ref = [p for p in ref if p is not None]
    assert len(lanes) == len(ref)
    for a, b in zip(lanes, ref):
        np.testing.assert_array_equal(a.points, b)
    assert det.prior_ys.dtype == torch.float32                               # no float64 mutation (Router4OL.py:398-399)


def test_compat_install_registers_reference_import_paths():
    import sys
    import phnet_amd
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "libs" or k.startswith("libs.")}
    try:
        done = phnet_amd.install()
        assert "libs.models.Router4OL" in done and "libs.ops" in done
        from libs.models.Router4OL import RouterOL          # noqa: F401
        from libs.ops import nms                             # noqa: F401
        from libs.utils.loss4OLV3 import Criterion4OL        # noqa: F401
        assert RouterOL.__module__.startswith("phnet_amd")
    finally:
        for k in [k for k in sys.modules if k == "libs" or k.startswith("libs.")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})


def test_grad_sink_collects_per_use_gradients_once():
    """arena.grad_sink: uses accumulate into the side buffer and return no gradient; autograd gets the sum exactly once."""
    from phnet_amd.arena import direct_grad, grad_sink

    class Use(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            ctx.sink = direct_grad(w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            ctx.sink.add_(g.t() @ x)
            return g @ w, None

    torch.manual_seed(0)
    a, b = torch.randn(4, 3, requires_grad=True), torch.randn(4, 3, requires_grad=True)
    x = torch.randn(5, 3)
    w = grad_sink(a * b)
    assert direct_grad(w) is not None and torch.equal(w, a * b)
    sum(Use.apply(x * (i + 1), w).sum() for i in range(3)).backward()
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    sum(((x * (i + 1)) @ (a2 * b2).t()).sum() for i in range(3)).backward()
    assert torch.allclose(a.grad, a2.grad) and torch.allclose(b.grad, b2.grad)
    with torch.no_grad():
        assert grad_sink(a * b)._phnet_sink is None if hasattr(grad_sink(a * b), "_phnet_sink") else True


def test_lines_txt_wire_format(tmp_path):
    """evaluation/generate_lane.py:46-61 of the reference: '%.1f %.1f ' pairs, points last-to-first, lanes with <= 2 points
    dropped, x = tx*W/2, y = (ty*H + 480)/2.  Expected text written out by hand from that formula."""
    import numpy as np
    from phnet_amd.evaluation.generate_lane import format_pred_lines, generate_predV2
    from phnet_amd.libs.utils.lane import Lane
    size = (800, 1920)                                         # (H after the 480-row crop, W) of an OpenLane image
    a = Lane(points=np.array([[0.75, 0.0], [0.5, 0.5], [0.25, 1.0]]))
    short = Lane(points=np.array([[0.2, 0.8], [0.1, 0.9]]))     # two points: not written
    b = Lane(points=np.array([[0.5, 0.8], [0.3333, 0.85], [0.10004, 0.9], [0.1, 0.95]]))
    text = format_pred_lines([a, short, b], size)
    assert text == ("240.0 640.0 480.0 440.0 720.0 240.0 \n"
                    "96.0 620.0 96.0 600.0 320.0 580.0 480.0 560.0 \n")
    info = {"name": "clip7", "ImgName": ["f0", "f1"], "size": size}
    path = generate_predV2(info, [a], 1, str(tmp_path))
    assert path.endswith("clip7/f1.lines.txt") and open(path).read() == "240.0 640.0 480.0 440.0 720.0 240.0 \n"
