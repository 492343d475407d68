"""GPU parity of the evaluator's pixel work (csrc/lane_iou.hip through the C-ABI) against oracle/culane_cpu.py: masks bit for bit,
areas / intersections exact, and the whole evaluation (phnet_amd.evaluation.culane.evaluate) against the oracle's on the same files."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import culane_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from phnet_amd import hip_ops
    return hip_ops


def _unpack(masks, width):
    m = masks.cpu().numpy().view(np.uint32)
    bits = ((m[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    return bits.reshape(m.shape[0], m.shape[1], -1)[:, :, :width]


@pytest.mark.parametrize("h,w,lw", [(40, 50, 1), (64, 96, 2), (200, 300, 30), (200, 333, 31), (97, 65, 7)])
def test_lane_masks_and_counts_bit_exact(ops, h, w, lw):
    rng = np.random.default_rng(h * w + lw)
    lanes = []
    for _ in range(5):
        segs = [tuple(int(v) for v in rng.integers(-40, max(h, w) + 40, 4)) for _ in range(6)]
        segs += [(7, 9, 7, 9), (-300, -300, -250, -280), (0, 0, w - 1, h - 1)]          # a point, fully outside, the diagonal
        lanes.append(segs)
    rows = np.array([s + (l,) for l, segs in enumerate(lanes) for s in segs], dtype=np.int32)
    masks = ops.lane_raster(torch.from_numpy(rows).cuda(), len(lanes), h, w, lw)
    got = _unpack(masks, w)
    want = np.stack([O.raster_lane(segs, h, w, lw) for segs in lanes])
    assert np.array_equal(got, want)
    assert not _unpack(masks, ((w + 31) // 32) * 32)[:, :, w:].any()                      # no bits past the image edge
    pairs = np.array([(i, j) for i in range(5) for j in range(5) if i != j] + [(2, 2)], dtype=np.int32)
    area, inter = ops.lane_mask_stats(masks, torch.from_numpy(pairs).cuda(), w)
    assert area.cpu().tolist() == [int(m.sum()) for m in want]
    assert inter.cpu().tolist() == [int((want[i] & want[j]).sum()) for i, j in pairs]


def test_full_size_canvas_and_abi_limits(ops):
    from phnet_amd._lib import lib
    h, w, lw = 1280, 1920, 30
    rng = np.random.default_rng(9)
    ys = np.linspace(1270, 500, 12)
    lane = np.stack([900 + np.cumsum(rng.normal(0, 20, 12)), ys], 1).astype(np.float32)
    segs = O.segments_of(O.lane_polyline([tuple(p) for p in lane]))
    rows = torch.tensor([s + (0,) for s in segs], dtype=torch.int32).cuda()
    masks = ops.lane_raster(rows, 1, h, w, lw)
    assert np.array_equal(_unpack(masks, w)[0], O.raster_lane(segs, h, w, lw))
    assert lib().phnet_lane_raster(rows.data_ptr(), rows.shape[0], masks.data_ptr(), 1, 5000, w, lw, None) == -1
    assert lib().phnet_lane_raster(rows.data_ptr(), rows.shape[0], masks.data_ptr(), 1, h, w, 0, None) == -1
    assert lib().phnet_lane_raster(None, 0, None, 0, h, w, lw, None) == 0                 # nothing to draw


def _write(path, lanes):
    path.write_text("".join(" ".join(f"{x:.2f} {y:.2f}" for x, y in l) + " \n" for l in lanes))


def test_evaluation_equals_the_oracle_on_the_same_files(tmp_path):
    from phnet_amd.evaluation import culane as P
    rng = np.random.default_rng(5)
    H, W = 320, 480
    (tmp_path / "anno" / "v").mkdir(parents=True); (tmp_path / "det" / "v").mkdir(parents=True)
    names = []
    for i in range(12):
        n_a, n_d = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        anno = []
        for _ in range(n_a):
            n = int(rng.integers(2, 9))
            ys = np.sort(rng.uniform(80, H, n))[::-1]
            anno.append(np.stack([rng.uniform(60, W - 60) + np.cumsum(rng.normal(0, 12, n)), ys], 1))
        det = [a + rng.normal(0, 6, a.shape) for a in anno[:n_d]] + \
              [np.stack([rng.uniform(0, W, 4), np.sort(rng.uniform(0, H, 4))[::-1]], 1) for _ in range(max(0, n_d - n_a))]
        if i == 3:
            det.append(np.array([[10.0, 10.0]]))                         # a one-point lane: similarity 0
        if i == 4:
            anno.append(np.array([[-500.0, -500.0], [-400.0, -450.0]]))  # drawn entirely off the canvas
        _write(tmp_path / "anno" / "v" / f"{i:03d}.lines.txt", anno)
        if i != 7:                                                       # a frame without a detection file
            _write(tmp_path / "det" / "v" / f"{i:03d}.lines.txt", det)
        names.append(f"/v/{i:03d}.jpg")
    a_dir, d_dir = str(tmp_path / "anno"), str(tmp_path / "det")
    seen = []
    for lw, thr in ((30, 0.5), (10, 0.4), (7, 0.8)):
        want = O.evaluate(a_dir, d_dir, names, W, H, lw, thr, str(tmp_path / "o.txt"))
        got = P.evaluate(a_dir, d_dir, names, W, H, lw, thr, str(tmp_path / "g.txt"), batch_images=5)
        assert (got["tp"], got["fp"], got["fn"]) == (want["tp"], want["fp"], want["fn"])
        for k in ("precision", "recall", "miou", "Fmeasure"):
            assert got[k] == want[k] or (got[k] != got[k] and want[k] != want[k]), (k, got[k], want[k])
        assert (tmp_path / "g.txt").read_text().split("\n")[1:] == (tmp_path / "o.txt").read_text().split("\n")[1:]
        seen.append(want)
    assert seen[0]["tp"] > 0 and seen[0]["fp"] > 0 and seen[0]["fn"] > 0 and 0 < seen[0]["miou"] < 1   # the cases are not degenerate
    # the command line of the binary
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    rc = P.main(["-a", a_dir, "-d", d_dir, "-i", "unused/", "-l", str(tmp_path / "list.txt"), "-w", "7", "-t", "0.8",
                 "-c", str(W), "-r", str(H), "-f", "1", "-o", str(tmp_path / "cli.txt")])
    assert rc == 0 and (tmp_path / "cli.txt").read_text().split("\n")[1:] == (tmp_path / "o.txt").read_text().split("\n")[1:]
