"""CPU tests of the CULane-style evaluator's restatement (oracle/culane_cpu.py) and of the product's host arithmetic
(phnet_amd/evaluation/culane.py: file parsing, spline, matching, counters - everything but the pixel work, which is HIP only).
The reference holds no vectors for its evaluator and its rasteriser is OpenCV (absent): the pieces restated from the reference's
own sources are pinned against independent implementations (scipy's natural cubic spline and assignment solver, an exact
rational rasteriser) and hand-computed cases; parity of the drawn pixels against cv::line itself stays unpinned."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import culane_cpu as O
from phnet_amd.evaluation import culane as P


def _random_lane(rng, n, w=1920, h=1280):
    ys = np.sort(rng.uniform(0.3 * h, h, n))[::-1]
    xs = rng.uniform(0.2 * w, 0.8 * w) + np.cumsum(rng.normal(0, 25, n))
    return np.stack([xs, ys], 1).astype(np.float32)


def test_spline_is_the_natural_cubic_spline_in_chord_length():
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(0)
    for n in (3, 4, 7, 20):
        lane = _random_lane(rng, n)
        got = np.asarray(O.spline_interp_times([tuple(p) for p in lane], 50), dtype=np.float64)
        assert got.shape == ((n - 1) * 50 + 1, 2)
        p = lane.astype(np.float64)
        t = np.concatenate([[0.0], np.cumsum(np.hypot(*(p[1:] - p[:-1]).T))])
        cs = CubicSpline(t, p, bc_type="natural")
        tt = np.concatenate([t[i] + (t[i + 1] - t[i]) / 50 * np.arange(50) for i in range(n - 1)] + [[t[-1]]])
        assert np.abs(got - cs(tt)).max() < 2e-3                      # float32 storage of coordinates up to 1920
    two = O.spline_interp_times([(np.float32(0), np.float32(0)), (np.float32(10), np.float32(5))], 50)
    assert len(two) == 51 and two[50] == (np.float32(10), np.float32(5)) and two[25] == (np.float32(5), np.float32(2.5))


def test_product_spline_and_segments_equal_the_oracle_bit_for_bit():
    rng = np.random.default_rng(1)
    for n in (2, 3, 4, 5, 9, 14, 30):
        for _ in range(5):
            lane = _random_lane(rng, n)
            want = np.asarray(O.lane_polyline([tuple(p) for p in lane]), dtype=np.float32).reshape(-1, 2)
            got = P.lane_polyline(lane)
            assert got.dtype == np.float32 and np.array_equal(got, want)
            assert np.array_equal(P.lane_segments(lane), np.asarray(O.segments_of([tuple(p) for p in want]), dtype=np.int32).reshape(-1, 4))
    dup = np.array([[5, 5], [5, 5], [9, 9]], np.float32)              # coincident points: h = 0 -> nan, like the reference's doubles
    assert np.array_equal(P.lane_polyline(dup), np.asarray(O.lane_polyline([tuple(p) for p in dup]), np.float32), equal_nan=True)
    assert np.array_equal(P.lane_segments(dup), np.asarray(O.segments_of(O.lane_polyline([tuple(p) for p in dup])), np.int32))


def _exact_mask(segs, h, w, lw):
    out = np.zeros((h, w), bool)
    for y in range(h):
        for x in range(w):
            for x0, y0, x1, y1 in segs:
                dx, dy = x1 - x0, y1 - y0
                L2 = dx * dx + dy * dy
                t = Fraction(0) if L2 == 0 else max(Fraction(0), min(Fraction(1), Fraction((x - x0) * dx + (y - y0) * dy, L2)))
                px, py = x0 + t * dx, y0 + t * dy
                if 4 * ((x - px) ** 2 + (y - py) ** 2) <= lw * lw:
                    out[y, x] = True
                    break
    return out


def test_raster_rule_against_exact_rational_distances():
    rng = np.random.default_rng(2)
    h, w = 36, 48
    for lw in (1, 2, 5, 8):
        segs = [tuple(int(v) for v in rng.integers(-10, 60, 4)) for _ in range(4)] + [(5, 5, 5, 5), (-30, -30, -20, -25)]
        assert np.array_equal(O.raster_lane(segs, h, w, lw), _exact_mask(segs, h, w, lw))
    # hand count: horizontal segment (10,20)-(30,20), width 4: 21 x 5 rectangle + 4 pixels in each end cap
    assert int(O.raster_lane([(10, 20, 30, 20)], 40, 50, 4).sum()) == 113
    assert O.cv_round(2.5) == 2 and O.cv_round(3.5) == 4 and O.cv_round(-0.5) == 0 and O.cv_round(1e9) == O.COORD_LIMIT


def test_matching_is_the_reference_kuhn_munkres_and_near_optimal():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(3)
    for m, n in [(1, 1), (2, 4), (4, 2), (4, 4), (3, 5), (6, 3)]:
        for k in range(20):
            sim = rng.uniform(0, 1, (m, n))
            if k % 3 == 0:
                sim[rng.uniform(size=sim.shape) < 0.5] = 0.0            # lanes that do not overlap at all
            if k % 5 == 0:
                sim = np.round(sim, 1)                                  # ties
            want = O.make_match(sim.tolist())
            got = P.make_match(sim)
            assert list(got[0]) == list(want[0]) and list(got[1]) == list(want[1])
            a = got[0]
            matched = [(i, j) for i, j in enumerate(a) if j >= 0]
            assert len({j for _, j in matched}) == len(matched)
            r, c = linear_sum_assignment(sim, maximize=True)
            if len(matched) == min(m, n):
                assert sum(sim[i, j] for i, j in matched) >= sim[r, c].sum() - 1e-2 * min(m, n) - 1e-9
    # the worked case: one good pair and one poor pair
    a, d = P.make_match(np.array([[0.9, 0.1], [0.2, 0.3]]))
    assert list(a) == [0, 1] and list(d) == [0, 1]


def test_count_im_pair_cases():
    sim = np.array([[0.9, 0.1], [0.2, 0.3]])
    assert P.count_im_pair(None, 0, 0, 0.5)[1:] == (0, 0, 0, 0, 1.0)
    assert P.count_im_pair(None, 0, 3, 0.5)[1:] == (0, 3, 0, 0, 0.0)
    assert P.count_im_pair(None, 2, 0, 0.5)[1:] == (0, 0, 0, 2, 0.0)
    match, tp, fp, tn, fn, iou = P.count_im_pair(sim, 2, 2, 0.5)
    assert (match, tp, fp, tn, fn) == ([0, -1], 1, 1, 0, 1) and abs(iou - (0.9 + 0.3) / 2) < 1e-15   # the IoU sum keeps the poor match
    fake = lambda a, d, *_: sim[a][d]
    assert O.count_im_pair([0, 1], [0, 1], 10, 10, 3, 0.5, similarity=fake) == (match, tp, fp, tn, fn, iou)


def test_lane_files_and_output_text(tmp_path):
    f = tmp_path / "a.lines.txt"
    f.write_text("1.5 2 3 4 5 6 \n\n7 8 9\n10 11 x 12\n")
    for lanes in (O.read_lane_file(str(f)), P.read_lane_file(str(f))):
        assert [len(l) for l in lanes] == [3, 0, 1, 1]
        assert tuple(float(v) for v in lanes[0][0]) == (1.5, 2.0)
    assert O.read_lane_file(str(tmp_path / "missing.txt")) == [] and P.read_lane_file(str(tmp_path / "missing.txt")) == []
    out = tmp_path / "out.txt"
    res = P.summarize(7, 3, 1, 4.5, 6, str(out))
    assert res == O.summarize(7, 3, 1, 4.5, 6, str(tmp_path / "out_o.txt"))
    assert out.read_text() == f"file: {out}\ntp: 7 fp: 3 fn: 1\nprecision: 0.7\nrecall: 0.875\nmiou: 0.75\nFmeasure: 0.777778\n\n"
    h = P.read_helper(str(out))
    assert h["tp"] == "7" and float(h["Fmeasure"]) == 0.777778
    agg = P.aggregate({"v1": h, "v2": h})
    assert abs(agg["F1"] - 2 * 0.7 * 0.875 / 1.575) < 1e-12 and agg["miou"] == 0.75
    none = P.summarize(0, 0, 0, 0.0, 1)
    assert none["precision"] == -1.0 and none["recall"] == -1.0 and none["Fmeasure"] == -1.0      # 2 * (-1) * (-1) / (-2)


def test_oracle_evaluate_on_identical_and_shifted_lanes(tmp_path):
    rng = np.random.default_rng(4)
    (tmp_path / "anno").mkdir(); (tmp_path / "det").mkdir()
    names = []
    for i in range(3):
        lanes = [_random_lane(rng, 6, 300, 200) for _ in range(2)]
        txt = "".join(" ".join(f"{x:.3f} {y:.3f}" for x, y in l) + " \n" for l in lanes)
        (tmp_path / "anno" / f"f{i}.lines.txt").write_text(txt)
        shifted = "".join(" ".join(f"{x + (200 if i == 2 else 0):.3f} {y:.3f}" for x, y in l) + " \n" for l in lanes)
        (tmp_path / "det" / f"f{i}.lines.txt").write_text(shifted)
        names.append(f"f{i}.jpg")
    res = O.evaluate(str(tmp_path / "anno") + "/", str(tmp_path / "det") + "/", names, 300, 200, 10, 0.5)
    assert (res["tp"], res["fp"], res["fn"]) == (4, 2, 2)               # the shifted frame matches nothing
    assert abs(res["miou"] - 2 / 3) < 0.05 and abs(res["Fmeasure"] - 2 / 3) < 1e-12
