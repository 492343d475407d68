"""Known-answer and cross-restatement tests of the lane-NMS oracle (CPU).
The reference ships no vectors for libs/ops; the KATs below are derived by hand
from libs/ops/csrc/nms_kernel.cu:26-48,99-143."""
import numpy as np
import pytest

from oracle import lane_nms as N

S = 36


def row(start_y, length, xs, cls=(0.0, 1.0), sx=0.0):
    r = np.zeros(5 + S, dtype=np.float32)
    r[0], r[1], r[2], r[3], r[4] = cls[0], cls[1], start_y, sx, length
    r[5:] = xs
    return r


def test_kat_two_parallel_lanes_far_apart_are_both_kept():
    a = row(0.0, 36, np.full(S, 100.0))
    b = row(0.0, 36, np.full(S, 151.0))          # mean |dx| = 51 >= 50 -> not similar
    keep, num, parent = N.lane_nms(np.stack([a, b]), np.array([0.9, 0.8], np.float32), 50.0, 4)
    assert num == 2 and keep.tolist() == [0, 1] and parent.tolist() == [1, 2]


def test_kat_close_lane_is_suppressed_by_higher_score():
    a = row(0.0, 36, np.full(S, 100.0))
    b = row(0.0, 36, np.full(S, 149.0))          # mean |dx| = 49 < 50 -> similar
    keep, num, parent = N.lane_nms(np.stack([a, b]), np.array([0.3, 0.8], np.float32), 50.0, 4)
    assert num == 1 and keep.tolist() == [1, 0] and parent.tolist() == [1, 1]


def test_kat_disjoint_vertical_extents_never_similar():
    a = row(0.0, 10, np.full(S, 100.0))          # strips 0..9
    b = row(20 / 35, 10, np.full(S, 100.0))      # strips 20..29
    keep, num, parent = N.lane_nms(np.stack([a, b]), np.array([0.9, 0.8], np.float32), 50.0, 4)
    assert num == 2 and parent.tolist() == [1, 2]


def test_kat_overlap_window_only():
    xa = np.full(S, 100.0); xb = np.full(S, 100.0)
    xb[:10] = 1000.0                              # differs only below b's start
    a = row(0.0, 36, xa)
    b = row(10 / 35, 20, xb)                      # start 10, end 29 -> compared on 10..29 only
    keep, num, _ = N.lane_nms(np.stack([a, b]), np.array([0.9, 0.8], np.float32), 50.0, 4)
    assert num == 1 and keep[0] == 0


def test_kat_zero_length_lane():
    a = row(0.0, 36, np.full(S, 100.0))
    z = row(0.0, 0.0, np.full(S, 100.0))          # end = start - 1 - 1 + .5 -> trunc(-1.5)... => end < start
    keep, num, parent = N.lane_nms(np.stack([a, z]), np.array([0.9, 0.8], np.float32), 50.0, 4)
    assert num == 2 and parent.tolist() == [1, 2]


def test_kat_top_k_stops_sweep_and_zero_fills():
    rows = np.stack([row(0.0, 36, np.full(S, 100.0 + 60 * i)) for i in range(6)])
    sc = np.linspace(0.9, 0.4, 6).astype(np.float32)
    keep, num, parent = N.lane_nms(rows, sc, 50.0, 4)
    assert num == 4 and keep.tolist() == [0, 1, 2, 3, 0, 0] and parent.tolist() == [1, 2, 3, 4, 0, 0]


def test_empty_and_single():
    keep, num, parent = N.lane_nms(np.zeros((0, 5 + S), np.float32), np.zeros(0, np.float32), 50.0, 4)
    assert num == 0 and keep.shape == (0,)
    keep, num, parent = N.lane_nms(row(0.2, 5, np.zeros(S))[None], np.array([0.7], np.float32), 50.0, 4)
    assert num == 1 and keep.tolist() == [0] and parent.tolist() == [1]


@pytest.mark.parametrize("K,n_off,seed", [(1, 36, 0), (63, 36, 1), (64, 36, 2), (65, 36, 3), (240, 36, 4), (130, 72, 5)])
def test_c_matches_numpy_restatement_on_random_lanes(K, n_off, seed):
    r = np.random.default_rng(seed)
    rows = np.zeros((K, 5 + n_off), np.float32)
    base = r.uniform(0, 800, (K, 1)) + r.normal(0, 12, (K, n_off)).cumsum(1)
    # clusters so that suppression actually happens
    base[K // 2:] = base[:K - K // 2] + r.normal(0, 30, (K - K // 2, 1))
    rows[:, 5:] = base
    rows[:, 2] = r.uniform(-0.3, 1.2, K)          # includes negative / >1 starts (unsigned-char loop edge)
    rows[:, 4] = r.uniform(-2, 40, K)
    scores = r.permutation(K).astype(np.float32) / K
    a = N.lane_nms(rows, scores, 50.0, 4 if K < 200 else 1000)
    b = N.lane_nms_numpy(rows, scores, 50.0, 4 if K < 200 else 1000)
    assert a[1] == b[1] and a[0].tolist() == b[0].tolist() and a[2].tolist() == b[2].tolist()


def test_kat_nan_and_tied_scores_have_a_total_order():
    """csrc/nms.cpp:51 sorts with ATen, which orders NaN above every number: a NaN-scored row is visited first (and, here,
    suppresses its near-duplicate); equal scores fall back to the lower index.  A total order is also what keeps the HIP
    kernel's rank -> row table fully written."""
    import torch
    sc = np.array([0.5, np.nan, 0.9, 0.5, np.nan, -np.inf, np.inf], np.float32)
    order = N.score_order(sc)
    assert order.tolist() == [1, 4, 6, 2, 0, 3, 5]
    assert order.tolist() == torch.sort(torch.from_numpy(sc), descending=True, stable=True)[1].tolist()
    rows = np.stack([row(0.0, 36, np.full(S, 100.0 + 60 * i)) for i in range(7)])
    rows[0, 5:] = rows[1, 5:] + 1.0                 # row 0 is a near-duplicate of the NaN-scored row 1
    keep, num, parent = N.lane_nms(rows, sc, 50.0, 4)
    assert num == 4 and keep.tolist()[:4] == [1, 4, 6, 2] and parent[0] == 1 and parent[1] == 1
    a = N.lane_nms_numpy(rows, sc, 50.0, 4)
    assert a[1] == num and a[0].tolist() == keep.tolist() and a[2].tolist() == parent.tolist()
