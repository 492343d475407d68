"""ctypes + numpy front-ends of the lane-NMS oracle  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

`lane_nms` mirrors the reference call `libs.ops.nms(boxes, scores, overlap, top_k)`
(libs/ops/nms.py:32-33 -> csrc/nms.cpp:44-57): sort scores descending, then run the
C restatement in oracle/lane_nms.c.  `lane_nms_numpy` is an independent pure-numpy
restatement of the same source lines, used to cross-check the C on small cases.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblane_nms_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lane_nms.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        # -ffp-contract=off: the reference arithmetic has no fused multiply-adds
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.phnet_oracle_lane_nms.restype = ctypes.c_int
        _lib.phnet_oracle_lane_similar.restype = ctypes.c_int
    return _lib


def lane_nms_sorted(rows: np.ndarray, order: np.ndarray, thresh: float, top_k: int):
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    order = np.ascontiguousarray(order, dtype=np.int64)
    k, prop = rows.shape
    keep = np.zeros(k, dtype=np.int64)
    parent = np.zeros(k, dtype=np.int64)
    num = np.zeros(1, dtype=np.int64)
    rc = _load().phnet_oracle_lane_nms(
        rows.ctypes.data_as(ctypes.c_void_p), order.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_int64(k), ctypes.c_int(prop - 5), ctypes.c_float(thresh), ctypes.c_int64(top_k),
        keep.ctypes.data_as(ctypes.c_void_p), num.ctypes.data_as(ctypes.c_void_p),
        parent.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return keep, int(num[0]), parent


def score_order(scores: np.ndarray) -> np.ndarray:
    """Descending score order (csrc/nms.cpp:51); ties (unspecified in the reference) by lower index; a NaN score sorts
    above every number (ATen's sort treats NaN as the largest value), NaNs among themselves by index."""
    s = np.asarray(scores, dtype=np.float32)
    nan = np.isnan(s)
    with np.errstate(invalid="ignore"):
        key = np.where(nan, np.float32(0), -s)
    return np.lexsort((np.arange(s.size), key, ~nan)).astype(np.int64)       # primary: NaN first; then -score; then index


def lane_nms(rows, scores, overlap: float, top_k: int):
    """Accepts numpy arrays or CPU torch tensors; returns the same container kind."""
    is_torch = hasattr(rows, "detach")
    r = rows.detach().cpu().numpy() if is_torch else np.asarray(rows)
    s = scores.detach().cpu().numpy() if is_torch else np.asarray(scores)
    keep, num, parent = lane_nms_sorted(r, score_order(s), float(overlap), int(top_k))
    if is_torch:
        import torch
        return torch.from_numpy(keep), torch.tensor(num, dtype=torch.int64), torch.from_numpy(parent)
    return keep, num, parent


# ---------------------------------------------------------------------------------------------
# independent numpy restatement (nms_kernel.cu:26-48, 99-143), float32 scalar arithmetic
# ---------------------------------------------------------------------------------------------
def _similar_numpy(a: np.ndarray, b: np.ndarray, n_offsets: int, thr: np.float32) -> bool:
    f32 = np.float32
    n_strips = n_offsets - 1

    def extent(v):
        start = int(np.float64(v[2] * f32(n_strips)) + 0.5)          # C truncation == int() toward zero
        f = f32(f32(f32(start) + v[4]) - f32(1.0))
        end = int(np.float64(f) + 0.5 - (1.0 if f32(v[4] - f32(1.0)) < 0 else 0.0))
        return start, end
    sa, ea = extent(a)
    sb, eb = extent(b)
    start, end = max(sa, sb), min(ea, eb, n_offsets - 1)
    if end < start:
        return False
    dist = f32(0.0)
    i = (5 + start) & 0xFF                                             # unsigned char loop counter
    while i <= 5 + end:
        dist = f32(dist + (f32(b[i] - a[i]) if a[i] < b[i] else f32(a[i] - b[i])))
        i = (i + 1) & 0xFF
    return bool(dist < f32(thr * f32(end - start + 1)))


def lane_nms_numpy(rows: np.ndarray, scores: np.ndarray, thresh: float, top_k: int):
    rows = np.asarray(rows, dtype=np.float32)
    k, prop = rows.shape
    order = score_order(scores)
    keep = np.zeros(k, dtype=np.int64)
    parent = np.zeros(k, dtype=np.int64)
    removed = np.zeros(k, dtype=bool)
    kept = 0
    with np.errstate(all="ignore"):
        for i in range(k):
            if removed[i]:
                continue
            keep[kept] = order[i]
            for j in range(i + 1, k):
                if _similar_numpy(rows[order[i]], rows[order[j]], prop - 5, np.float32(thresh)):
                    removed[j] = True
                    parent[order[j]] = kept + 1
            parent[order[i]] = kept + 1
            kept += 1
            if kept == top_k:
                break
    return keep, min(top_k, kept), parent
