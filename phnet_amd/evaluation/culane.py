"""CULane-style F1 / mIoU evaluator of the OpenLane-V results: the `./culane/culane_evaluator` binary that
evaluation/evaluate_iou4OL.py:19-62 runs per video (sources: evaluation/culane/src/{evaluate,counter,lane_compare,spline}.cpp,
include/hungarianGraph.hpp), with the pixel work on the GPU.

    python -m phnet_amd.evaluation.culane -a ANNO_DIR -d DETECT_DIR -i IM_DIR -l LIST -w 30 -t 0.5 -c 1920 -r 1280 -f 1 -o OUT

takes the binary's options (evaluate.cpp:56-99) and writes its output file (evaluate.cpp:221-230); `evaluate()` is the same as a
function, `aggregate()` the totals evaluate_iou4OL.py:64-80 prints over the per-video files.

Per image the reference draws every annotated and every detected lane (spline through the points, 50 segments per interval,
cv::line of `lane_width`) on two canvases per PAIR and counts pixels on the host (OpenMP over images).  Here the lanes of a
batch of images are rasterised ONCE each into bit masks in HBM (`phnet_lane_raster`, one workgroup per segment) and the areas /
pairwise intersections are bit counts (`phnet_lane_mask_stats`); spline, matching (the reference's Kuhn-Munkres with its 1e-2
slack) and the counters are host arithmetic as in the reference.  No CPU fallback: without the HIP library this raises.
The rasterisation rule (parity against OpenCV unpinned) is documented in include/phnet_hip.h and oracle/culane_cpu.py.
"""
import math
import os
import sys
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import hip_ops as K

COORD_LIMIT = 1 << 13
Lane = np.ndarray           # [n, 2] float32 (cv::Point2f)


# ---------------------------------------------------------------------------------------------- files (evaluate.cpp:236-263)
def read_lane_file(path: str) -> List[Lane]:
    """One lane per text line ("x y x y ..."); a missing file is an image without lanes; an empty line is a lane without points."""
    try:
        with open(path, "r") as fh:
            lines = fh.read().split("\n")
    except OSError:
        return []
    if lines and lines[-1] == "":
        lines.pop()
    lanes = []
    for line in lines:
        vals = []
        for tok in line.split():
            try:
                vals.append(float(tok))
            except ValueError:
                break
        n = len(vals) // 2
        lanes.append(np.asarray(vals[:2 * n], dtype=np.float64).astype(np.float32).reshape(n, 2))
    return lanes


# ---------------------------------------------------------------------------------------------- spline (spline.cpp)
def spline_interp_times(lane: Lane, times: int = 50) -> Lane:
    """Natural cubic spline in chord length through the points, `times` points per interval plus the last point
    (spline.cpp:9-47, cal_fun :118-178: the tridiagonal system is solved as written, in double; points come back as float)."""
    n = len(lane)
    p = lane.astype(np.float64)
    if n == 2:
        k = np.arange(times + 1, dtype=np.float64)
        return (p[0] + (p[1] - p[0]) * k[:, None] / times).astype(np.float32)
    if n < 2:
        return np.zeros((0, 2), np.float32)
    with np.errstate(all="ignore"):
        d = p[1:] - p[:-1]
        h = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2)
        A, B, C = h[:-1].copy(), 2 * (h[:-1] + h[1:]), h[1:].copy()
        D = 6 * (d[1:] / h[1:, None] - d[:-1] / h[:-1, None])            # [n-2][2] = (Dx, Dy)
        C[0] = C[0] / B[0]
        D[0] = D[0] / B[0]
        for i in range(1, n - 2):
            tmp = B[i] - A[i] * C[i - 1]
            C[i] = C[i] / tmp
            D[i] = (D[i] - A[i] * D[i - 1]) / tmp
        M = np.zeros((n, 2))
        M[n - 2] = D[n - 3]
        for i in range(n - 4, -1, -1):
            M[i + 1] = D[i] - C[i] * M[i + 2]
        M[0] = 0.0
        M[n - 1] = 0.0
        hh = h[:, None]
        a = p[:-1]
        b = d / hh - (2 * hh * M[:-1] + hh * M[1:]) / 6
        c = M[:-1] / 2
        dd = (M[1:] - M[:-1]) / (6 * hh)
        t = (hh / times) * np.arange(times, dtype=np.float64)[None, :]    # [n-1][times]: t1 = delta * k
        t = t[:, :, None]
        pts = a[:, None, :] + b[:, None, :] * t + c[:, None, :] * np.power(t, 2) + dd[:, None, :] * np.power(t, 3)
    return np.concatenate([pts.reshape(-1, 2).astype(np.float32), lane[-1:].astype(np.float32)], axis=0)


def lane_polyline(lane: Lane) -> Lane:
    """lane_compare.cpp:22-39: a two-point lane is drawn as it is, a longer one through the spline."""
    return lane if len(lane) == 2 else spline_interp_times(lane, 50)


def lane_segments(lane: Lane) -> np.ndarray:
    """[s, 4] int32 end points of the cv::line calls of one lane: cv::line takes integer Points = cvRound of the Point2f
    (nearest, ties to even), clamped to +-2^13 here."""
    poly = lane_polyline(lane).astype(np.float64)
    poly = np.where(np.isnan(poly), 0.0, poly)
    q = np.clip(np.rint(poly), -COORD_LIMIT, COORD_LIMIT).astype(np.int32)
    if len(q) < 2:
        return np.zeros((0, 4), np.int32)
    return np.concatenate([q[:-1], q[1:]], axis=1)


# ---------------------------------------------------------------------------------------------- lane IoU on the device
def similarity_matrices(images: Sequence, height: int, width: int, lane_width: int, device="cuda") -> List[np.ndarray]:
    """images: [(anno_lanes, detect_lanes)] -> per image the [len(anno)][len(detect)] float64 IoU matrix of
    LaneCompare::get_lane_similarity (0 where a lane has fewer than two points).  Every lane is drawn once."""
    seg_rows, pairs, layout = [], [], []
    n_lanes = 0
    for anno, det in images:
        ids = []
        for lane in list(anno) + list(det):
            if len(lane) < 2:
                ids.append(-1)
                continue
            s = lane_segments(lane)
            seg_rows.append(np.concatenate([s, np.full((len(s), 1), n_lanes, np.int32)], axis=1))
            ids.append(n_lanes)
            n_lanes += 1
        ia, idt = ids[:len(anno)], ids[len(anno):]
        first = len(pairs)
        for i in ia:
            for j in idt:
                if i >= 0 and j >= 0:
                    pairs.append((i, j))
        layout.append((ia, idt, first))
    out = [np.zeros((len(a), len(d)), np.float64) for a, d in images]
    if not pairs:
        return out
    dev = torch.device(device)
    segs = torch.from_numpy(np.concatenate(seg_rows, axis=0)).to(dev)
    area, inter = K.lane_mask_iou(segs, n_lanes, torch.tensor(pairs, dtype=torch.int32, device=dev), height, width, lane_width)
    area, inter = area.cpu().numpy().astype(np.float64), inter.cpu().numpy().astype(np.float64)
    for (ia, idt, first), m in zip(layout, out):
        p = first
        for r, i in enumerate(ia):
            for c, j in enumerate(idt):
                if i >= 0 and j >= 0:
                    with np.errstate(all="ignore"):
                        m[r, c] = inter[p] / (area[i] + area[j] - inter[p])      # 0 / 0 = nan, as in the reference
                    p += 1
    return out


# ---------------------------------------------------------------------------------------------- matching (hungarianGraph.hpp, counter.cpp)
def make_match(sim: np.ndarray):
    """Counter::makeMatch (counter.cpp:143-161) on pipartiteGraph::match (hungarianGraph.hpp:39-67): Kuhn-Munkres on the
    similarity matrix (transposed when there are more rows than columns), equality within 1e-2, early return when no slack is
    left.  -> (anno_match, detect_match), -1 = unmatched."""
    m, n = sim.shape
    swap = m > n
    mat = sim.T if swap else sim
    L, R = mat.shape
    left_match, right_match = [-1] * L, [-1] * R
    lw = [-1e5] * L
    for i in range(L):                                   # `if (leftWeight < mat) leftWeight = mat`: a nan entry never raises the maximum
        w = -1e5
        for j in range(R):
            if w < mat[i, j]:
                w = float(mat[i, j])
        lw[i] = w
    rw = [0.0] * R
    done = False
    for u in range(L):
        while True:
            left_used, right_used = [False] * L, [False] * R
            # iterative form of matchDfs: (vertex, next column to try)
            ok = _augment(u, mat, lw, rw, left_used, right_used, left_match, right_match)
            if ok:
                break
            dmin = 1e10
            for i in range(L):
                if left_used[i]:
                    for j in range(R):
                        if not right_used[j]:
                            dmin = min(dmin, lw[i] + rw[j] - float(mat[i, j]))
            if dmin == 1e10:
                done = True
                break
            for i in range(L):
                if left_used[i]:
                    lw[i] -= dmin
            for j in range(R):
                if right_used[j]:
                    rw[j] += dmin
        if done:
            break
    return (right_match, left_match) if swap else (left_match, right_match)


def _augment(u, mat, lw, rw, left_used, right_used, left_match, right_match) -> bool:
    """pipartiteGraph::matchDfs(u) without recursion: same visiting order, same marks."""
    R = mat.shape[1]
    stack = [[u, 0, -1]]                                  # vertex, next column, column through which the child was entered
    left_used[u] = True
    while stack:
        top = stack[-1]
        x, v = top[0], top[1]
        advanced = False
        while v < R:
            if not right_used[v] and abs(lw[x] + rw[v] - float(mat[x, v])) < 1e-2:
                right_used[v] = True
                if right_match[v] == -1:
                    # success: unwind, re-matching along the path
                    right_match[v] = x
                    left_match[x] = v
                    stack.pop()
                    while stack:
                        px, _, pv = stack.pop()
                        right_match[pv] = px
                        left_match[px] = pv
                    return True
                top[1] = v + 1
                top[2] = v
                child = right_match[v]
                left_used[child] = True
                stack.append([child, 0, -1])
                advanced = True
                break
            v += 1
        if not advanced:
            stack.pop()                                   # this vertex failed: the parent continues with its next column
    return False


def count_im_pair(sim: Optional[np.ndarray], n_anno: int, n_detect: int, threshold: float):
    """Counter::count_im_pair (counter.cpp:83-141) -> (anno_match, tp, fp, tn, fn, iou)."""
    anno_match = [-1] * n_anno
    if n_anno == 0 and n_detect == 0:
        return anno_match, 0, 0, 0, 0, 1.0
    if n_anno == 0:
        return anno_match, 0, n_detect, 0, 0, 0.0
    if n_detect == 0:
        return anno_match, 0, 0, 0, n_anno, 0.0
    anno_match = list(make_match(sim)[0])
    tp, iou = 0, 0.0
    for i in range(n_anno):
        j = anno_match[i]
        if j >= 0:
            iou += float(sim[i, j])
        if j >= 0 and sim[i, j] > threshold:
            tp += 1
        else:
            anno_match[i] = -1
    return anno_match, tp, n_detect - tp, 0, n_anno - tp, iou / n_detect


# ---------------------------------------------------------------------------------------------- driver (evaluate.cpp:42-233)
def evaluate(anno_dir: str, detect_dir: str, names: Sequence[str], width: int = 1920, height: int = 1080, lane_width: int = 10,
             threshold: float = 0.4, output_file: Optional[str] = None, batch_images: int = 64, device="cuda") -> dict:
    """`names`: the lines of the -l list file (image paths; the label files are `<dir><name without extension>.lines.txt`,
    plain string concatenation as in evaluate.cpp:156-160).  Defaults are the binary's."""
    if lane_width < 1:
        raise ValueError("width_lane must be positive")              # evaluate.cpp:116-121
    tp = fp = fn = 0
    iou = 0.0
    names = list(names)
    for b in range(0, len(names), batch_images):
        images = []
        for name in names[b:b + batch_images]:
            stem = (name[:name.rfind(".")] if "." in name else name) + ".lines.txt"      # substr(0, find_last_of("."))
            images.append((read_lane_file(anno_dir + stem), read_lane_file(detect_dir + stem)))
        sims = similarity_matrices(images, height, width, lane_width, device)
        for (anno, det), sim in zip(images, sims):
            _, a, b_, _, c, d = count_im_pair(sim, len(anno), len(det), threshold)
            tp += a; fp += b_; fn += c; iou += d
    return summarize(tp, fp, fn, iou, len(names), output_file)


def summarize(tp: int, fp: int, fn: int, iou: float, n_images: int, output_file: Optional[str] = None) -> dict:
    """evaluate.cpp:192-230; Counter::get_precision / get_recall return -1 when undefined."""
    with np.errstate(all="ignore"):
        miou = float(np.float64(iou) / np.float64(n_images))
        precision = -1.0 if tp + fp == 0 else tp / float(tp + fp)
        recall = -1.0 if tp + fn == 0 else tp / float(tp + fn)
        f = float(np.float64(2 * precision * recall) / np.float64(precision + recall))
    res = dict(tp=tp, fp=fp, fn=fn, precision=precision, recall=recall, miou=miou, Fmeasure=f)
    if output_file:
        g = lambda v: "nan" if v != v else "%g" % v              # ofstream << double at default precision
        with open(output_file, "w") as fh:
            fh.write(f"file: {output_file}\ntp: {tp} fp: {fp} fn: {fn}\nprecision: {g(precision)}\nrecall: {g(recall)}\n"
                     f"miou: {g(miou)}\nFmeasure: {g(f)}\n\n")
    return res


def read_helper(path: str) -> dict:
    """evaluate_iou4OL.py:9-16: the key / value pairs of an output file (first line skipped)."""
    toks = " ".join(open(path, "r").readlines()[1:]).split(" ")
    return {k[:-1]: v for k, v in zip(toks[0::2], toks[1::2])}


def aggregate(results: dict) -> dict:
    """evaluate_iou4OL.py:64-80: totals over the per-video results ({name: read_helper(file)})."""
    tp = sum(int(v["tp"]) for v in results.values())
    fp = sum(int(v["fp"]) for v in results.values())
    fn = sum(int(v["fn"]) for v in results.values())
    miou = sum(float(v["miou"]) for v in results.values())
    p, r = tp * 1.0 / (tp + fp), tp * 1.0 / (tp + fn)
    return {"miou": miou / len(results), "F1": 2 * p * r / (p + r), "p": p, "r": r}


def main(argv=None) -> int:
    """The binary's command line (evaluate.cpp:42-99; -i, -s and -f are accepted and, like -i and -f there, change nothing)."""
    import getopt
    opts, _ = getopt.getopt(sys.argv[1:] if argv is None else argv, "ha:d:i:l:w:t:c:r:sf:o:")
    o = {"-a": "/data/driving/eval_data/anno_label/", "-d": "/data/driving/eval_data/predict_label/",
         "-l": "/data/driving/eval_data/img/all.txt", "-w": "10", "-t": "0.4", "-c": "1920", "-r": "1080", "-o": "./output.txt"}
    o.update(dict(opts))
    if "-h" in o:
        print(__doc__)
        return 0
    if not os.path.exists(o["-l"]):
        print(f"Error: file {o['-l']} not exist!", file=sys.stderr)
        return 1
    names = open(o["-l"]).read().split("\n")
    if names and names[-1] == "":
        names.pop()
    res = evaluate(o["-a"], o["-d"], names, int(o["-c"]), int(o["-r"]), int(o["-w"]), float(o["-t"]), o["-o"])
    print(f"tp: {res['tp']} fp: {res['fp']} fn: {res['fn']}", file=sys.stderr)
    for k in ("precision", "recall", "miou", "Fmeasure"):
        print(f"{k}: {res[k]:g}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
