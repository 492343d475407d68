"""CPU restatement of the reference's CULane-style evaluator (evaluation/culane/src/*.cpp, driven by
evaluation/evaluate_iou4OL.py:19-88) - TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in phnet_amd/).

What is restated, line by line, from the reference's own sources (plain Python / numpy float64, results stored as float32
where the reference stores cv::Point2f):
  * read_lane_file            evaluate.cpp:236-263   (one lane per text line, x y pairs; every line - even an empty one - is a lane)
  * spline_interp_times       spline.cpp:9-47        (natural cubic spline in chord length, `times` points per interval + last point)
  * cal_fun                   spline.cpp:118-178     (tridiagonal solve as written)
  * BipartiteGraph.match      include/hungarianGraph.hpp:6-68 (Kuhn-Munkres with the reference's 1e-2 equality slack and early return)
  * make_match                counter.cpp:143-161    (transpose when there are more annotated than detected lanes)
  * count_im_pair             counter.cpp:83-141     (the empty-side cases, matched-IoU sum over ALL matches, threshold on tp)
  * evaluate                  evaluate.cpp:127-230   (totals, precision / recall / F, miou = sum / number of images, output text)

What is NOT pinned ("parity unpinned"): lane_similarity (lane_compare.cpp:11-57) draws every poly-line with OpenCV's
cv::line(thickness = lane_width) and counts pixels.  OpenCV is neither in the reference tree nor in this image, the
reference's evaluator is a prebuilt binary (never run) whose sources need OpenCV to build - so its rasteriser cannot be
executed or compiled here.  cv::line draws a thick segment as a convex quadrilateral (normal offset thickness/2) plus filled
end circles; this file (and the HIP kernel it checks) uses the ideal shape those approximate: pixel (x, y) belongs to a segment
iff its centre lies within lane_width / 2 of the segment between the ROUNDED end points (cv::line takes integer Points:
cvRound of the Point2f), in exact integer arithmetic.  Boundary pixels can differ from OpenCV's scan conversion, so IoU values
can differ in the third decimal; everything downstream of the IoU matrix is the reference's arithmetic.
"""
import math

import numpy as np

COORD_LIMIT = 1 << 13            # end points are clamped to +-2^13 (images up to 4096 x 4096) so that the int64 distance tests cannot overflow


# ---------------------------------------------------------------------------------------------- files
def read_lane_file(path):
    """evaluate.cpp:236-263.  A missing file is an image without lanes (the reference prints "fail" and returns no lanes)."""
    lanes = []
    try:
        fh = open(path, "r")
    except OSError:
        return lanes
    with fh:
        lines = fh.read().split("\n")
    if lines and lines[-1] == "":                  # std::getline: a trailing newline does not open another line
        lines.pop()
    return [_parse_line(line) for line in lines]


def _parse_line(line):
    pts = []
    toks = line.split()
    i = 0
    while i + 1 < len(toks):                      # `while (ss >> x >> y)`: stops at the first token that is not a number
        try:
            x, y = float(toks[i]), float(toks[i + 1])
        except ValueError:
            break
        pts.append((np.float32(x), np.float32(y)))
        i += 2
    return pts


# ---------------------------------------------------------------------------------------------- spline
def cal_fun(pts):
    """spline.cpp:118-178: per interval (a, b, c, d) for x and y and the chord length h."""
    n = len(pts)
    if n <= 2:
        return []
    px = [float(p[0]) for p in pts]
    py = [float(p[1]) for p in pts]
    h = [math.sqrt((px[i + 1] - px[i]) ** 2 + (py[i + 1] - py[i]) ** 2) for i in range(n - 1)]
    A = [0.0] * (n - 2); B = [0.0] * (n - 2); C = [0.0] * (n - 2); Dx = [0.0] * (n - 2); Dy = [0.0] * (n - 2)
    with np.errstate(all="ignore"):
        for i in range(n - 2):
            A[i] = h[i]
            B[i] = 2 * (h[i] + h[i + 1])
            C[i] = h[i + 1]
            Dx[i] = 6 * (_div(px[i + 2] - px[i + 1], h[i + 1]) - _div(px[i + 1] - px[i], h[i]))
            Dy[i] = 6 * (_div(py[i + 2] - py[i + 1], h[i + 1]) - _div(py[i + 1] - py[i], h[i]))
        C[0] = _div(C[0], B[0]); Dx[0] = _div(Dx[0], B[0]); Dy[0] = _div(Dy[0], B[0])
        for i in range(1, n - 2):
            tmp = B[i] - A[i] * C[i - 1]
            C[i] = _div(C[i], tmp)
            Dx[i] = _div(Dx[i] - A[i] * Dx[i - 1], tmp)
            Dy[i] = _div(Dy[i] - A[i] * Dy[i - 1], tmp)
        Mx = [0.0] * n; My = [0.0] * n
        Mx[n - 2] = Dx[n - 3]; My[n - 2] = Dy[n - 3]
        for i in range(n - 4, -1, -1):
            Mx[i + 1] = Dx[i] - C[i] * Mx[i + 2]
            My[i + 1] = Dy[i] - C[i] * My[i + 2]
        Mx[0] = Mx[n - 1] = My[0] = My[n - 1] = 0.0
        out = []
        for i in range(n - 1):
            out.append(dict(
                a_x=px[i], b_x=_div(px[i + 1] - px[i], h[i]) - (2 * h[i] * Mx[i] + h[i] * Mx[i + 1]) / 6, c_x=Mx[i] / 2,
                d_x=_div(Mx[i + 1] - Mx[i], 6 * h[i]),
                a_y=py[i], b_y=_div(py[i + 1] - py[i], h[i]) - (2 * h[i] * My[i] + h[i] * My[i + 1]) / 6, c_y=My[i] / 2,
                d_y=_div(My[i + 1] - My[i], 6 * h[i]), h=h[i]))
    return out


def _div(a, b):
    """C++ double division (inf / nan instead of ZeroDivisionError: two coincident points give h = 0)."""
    return float(np.float64(a) / np.float64(b))


def spline_interp_times(pts, times=50):
    """spline.cpp:9-47."""
    res = []
    if len(pts) == 2:
        x1, y1, x2, y2 = float(pts[0][0]), float(pts[0][1]), float(pts[1][0]), float(pts[1][1])
        for k in range(times + 1):
            res.append((np.float32(x1 + (x2 - x1) * k / times), np.float32(y1 + (y2 - y1) * k / times)))
    elif len(pts) > 2:
        for f in cal_fun(pts):
            delta = f["h"] / times
            for k in range(times):
                t1 = delta * k
                with np.errstate(all="ignore"):
                    x1 = np.float64(f["a_x"]) + np.float64(f["b_x"]) * t1 + np.float64(f["c_x"]) * t1 ** 2 + np.float64(f["d_x"]) * t1 ** 3
                    y1 = np.float64(f["a_y"]) + np.float64(f["b_y"]) * t1 + np.float64(f["c_y"]) * t1 ** 2 + np.float64(f["d_y"]) * t1 ** 3
                res.append((np.float32(x1), np.float32(y1)))
        res.append(pts[-1])
    return res


def lane_polyline(pts):
    """lane_compare.cpp:22-39: two-point lanes are drawn as they are, longer ones through the spline (50 points per interval)."""
    return list(pts) if len(pts) == 2 else spline_interp_times(pts, 50)


# ---------------------------------------------------------------------------------------------- raster (parity unpinned, see header)
def cv_round(v):
    """cv::saturate_cast<int>(float) = cvRound: nearest, ties to even; NaN / out of range -> clamped here."""
    v = float(v)
    if not math.isfinite(v):
        return 0 if math.isnan(v) else (COORD_LIMIT if v > 0 else -COORD_LIMIT)
    return int(max(-COORD_LIMIT, min(COORD_LIMIT, np.rint(v))))


def segments_of(polyline):
    """Integer end points (x0, y0, x1, y1) of the cv::line calls of one lane (lane_compare.cpp:42-49)."""
    p = [(cv_round(x), cv_round(y)) for x, y in polyline]
    return [(p[i][0], p[i][1], p[i + 1][0], p[i + 1][1]) for i in range(len(p) - 1)]


def raster_lane(segs, height, width, lane_width):
    """bool [height][width]: pixels whose centre is within lane_width / 2 of one of the segments (4 d^2 <= lane_width^2, int64)."""
    mask = np.zeros((height, width), dtype=bool)
    r = (lane_width + 1) // 2
    w2 = np.int64(lane_width) * lane_width
    for x0, y0, x1, y1 in segs:
        xa, xb = max(0, min(x0, x1) - r), min(width - 1, max(x0, x1) + r)
        ya, yb = max(0, min(y0, y1) - r), min(height - 1, max(y0, y1) + r)
        if xa > xb or ya > yb:
            continue
        ys, xs = np.mgrid[ya:yb + 1, xa:xb + 1].astype(np.int64)
        dx, dy = np.int64(x1 - x0), np.int64(y1 - y0)
        qx, qy = xs - x0, ys - y0
        L2 = dx * dx + dy * dy
        dot = qx * dx + qy * dy
        d0 = qx * qx + qy * qy                                  # squared distance to the first end point
        ex, ey = xs - x1, ys - y1
        d1 = ex * ex + ey * ey
        cross = qx * dy - qy * dx
        inside = np.where(dot <= 0, 4 * d0 <= w2, np.where(dot >= L2, 4 * d1 <= w2, 4 * cross * cross <= w2 * L2))
        mask[ya:yb + 1, xa:xb + 1] |= inside
    return mask


def lane_similarity(lane1, lane2, height, width, lane_width):
    """lane_compare.cpp:11-57: IoU of the two drawn lanes (0 when a lane has fewer than 2 points; 0/0 = nan as in C++)."""
    if len(lane1) < 2 or len(lane2) < 2:
        return 0.0
    m1 = raster_lane(segments_of(lane_polyline(lane1)), height, width, lane_width)
    m2 = raster_lane(segments_of(lane_polyline(lane2)), height, width, lane_width)
    s1, s2, inter = float(m1.sum()), float(m2.sum()), float((m1 & m2).sum())
    with np.errstate(all="ignore"):
        return float(np.float64(inter) / np.float64(s1 + s2 - inter))


# ---------------------------------------------------------------------------------------------- matching
class BipartiteGraph:
    """include/hungarianGraph.hpp:6-68 (`pipartiteGraph`)."""

    def __init__(self, mat):
        self.mat = mat
        self.left, self.right = len(mat), len(mat[0])

    def _dfs(self, u):
        self.left_used[u] = True
        for v in range(self.right):
            if not self.right_used[v] and abs(self.lw[u] + self.rw[v] - self.mat[u][v]) < 1e-2:
                self.right_used[v] = True
                if self.right_match[v] == -1 or self._dfs(self.right_match[v]):
                    self.right_match[v] = u
                    self.left_match[u] = v
                    return True
        return False

    def match(self):
        self.left_match = [-1] * self.left
        self.right_match = [-1] * self.right
        self.rw = [0.0] * self.right
        self.lw = []
        for i in range(self.left):
            w = -1e5
            for j in range(self.right):
                if w < self.mat[i][j]:
                    w = self.mat[i][j]
            self.lw.append(w)
        for u in range(self.left):
            while True:
                self.left_used = [False] * self.left
                self.right_used = [False] * self.right
                if self._dfs(u):
                    break
                d = 1e10
                for i in range(self.left):
                    if self.left_used[i]:
                        for j in range(self.right):
                            if not self.right_used[j]:
                                d = min(d, self.lw[i] + self.rw[j] - self.mat[i][j])
                if d == 1e10:
                    return
                for i in range(self.left):
                    if self.left_used[i]:
                        self.lw[i] -= d
                for j in range(self.right):
                    if self.right_used[j]:
                        self.rw[j] += d


def make_match(sim):
    """counter.cpp:143-161 -> (anno_match, detect_match)."""
    m, n = len(sim), len(sim[0])
    swap = m > n
    mat = [[sim[j][i] for j in range(m)] for i in range(n)] if swap else [list(r) for r in sim]
    g = BipartiteGraph(mat)
    g.match()
    return (g.right_match, g.left_match) if swap else (g.left_match, g.right_match)


def count_im_pair(anno, detect, height, width, lane_width, threshold, similarity=lane_similarity):
    """counter.cpp:83-141 -> (anno_match, tp, fp, tn, fn, iou)."""
    anno_match = [-1] * len(anno)
    if not anno and not detect:
        return anno_match, 0, 0, 0, 0, 1.0
    if not anno:
        return anno_match, 0, len(detect), 0, 0, 0.0
    if not detect:
        return anno_match, 0, 0, 0, len(anno), 0.0
    sim = [[similarity(a, d, height, width, lane_width) for d in detect] for a in anno]
    anno_match, _ = make_match(sim)
    anno_match = list(anno_match)
    tp, iou = 0, 0.0
    for i in range(len(anno)):
        if anno_match[i] >= 0:
            iou += sim[i][anno_match[i]]
        if anno_match[i] >= 0 and sim[i][anno_match[i]] > threshold:
            tp += 1
        else:
            anno_match[i] = -1
    return anno_match, tp, len(detect) - tp, 0, len(anno) - tp, iou / len(detect)


# ---------------------------------------------------------------------------------------------- driver
def evaluate(anno_dir, detect_dir, names, width=1920, height=1080, lane_width=10, threshold=0.4, output_file=None):
    """evaluate.cpp:42-233 for the images listed in `names` (the lines of the -l list file) -> dict of the printed numbers."""
    tp = fp = fn = 0
    iou = 0.0
    for name in names:
        stem = (name[:name.rfind(".")] if "." in name else name) + ".lines.txt"        # substr(0, find_last_of("."))
        anno = read_lane_file(anno_dir + stem)
        det = read_lane_file(detect_dir + stem)
        _, a, b, _, c, d = count_im_pair(anno, det, height, width, lane_width, threshold)
        tp += a; fp += b; fn += c; iou += d
    return summarize(tp, fp, fn, iou, len(names), output_file)


def summarize(tp, fp, fn, iou, n_images, output_file=None):
    """evaluate.cpp:192-230 (Counter::get_precision / get_recall: -1 when undefined)."""
    with np.errstate(all="ignore"):
        miou = float(np.float64(iou) / np.float64(n_images))
        precision = -1.0 if tp + fp == 0 else tp / float(tp + fp)
        recall = -1.0 if tp + fn == 0 else tp / float(tp + fn)
        f = float(np.float64(2 * precision * recall) / np.float64(precision + recall))
    res = dict(tp=tp, fp=fp, fn=fn, precision=precision, recall=recall, miou=miou, Fmeasure=f)
    if output_file:
        with open(output_file, "w") as fh:
            fh.write(f"file: {output_file}\ntp: {tp} fp: {fp} fn: {fn}\nprecision: {_g(precision)}\nrecall: {_g(recall)}\n"
                     f"miou: {_g(miou)}\nFmeasure: {_g(f)}\n\n")
    return res


def _g(v):
    """operator<<(double) of an ofstream at default precision = printf("%g")."""
    return "nan" if v != v else "%g" % v
