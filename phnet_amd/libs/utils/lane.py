"""Result object returned by eval forward (reference: libs/utils/lane.py:4-48): a polyline in normalised image
coordinates with a spline through it."""
import numpy as np
from scipy.interpolate import InterpolatedUnivariateSpline


class Lane:
    def __init__(self, points=None, invalid_value=-2., metadata=None):
        self.curr_iter = 0
        self.points = points
        self.invalid_value = invalid_value
        self.function = InterpolatedUnivariateSpline(points[:, 1], points[:, 0], k=min(3, len(points) - 1))
        self.min_y = points[:, 1].min() - 0.01
        self.max_y = points[:, 1].max() + 0.01
        self.metadata = metadata or {}

    def __repr__(self):
        return "[Lane]\n" + str(self.points) + "\n[/Lane]"

    def __call__(self, lane_ys):
        xs = self.function(lane_ys)
        xs[(lane_ys < self.min_y) | (lane_ys > self.max_y)] = self.invalid_value
        return xs

    def to_array(self, cfg):
        ys = np.array(cfg.sample_y) / float(cfg.ori_img_h)
        xs = self(ys)
        ok = (xs >= 0) & (xs < 1)
        return np.stack([xs[ok] * cfg.ori_img_w, ys[ok] * cfg.ori_img_h], axis=1)

    def __iter__(self):
        return self

    def __next__(self):
        if self.curr_iter < len(self.points):
            self.curr_iter += 1
            return self.points[self.curr_iter - 1]
        self.curr_iter = 0
        raise StopIteration
