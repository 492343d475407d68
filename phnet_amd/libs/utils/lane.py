"""Result object handed back by the eval forward: one detected lane as a polyline in normalised image coordinates
(x, y in [0, 1]) plus an interpolating spline x(y).  Same public surface as the reference's `libs.utils.lane.Lane`
(`points`, `metadata`, calling the object with an array of y's, `to_array(cfg)`, iteration over the points), so the
reference's evaluation writers (evaluation/generate_lane.py) keep working."""
from typing import Dict, Optional

import numpy as np
from scipy.interpolate import InterpolatedUnivariateSpline

_Y_MARGIN = 0.01


class Lane:
    def __init__(self, points: Optional[np.ndarray] = None, invalid_value: float = -2.0, metadata: Optional[Dict] = None):
        self.points = points
        self.invalid_value = invalid_value
        self.metadata = {} if metadata is None else metadata
        xs, ys = points[:, 0], points[:, 1]
        order = min(3, len(points) - 1)                       # cubic where there are enough points
        self.function = InterpolatedUnivariateSpline(ys, xs, k=order)
        self.min_y, self.max_y = float(ys.min()) - _Y_MARGIN, float(ys.max()) + _Y_MARGIN
        self.curr_iter = 0

    def __call__(self, lane_ys: np.ndarray) -> np.ndarray:
        """x at the given y's; `invalid_value` outside the lane's own y-range."""
        xs = self.function(lane_ys)
        outside = np.logical_or(lane_ys < self.min_y, lane_ys > self.max_y)
        xs[outside] = self.invalid_value
        return xs

    def to_array(self, cfg) -> np.ndarray:
        """Pixel-space [n,2] samples at cfg.sample_y (rows of the original image)."""
        ys = np.asarray(cfg.sample_y, dtype=np.float64) / float(cfg.ori_img_h)
        xs = self(ys)
        inside = np.logical_and(xs >= 0, xs < 1)
        return np.column_stack((xs[inside] * cfg.ori_img_w, ys[inside] * cfg.ori_img_h))

    def __iter__(self):
        return self

    def __next__(self):
        if self.curr_iter >= len(self.points):
            self.curr_iter = 0
            raise StopIteration
        p = self.points[self.curr_iter]
        self.curr_iter += 1
        return p

    def __repr__(self) -> str:
        return f"[Lane]\n{self.points}\n[/Lane]"
