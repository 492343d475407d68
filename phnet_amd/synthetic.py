"""Synthetic clips / lane labels in the reference's input contract: frames float32 N(0,1)
[T,3,H,W]; labels [T,4,6+S] in the layout of libs/dataset/openlane/transforms.py:264-297
(neg flag, pos flag, start_y, start_x/(W-1), theta, len/n_strips, S x-coordinates in pixels, -1e5 = invalid)."""
import math

import numpy as np
import torch


def make_clip(img_h: int, img_w: int, T: int, seed: int = 3407) -> torch.Tensor:
    r = np.random.default_rng([seed, T, img_h, img_w])
    return torch.from_numpy(r.standard_normal((T, 3, img_h, img_w), dtype=np.float32))


def make_targets(img_h: int, img_w: int, T: int, num_points: int = 36, n_lanes: int = 3, max_lanes: int = 4) -> torch.Tensor:
    S, W, H = num_points, img_w, img_h
    strip = H / (S - 1)
    out = np.full((T, max_lanes, 6 + S), -1e5, dtype=np.float32)
    out[:, :, 0], out[:, :, 1] = 1, 0
    x0s, slopes = (0.2, 0.45, 0.7, 0.85), (4.0, 0.5, -4.0, -6.0)
    for t in range(T):
        for j in range(min(n_lanes, max_lanes)):
            xs = x0s[j] * W + 5.0 * t + slopes[j] * np.arange(S)
            valid = (xs >= 0) & (xs < W)
            n = int(np.argmin(valid)) if not valid.all() else S
            n = min(n, S - 4 - j)
            xs = xs[:n]
            th = [math.atan(i * strip / (xs[i] - xs[0] + 1e-5)) / math.pi for i in range(1, n)]
            th = [v if v > 0 else 1 - abs(v) for v in th]
            out[t, j, 0], out[t, j, 1], out[t, j, 2] = 0, 1, 0.0
            out[t, j, 3] = xs[0] / (W - 1)
            out[t, j, 4] = sum(th) / len(th)
            out[t, j, 5] = n / (S - 1)
            out[t, j, 6:6 + n] = xs
    return torch.from_numpy(out)


def spread_scores_(model, std: float = 0.5, seed: int = 11) -> None:
    """Random-init PHNet scores every anchor at ~0.5 (the reference initialises its class heads with std 1e-3,
    Router4OL.py:96-100), so an eval clip keeps no lane, the NMS sees K = 0 and the memory holds no positive token.
    For inference benchmarks: redraw the class heads of both branches with a visible spread, so that about half of the 240
    anchors of every frame pass conf_threshold (K ~ 120 candidates into the NMS, max_lanes keepers, positives in memory)."""
    g = torch.Generator().manual_seed(seed)
    det = model.detNet
    with torch.no_grad():
        for lin in (det.cls_layers, det.cls_layers_sec):
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=g).to(lin.weight) * std)
            lin.bias.copy_(torch.randn(lin.bias.shape, generator=g).to(lin.bias) * 0.1)
