"""Clip pre-processing on the GPU: decoded uint8 RGB frames -> the normalised float clip the model consumes.

Replaces the image half of the reference's per-frame CPU pipeline (libs/dataset/openlane/datasetOL.py:40-52 crop / flip,
transforms.py:150-156 iaa.Resize = cv2 INTER_CUBIC on uint8, datasetOL.py:63-75 ToTensor + Normalize, :11-17 stacking) with ONE
launch per clip (csrc/preprocess.hip).  The label half (lane resampling, transforms.py:100-147 / 264-297) is host-side
geometry on a handful of points and stays where it is.  No CPU path: the frames must already be on the device.

Arithmetic: OpenCV's 8-bit bicubic resize as published (half-pixel centres, a = -0.75 kernel, 11-bit fixed-point taps that
sum to 2048, replicated borders, rounding 22-bit shift, saturation).  PARITY UNPINNED: cv2 / imgaug are not installed here
and the reference ships no image fixture; the kernel is held bit-exactly to a numpy restatement of the same algorithm
(oracle/preprocess_cpu.py)."""
import ctypes
from typing import Sequence

import numpy as np
import torch

from phnet_amd._lib import check, lib

_COEF_BITS = 11


def _axis_table(n_dst: int, n_src: int):
    """Clamped source indices [n_dst,4] int32 and fixed-point cubic taps [n_dst,4] int16 of one axis, as OpenCV's published
    imgproc/src/resize.cpp builds them for 8-bit INTER_CUBIC (the 4.x sources: cv::resize -> the generic path's coefficient
    loop + interpolateCubic): `fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx;` - the source coordinate
    is ROUNDED TO FLOAT before its floor is taken and the fraction is a float subtraction - and every tap is stored as
    `saturate_cast<short>(c * 2048)` on its own: the four taps are NOT renormalised to sum to 2048 (they sum to 2047..2049)."""
    a = np.float32(-0.75)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (np.float64(n_src) / n_dst) - 0.5).astype(np.float32)
    s = np.floor(f)
    fx = (f - s).astype(np.float32)
    c0 = ((a * (fx + 1) - 5 * a) * (fx + 1) + 8 * a) * (fx + 1) - 4 * a
    c1 = ((a + 2) * fx - (a + 3)) * fx * fx + 1
    c2 = ((a + 2) * (1 - fx) - (a + 3)) * (1 - fx) * (1 - fx) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    q = np.clip(np.rint(np.stack([c0, c1, c2, c3], axis=1).astype(np.float32) * np.float32(1 << _COEF_BITS)), -32768, 32767).astype(np.int32)
    idx = np.clip(s.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1).astype(np.int32)
    return idx, q.astype(np.int16)


class ClipPreprocessor:
    """cfg-like arguments as in options/options4OL.py:101-108 (org 1280x1920, crop_size 480, mean / std of ImageNet)."""

    def __init__(self, out_h: int, out_w: int, src_h: int = 1280, src_w: int = 1920, crop_size: int = 480,
                 mean: Sequence[float] = (0.485, 0.456, 0.406), std: Sequence[float] = (0.229, 0.224, 0.225), device="cuda"):
        self.out_h, self.out_w, self.src_h, self.src_w, self.crop = int(out_h), int(out_w), int(src_h), int(src_w), int(crop_size)
        if not 0 <= self.crop < self.src_h:
            raise ValueError("crop_size must leave at least one row")
        xi, xc = _axis_table(self.out_w, self.src_w)
        yi, yc = _axis_table(self.out_h, self.src_h - self.crop)
        dev = torch.device(device)
        self.xi, self.xc = torch.from_numpy(xi).to(dev), torch.from_numpy(xc).to(dev)
        self.yi, self.yc = torch.from_numpy(yi).to(dev), torch.from_numpy(yc).to(dev)
        self._mean = (ctypes.c_float * 3)(*[float(v) for v in mean])
        self._std = (ctypes.c_float * 3)(*[float(v) for v in std])

    def __call__(self, frames_u8: torch.Tensor, flip: bool = False, layout: str = "nchw", return_u8: bool = False):
        """frames_u8 [T,src_h,src_w,3] uint8 on the device -> float32 [T,3,out_h,out_w] ("nchw", the reference's `img`) or
        [T,out_h,out_w,4] ("nhwc4", the stem's staging layout); with return_u8 also the resized 8-bit frames."""
        if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or not frames_u8.is_contiguous():
            raise RuntimeError("ClipPreprocessor: contiguous uint8 CUDA(HIP) frames [T,H,W,3] expected; phnet_amd has no CPU path")
        t, h, w, c = frames_u8.shape
        if (h, w, c) != (self.src_h, self.src_w, 3):
            raise ValueError(f"frames of {h}x{w}x{c}, built for {self.src_h}x{self.src_w}x3")
        if layout not in ("nchw", "nhwc4"):
            raise ValueError("layout must be 'nchw' or 'nhwc4'")
        dev = frames_u8.device
        out = torch.empty((t, 3, self.out_h, self.out_w) if layout == "nchw" else (t, self.out_h, self.out_w, 4), dtype=torch.float32, device=dev)
        u8 = torch.empty((t, self.out_h, self.out_w, 3), dtype=torch.uint8, device=dev) if return_u8 else None
        check(lib().phnet_preprocess_u8(frames_u8.data_ptr(), out.data_ptr(), None if u8 is None else u8.data_ptr(),
                                        self.xi.data_ptr(), self.xc.data_ptr(), self.yi.data_ptr(), self.yc.data_ptr(),
                                        t, h, w, self.crop, self.out_h, self.out_w, int(flip), 0 if layout == "nchw" else 1,
                                        ctypes.cast(self._mean, ctypes.c_void_p), ctypes.cast(self._std, ctypes.c_void_p),
                                        torch.cuda.current_stream().cuda_stream), "phnet_preprocess_u8")
        return (out, u8) if return_u8 else out


def multibatch_collate_fn(batch):
    """datasetOL.py:11-17 for already pre-processed samples: [([{img, lane_line}, ...], info), ...] -> (frames [B,T,3,H,W],
    lanes [B,T,4,6+S], infos)."""
    outs, infos = [s[0] for s in batch], [s[1] for s in batch]
    frames = torch.stack([torch.stack([d["img"] for d in sample]) for sample in outs])
    lanes = torch.stack([torch.stack([torch.as_tensor(d["lane_line"]) for d in sample]) for sample in outs])
    return frames, lanes, infos
