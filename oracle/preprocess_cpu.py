"""CPU oracle of the input pre-processing  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates the image half of the reference's per-frame pipeline:
  libs/dataset/openlane/datasetOL.py:40-52   cropping(): drop the top `crop_size` rows, optional left-right flip
  libs/dataset/openlane/transforms.py:150-156  iaa.Resize(size=(img_h, img_w)) - imgaug's default interpolation "cubic",
                                               i.e. cv2.resize(..., interpolation=cv2.INTER_CUBIC) on the uint8 image
  libs/dataset/openlane/datasetOL.py:63-75   ToTensor (uint8 HWC -> float32 CHW / 255) and Normalize(mean, std)
  libs/dataset/openlane/datasetOL.py:11-17   multibatch_collate_fn: stack the frames of a clip

**PARITY UNPINNED**: cv2 / imgaug are not installed in this image and the reference ships no image fixtures, so nothing
executed from the reference can confirm this file.  The bicubic resampling restates OpenCV's published algorithm for 8-bit
images (the 4.x sources, modules/imgproc/src/resize.cpp: the generic path of cv::resize - half-pixel centres with the
source coordinate rounded to float before its floor, `fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx`;
interpolateCubic with A = -0.75; every tap stored on its own as saturate_cast<short>(c * 2048), no renormalisation;
replicated borders; HResizeCubic in 32-bit integers, VResizeCubic with FixedPtCast<int, uchar, 22>: (v + 2^21) >> 22,
saturated) as recalled from that source; the SIMD paths of a real cv2 build are documented to give the same integers, but
an off-by-one-LSB difference against one cannot be excluded.
Only tests/ may import this.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def cubic_coeffs(fx: np.ndarray) -> np.ndarray:
    """[n,4] float32 cubic-convolution weights (a = -0.75) for fractional offsets fx in [0,1)."""
    a = np.float32(-0.75)
    fx = fx.astype(np.float32)
    c0 = ((a * (fx + 1) - 5 * a) * (fx + 1) + 8 * a) * (fx + 1) - 4 * a
    c1 = ((a + 2) * fx - (a + 3)) * fx * fx + 1
    c2 = ((a + 2) * (1 - fx) - (a + 3)) * (1 - fx) * (1 - fx) + 1
    c3 = np.float32(1.0) - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=1).astype(np.float32)


def fixed_coeffs(fx: np.ndarray) -> np.ndarray:
    """[n,4] int16 coefficients: saturate_cast<short>(c * 2048) per tap (round half to even, as cvRound), NOT renormalised."""
    c = cubic_coeffs(fx)
    q = np.clip(np.rint(c * np.float32(COEF_SCALE)), -32768, 32767)
    return q.astype(np.int16)


def axis_table(n_dst: int, n_src: int):
    """(idx [n_dst,4] int32 clamped source indices, coef [n_dst,4] int16)."""
    scale = np.float64(n_src) / n_dst
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)     # (float)((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f)                                                                        # cvFloor of the FLOAT coordinate
    fx = (f - s).astype(np.float32)
    idx = np.clip(s.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1).astype(np.int32)
    return idx, fixed_coeffs(fx)


def resize_cubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C]."""
    h, w, _ = img.shape
    xi, xc = axis_table(out_w, w)
    yi, yc = axis_table(out_h, h)
    src = img.astype(np.int32)
    rows = (src[:, xi, :] * xc[None, :, :, None].astype(np.int32)).sum(axis=2)            # [H,out_w,C] int32, scale 2^11
    out = (rows[yi, :, :] * yc[:, :, None, None].astype(np.int32)).sum(axis=1)            # [out_h,out_w,C], scale 2^22
    out = (out + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def preprocess_clip(frames_u8: np.ndarray, crop_top: int, out_h: int, out_w: int, mean, std, flip: bool = False):
    """frames uint8 [T,H0,W0,3] (RGB) -> (float32 [T,3,out_h,out_w] normalised, uint8 [T,out_h,out_w,3] resized)."""
    mean = np.asarray(mean, np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(std, np.float32).reshape(1, 3, 1, 1)
    res = []
    for f in frames_u8:
        f = f[crop_top:]
        if flip:
            f = f[:, ::-1]
        res.append(resize_cubic_u8(np.ascontiguousarray(f), out_h, out_w))
    res = np.stack(res)
    x = res.astype(np.float32).transpose(0, 3, 1, 2) / np.float32(255.0)
    return ((x - mean) / std).astype(np.float32), res
