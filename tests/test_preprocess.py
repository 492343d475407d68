"""Input pre-processing: the numpy oracle against closed-form cases on CPU, the HIP kernel
bit-exactly against the oracle on the GPU.  Parity against the reference's own cv2 / imgaug resize is UNPINNED (those
packages are not installed in this image and the reference ships no image fixture)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_cpu as P

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def test_oracle_taps_sum_to_one_and_identity_resize_is_exact():
    idx, coef = P.axis_table(37, 91)
    sums = coef.astype(np.int32).sum(1)                    # each tap is saturate_cast<short>(c * 2048) on its own: no renormalisation
    assert sums.min() >= 2047 and sums.max() <= 2049 and idx.min() == 0 and idx.max() == 90
    # the source coordinate is rounded to float BEFORE its floor (OpenCV: fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx))
    f32 = ((np.arange(37, dtype=np.float64) + 0.5) * (91 / 37) - 0.5).astype(np.float32)
    assert np.array_equal(idx[:, 1], np.clip(np.floor(f32).astype(np.int64), 0, 90))
    from phnet_amd.libs.dataset.openlane.preprocess import _axis_table
    hidx, hcoef = _axis_table(37, 91)                        # the host-side table builder of the product restates the same rule
    assert np.array_equal(hidx, idx) and np.array_equal(hcoef, coef)
    r = np.random.default_rng(0)
    img = r.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_cubic_u8(img, 23, 31), img)                      # fx = 0: taps (0, 2048, 0, 0)
    flat = np.full((40, 50, 3), 137, np.uint8)
    assert (P.resize_cubic_u8(flat, 17, 29) == 137).all()                           # partition of unity, both passes
    ramp = np.broadcast_to((np.arange(64, dtype=np.float32) * 3)[None, :, None], (8, 64, 3)).astype(np.uint8)
    half = P.resize_cubic_u8(np.ascontiguousarray(ramp), 8, 32)
    want = (np.arange(32) * 2 + 0.5) * 3                                            # a cubic kernel reproduces a linear ramp
    assert np.abs(half[4, 2:-2, 0].astype(np.float64) - want[2:-2]).max() <= 1.0


def test_oracle_pipeline_crop_flip_normalise():
    r = np.random.default_rng(1)
    frames = r.integers(0, 256, (2, 40, 48, 3), dtype=np.uint8)
    x, u8 = P.preprocess_clip(frames, 8, 32, 48, MEAN, STD)
    assert x.shape == (2, 3, 32, 48) and u8.shape == (2, 32, 48, 3)
    assert np.array_equal(u8[0], frames[0, 8:])                                     # same size after the crop: identity resize
    want = (frames[:, 8:].astype(np.float32).transpose(0, 3, 1, 2) / 255 - np.asarray(MEAN, np.float32).reshape(1, 3, 1, 1)) / np.asarray(STD, np.float32).reshape(1, 3, 1, 1)
    assert np.abs(x - want).max() < 1e-6
    xf, u8f = P.preprocess_clip(frames, 8, 32, 48, MEAN, STD, flip=True)
    assert np.array_equal(u8f, u8[:, :, ::-1])


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(1280, 1920, 480, 320, 800), (1280, 1920, 480, 384, 768), (97, 131, 13, 64, 160), (64, 160, 0, 64, 160)])
@pytest.mark.parametrize("flip", [False, True])
def test_hip_preprocess_is_bit_exact_vs_oracle(geom, flip):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from phnet_amd.libs.dataset.openlane.preprocess import ClipPreprocessor
    h0, w0, crop, oh, ow = geom
    T = 3
    r = np.random.default_rng(h0 + ow)
    frames = r.integers(0, 256, (T, h0, w0, 3), dtype=np.uint8)
    frames[0, crop:crop + 5] = 255; frames[1, -7:] = 0                                # saturation / overshoot at hard edges
    want, want_u8 = P.preprocess_clip(frames, crop, oh, ow, MEAN, STD, flip=flip)
    pre = ClipPreprocessor(oh, ow, src_h=h0, src_w=w0, crop_size=crop, mean=MEAN, std=STD)
    dev = torch.from_numpy(frames).cuda()
    got, got_u8 = pre(dev, flip=flip, return_u8=True)
    assert np.array_equal(got_u8.cpu().numpy(), want_u8)                              # the resampled 8-bit image: exact
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-6                             # (q/255 - mean)/std in f32
    nhwc = pre(dev, flip=flip, layout="nhwc4")
    assert torch.equal(nhwc[..., :3].permute(0, 3, 1, 2), got) and float(nhwc[..., 3].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        pre(torch.from_numpy(frames))                                                 # no CPU path


@pytest.mark.gpu
def test_preprocessed_clip_feeds_the_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from phnet_amd.config import make_cfg
    from phnet_amd.libs.dataset.openlane.preprocess import ClipPreprocessor
    from phnet_amd.libs.models.Router4OL import RouterOL
    cfg = make_cfg(img_h=64, img_w=160, arch="resnet18")
    model = RouterOL(cfg, None).cuda().eval()
    r = np.random.default_rng(3)
    raw = torch.from_numpy(r.integers(0, 256, (2, 128, 320, 3), dtype=np.uint8)).cuda()
    clip = ClipPreprocessor(64, 160, src_h=128, src_w=320, crop_size=48)(raw)
    with torch.no_grad():
        out = model({"frame": clip, "lanes": None})
    assert len(out["lane_lines"]) == 2
