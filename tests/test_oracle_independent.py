"""The oracle's hand-rolled TF-semantics ops against INDEPENDENT implementations this container already has
(Next 6).  The reference holds no value fixtures for what TensorFlow executes and TF cannot be installed,
so this cannot pin the oracle to TF - parity stays "unpinned" - but it removes the single-author risk: every op below is
computed a second time by code that shares nothing with oracle/ (torch's own resamplers and pooling, or a literal
definition-level loop written from the op's documentation), on up- and down-scaling, odd and even sizes, every kernel
size / stride the network uses.

  bilinear resize (tf.image.resize v2, half-pixel centres, no antialias)  == F.interpolate(mode="bilinear", align_corners=False)
  nearest upsample (half_pixel_centers=False: src = floor(dst * in/out))  == F.interpolate(mode="nearest")
  SAME max-pool (padding never wins)                                      == F.max_pool2d on explicitly -inf-padded input
  SAME conv / depthwise, k in {3,5}, s in {1,2}, even and odd inputs      == a direct loop over the SAME definition
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import effdet_ref as E, preprocess_ref as PP


@pytest.mark.parametrize("h,w,oh,ow", [(375, 1242, 386, 1280), (370, 1224, 386, 1280), (37, 53, 128, 183), (100, 180, 106, 192),
                                        (300, 250, 128, 106), (97, 131, 41, 67), (5, 7, 5, 20), (64, 64, 128, 128), (9, 4, 3, 2)])
def test_bilinear_resize_equals_torch_interpolate(h, w, oh, ow):
    img = np.random.default_rng(h * 131 + w).uniform(-3, 3, (h, w, 3)).astype(np.float32)
    got = PP.resize_bilinear(img, oh, ow)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    want = F.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    # Same sampling positions and weights.  Both sides compute the source coordinate (dst + 0.5) * in/out - 0.5 in float32,
    # whose ulp at x ~ 1200 is 1.2e-4: the lerp weight of a far-right pixel differs by up to ~2e-4 between two correct
    # float32 evaluations (0.2 % of the elements, values of +-3) - a geometry error (corner alignment, missing half pixel,
    # antialias) would be O(0.1 - 1).
    np.testing.assert_allclose(got, want, rtol=0, atol=1.5e-3)
    assert np.mean(np.abs(got - want) > 2e-5) < 0.01
    # the sampling geometry itself, exactly: a ramp image is reproduced wherever no clamping happens
    ramp = np.arange(w, dtype=np.float32)[None, :, None].repeat(h, 0)
    r = PP.resize_bilinear(ramp, oh, ow)[0, :, 0]
    src = (np.arange(ow) + 0.5) * (w / ow) - 0.5
    np.testing.assert_allclose(r, np.clip(src, 0, w - 1), atol=2e-4 * max(1, w / 64))


@pytest.mark.parametrize("h,w,th,tw", [(6, 10, 12, 20), (12, 20, 24, 40), (3, 5, 6, 10), (4, 4, 8, 8), (3, 5, 7, 9), (5, 3, 13, 4)])
def test_nearest_upsample_equals_torch_interpolate(h, w, th, tw):
    x = torch.from_numpy(np.random.default_rng(h + 7 * w).normal(size=(2, 3, h, w)).astype(np.float32))
    got = E.nearest_upsample(x, th, tw)
    want = F.interpolate(x, size=(th, tw), mode="nearest")          # legacy nearest: src = floor(dst * in / out)
    assert torch.equal(got, want)


@pytest.mark.parametrize("h,w", [(96, 160), (12, 20), (5, 5), (6, 10), (3, 5), (7, 4), (1, 1)])
def test_same_max_pool_equals_torch_on_explicit_padding(h, w):
    """Pool size = stride + 1 = 3, stride 2 (efficientdet_keras.py:282-290): TF SAME pads bottom/right-heavy with -inf."""
    x = torch.from_numpy(-np.abs(np.random.default_rng(h * 17 + w).normal(size=(2, 4, h, w))).astype(np.float32))   # all negative
    got = E.max_pool_same(x, 3, 2)
    oh, ow = -(-h // 2), -(-w // 2)
    ph, pw = max((oh - 1) * 2 + 3 - h, 0), max((ow - 1) * 2 + 3 - w, 0)
    xp = torch.full((2, 4, h + ph, w + pw), float("-inf"))
    xp[:, :, ph // 2:ph // 2 + h, pw // 2:pw // 2 + w] = x             # independent placement of the padding
    want = F.max_pool2d(xp, 3, 2)
    assert got.shape == (2, 4, oh, ow) and torch.equal(got, want) and torch.isfinite(got).all()


def _same_conv_loops(x, kern, stride, depthwise):
    """Direct statement of TF's SAME convolution (tf.nn.conv2d / depthwise_conv2d docs): out = ceil(in / s); total padding
    max((out-1) s + k - in, 0), floor(half) before and the rest after; zero outside.  float64 accumulation."""
    n, c, H, W = x.shape
    k = kern.shape[0]
    oh, ow = -(-H // stride), -(-W // stride)
    pt = max((oh - 1) * stride + k - H, 0) // 2
    pl = max((ow - 1) * stride + k - W, 0) // 2
    co = c if depthwise else kern.shape[3]
    out = np.zeros((n, co, oh, ow), np.float64)
    for i in range(oh):
        for j in range(ow):
            for di in range(k):
                for dj in range(k):
                    y, xx = i * stride + di - pt, j * stride + dj - pl
                    if 0 <= y < H and 0 <= xx < W:
                        v = x[:, :, y, xx].astype(np.float64)
                        if depthwise:
                            out[:, :, i, j] += v * kern[di, dj, :, 0]
                        else:
                            out[:, :, i, j] += v @ kern[di, dj].astype(np.float64)
    return out


@pytest.mark.parametrize("k", [3, 5])
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("h,w", [(8, 10), (7, 9), (6, 5), (3, 2), (1, 4)])
def test_same_conv_and_depthwise_equal_the_definition(k, stride, h, w):
    rng = np.random.default_rng(k * 100 + stride * 10 + h)
    x = rng.normal(size=(2, 3, h, w)).astype(np.float32)
    kc = rng.normal(size=(k, k, 3, 4)).astype(np.float32)
    kd = rng.normal(size=(k, k, 3, 1)).astype(np.float32)
    got = E.conv2d(torch.from_numpy(x), kc, stride).numpy()
    want = _same_conv_loops(x, kc, stride, False)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    gd = E.depthwise(torch.from_numpy(x), kd, stride).numpy()
    wd = _same_conv_loops(x, kd, stride, True)
    assert gd.shape == wd.shape
    np.testing.assert_allclose(gd, wd, rtol=1e-5, atol=1e-5)


def test_even_input_stride2_differs_from_symmetric_padding():
    """The case PyTorch's padding=1 gets wrong for TF weights: even input, k = 3, s = 2 -> 0 before, 1 after."""
    x = np.random.default_rng(0).normal(size=(1, 1, 8, 8)).astype(np.float32)
    kern = np.random.default_rng(1).normal(size=(3, 3, 1, 1)).astype(np.float32)
    tf_same = E.conv2d(torch.from_numpy(x), kern, 2).numpy()
    sym = F.conv2d(torch.from_numpy(x), torch.from_numpy(kern).permute(3, 2, 0, 1), stride=2, padding=1).numpy()
    np.testing.assert_allclose(tf_same, _same_conv_loops(x, kern, 2, False), rtol=1e-5, atol=1e-5)
    assert np.abs(tf_same - sym).max() > 0.1
