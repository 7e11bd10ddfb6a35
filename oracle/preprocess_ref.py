"""ORACLE (test infrastructure): input preprocessing on the CPU.

Restates `InputProcessor.normalize_image / set_scale_factors_to_output_size /
resize_and_crop_image` (src/dataloader.py:69-75,123-152) as called from
`EfficientDetModel._preprocessing` (src/efficientdet_keras.py:1076-1100):

  image = (float32(uint8) - mean_rgb) / stddev_rgb
  scale = min(H_out / h, W_out / w)           (float32 arithmetic)
  scaled size = (int(h * scale), int(w * scale))
  bilinear resize, half-pixel centres, no antialias (tf.image.resize v2):
      src = (dst + 0.5) * (in / out) - 0.5;  lo = max(floor(src), 0);  hi = min(ceil(src), in-1)
      lerp = src - floor(src);  value = top + (bottom - top) * ly,  top = tl + (tr - tl) * lx
  crop to the output size (offset 0), zero-pad bottom/right to (H_out, W_out)
  returned image scale = 1 / scale

TF is not installable here: **parity unpinned** (TF op semantics restated from its docs/kernels).
"""
import numpy as np


def _interp_axis(out_size, in_size):
    scale = np.float32(in_size) / np.float32(out_size)
    src = (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(src)
    lo = np.maximum(fl, 0).astype(np.int64)
    hi = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
    return lo, hi, (src - fl).astype(np.float32)


def resize_bilinear(img, oh, ow):
    """img float32 [h, w, c] -> [oh, ow, c]."""
    h, w = img.shape[:2]
    if (oh, ow) == (h, w):
        return img.copy()
    ylo, yhi, ly = _interp_axis(oh, h)
    xlo, xhi, lx = _interp_axis(ow, w)
    lx = lx[None, :, None]
    ly = ly[:, None, None]
    tl, tr = img[ylo][:, xlo], img[ylo][:, xhi]
    bl, br = img[yhi][:, xlo], img[yhi][:, xhi]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    return (top + (bot - top) * ly).astype(np.float32)


def preprocess(images, image_size_hw, mean_rgb, stddev_rgb):
    """uint8 [N,h,w,3] (or list of [h,w,3]) -> (float32 [N,H,W,3], float32 scales [N])."""
    H, W = image_size_hw
    mean = np.asarray(mean_rgb, dtype=np.float32).reshape(1, 1, 3)
    std = np.asarray(stddev_rgb, dtype=np.float32).reshape(1, 1, 3)
    outs, scales = [], []
    for im in images:
        im = (np.asarray(im).astype(np.float32) - mean) / std
        h, w = im.shape[:2]
        sy = np.float32(H) / np.float32(h)
        sx = np.float32(W) / np.float32(w)
        s = np.minimum(sx, sy)
        sh, sw = int(np.float32(h) * s), int(np.float32(w) * s)
        scaled = resize_bilinear(im, sh, sw)[:H, :W]
        out = np.zeros((H, W, 3), dtype=np.float32)
        out[:scaled.shape[0], :scaled.shape[1]] = scaled
        outs.append(out)
        scales.append(np.float32(1.0) / s)
    return np.stack(outs), np.asarray(scales, dtype=np.float32)
