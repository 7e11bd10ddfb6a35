"""Op-level GPU parity of the pointwise (1x1) convolution kernels, through the C ABI (uda_debug_pw),
against a float64 numpy restatement of the same op (reference call sites:
backbone/efficientnet_model.py:358-373,403-418,471-486; efficientdet_keras.py:207-227).

Tolerances (relative to max|ref| of the op's output, stated per variant):
  f32-input MFMA (terms 0)      2e-6   (float32 summation order)
  split-bf16, 6 cross terms     2e-6   (float32-equivalent)
  split-fp16, 3 cross terms     2e-6   (terms = 16: two fp16 pieces, ~2^-22 per product - the shipped default; SAME bar as six terms)
  split-bf16, 3 cross terms     4e-5   (~2^-17 per product; the network-level bar stays 2e-4)
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {0: 2e-6, 6: 2e-6, 16: 2e-6, 3: 4e-5}


def _ref(x, w, bias, sc, sh, se, mask, res, in_div, act):
    rows = mask.shape[0] if mask is not None else x.shape[0] * in_div
    xi = np.repeat(x.astype(np.float64), in_div, axis=0)
    if se is not None:
        xi = xi * np.repeat(se.astype(np.float64), in_div, axis=0)[:, None, :]
    y = xi @ w.astype(np.float64)
    if bias is not None:
        y = y + bias
    if sc is not None:
        y = y * sc + sh
    if act:
        y = y / (1.0 + np.exp(-y))
    if mask is not None:
        y = y * mask[:, None, :]
    if res is not None:
        y = y + res
    assert y.shape[0] == rows
    return y


def _run(x, w, bias, sc, sh, se, mask, res, in_div, act, terms, reps=0, expect_rc=None):
    from uda_amd import capi
    lib = capi.load()
    rows = x.shape[0] * in_div
    hw, cin = x.shape[1], x.shape[2]
    cout = w.shape[1]
    out = np.empty((rows, hw, cout), np.float32)
    ms = C.c_float(0)
    arrs = [np.ascontiguousarray(a, np.float32) if a is not None else None for a in (x, w, bias, sc, sh, se, mask, res)]
    ptr = [a.ctypes.data if a is not None else None for a in arrs]
    rc = lib.uda_debug_pw(0, *ptr, rows, in_div, hw, cin, cout, act, terms, reps, out.ctypes.data, C.byref(ms))
    if expect_rc is not None:
        return rc, (lib.uda_last_error(None) or b"").decode()
    assert rc == 0, lib.uda_last_error(None)
    return out, ms.value


CASES = [
    # rows_in, in_div, hw, cin, cout, flags
    (2, 1, 300, 32, 16, "bn"),
    (2, 1, 513, 96, 24, "bn,se,res"),
    (1, 3, 200, 16, 96, "bn,act,mask"),
    (2, 1, 130, 112, 672, "bn,act,mask"),
    (2, 1, 129, 672, 112, "bn,se,res"),
    (1, 1, 96, 1152, 320, "bn,se"),
    (2, 1, 257, 64, 64, "bias,bn"),
    (2, 2, 100, 64, 63, "bias"),
    (2, 1, 100, 64, 72, "bias"),
    (1, 1, 77, 24, 144, "bn,act"),
    (1, 1, 64, 40, 240, "bn,act,mask"),
    (3, 1, 31, 88, 528, "bn,act"),
    (1, 1, 40, 208, 1248, "bn,act,mask"),
    (1, 1, 1, 8, 4, ""),
    (2, 5, 300, 32, 16, "bn,se"),            # shared-input kernel: one tile load for the 5 sample rows of an image
    (1, 4, 131, 16, 24, "bn,res,mask"),
]


@pytest.mark.parametrize("terms", [0, 3, 6, 16])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_hw%d_%d-%d_%s" % c)
def test_pointwise_matches_float64(case, terms):
    rows_in, in_div, hw, cin, cout, flags = case
    f = set(flags.split(",")) if flags else set()
    rng = np.random.default_rng(hash(case) & 0xFFFF)
    rows = rows_in * in_div
    x = rng.normal(0, 1, (rows_in, hw, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32) if "bias" in f else None
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32) if "bn" in f else None
    sh = rng.normal(0, 0.3, cout).astype(np.float32) if "bn" in f else None
    se = rng.uniform(0.1, 1.0, (rows_in, cin)).astype(np.float32) if "se" in f else None
    mask = (rng.uniform(0, 1, (rows, cout)) >= 0.1).astype(np.float32) / 0.9 if "mask" in f else None
    mask = mask.astype(np.float32) if mask is not None else None
    res = rng.normal(0, 1, (rows, hw, cout)).astype(np.float32) if "res" in f else None
    got, _ = _run(x, w, bias, sc, sh, se, mask, res, in_div, int("act" in f), terms)
    want = _ref(x, w, bias, sc, sh, se, mask, res, in_div, "act" in f)
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= TOL[terms] * scale + 1e-7, (err, scale, err / scale)


def test_pointwise_a_identity_asymmetric_b():
    """A = I with an asymmetric integer B: catches a transposed operand or output map exactly."""
    cin = cout = 64
    x = np.eye(64, dtype=np.float32)[None]                       # [1, 64 pixels, 64 channels]
    w = (np.arange(64)[:, None] * 3 + np.arange(64)[None, :] * 7 % 11).astype(np.float32)
    for terms in (0, 3, 6, 16):
        got, _ = _run(x, w, None, None, None, None, None, None, 1, 0, terms)
        np.testing.assert_array_equal(got[0], w)


@pytest.mark.parametrize("w_std,x_std", [(0.3, 1.0), (0.04, 1.0), (0.04, 0.05), (0.005, 1.0), (1e-4, 30.0), (3.0, 300.0)])
def test_fp16_pieces_hold_the_float32_bar_across_operand_magnitudes(w_std, x_std):
    """Two fp16 pieces resolve an operand to 2^-22 only while its low piece is a normal fp16 number (|x| >= 2^-3); below,
    the resolution is 2^-25 ABSOLUTE.  The 1x1 / separable kernels therefore pre-scale the weights by a power of two on
    the host (largest entry in [2^13, 2^14), undone exactly in the epilogue), which makes the result independent of the
    weights' magnitude; what is left is the activations' absolute 2^-25: the six-term bar (2e-6 of max|ref|) holds for
    swish-shaped activations from 0.05 to 300 rms and weights from 1e-4 to 3 rms, K = 1152."""
    rng = np.random.default_rng(11)
    hw, cin, cout = 96, 1152, 320
    pre = rng.normal(0, x_std, (1, hw, cin))
    x = (pre / (1.0 + np.exp(-pre))).astype(np.float32)
    w = rng.normal(0, w_std, (cin, cout)).astype(np.float32)
    got, _ = _run(x, w, None, None, None, None, None, None, 1, 0, 16)
    want = _ref(x, w, None, None, None, None, None, None, 1, False)
    six, _ = _run(x, w, None, None, None, None, None, None, 1, 0, 6)
    scale = np.abs(want).max()
    err, err6 = np.abs(got - want).max(), np.abs(six - want).max()
    assert err <= TOL[16] * scale, (err / scale, err6 / scale)


def test_fp16_pieces_report_an_operand_above_65504():
    """fp16 ends at 65504: an activation above it cannot be split.  The kernels track the largest operand they split and
    raise the launch's range flag - the call fails with a message, it does not return infinities (or a flushed value)."""
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, (1, 200, 64)).astype(np.float32)
    w = (rng.normal(0, 1, (64, 32)) / 8).astype(np.float32)
    rc, msg = _run(x, w, None, None, None, None, None, None, 1, 0, 16, expect_rc=True)
    assert rc == 0
    x[0, 137, 5] = 7.0e4
    rc, msg = _run(x, w, None, None, None, None, None, None, 1, 0, 16, expect_rc=True)
    assert rc != 0 and "65504" in msg, (rc, msg)
    got, _ = _run(x, w, None, None, None, None, None, None, 1, 0, 6)          # three bf16 pieces: float32's exponent range
    want = _ref(x, w, None, None, None, None, None, None, 1, False)
    assert np.abs(got - want).max() <= TOL[6] * np.abs(want).max()


# ---- few pixels x many input channels: pws_kernel (one wave per tile, operands straight into registers) must be the SAME
# function as pwb_kernel bit for bit - a serve of one image equals the same image inside a batch of 32 because of it
SKINNY_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import test_gpu_ops as T
out = {}
for i, (rows_in, in_div, hw, cin, cout, flags, terms) in enumerate(%(cases)r):
    f = set(flags.split(",")) if flags else set()
    rng = np.random.default_rng(100 + i)
    rows = rows_in * in_div
    x = rng.normal(0, 1, (rows_in, hw, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (cin, cout)) / np.sqrt(cin)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32) if "bias" in f else None
    sc = rng.uniform(0.5, 1.5, cout).astype(np.float32) if "bn" in f else None
    sh = rng.normal(0, 0.3, cout).astype(np.float32) if "bn" in f else None
    se = rng.uniform(0.1, 1.0, (rows_in, cin)).astype(np.float32) if "se" in f else None
    mask = ((rng.uniform(0, 1, (rows, cout)) >= 0.1) / 0.9).astype(np.float32) if "mask" in f else None
    res = rng.normal(0, 1, (rows, hw, cout)).astype(np.float32) if "res" in f else None
    got, ms = T._run(x, w, bias, sc, sh, se, mask, res, in_div, int("act" in f), terms, reps=20)
    out["y%%d" %% i] = got
    out["ms%%d" %% i] = np.float32(ms)
np.savez(%(dst)r, **out)
"""

SKINNY_CASES = [
    (1, 1, 960, 1152, 192, "bn,se,res", 16),      # blocks 12-14 projection of ONE image (24 x 40), head-only MC: rows = 1
    (1, 1, 960, 1152, 320, "bn,se", 16),          # block 15
    (1, 1, 960, 672, 192, "bn,se", 16),           # block 11: 42 k-steps = 10 groups + 2
    (1, 1, 3840, 672, 112, "bn,se,res", 16),      # blocks 9-10 (48 x 80): two column tiles per wave
    (1, 1, 3840, 480, 80, "bn,se,res,mask", 16),
    (2, 2, 333, 288, 63, "bias,act,mask", 16),    # ragged rows / columns, shared input rows
    (1, 1, 960, 1152, 192, "bn,se,res", 6),
    (1, 1, 960, 1152, 192, "bn,se,res", 3),
    (1, 3, 100, 256, 40, "bn,res", 16),           # KS = 16: four whole groups, none left
    (1, 1, 64, 272, 24, "bn", 16),                # KS = 17: four groups + 1
]


def test_skinny_pointwise_kernel_is_bit_identical_to_the_tiled_one(tmp_path):
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("60000", "0"):
        dst = str(tmp_path / ("pw_%s.npz" % mode))
        env = dict(os.environ, UDA_PW_SKINNY=mode)
        r = subprocess.run([sys.executable, "-c", SKINNY_WORKER % dict(root=root, cases=SKINNY_CASES, dst=dst)], env=env, cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = np.load(dst)
    for i, c in enumerate(SKINNY_CASES):
        a, b = res["60000"]["y%d" % i], res["0"]["y%d" % i]
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b, err_msg=str(c))
        print("case %s: skinny %.1f us, tiled %.1f us" % (c, 1e3 * float(res["60000"]["ms%d" % i]), 1e3 * float(res["0"]["ms%d" % i])))
