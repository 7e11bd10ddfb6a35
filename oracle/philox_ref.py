"""ORACLE (test infrastructure): Philox4x32-10 and the MC-dropout mask stream.

TF's stateful RNG stream cannot be reproduced (SURVEY §7 'hard parts'), so the
build defines its own counter-based stream and the oracle and the HIP path
share the definition:

    site s (index in `effdet_ref.dropout_sites`), sample row b = n*T + t, channel c
    counter = (c >> 2, b, s, 0)    key = (seed & 0xffffffff, seed >> 32)
    word    = philox4x32_10(counter, key)[c & 3]
    u       = (word >> 8) * 2**-24                      in [0, 1)
    keep    = u >= rate          (SpatialDropout2D keeps iff uniform >= rate)
    scale   = keep ? 1/(1-rate) : 0        (float32)

Known-answer vectors for the generator itself are the Random123 ones
(tests/test_philox.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def site_mask(seed, site, rate, n_rows, channels, row_base=0):
    """float32 [n_rows, channels] keep-scales of one dropout site (rows row_base .. row_base+n_rows)."""
    b = (np.arange(n_rows, dtype=np.uint32) + np.uint32(row_base))[:, None]
    c = np.arange(channels, dtype=np.uint32)[None, :]
    out = philox4x32_10(c >> np.uint32(2), b, np.uint32(site), np.uint32(0),
                        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    sel = (c & np.uint32(3)) + np.zeros_like(b)
    word = np.choose(sel, out)
    u = (word >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    keep = u >= np.float32(rate)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    return np.where(keep, scale, np.float32(0.0)).astype(np.float32)


def make_masks(sites, seed, N, T, first_image=0):
    """{site name: float32 [N, T, C]} for `effdet_ref.forward`; `first_image` = index of image 0
    in the global batch (image shards)."""
    return {name: site_mask(seed, s, rate, N * T, ch, first_image * T).reshape(N, T, ch)
            for s, (name, ch, rate) in enumerate(sites)}


def normal2(seed, i0, i1, i2, tag):
    """Both Box-Muller values of Philox4x32-10(counter = (i0, i1, i2, tag), key = seed), float64, vectorised
    (csrc/uda_internal.h philox_normal2): u1 = ((w0 >> 8) + 0.5) 2^-24, u2 = (w1 >> 8) 2^-24,
    z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = ... sin(2 pi u2)."""
    w = philox4x32_10(i0, i1, i2, np.uint32(tag), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = ((w[0] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u2 = (w[1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    r, th = np.sqrt(-2.0 * np.log(u1)), 2.0 * np.pi * u2
    return r * np.cos(th), r * np.sin(th)
