"""Philox4x32-10 (oracle/philox_ref.py) against the Random123 known-answer vectors, and the
mask stream's basic properties."""
import numpy as np

from oracle.philox_ref import philox4x32_10, site_mask


def test_random123_kats():
    f = 0xFFFFFFFF
    assert [int(v) for v in philox4x32_10(0, 0, 0, 0, 0, 0)] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert [int(v) for v in philox4x32_10(f, f, f, f, f, f)] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    got = philox4x32_10(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)
    assert [int(v) for v in got] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_mask_stream_properties():
    m = site_mask(12345, 7, 0.05, 640, 96)
    assert m.dtype == np.float32 and m.shape == (640, 96)
    vals = np.unique(m)
    assert set(vals.tolist()) <= {0.0, float(np.float32(1) / (np.float32(1) - np.float32(0.05)))}
    assert abs((m == 0).mean() - 0.05) < 0.01
    # rows (samples) and sites are independent streams; the same (seed, site) repeats exactly
    assert not np.array_equal(m[0], m[1])
    assert not np.array_equal(m, site_mask(12345, 8, 0.05, 640, 96))
    np.testing.assert_array_equal(m, site_mask(12345, 7, 0.05, 640, 96))
    # chunking invariance: the first rows of a longer stream are the shorter stream
    np.testing.assert_array_equal(m[:100], site_mask(12345, 7, 0.05, 100, 96))
    assert np.all(site_mask(1, 0, 0.0, 16, 8) == 1.0)
