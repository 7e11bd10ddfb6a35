"""TensorFlow-checkpoint plumbing of `utils_keras.restore_ckpt` on the HIP path (src/utils_keras.py:125-235), CPU only.
No TensorFlow and no checkpoint file exist here: the bundle format is restated from its published layout and checked
against this package's own writer (format parity unpinned, see ckpt_reader.py); what IS checked independently are the
CRC-32C known answer, the LevelDB masking rule, the footer magic, and the restore rules (names, EMA shadows,
mismatch / missing handling) the reference applies."""
import os
import struct

import numpy as np
import pytest

from common import make_params
from uda_amd import ckpt_reader as CR, utils_keras, weights as W


def test_crc32c_known_answers():
    assert CR._crc32c(b"123456789") == 0xE3069283                    # the standard CRC-32C check value
    assert CR._crc32c(b"") == 0 and CR._crc32c(bytes(32)) == 0x8A9136AA
    # the library's host-side slicing-by-8 CRC (used when the library is built) and the pure-Python table CRC agree, also
    # on unaligned lengths and when continued from a running value
    data = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    assert CR._crc32c_py(b"123456789") == 0xE3069283 and CR._crc32c(data) == CR._crc32c_py(data)
    assert CR._crc32c(data[1000:], CR._crc32c(data[:1000])) == CR._crc32c_py(data)
    assert CR._masked(0) == 0xA282EAD8                               # LevelDB: rotate right by 15, add the constant


def _with_shadows(w, **changed):
    """The bundle a training run with moving_average_decay > 0 leaves: every variable (the trainable ones and the BN moving
    statistics, utils_keras.py:85-97) has its `/ExponentialMovingAverage` shadow; `changed` overrides single shadows."""
    b = dict(w)
    for k, v in w.items():
        b[k + "/ExponentialMovingAverage"] = changed.get(k, v)
    return b


def test_name_based_checkpoint_round_trip(tmp_path):
    p = make_params(loss_attenuation=True)
    w = W.init_weights(p, seed=3)
    prefix = CR.save_checkpoint(str(tmp_path / "model"), _with_shadows(w), checksum=True)     # every restored tensor is CRC-checked
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and os.path.getsize(prefix + ".data-00000-of-00001") == 2 * sum(v.nbytes for v in w.values())
    r = CR.BundleReader(prefix)
    assert set(r.entries) == set(_with_shadows(w)) and r.variable_to_shape_map()["efficientnet-b0/stem/conv2d/kernel"] == (3, 3, 3, 32)
    got = CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=False)
    assert set(got) == set(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k], err_msg=k)
    # the directory form goes through the `checkpoint` state file (tf.train.latest_checkpoint)
    assert CR.latest_checkpoint(str(tmp_path)) == prefix
    got2 = W.resolve_weights(str(tmp_path), p)
    np.testing.assert_array_equal(got2["class_net/class-predict/bias"], w["class_net/class-predict/bias"])

    class Model:
        config = type("C", (), {"as_dict": staticmethod(lambda: p)})()

        def load_weights(self, ws):
            self.ws = ws
    m = Model()
    utils_keras.restore_ckpt(m, str(tmp_path), 0.9998, skip_mismatch=False)
    np.testing.assert_array_equal(m.ws["box_net/box-0/pointwise_kernel"], w["box_net/box-0/pointwise_kernel"])
    utils_keras.restore_ckpt(m, "_")                                 # "running test: do not load any ckpt"


def test_restore_rules_ema_missing_and_mismatch(tmp_path):
    p = make_params()
    w = W.init_weights(p, seed=4)
    k = "efficientnet-b0/blocks_1/conv2d/kernel"
    bundle = dict(w)
    bundle[k + "/ExponentialMovingAverage"] = w[k] * 2                 # an EMA shadow: preferred when ema_decay > 0
    del bundle["class_net/class-predict/bias"]                          # a missing variable
    bundle["box_net/box-predict/bias"] = np.zeros(7, np.float32)        # a shape mismatch
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-3"), bundle, checksum=False)
    got = CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=True)
    np.testing.assert_array_equal(got[k], w[k] * 2)
    np.testing.assert_array_equal(CR.load_checkpoint(prefix, p, use_ema=False, skip_mismatch=True)[k], w[k])
    ref = W.init_weights(p, seed=0)                                     # skipped variables keep their initial value
    np.testing.assert_array_equal(got["class_net/class-predict/bias"], ref["class_net/class-predict/bias"])
    np.testing.assert_array_equal(got["box_net/box-predict/bias"], ref["box_net/box-predict/bias"])
    with pytest.raises((KeyError, ValueError)):
        CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=False)


def test_mis_shaped_ema_shadow_falls_back_to_the_plain_variable_under_skip_mismatch(tmp_path):
    """restore_ckpt (utils_keras.py:213-235) assigns the plain variable first and only WARNS when the ExponentialMovingAverage
    entry has the wrong shape: the variable keeps the checkpoint's plain value - not its random initial one."""
    p = make_params()
    w = W.init_weights(p, seed=4)
    k = "efficientnet-b0/blocks_2/conv2d/kernel"
    bundle = dict(w)
    for name in list(w):
        bundle[name + "/ExponentialMovingAverage"] = w[name] * 2
    bundle[k + "/ExponentialMovingAverage"] = np.zeros((1, 1, 3, 5), np.float32)       # a shadow of the wrong shape
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-5"), bundle, checksum=False)
    got = CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=True)
    np.testing.assert_array_equal(got[k], w[k])                         # the PLAIN tensor of the checkpoint
    other = "efficientnet-b0/blocks_3/conv2d/kernel"
    np.testing.assert_array_equal(got[other], w[other] * 2)             # everything else: its shadow
    with pytest.raises(ValueError):
        CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=False)


def test_checkpoint_without_shadows_is_refused_when_ema_is_on(tmp_path, caplog):
    """With moving_average_decay > 0 the reference's restore map holds the plain AND the shadow name of every variable and
    raises `Not found ...` for a missing shadow (utils_keras.py:176-235): a checkpoint saved without EMA must not load as if it
    had one.  skip_mismatch=True restores the plain variables and says so; use_ema=False never looks for shadows."""
    p = make_params()
    w = W.init_weights(p, seed=12)
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-9"), w, checksum=False)
    with pytest.raises(KeyError, match="ExponentialMovingAverage"):
        CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=False)
    with pytest.raises(KeyError, match="ExponentialMovingAverage"):
        W.resolve_weights(prefix, dict(p, moving_average_decay=0.9998))          # the drivers restore strictly (infer_lib.py:435)
    import logging
    with caplog.at_level(logging.WARNING):
        got = CR.load_checkpoint(prefix, p, use_ema=True, skip_mismatch=True)
    assert "no ExponentialMovingAverage shadow" in caplog.text
    k = "efficientnet-b0/blocks_1/conv2d/kernel"
    np.testing.assert_array_equal(got[k], w[k])
    np.testing.assert_array_equal(CR.load_checkpoint(prefix, p, use_ema=False, skip_mismatch=False)[k], w[k])
    np.testing.assert_array_equal(W.resolve_weights(prefix, dict(p, moving_average_decay=0))[k], w[k])


def test_tf2_object_graph_checkpoint_names(tmp_path):
    """A TF2 checkpoint stores variables under object-path keys and the variable names in the serialized
    TrackableObjectGraph (`full_name`): build such a bundle with the wire-format primitives and resolve it."""
    arrays = {"a/b/kernel": np.arange(6, dtype=np.float32).reshape(2, 3), "a/bn/gamma": np.ones(4, np.float32)}
    keys = {"a/b/kernel": "model/layer-0/kernel/.ATTRIBUTES/VARIABLE_VALUE", "a/bn/gamma": "model/layer-1/gamma/.ATTRIBUTES/VARIABLE_VALUE"}
    node = lambda full, key: CR._field(1, 2, CR._field(2, 2, CR._field(1, 2, b"VARIABLE_VALUE") + CR._field(2, 2, (full + ":0").encode())
                                                      + CR._field(3, 2, key.encode())))
    graph = b"".join(node(f, k) for f, k in keys.items())
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-1"), {keys[n]: a for n, a in arrays.items()}, checksum=True)
    # append the string tensor by hand: rewrite the bundle with the graph entry (dtype 7, scalar: varint length, 4-byte crc, bytes)
    r = CR.BundleReader(prefix)
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    payload = CR._enc_varint(len(graph)) + b"\0\0\0\0" + graph
    open(prefix + ".data-00000-of-00001", "wb").write(data + payload)
    entry = CR._field(1, 0, 7) + CR._field(2, 2, b"") + CR._field(4, 0, len(data)) + CR._field(5, 0, len(payload))
    pairs = [(b"", CR._field(1, 0, 1))] + sorted([(k.encode(), v) for k, v in
                                                  [("_CHECKPOINTABLE_OBJECT_GRAPH", entry)] + [(kk, _entry_bytes(r, kk)) for kk in r.entries]])
    _write_index(prefix, pairs)
    r2 = CR.BundleReader(prefix)
    names = r2.name_map()
    assert names == keys
    for n, a in arrays.items():
        np.testing.assert_array_equal(r2.get_tensor(names[n]), a)


def _entry_bytes(reader, key):
    e = reader.entries[key]
    shape = b"".join(CR._field(2, 2, CR._field(1, 0, int(d))) for d in e["shape"])
    out = CR._field(1, 0, e["dtype"]) + CR._field(2, 2, shape)
    if e["offset"]:
        out += CR._field(4, 0, e["offset"])
    return out + CR._field(5, 0, e["size"])


def _write_index(prefix, pairs):
    out = bytearray()

    def emit(block):
        h = CR._enc_varint(len(out)) + CR._enc_varint(len(block))
        out.extend(block + b"\x00" + struct.pack("<I", CR._masked(CR._crc32c(block + b"\x00"))))
        return h
    d = emit(CR._block(pairs))
    meta = emit(CR._block([]))
    index = emit(CR._block([(pairs[-1][0], d)]))
    footer = meta + index
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57))
    open(prefix + ".index", "wb").write(bytes(out))


def test_prefix_compressed_blocks_are_read():
    """LevelDB blocks share key prefixes between restart points; the writer here never does, so decode one by hand."""
    ent = lambda shared, suffix, val: CR._enc_varint(shared) + CR._enc_varint(len(suffix)) + CR._enc_varint(len(val)) + suffix + val
    body = ent(0, b"conv/kernel", b"A") + ent(5, b"bias", b"B") + ent(0, b"dense", b"C")
    block = body + struct.pack("<III", 0, len(body) - len(ent(0, b"dense", b"C")), 2)
    assert CR._block_entries(block) == [(b"conv/kernel", b"A"), (b"conv/bias", b"B"), (b"dense", b"C")]


def test_ema_shadow_wins_for_bn_statistics_too(tmp_path):
    """`get_ema_vars` adds the BN moving statistics (utils_keras.py:85-97) and `restore_ckpt` assigns the shadow after the
    plain name (:183-196): a bundle holding both keys for a BN statistic restores the shadow."""
    p = make_params()
    w = W.init_weights(p, seed=6)
    bundle = _with_shadows(w)
    for f in ("moving_mean", "moving_variance", "gamma"):
        k = "efficientnet-b0/stem/tpu_batch_normalization/" + f
        bundle[k + "/ExponentialMovingAverage"] = w[k] + 3
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-7"), bundle, checksum=False)
    got = CR.load_checkpoint(prefix, p, use_ema=True, verify_crc=False)
    plain = CR.load_checkpoint(prefix, p, use_ema=False, verify_crc=False)
    for f in ("moving_mean", "moving_variance", "gamma"):
        k = "efficientnet-b0/stem/tpu_batch_normalization/" + f
        np.testing.assert_array_equal(got[k], w[k] + 3)
        np.testing.assert_array_equal(plain[k], w[k])


def test_driver_restore_is_strict_and_follows_moving_average_decay(tmp_path, caplog):
    """`resolve_weights` restores as the reference's KerasDriver does (infer_lib.py:435): skip_mismatch=False, EMA shadows
    only when config.moving_average_decay > 0; an explicit skip_mismatch logs every variable it skips."""
    p = make_params()
    w = W.init_weights(p, seed=8)
    k = "efficientnet-b0/blocks_2/conv2d/kernel"
    bundle = _with_shadows(w)
    bundle[k + "/ExponentialMovingAverage"] = w[k] * 3
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-1"), bundle, checksum=False)
    np.testing.assert_array_equal(W.resolve_weights(prefix, dict(p, moving_average_decay=0.9998))[k], w[k] * 3)
    np.testing.assert_array_equal(W.resolve_weights(prefix, dict(p, moving_average_decay=0))[k], w[k])
    del bundle["class_net/class-predict/bias"]
    bundle["box_net/box-predict/bias"] = np.zeros((6, 6), np.float32)     # same element count, wrong layout: still a mismatch
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-2"), bundle, checksum=False)
    with pytest.raises((KeyError, ValueError)):
        W.resolve_weights(prefix, p)
    import logging
    with caplog.at_level(logging.WARNING):
        got = CR.load_checkpoint(prefix, p, skip_mismatch=True, verify_crc=False)
    text = caplog.text
    assert "class_net/class-predict/bias" in text and "box_net/box-predict/bias" in text
    assert got["box_net/box-predict/bias"].shape == (36,)


def test_truncated_or_corrupt_data_shard_raises(tmp_path):
    arrays = {"a/kernel": np.arange(64, dtype=np.float32), "b/kernel": np.arange(32, dtype=np.float32)}
    prefix = CR.save_checkpoint(str(tmp_path / "ckpt-5"), arrays, checksum=True)
    data = prefix + ".data-00000-of-00001"
    raw = bytearray(open(data, "rb").read())
    r = CR.BundleReader(prefix)
    np.testing.assert_array_equal(r.get_tensor("a/kernel", verify_crc=True), arrays["a/kernel"])
    raw[5] ^= 0x40                                                         # one flipped bit in a/kernel
    open(data, "wb").write(bytes(raw))
    r = CR.BundleReader(prefix)
    with pytest.raises(ValueError, match="crc32c mismatch"):
        r.get_tensor("a/kernel", verify_crc=True)
    np.testing.assert_array_equal(r.get_tensor("b/kernel", verify_crc=True), arrays["b/kernel"])
    open(data, "wb").write(bytes(raw[:-16]))                               # truncated: b/kernel's range runs past the end
    r = CR.BundleReader(prefix)
    with pytest.raises(ValueError, match="truncated"):
        r.get_tensor("b/kernel")
