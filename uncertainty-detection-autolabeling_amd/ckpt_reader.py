"""TensorFlow checkpoints without TensorFlow: what `utils_keras.restore_ckpt` needs on the HIP path.

The reference restores its Keras model from a TF checkpoint (src/utils_keras.py:125-235): a name-based one (TF1
style, the published EfficientDet checkpoints; keys = variable names, EMA shadows under
`<name>/ExponentialMovingAverage`, :176-235) or a TF2 object-graph one (keys `.../.ATTRIBUTES/VARIABLE_VALUE` plus the
`_CHECKPOINTABLE_OBJECT_GRAPH` entry, :148-175).  Both are "tensor bundles": `<prefix>.index`, an SSTable (LevelDB
table format: prefix-compressed key blocks, an index block, a 48-byte footer ending in the magic 0xdb4775248b80fb57)
whose values are `BundleEntryProto` messages (dtype, shape, shard, offset, size, crc32c), and
`<prefix>.data-0000N-of-0000M` holding the raw little-endian tensor bytes.  This module reads that format with numpy
and a 40-line protobuf wire decoder, maps the variables to the reference names `weights.variable_specs` lists
(through the object graph's `full_name` fields for TF2 checkpoints) and returns the weight-set dict the drivers take.
`save_checkpoint` writes a name-based bundle (the counterpart, used by the tests and to hand weights back to TF code).

Pinning: there is no TensorFlow and no checkpoint file in this environment, so the format is restated from
the published TensorBundle / LevelDB table layouts and exercised only against this module's own writer -
**format parity unpinned**.  Integrity on read: a tensor whose byte range runs past the end of its data shard raises
(truncated files never load silently); the per-tensor crc32c (masked as LevelDB does) is verified on every restore through
the library's host-side `uda_crc32c` (slicing-by-8; without the built library: bundles up to 4 MB with the pure-Python
table CRC, ~1 s per MB, and a log line says so when it is skipped).
"""
import logging
import os
import re
import struct

import numpy as np

from . import weights as weights_mod

_MAGIC = 0xDB4775248B80FB57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DT_STRING, _DT_BFLOAT16 = 7, 14


# ---------------------------------------------------------------- protobuf wire format (the three messages needed)
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """[(field number, wire type, value)]: varints as int, length-delimited as bytes, fixed32/64 as int."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _bundle_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard, offset, size)."""
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, sliced=False, crc32c=None)
    for num, _, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            for n2, _, v2 in _fields(v):
                if n2 == 2:                                   # TensorShapeProto.dim
                    e["shape"].append(next((_signed(x) for n3, _, x in _fields(v2) if n3 == 1), 0))
        elif num == 3:
            e["shard"] = v
        elif num == 4:
            e["offset"] = v
        elif num == 5:
            e["size"] = v
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


def _object_graph_names(buf):
    """TrackableObjectGraph -> {variable full_name: checkpoint_key} for every VARIABLE_VALUE attribute."""
    out = {}
    for num, _, node in _fields(buf):
        if num != 1:
            continue
        for n2, _, attr in _fields(node):
            if n2 != 2:
                continue
            f = {n3: v for n3, _, v in _fields(attr)}
            if f.get(1, b"") == b"VARIABLE_VALUE" and 2 in f and 3 in f:
                out[f[2].decode()] = f[3].decode()
    return out


# ---------------------------------------------------------------- SSTable (LevelDB table format)
def _block_entries(block):
    """(key, value) pairs of one table block (prefix-compressed keys, restart array at the end)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(path):
    """{key bytes: value bytes} of a `.index` file."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != _MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-48:]
    _, p = _varint(footer, 0)
    _, p = _varint(footer, p)                                   # metaindex handle: unused
    ioff, p = _varint(footer, p)
    isize, p = _varint(footer, p)
    out = {}
    for _, handle in _block_entries(data[ioff:ioff + isize]):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        if data[boff + bsize] != 0:
            raise ValueError("compressed table blocks (type %d) are not supported" % data[boff + bsize])
        out.update(_block_entries(data[boff:boff + bsize]))
    return out


class BundleReader:
    """Name -> tensor access to one checkpoint prefix."""

    def __init__(self, prefix):
        self.prefix = prefix
        raw = read_index(prefix + ".index")
        header = {n: v for n, _, v in _fields(raw.get(b"", b""))}
        self.num_shards = header.get(1, 1)
        if header.get(2, 0) != 0:
            raise ValueError("big-endian checkpoints are not supported")
        self.entries = {k.decode(): _bundle_entry(v) for k, v in raw.items() if k != b""}
        self._shards = {}

    def _shard(self, i):
        if i not in self._shards:
            self._shards[i] = np.memmap("%s.data-%05d-of-%05d" % (self.prefix, i, self.num_shards), dtype=np.uint8, mode="r")
        return self._shards[i]

    def variable_to_shape_map(self):
        return {k: tuple(e["shape"]) for k, e in self.entries.items()}

    def _raw(self, name):
        e = self.entries[name]
        shard = self._shard(e["shard"])
        if e["offset"] + e["size"] > shard.shape[0]:
            raise ValueError("checkpoint %s is truncated: tensor %s needs bytes [%d, %d) of a %d-byte shard"
                             % (self.prefix, name, e["offset"], e["offset"] + e["size"], shard.shape[0]))
        return shard[e["offset"]:e["offset"] + e["size"]]

    def get_bytes(self, name):
        return bytes(self._raw(name))

    def total_bytes(self):
        return sum(e["size"] for e in self.entries.values())

    def get_tensor(self, name, verify_crc=False):
        e = self.entries[name]
        if e["sliced"]:
            raise ValueError("partitioned variable %s is not supported" % name)
        raw = self._raw(name)
        if verify_crc and e["crc32c"] and e["dtype"] != _DT_STRING:      # (0 = written without a checksum: save_checkpoint(checksum=False))
            if _masked(_crc32c(raw)) != e["crc32c"]:
                raise ValueError("checkpoint %s: crc32c mismatch in tensor %s (corrupt data shard)" % (self.prefix, name))
        if e["dtype"] == _DT_STRING:
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            buf, pos, lens = bytes(raw), 0, []
            for _ in range(n):
                ln, pos = _varint(buf, pos)
                lens.append(ln)
            pos += 4                                            # masked crc32c of the lengths
            out = []
            for ln in lens:
                out.append(buf[pos:pos + ln])
                pos += ln
            return out[0] if not e["shape"] else out
        if e["dtype"] == _DT_BFLOAT16:
            return (np.frombuffer(raw, np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(e["shape"])
        if e["dtype"] not in _DTYPES:
            raise ValueError("tensor %s has unsupported dtype %d" % (name, e["dtype"]))
        return np.frombuffer(raw, _DTYPES[e["dtype"]]).reshape(e["shape"]).copy()

    def name_map(self):
        """{variable name as the model knows it: key in this bundle}.  TF2 object-graph checkpoints are resolved through
        the graph's `full_name` fields; name-based checkpoints map to themselves."""
        if "_CHECKPOINTABLE_OBJECT_GRAPH" in self.entries:
            graph = self.get_tensor("_CHECKPOINTABLE_OBJECT_GRAPH")
            return {re.sub(r":0$", "", full): key for full, key in _object_graph_names(graph).items()}
        return {k: k for k in self.entries}


# ---------------------------------------------------------------- the restore the drivers use
def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by `model_checkpoint_path` in <directory>/checkpoint."""
    state = os.path.join(str(directory), "checkpoint")
    if not os.path.exists(state):
        return None
    for line in open(state):
        m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
        if m:
            p = m.group(1)
            return p if os.path.isabs(p) else os.path.join(str(directory), p)
    return None


_CRC_AUTO_BYTES = 4 << 20


def load_checkpoint(path, config, use_ema=True, skip_mismatch=False, verify_crc=None):
    """Weight set (reference variable name -> float32 array) of the detector described by `config`.

    Every variable `weights.variable_specs(config)` lists is looked up by name; with `use_ema` the shadow
    `<name>/ExponentialMovingAverage` wins for EVERY variable, BN moving statistics included, as `get_ema_vars` +
    `restore_ckpt` assign it after the plain name (utils_keras.py:85-97,183-196).  A missing variable or a shape
    mismatch raises (KeyError / ValueError, :213-233) - the reference's driver restores with `skip_mismatch=False`
    (infer_lib.py:435).  With `skip_mismatch` the variable keeps its initial value and every such variable is logged.
    Shapes must be equal; only the scalar fusion weights (WSM) may be stored as () or (1,).
    verify_crc: None = check the per-tensor crc32c whenever the HIP library is built (its host-side `uda_crc32c`), and for
    bundles up to 4 MB with the pure-Python fallback; True / False = always / never."""
    path = str(path)
    if os.path.isdir(path):
        path = latest_checkpoint(path) or path
    for ext in (".index", ".data-00000-of-00001"):
        if path.endswith(ext):
            path = path[:-len(ext)]
    reader = BundleReader(path)
    names = reader.name_map()
    if verify_crc is None:
        verify_crc = reader.total_bytes() <= _CRC_AUTO_BYTES or _native_crc() is not None
        if not verify_crc:
            logging.getLogger(__name__).warning(
                "%s: per-tensor crc32c NOT verified (%.1f MB; pass verify_crc=True to check, ~1 s per MB)",
                path, reader.total_bytes() / 2 ** 20)
    init = None
    out = {}
    for name, shape, kind in weights_mod.variable_specs(config):
        fields = [(name + "/" + f, shape) for f in weights_mod.BN_FIELDS] if kind == "bn" else [(name, shape)]
        for var, shp in fields:
            key = None
            problem = None
            if use_ema:
                # the reference puts BOTH names into its restore map when moving_average_decay > 0 (utils_keras.py:200-235) and
                # raises when the shadow is missing: a checkpoint saved without EMA does not load silently as if it had one
                key = names.get(var + "/ExponentialMovingAverage")
                if key is None and names.get(var) is not None:
                    if not skip_mismatch:
                        raise KeyError("Not found %s/ExponentialMovingAverage in %s (the checkpoint holds %s without its moving "
                                       "average; restore with moving_average_decay = 0 or skip_mismatch=True)" % (var, path, var))
                    logging.getLogger(__name__).warning("skip_mismatch: %s has no ExponentialMovingAverage shadow in %s: the plain "
                                                        "variable is restored", var, path)
            if use_ema and key is not None:
                # ... and the plain name is in that map too: it must exist with the right shape although the shadow is what is kept
                plain = names.get(var)
                if plain is None:
                    problem = KeyError("Not found %s in %s" % (var, path))
                else:
                    gotp = tuple(reader.entries[plain]["shape"])
                    wantp = tuple(shp) if kind != "wsm" else ()
                    if gotp != wantp and not (kind == "wsm" and gotp in ((), (1,))):
                        problem = ValueError("Shape mismatch: %s, expected %s, but got %s" % (var, wantp, gotp))
                if problem is not None and skip_mismatch:
                    logging.getLogger(__name__).warning("skip_mismatch: %s (the ExponentialMovingAverage shadow is restored)", problem)
                    problem = None
            key = key or names.get(var)
            if key is None:
                problem = KeyError("Not found %s in %s" % (var, path))
            elif problem is None:
                got = tuple(reader.entries[key]["shape"])
                want = tuple(shp) if kind != "wsm" else ()
                if got != want and not (kind == "wsm" and got in ((), (1,))):
                    problem = ValueError("Shape mismatch: %s, expected %s, but got %s" % (var, want, got))
                    # restore_ckpt (utils_keras.py:213-235) assigns the plain variable first and only warns when the
                    # ExponentialMovingAverage shadow has the wrong shape: the variable keeps the checkpoint's PLAIN value
                    plain = names.get(var)
                    if use_ema and skip_mismatch and plain is not None and plain != key:
                        gotp = tuple(reader.entries[plain]["shape"])
                        if gotp == want or (kind == "wsm" and gotp in ((), (1,))):
                            logging.getLogger(__name__).warning("skip_mismatch: %s: the plain variable is restored", problem)
                            key, problem = plain, None
            if problem is not None:
                if not skip_mismatch:
                    raise problem
                logging.getLogger(__name__).warning("skip_mismatch: %s keeps its initial value (%s)", var, problem)
                if init is None:
                    init = weights_mod.init_weights(config, seed=int(config.get("uda_seed", 0)))
                out[var] = init[var]
                continue
            out[var] = np.ascontiguousarray(reader.get_tensor(key, verify_crc), dtype=np.float32).reshape(tuple(shp) if kind != "wsm" else ())
    return out


# ---------------------------------------------------------------- writer (name-based bundle)
_CRC_TABLE = None


def _native_crc():
    """uda_crc32c of the HIP library (host code, slicing-by-8: hundreds of MB/s) when the library is built; else None."""
    try:
        from . import capi
        return capi.load().uda_crc32c
    except Exception:
        return None


def _crc32c(data, crc=0):
    fn = _native_crc()
    if fn is not None:
        buf = np.frombuffer(bytes(data) if not isinstance(data, (bytes, bytearray, np.ndarray)) else data, np.uint8)
        buf = np.ascontiguousarray(buf)
        return int(fn(buf.ctypes.data, buf.size, crc))
    return _crc32c_py(data, crc)


def _crc32c_py(data, crc=0):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    crc ^= 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field(num, wt, payload):
    tag = _enc_varint((num << 3) | wt)
    if wt == 2:
        return tag + _enc_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    return tag + _enc_varint(payload)


def _block(pairs):
    """One table block, a restart point at every entry (no prefix sharing: valid, just not compact)."""
    body, restarts = bytearray(), []
    for k, v in pairs:
        restarts.append(len(body))
        body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
    for r in restarts or [0]:
        body += struct.pack("<I", r)
    body += struct.pack("<I", max(len(restarts), 1))
    return bytes(body)


def save_checkpoint(prefix, weights, checksum=True):
    """Write `weights` (name -> array) as a name-based TensorFlow checkpoint (one shard) and a `checkpoint` state file.
    checksum=False skips the per-tensor crc32c (pure-Python CRC: ~1 s per MB), which TensorFlow's reader would reject."""
    names = sorted(weights)
    data_path = prefix + ".data-00000-of-00001"
    entries, off = [], 0
    with open(data_path, "wb") as f:
        for n in names:
            a = np.ascontiguousarray(weights[n], dtype=np.float32)
            raw = a.tobytes()
            f.write(raw)
            shape = b"".join(_field(2, 2, _field(1, 0, int(d))) for d in a.shape)
            e = _field(1, 0, 1) + _field(2, 2, shape)
            if off:
                e += _field(4, 0, off)
            e += _field(5, 0, len(raw)) + _field(6, 5, _masked(_crc32c(raw)) if checksum else 0)
            entries.append((n.encode(), e))
            off += len(raw)
    header = _field(1, 0, 1) + _field(3, 2, _field(1, 0, 1))                  # num_shards = 1, little endian, version.producer = 1
    pairs = [(b"", header)] + entries
    out = bytearray()

    def emit(block):
        h = _enc_varint(len(out)) + _enc_varint(len(block))
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", _masked(_crc32c(block + b"\x00"))))
        return h

    index_pairs = []
    step = 64
    for i in range(0, len(pairs), step):
        chunk = pairs[i:i + step]
        index_pairs.append((chunk[-1][0], emit(_block(chunk))))
    meta = emit(_block([]))
    index = emit(_block(index_pairs))
    footer = meta + index
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (os.path.basename(prefix), os.path.basename(prefix)))
    return prefix
