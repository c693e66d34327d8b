"""Reader / writer of TensorFlow "V2" checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`), the
format `tf.train.Saver` writes and restores in the reference (train_cloudAAE_ycbv.py:276, :418-430;
evaluate_cloudAAE_ycbv.py:495-499; the shipped snapshot trained_network/20200908-204328/model.ckpt.*).
Host side, no TensorFlow and no protobuf runtime -- like tfrecord_io.py it decodes the wire formats by hand:

  .index   a LevelDB-style sorted table (tensorflow/core/lib/io/table*): data blocks of prefix-compressed
           (key, value) entries + restart array, each followed by a 1-byte compression tag and a masked
           crc32c; an index block; a 48-byte footer ending in the magic 0xdb4775248b80fb57.
           key ""      -> BundleHeaderProto  (num_shards, endianness, version)
           key <name>  -> BundleEntryProto   (dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6)
  .data-*  the raw little-endian tensor bytes at [offset, offset + size) of shard shard_id.

`TrainGraph.restore()` accepts such a prefix; `write_checkpoint` produces one from name -> array (what
`TrainGraph.checkpoint()` returns), so a run of this package can be restored by the reference and vice versa.
"""
import os
import struct
from collections import OrderedDict

import numpy as np

from .tfrecord_io import _varint, crc32c, masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def _fields(buf):
    """(field number, wire type, value) of a serialized protobuf message; value = int or bytes."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise IOError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _shape(buf):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }."""
    dims = []
    for num, _, v in _fields(buf):
        if num == 2:
            size = 0
            for n2, _, v2 in _fields(v):
                if n2 == 1:
                    size = v2
            dims.append(int(size))
    return tuple(dims)


def _entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=0)
    for num, wt, v in _fields(buf):
        if num == 1:
            e["dtype"] = int(v)
        elif num == 2:
            e["shape"] = _shape(v)
        elif num == 3:
            e["shard_id"] = int(v)
        elif num == 4:
            e["offset"] = int(v)
        elif num == 5:
            e["size"] = int(v)
        elif num == 6:
            e["crc32c"] = struct.unpack("<I", v)[0] if wt == 5 else int(v)
    return e


def _block(data, offset, size, verify):
    """The (key, value) pairs of the table block at [offset, offset + size) (+ 5-byte trailer)."""
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise IOError("compressed table blocks (type %d) are not supported" % ctype)
    if verify:
        want = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
        if masked_crc32c(raw + bytes([ctype])) != want:
            raise IOError("table block checksum mismatch at offset %d" % offset)
    nrestart = struct.unpack("<I", raw[-4:])[0]
    end = len(raw) - 4 - 4 * nrestart
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(raw, pos)
        non_shared, pos = _varint(raw, pos)
        vlen, pos = _varint(raw, pos)
        key = key[:shared] + raw[pos:pos + non_shared]
        pos += non_shared
        out.append((key, raw[pos:pos + vlen]))
        pos += vlen
    return out


def _handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_index(prefix, verify=True):
    """name -> {dtype (numpy), shape, shard_id, offset, size, crc32c} of `<prefix>.index`, plus the header
    under the key "" ({num_shards, little_endian})."""
    path = prefix if prefix.endswith(".index") else prefix + ".index"
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise IOError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = data[-48:]
    _, _, p = _handle(footer, 0)                    # metaindex (unused)
    ioff, isize, _ = _handle(footer, p)
    out = OrderedDict()
    for _, hv in _block(data, ioff, isize, verify):  # index block: last key of a data block -> its handle
        boff, bsize, _ = _handle(hv, 0)
        for key, value in _block(data, boff, bsize, verify):
            if key == b"":
                hdr = dict(num_shards=1, little_endian=True)
                for num, _, v in _fields(value):
                    if num == 1:
                        hdr["num_shards"] = int(v)
                    elif num == 2:
                        hdr["little_endian"] = int(v) == 0
                out[""] = hdr
                continue
            e = _entry(value)
            if e["dtype"] not in _DTYPES:
                raise IOError("variable %s has unsupported dtype %d" % (key.decode(), e["dtype"]))
            e["dtype"] = np.dtype(_DTYPES[e["dtype"]])
            out[key.decode("utf-8")] = e
    return out


def _shard_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix[:-6] if prefix.endswith(".index") else prefix, shard, num_shards)


def load_checkpoint(prefix, names=None, verify=True):
    """name -> numpy array for every (or the named) variable of the checkpoint."""
    index = read_index(prefix, verify)
    hdr = index.pop("", dict(num_shards=1, little_endian=True))
    if not hdr["little_endian"]:
        raise IOError("big-endian checkpoints are not supported")
    shards, out = {}, OrderedDict()
    for name, e in index.items():
        if names is not None and name not in names:
            continue
        if e["shard_id"] not in shards:
            p = _shard_path(prefix, e["shard_id"], hdr["num_shards"])
            if not os.path.exists(p):
                raise IOError("checkpoint data shard %s is missing" % p)
            shards[e["shard_id"]] = np.memmap(p, dtype=np.uint8, mode="r")
        raw = bytes(shards[e["shard_id"]][e["offset"]:e["offset"] + e["size"]])
        if len(raw) != e["size"]:
            raise IOError("variable %s lies outside its data shard" % name)
        if verify and e["crc32c"] and masked_crc32c(raw) != e["crc32c"]:
            raise IOError("variable %s: data checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=e["dtype"]).reshape(e["shape"]).copy()
    return out


# ---- writer -----------------------------------------------------------------------------------------
def _vint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_varint(num, v):
    return _vint(num << 3) + _vint(v)


def _pb_bytes(num, payload):
    return _vint(num << 3 | 2) + _vint(len(payload)) + payload


def _write_block(entries):
    """One table block: every entry is a restart point (no prefix sharing: valid, and the files are tiny)."""
    body, restarts = bytearray(), []
    for key, value in entries:
        restarts.append(len(body))
        body += _vint(0) + _vint(len(key)) + _vint(len(value)) + key + value
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    raw = bytes(body)
    return raw + b"\x00" + struct.pack("<I", masked_crc32c(raw + b"\x00"))


def write_checkpoint(prefix, arrays):
    """Write name -> array as `<prefix>.index` + `<prefix>.data-00000-of-00001` (one shard, little endian)."""
    names = sorted(arrays, key=lambda s: s.encode("utf-8"))
    data, entries = bytearray(), [(b"", _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1)))]   # header: num_shards 1, version{producer 1}
    for name in names:
        a = np.asarray(arrays[name])
        a = a if a.flags.c_contiguous else np.ascontiguousarray(a)      # (ascontiguousarray makes 0-d arrays 1-d)
        if a.dtype not in _DTYPE_IDS:
            raise ValueError("variable %s has unsupported dtype %s" % (name, a.dtype))
        raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
        shape = b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in a.shape)
        value = _pb_varint(1, _DTYPE_IDS[a.dtype]) + _pb_bytes(2, shape)
        if len(data):
            value += _pb_varint(4, len(data))
        value += _pb_varint(5, len(raw)) + _vint(6 << 3 | 5) + struct.pack("<I", masked_crc32c(raw))
        entries.append((name.encode("utf-8"), value))
        data += raw
    block = _write_block(entries)
    # index block: one entry, key >= the last key of the data block, value = its handle (offset 0, size w/o trailer)
    index = _write_block([(entries[-1][0] + b"\x00", _vint(0) + _vint(len(block) - 5))])
    meta = _write_block([])
    moff, ioff = len(block), len(block) + len(meta)
    footer = _vint(moff) + _vint(len(meta) - 5) + _vint(ioff) + _vint(len(index) - 5)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    tmp = prefix + ".tmp"
    with open(tmp + ".data", "wb") as f:
        f.write(bytes(data))
    with open(tmp + ".index", "wb") as f:
        f.write(block + meta + index + footer)
    os.replace(tmp + ".data", _shard_path(prefix, 0, 1))
    os.replace(tmp + ".index", prefix + ".index")
    return prefix


__all__ = ["read_index", "load_checkpoint", "write_checkpoint", "crc32c"]
