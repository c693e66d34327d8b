"""Readers for the reference's data files -- the input side of the hot path
(train_cloudAAE_ycbv.py:36-65, 177): TFRecord framing + tf.train.Example decoding, without
TensorFlow (neither TF nor a crc32c module exists in this image).

  TFRecord framing:
      u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)
  Example  { Features features = 1 }            Features { map<string, Feature> feature = 1 }
  Feature  { oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 } }
  FloatList/Int64List { repeated value = 1 [packed] }

  train_syn/<cls>_syn.tfrecords : class_id int64[1], translation float[3], axisangle float[3]
  obj_models.tfrecords          : label int64[1], model float[2048*6]   (xyz metres + rgb)

This is host-side IO (the reference does it in tf.data on /cpu:0 as well); nothing here is
arithmetic on the training path.
"""
import struct

import numpy as np

# ---- masked CRC32C (Castagnoli), table-driven; only used when verify=True --------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        poly = 0x82F63B78
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ poly if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    return _CRC_TABLE


_native_crc = None


def _native():
    """cloudaae_crc32c of libcloudaae_hip.so (host code: SSE4.2 crc32, ~GB/s) when the library is built."""
    global _native_crc
    if _native_crc is None:
        _native_crc = False
        try:
            import ctypes
            import os
            path = os.environ.get("CLOUDAAE_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                   "libcloudaae_hip.so"))
            if os.path.exists(path):
                import torch  # noqa: F401  (one HIP runtime per process: torch's is loaded first, as _lib does)
                fn = ctypes.CDLL(path).cloudaae_crc32c
                fn.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_uint]
                fn.restype = ctypes.c_uint
                _native_crc = fn
        except (OSError, AttributeError, ImportError):
            _native_crc = False
    return _native_crc


def crc32c(data, native=True):
    """CRC-32C (Castagnoli) of a bytes-like object; native=False forces the pure-Python table loop (tests)."""
    fn = _native() if native and len(data) >= 64 else None
    if fn:
        return int(fn(bytes(data) if not isinstance(data, bytes) else data, len(data), 0))
    tab = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def tf_record_iterator(path, verify=False):
    """Yields the payload bytes of every record (tf.python_io.tf_record_iterator, train...:49)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) != 12:
                raise IOError("%s: truncated record header" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            payload = f.read(length)
            tail = f.read(4)
            if len(payload) != length or len(tail) != 4:
                raise IOError("%s: truncated record" % path)
            if verify:
                if masked_crc32c(head[:8]) != lcrc or masked_crc32c(payload) != struct.unpack("<I", tail)[0]:
                    raise IOError("%s: CRC mismatch" % path)
            yield payload


# ---- minimal protobuf wire decoding --------------------------------------------------------
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """(field_number, wire_type, value) triples of one message; value is an int for
    varint/fixed fields and a memoryview for length-delimited ones."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield num, wt, v


def _decode_feature(buf):
    for num, wt, v in _fields(buf):
        if num == 1:      # BytesList
            return [bytes(x) for n2, _, x in _fields(v) if n2 == 1]
        if num == 2:      # FloatList: packed (wire type 2) or repeated fixed32
            out = []
            for n2, w2, x in _fields(v):
                if n2 != 1:
                    continue
                if w2 == 2:
                    out.append(np.frombuffer(bytes(x), dtype="<f4"))
                else:
                    out.append(np.frombuffer(x, dtype="<f4"))
            return np.concatenate(out) if out else np.zeros(0, np.float32)
        if num == 3:      # Int64List: packed varints or repeated varint
            out = []
            for n2, w2, x in _fields(v):
                if n2 != 1:
                    continue
                if w2 == 2:
                    p, m = 0, len(x)
                    while p < m:
                        val, p = _varint(x, p)
                        out.append(val)
                else:
                    out.append(x)
            arr = np.array(out, dtype=np.uint64).astype(np.int64)   # two's complement for negatives
            return arr
    return None


def parse_example(serialized):
    """tf.parse_single_example without a schema: {feature name: ndarray / list of bytes}."""
    out = {}
    for num, _, features in _fields(serialized):
        if num != 1:
            continue
        for n2, _, entry in _fields(features):
            if n2 != 1:
                continue
            key = val = None
            for n3, _, x in _fields(entry):      # map entry: key = 1, value = 2
                if n3 == 1:
                    key = bytes(x).decode("utf-8")
                elif n3 == 2:
                    val = _decode_feature(x)
            out[key] = val
    return out


# ---- the two file kinds of the reference ----------------------------------------------------
def decode(serialized_example):
    """train_cloudAAE_ycbv.py:57-65: translation float[3], axisangle float[3], class_id int64."""
    ex = parse_example(serialized_example)
    t, a, c = ex["translation"], ex["axisangle"], ex["class_id"]
    if t.shape != (3,) or a.shape != (3,) or c.shape != (1,):
        raise ValueError("unexpected feature shapes in pose record")
    return {"translation": t.astype(np.float32), "axisangle": a.astype(np.float32), "class_id": np.int64(c[0])}


def read_and_decode_obj_model(filename):
    """train_cloudAAE_ycbv.py:42-54: (models [n,2048,6] float32, labels [n] int64)."""
    models, labels = [], []
    for rec in tf_record_iterator(filename):
        ex = parse_example(rec)
        m = ex["model"]
        if m.size != 2048 * 6:
            raise ValueError("object model record with %d floats" % m.size)
        models.append(m.reshape(2048, 6).astype(np.float32))
        labels.append(np.int64(ex["label"][0]))
    return np.stack(models), np.array(labels, np.int64)


def _decode_pose_group(body):
    """Vectorised decode of pose-record payloads body[n, length] that share ONE byte layout (same bytes
    outside the three value fields).  Returns (translation, axisangle, class_id) or None."""
    n, length = body.shape
    first = body[0].tobytes()
    d0 = decode(first)
    spans = {}
    for key in ("translation", "axisangle"):
        blob = d0[key].astype("<f4").tobytes()
        at = first.find(blob)
        if at < 0 or first.find(blob, at + 1) >= 0:
            return None
        spans[key] = (at, at + 12)
    # class_id is a one-byte varint: the byte whose change moves class_id (and nothing else) under the
    # generic decoder
    cid_at = None
    for pos in range(length):
        if any(lo <= pos < hi for lo, hi in spans.values()) or first[pos] != int(d0["class_id"]) or first[pos] >= 0x7f:
            continue
        probe = bytearray(first)
        probe[pos] = first[pos] + 1
        try:
            dp = decode(bytes(probe))
        except Exception:
            continue
        if (int(dp["class_id"]) == int(d0["class_id"]) + 1 and np.array_equal(dp["translation"], d0["translation"])
                and np.array_equal(dp["axisangle"], d0["axisangle"])):
            cid_at = pos
            break
    if cid_at is None:
        return None
    fixed = np.ones(length, bool)
    for lo, hi in spans.values():
        fixed[lo:hi] = False
    fixed[cid_at] = False
    if not (body[:, fixed] == body[0, fixed]).all() or (body[:, cid_at] >= 0x80).any():
        return None
    t = np.ascontiguousarray(body[:, spans["translation"][0]:spans["translation"][1]]).view("<f4").reshape(n, 3)
    a = np.ascontiguousarray(body[:, spans["axisangle"][0]:spans["axisangle"][1]]).view("<f4").reshape(n, 3)
    c = body[:, cid_at].astype(np.int64)
    for i in {0, n // 2, n - 1}:                         # spot check against the generic decoder
        d = decode(body[i].tobytes())
        if not (np.array_equal(d["translation"], t[i]) and np.array_equal(d["axisangle"], a[i]) and
                int(d["class_id"]) == int(c[i])):
            return None
    return t.astype(np.float32), a.astype(np.float32), c


def _decode_pose_file_fast(path):
    """Whole-file decode of a pose-record file: all records have one length, and the protobuf map of
    an Example is written in one of a few entry orders -- records are grouped by their bytes at the
    (few) positions that tell the orders apart and every group is decoded as one array operation
    (the shipped train_syn files: ~100x faster than record by record).  Returns (translation [n,3],
    axisangle [n,3], class_id [n]) in file order, or None when the file does not fit the pattern (the
    caller then decodes record by record)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 16:
        return None
    length = int(raw[:8].view("<u8")[0])
    stride = 12 + length + 4
    if length < 8 or raw.size % stride != 0:
        return None
    n = raw.size // stride
    recs = raw.reshape(n, stride)
    if not (recs[:, :8] == recs[0, :8]).all():          # every length field equal
        return None
    body = recs[:, 12:12 + length]
    # entry order signature: the length bytes of the three map entries (Features{ map entry* })
    o1 = 2
    l1 = body[:, o1 + 1].astype(np.int64)
    o2 = o1 + 2 + l1
    if (o2 + 1 >= length).any():
        return None
    rows = np.arange(n)
    l2 = body[rows, o2 + 1].astype(np.int64)
    sig = l1 * 256 + l2
    t = np.zeros((n, 3), np.float32)
    a = np.zeros((n, 3), np.float32)
    c = np.zeros((n,), np.int64)
    groups = np.unique(sig)
    if groups.size > 6:
        return None
    for gsig in groups:
        sel = np.nonzero(sig == gsig)[0]
        got = _decode_pose_group(np.ascontiguousarray(body[sel]))
        if got is None:
            return None
        t[sel], a[sel], c[sel] = got
    return t, a, c


class PoseRecords(object):
    """All pose records of a list of train_syn files in memory (381,553 records x 28 B for the
    shipped set), with the reference's epoch semantics: full shuffle (its shuffle buffer exceeds
    the dataset, :177) and drop_remainder batching (:114).  `shard(rank, world)` gives each
    data-parallel rank a disjoint strided subset."""

    def __init__(self, filenames, verify=False):
        t, a, c = [], [], []
        for fn in filenames:
            fast = None if verify else _decode_pose_file_fast(fn)
            if fast is not None:
                t.append(fast[0])
                a.append(fast[1])
                c.append(fast[2])
                continue
            ft, fa, fc = [], [], []
            for rec in tf_record_iterator(fn, verify=verify):
                d = decode(rec)
                ft.append(d["translation"])
                fa.append(d["axisangle"])
                fc.append(d["class_id"])
            if ft:
                t.append(np.stack(ft).astype(np.float32))
                a.append(np.stack(fa).astype(np.float32))
                c.append(np.array(fc, np.int64))
        self.translation = np.concatenate(t) if t else np.zeros((0, 3), np.float32)
        self.axisangle = np.concatenate(a) if a else np.zeros((0, 3), np.float32)
        self.class_id = np.concatenate(c) if c else np.zeros((0,), np.int64)

    def __len__(self):
        return len(self.class_id)

    def shard(self, rank, world):
        """Rank-strided shard, truncated to len // world records so that EVERY rank yields the same number
        of batches per epoch (one rank running an extra step would issue all-reduces nobody joins)."""
        n = len(self) // world
        out = object.__new__(PoseRecords)
        out.translation = self.translation[rank::world][:n]
        out.axisangle = self.axisangle[rank::world][:n]
        out.class_id = self.class_id[rank::world][:n]
        return out

    def epoch(self, batch_size, seed=None, shuffle=True):
        """Yields dicts of [batch_size, ...] arrays; the remainder is dropped."""
        n = len(self)
        order = np.random.default_rng(seed).permutation(n) if shuffle else np.arange(n)
        for i in range(0, n - batch_size + 1, batch_size):
            sel = order[i:i + batch_size]
            yield {"translation": self.translation[sel], "axisangle": self.axisangle[sel],
                   "class_id": self.class_id[sel]}
