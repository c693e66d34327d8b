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


def crc32c(data):
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


class PoseRecords(object):
    """All pose records of a list of train_syn files in memory (381,553 records x 28 B for the
    shipped set), with the reference's epoch semantics: full shuffle (its shuffle buffer exceeds
    the dataset, :177) and drop_remainder batching (:114).  `shard(rank, world)` gives each
    data-parallel rank a disjoint strided subset."""

    def __init__(self, filenames, verify=False):
        t, a, c = [], [], []
        for fn in filenames:
            for rec in tf_record_iterator(fn, verify=verify):
                d = decode(rec)
                t.append(d["translation"])
                a.append(d["axisangle"])
                c.append(d["class_id"])
        self.translation = np.stack(t).astype(np.float32) if t else np.zeros((0, 3), np.float32)
        self.axisangle = np.stack(a).astype(np.float32) if a else np.zeros((0, 3), np.float32)
        self.class_id = np.array(c, np.int64)

    def __len__(self):
        return len(self.class_id)

    def shard(self, rank, world):
        out = object.__new__(PoseRecords)
        out.translation = self.translation[rank::world]
        out.axisangle = self.axisangle[rank::world]
        out.class_id = self.class_id[rank::world]
        return out

    def epoch(self, batch_size, seed=None, shuffle=True):
        """Yields dicts of [batch_size, ...] arrays; the remainder is dropped."""
        n = len(self)
        order = np.random.default_rng(seed).permutation(n) if shuffle else np.arange(n)
        for i in range(0, n - batch_size + 1, batch_size):
            sel = order[i:i + batch_size]
            yield {"translation": self.translation[sel], "axisangle": self.axisangle[sel],
                   "class_id": self.class_id[sel]}
