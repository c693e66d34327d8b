"""CPU: the TFRecord / tf.train.Example reader against fixtures cut from the reference's data
files; expected values were produced by an independent decoder (the protobuf runtime,
oracle/make_golden_tfrecord.py)."""
import os

import numpy as np
import pytest

from cloudaae_amd import tfrecord_io as T


def test_pose_records(golden_dir):
    exp = np.load(os.path.join(golden_dir, "tfrecord_expected.npz"))
    path = os.path.join(golden_dir, "pose_records_cls0_first4.tfrecords")
    recs = list(T.tf_record_iterator(path, verify=True))       # CRCs of the real files check out
    assert [len(r) for r in recs] == [85] * 4                    # 85-byte payload
    for i, r in enumerate(recs):
        d = T.decode(r)
        assert d["class_id"] == exp["class_id"][i] == 0
        assert np.array_equal(d["translation"], exp["translation"][i])
        assert np.array_equal(d["axisangle"], exp["axisangle"][i])
        assert d["translation"].dtype == np.float32 and np.linalg.norm(d["axisangle"]) <= np.pi + 1e-6


def test_object_model(golden_dir):
    exp = np.load(os.path.join(golden_dir, "tfrecord_expected.npz"))
    models, labels = T.read_and_decode_obj_model(os.path.join(golden_dir, "obj_model_first1.tfrecords"))
    assert models.shape == (1, 2048, 6) and models.dtype == np.float32
    assert np.array_equal(models[0], exp["model"]) and labels[0] == exp["label"]
    assert np.abs(models[0][:, :3]).max() < 0.2 and 0 <= models[0][:, 3:].min() and models[0][:, 3:].max() <= 1


def test_crc_and_truncation(golden_dir, tmp_path):
    assert T.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    raw = open(os.path.join(golden_dir, "pose_records_cls0_first4.tfrecords"), "rb").read()
    bad = bytearray(raw)
    bad[20] ^= 0xFF
    p = tmp_path / "bad.tfrecords"
    p.write_bytes(bytes(bad))
    assert len(list(T.tf_record_iterator(str(p)))) == 4       # CRCs are skipped by default
    with pytest.raises(IOError):
        list(T.tf_record_iterator(str(p), verify=True))
    p.write_bytes(raw[:150])
    with pytest.raises(IOError):
        list(T.tf_record_iterator(str(p)))


def test_epoch_semantics_and_sharding(golden_dir):
    path = os.path.join(golden_dir, "pose_records_cls0_first4.tfrecords")
    ds = T.PoseRecords([path, path, path])                        # 12 records
    assert len(ds) == 12
    batches = list(ds.epoch(5, seed=0))
    assert len(batches) == 2 and batches[0]["translation"].shape == (5, 3)   # drop_remainder
    assert len(list(ds.epoch(13, seed=0))) == 0
    a, b = ds.shard(0, 2), ds.shard(1, 2)
    assert len(a) == len(b) == 6
    # every rank gets the same number of records (a longer shard would run a step nobody joins)
    odd = [ds.shard(r, 5) for r in range(5)]
    assert [len(x) for x in odd] == [2] * 5
    ds7 = T.PoseRecords([path, path])                                       # 8 records over 3 ranks
    assert [len(ds7.shard(r, 3)) for r in range(3)] == [2, 2, 2]
    seen = np.concatenate([x["translation"] for x in ds.epoch(4, seed=1)])
    assert seen.shape == (12, 3)                                            # a permutation of everything
    assert np.array_equal(np.sort(seen[:, 0]), np.sort(ds.translation[:, 0]))


def test_negative_int64_and_unpacked_lists():
    # hand-built Example: int64 -3 (10-byte varint), an unpacked float list
    def varint(v):
        v &= (1 << 64) - 1
        out = bytearray()
        while True:
            b = v & 0x7F
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def ld(num, payload):
        return varint(num << 3 | 2) + varint(len(payload)) + payload
    int_list = ld(3, ld(1, varint(-3) + varint(7)))
    import struct
    flt_list = ld(2, b"".join(varint(1 << 3 | 5) + struct.pack("<f", x) for x in (1.5, -2.0)))
    feats = ld(1, ld(1, b"ids") + ld(2, int_list)) + ld(1, ld(1, b"v") + ld(2, flt_list))
    ex = T.parse_example(ld(1, feats))
    assert ex["ids"].tolist() == [-3, 7] and ex["v"].tolist() == [1.5, -2.0]


def test_fast_pose_decode_equals_record_by_record(golden_dir, tmp_path):
    """PoseRecords decodes a whole file as array operations (records grouped by the order their
    protobuf map entries were written in); it must give exactly what the record-by-record decoder
    gives, and files that do not fit the pattern must still load through the slow path."""
    import numpy as np
    from cloudaae_amd import tfrecord_io as R
    fn = os.path.join(golden_dir, "pose_records_cls0_first4.tfrecords")
    fast = R._decode_pose_file_fast(fn)
    slow = [R.decode(x) for x in R.tf_record_iterator(fn, verify=True)]
    assert fast is not None
    assert np.array_equal(fast[0], np.stack([d["translation"] for d in slow]))
    assert np.array_equal(fast[1], np.stack([d["axisangle"] for d in slow]))
    assert np.array_equal(fast[2], np.array([d["class_id"] for d in slow]))
    recs = R.PoseRecords([fn])
    assert len(recs) == 4 and np.array_equal(recs.translation, fast[0])
    # a file with a ragged tail does not fit the fixed-stride pattern: fast path declines, slow path reads it
    raw = open(fn, "rb").read()
    extra = b"\x0a\x00"                                   # an Example with an empty feature map
    rec = len(extra).to_bytes(8, "little") + R.masked_crc32c(len(extra).to_bytes(8, "little")).to_bytes(4, "little") \
        + extra + R.masked_crc32c(extra).to_bytes(4, "little")
    odd = tmp_path / "odd.tfrecords"
    odd.write_bytes(raw + rec)
    assert R._decode_pose_file_fast(str(odd)) is None
    assert sum(1 for _ in R.tf_record_iterator(str(odd), verify=True)) == 5


def test_native_crc32c_equals_the_table_loop():
    """cloudaae_crc32c (host code of libcloudaae_hip.so, SSE4.2) against the byte-at-a-time definition and the
    standard check value of CRC-32C."""
    import os as _os
    assert T.crc32c(b"123456789", native=False) == 0xE3069283
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 255, 4096, 100003):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c(d) == T.crc32c(d, native=False), n
    assert T._native(), "libcloudaae_hip.so is built in this tree: the native checksum must load"
