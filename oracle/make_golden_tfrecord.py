"""Generates the TFRecord reader fixtures -- TEST INFRASTRUCTURE ONLY.
    python -m oracle.make_golden_tfrecord      (needs /root/reference)
Copies the first 4 records of train_syn/0_syn.tfrecords and record 0 of obj_models.tfrecords
(data files of the reference) byte-for-byte into tests/golden/, and stores their contents as
parsed by an INDEPENDENT decoder: the protobuf runtime with tf.train.Example's schema declared
here (tensorflow/core/example/{example,feature}.proto), not our hand-rolled wire parser."""
import os
import struct

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

REF = os.environ.get("CLOUDAAE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def example_class():
    fd = descriptor_pb2.FileDescriptorProto(name="ex.proto", package="t", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
        return m
    msg("BytesList", [("value", 1, T.TYPE_BYTES, T.LABEL_REPEATED, None)])
    msg("FloatList", [("value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, None)])
    msg("Int64List", [("value", 1, T.TYPE_INT64, T.LABEL_REPEATED, None)])
    msg("Feature", [("bytes_list", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t.BytesList"),
                    ("float_list", 2, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t.FloatList"),
                    ("int64_list", 3, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t.Int64List")])
    feats = msg("Features", [("feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, ".t.Features.FeatureEntry")])
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=T.TYPE_MESSAGE, label=T.LABEL_OPTIONAL, type_name=".t.Feature")
    msg("Example", [("features", 1, T.TYPE_MESSAGE, T.LABEL_OPTIONAL, ".t.Features")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("t.Example"))


def raw_records(path, count):
    out = []
    with open(path, "rb") as f:
        for _ in range(count):
            head = f.read(12)
            (n,) = struct.unpack("<Q", head[:8])
            body = f.read(n + 4)
            out.append(head + body)
    return out


def main():
    Example = example_class()
    os.makedirs(OUT, exist_ok=True)
    pose = raw_records(os.path.join(REF, "ycb_video_data_tfRecords/train_syn/0_syn.tfrecords"), 4)
    open(os.path.join(OUT, "pose_records_cls0_first4.tfrecords"), "wb").write(b"".join(pose))
    model = raw_records(os.path.join(REF, "object_model_tfrecord/obj_models.tfrecords"), 1)
    open(os.path.join(OUT, "obj_model_first1.tfrecords"), "wb").write(b"".join(model))
    t, a, c = [], [], []
    for rec in pose:
        ex = Example.FromString(rec[12:-4])
        f = ex.features.feature
        t.append(list(f["translation"].float_list.value))
        a.append(list(f["axisangle"].float_list.value))
        c.append(list(f["class_id"].int64_list.value))
    ex = Example.FromString(model[0][12:-4])
    f = ex.features.feature
    np.savez_compressed(os.path.join(OUT, "tfrecord_expected.npz"),
                        translation=np.array(t, np.float32), axisangle=np.array(a, np.float32),
                        class_id=np.array(c, np.int64)[:, 0],
                        model=np.array(f["model"].float_list.value, np.float32).reshape(2048, 6),
                        label=np.int64(f["label"].int64_list.value[0]),
                        payload_bytes=np.array([len(r) - 16 for r in pose + model]))
    print("wrote fixtures; pose payload sizes", [len(r) - 16 for r in pose], "model payload", len(model[0]) - 16)


if __name__ == "__main__":
    main()
