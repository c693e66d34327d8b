"""CPU: the TensorFlow V2 checkpoint reader / writer (cloudaae_amd/tf_checkpoint.py) -- against the index file of
the reference's own shipped snapshot (trained_network/20200908-204328/model.ckpt.index, kept as a DATA fixture:
tests/golden/reference_snapshot_20200908.ckpt.index; its data shard is missing from the reference checkout), and
the variable MANIFEST of this package against it: every name, shape and dtype tf.train.Saver stored for
get_model_dgcnn_mean_6d at num_point = 256 must be what TrainGraph.checkpoint() writes (checked here through the
oracle's variable list, which the GPU tests load 1:1 into the VariableStore with strict=True)."""
import os

import numpy as np
import pytest
import torch

from cloudaae_amd import tf_checkpoint as C
from cloudaae_amd import train_cloudAAE_ycbv as T


@pytest.fixture(scope="module")
def snapshot_index(golden_dir):
    return C.read_index(os.path.join(golden_dir, "reference_snapshot_20200908.ckpt.index"), verify=True)


def test_reads_the_reference_snapshot_index(snapshot_index):
    idx = dict(snapshot_index)
    assert idx.pop("") == {"num_shards": 1, "little_endian": True}
    assert len(idx) == 175                                  # 53 variables (3 of them scalars) + 22 BN moving averages + 2 x 50 Adam slots
    assert idx["Variable"]["shape"] == () and idx["beta1_power"]["dtype"] == np.float32
    assert idx["dgcnn_output/weights"]["shape"] == (1024, 3072)          # 12 * num_point, num_point = 256
    assert idx["dgcnn1/weights"]["shape"] == (1, 1, 48, 64)
    # entries tile the data shard without gaps, in key order
    offs = sorted((e["offset"], e["size"]) for e in idx.values())
    pos = 0
    for o, s in offs:
        assert o == pos
        pos += s
    assert pos == 4 * 20819413


def test_variable_manifest_equals_the_reference_snapshot(snapshot_index):
    from oracle import model_oracle as MO
    V = MO.Vars(seed=0)
    with torch.no_grad():
        MO.forward_losses(MO.synthetic_batch(2, 256, seed=1), V, 256, is_training=False)
    ours = {}
    for name, p in V.p.items():
        ours[name] = tuple(p.shape)
        ours[name + "/Adam"] = tuple(p.shape)
        ours[name + "/Adam_1"] = tuple(p.shape)
    for name, s in V.s.items():
        # the snapshot (like evaluate_cloudAAE_ycbv.py:436) was built under tf.name_scope('6d_pose')
        ours[T.tf_variable_name(name, "6d_pose")] = tuple(s.shape)
    ours.update({"beta1_power": (), "beta2_power": (), "Variable": ()})
    ref = {k: v["shape"] for k, v in snapshot_index.items() if k}
    assert set(ours) == set(ref), (sorted(set(ours) - set(ref))[:5], sorted(set(ref) - set(ours))[:5])
    assert ours == ref
    assert all(v["dtype"] == np.float32 for k, v in snapshot_index.items() if k)
    # and the names map back to the store's own
    for name in V.s:
        assert T.store_variable_name(T.tf_variable_name(name, "6d_pose")) == name
        assert T.store_variable_name(T.tf_variable_name(name)) == name


def test_write_then_read_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    arrays = {"dgcnn1/weights": rng.standard_normal((1, 1, 48, 64)).astype(np.float32),
              "dgcnn1/bn/decoder/dgcnn1/bn/moments/Squeeze/ExponentialMovingAverage": rng.standard_normal(64).astype(np.float32),
              "Variable": np.float32(7.0), "beta1_power": np.float32(0.5),
              "counts": np.arange(5, dtype=np.int64), "empty": np.zeros((0, 3), np.float32)}
    prefix = str(tmp_path / "model.ckpt")
    C.write_checkpoint(prefix, arrays)
    idx = C.read_index(prefix)
    assert list(idx)[1:] == sorted(arrays, key=lambda s: s.encode())
    back = C.load_checkpoint(prefix)
    assert set(back) == set(arrays)
    for k, a in arrays.items():
        assert back[k].dtype == np.asarray(a).dtype and back[k].shape == np.asarray(a).shape
        assert np.array_equal(back[k], a)
    # corruption is detected
    p = prefix + ".data-00000-of-00001"
    raw = bytearray(open(p, "rb").read())
    raw[10] ^= 0xFF
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        C.load_checkpoint(prefix)
    os.remove(p)
    with pytest.raises(IOError):
        C.load_checkpoint(prefix)
    with pytest.raises(IOError):
        C.read_index(p + ".nothing") if os.path.exists(p + ".nothing") else (_ for _ in ()).throw(IOError("missing"))


def test_index_of_something_else_is_rejected(tmp_path):
    p = tmp_path / "x.index"
    p.write_bytes(b"\x00" * 100)
    with pytest.raises(IOError):
        C.read_index(str(p))
