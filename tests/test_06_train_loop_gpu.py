"""GPU: the epoch loop of train_cloudAAE_ycbv.py:332-437 driven by the reference's own record files
(tests/golden fixtures: first object model, first four class-0 pose records), and the checkpoint
round trip (tf.train.Saver, :276 / :418-424) under the reference's variable names."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dataset(golden_dir):
    from cloudaae_amd import tfrecord_io as R
    models, _ = R.read_and_decode_obj_model(os.path.join(golden_dir, "obj_model_first1.tfrecords"))
    recs = R.PoseRecords([os.path.join(golden_dir, "pose_records_cls0_first4.tfrecords")])
    return torch.from_numpy(models).cuda(), recs


def _graph(num_point=256, batch=4):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    return T.TrainGraph({"num_point": num_point, "gpu": 0}, {"optimizer": "adam"},
                        {"batch_size": batch, "learning_rate": 0.0008})


def test_epoch_loop_on_reference_records(hip, dataset, tmp_path):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    obj_models, recs = dataset
    graph = _graph()
    lines = []
    clog = T.ClassLossLog(graph.device)
    first = None
    for epoch in range(6):
        n, out = T.train_graph(graph, recs, obj_models, epoch, clog, lines.append, str(tmp_path), seed=epoch)
        assert n == 1                                    # 4 records, batch 4, drop_remainder
        losses = [float(out[k].detach()) for k in ("xyz_loss", "trans_loss", "axag_loss")]
        assert all(np.isfinite(losses))
        first = first or losses
    assert float(graph.batch) == 6.0                     # global_step advanced once per batch
    assert losses[0] < first[0]                          # Chamfer loss goes down on 4 fixed poses
    assert os.path.exists(os.path.join(str(tmp_path), "model.ckpt.npz"))
    assert any(l.startswith("epoch 0 batch 0 xyz_loss") for l in lines)
    rows = clog.flush()
    assert [r[0] for r in rows] == [0] and rows[0][1] == 24      # class 0 only, 6 x 4 samples


def test_element_shapes_match_the_reference_pipeline(hip, dataset):
    """train...:210-211: visiblePoints is [B, 2048+1+400, 3], visiblePoints_org [B, 2048+1, 3]."""
    from cloudaae_amd import train_cloudAAE_ycbv as T
    obj_models, recs = dataset
    rec = next(recs.epoch(4, shuffle=False))
    el = T.get_small_data({k: torch.as_tensor(v).cuda() for k, v in rec.items()}, obj_models, seed=3)
    assert tuple(el["visiblePoints"].shape) == (4, 2449, 3)
    assert tuple(el["visiblePoints_org"].shape) == (4, 2049, 3)
    # num_point=512: 4N = 2048 of the 2049 rows (:214)
    graph = _graph(num_point=512, batch=4)
    out = graph.train_step(el)
    assert tuple(out["visiblePoints_org_final"].shape) == (4, 2048, 3)
    assert tuple(out["xyz_recon"].shape) == (4, 2048, 3)
    assert np.isfinite(float(out["total_loss"]))
    # num_point=1024 cannot be fed from this pipeline in the reference either: the slice of :214
    # keeps 2049 rows against 4096 reconstructed points and chamfer_loss.py:12 adds [B,4096] to
    # [B,2049].  Same failure here, raised loudly.
    graph = _graph(num_point=1024, batch=4)
    with pytest.raises(ValueError, match="chamfer_loss.py:12"):
        graph.train_step(el)


def test_checkpoint_round_trip(hip, dataset, tmp_path):
    from cloudaae_amd import train_cloudAAE_ycbv as T
    obj_models, recs = dataset
    g1 = _graph()
    el = T.get_small_data({k: torch.as_tensor(v).cuda() for k, v in next(recs.epoch(4, shuffle=False)).items()},
                          obj_models, seed=1)
    el["noise"] = torch.zeros((4, 256, 3), device="cuda")
    for _ in range(3):
        g1.train_step(el)
    path = g1.save(os.path.join(str(tmp_path), "model.ckpt"))
    ck = np.load(path)
    names = set(ck.files)
    # the names tf.train.Saver would write for this graph
    # (BN moving averages carry the tf.name_scope the model is built under, train...:223: see tf_variable_name)
    for n in ("dgcnn1/weights", "dgcnn1/bn/beta", "dgcnn1/bn/decoder/dgcnn1/bn/moments/Squeeze/ExponentialMovingAverage",
              "dgcnn1/weights/Adam", "dgcnn1/weights/Adam_1", "beta1_power", "beta2_power", "Variable"):
        assert n in names, n
    assert float(ck["Variable"]) == 3.0
    assert abs(float(ck["beta1_power"]) - 0.9 ** 4) < 1e-6

    g2 = _graph()
    g2.restore(path)
    assert torch.equal(g1.store.flat_params, g2.store.flat_params)
    assert torch.equal(g1.store.flat_state, g2.store.flat_state)
    assert torch.equal(g1.adam_m, g2.adam_m) and torch.equal(g1.adam_v, g2.adam_v)
    assert float(g2.batch) == 3.0
    # evaluation mode reads the restored moving averages
    e1, e2 = g1.eval_step(el), g2.eval_step(el)
    assert float(e1["xyz_loss"]) == float(e2["xyz_loss"])         # (the forward pass is bit-reproducible)
    # the restored graph continues exactly like the original: the same losses on the next step (bit for bit),
    # the same weights after it up to the fp32 atomics of the backward pass
    o1, o2 = g1.train_step(el), g2.train_step(el)
    for k in ("xyz_loss", "trans_loss", "axag_loss"):
        assert float(o1[k]) == float(o2[k]), k
    # (fp32 atomics in split-K GEMMs / the Chamfer gradient make two runs differ by round-off;
    # Adam turns round-off-sized gradients -- e.g. of the analytically dead conv biases in front of
    # a BN -- into +-lr-sized moves, so the bound is 2 lr for those few and ~0 for the rest)
    diff = (g1.store.flat_params - g2.store.flat_params).abs()
    assert float(diff.max()) <= 2.1 * 0.0008
    assert float((diff > 1e-5).float().mean()) < 0.01


def test_tf_checkpoint_save_restore(hip, tmp_path):
    """saver.save / saver.restore in the reference's own format (train...:418-430): TrainGraph.save(fmt='tf') writes a
    TensorFlow V2 checkpoint whose names are what tf.train.Saver uses (BN moving averages under the 'decoder' name
    scope of train...:223), restore() brings another graph to the identical state -- also from the '6d_pose' naming of
    the shipped snapshot / evaluate_cloudAAE_ycbv.py:436."""
    from cloudaae_amd import tf_checkpoint as C
    from cloudaae_amd import train_cloudAAE_ycbv as T
    g = T.TrainGraph({"num_point": 128, "gpu": 0}, {}, {"batch_size": 4})
    el = T.synthetic_element(4, 128, g.device, seed=2)
    for _ in range(3):
        g.train_step(el)
    for scope in ("decoder", "6d_pose"):
        prefix = str(tmp_path / ("model_%s.ckpt" % scope))
        assert g.save(prefix, fmt="tf", name_scope=scope) == prefix
        idx = C.read_index(prefix)
        assert "dgcnn1/bn/%s/dgcnn1/bn/moments/Squeeze/ExponentialMovingAverage" % scope in idx
        assert "dgcnn_output/weights/Adam_1" in idx and idx["Variable"]["shape"] == ()
        h = T.TrainGraph({"num_point": 128, "gpu": 0}, {}, {"batch_size": 4}, seed=99)
        assert not torch.equal(h.store.flat_params, g.store.flat_params)
        h.restore(prefix)
        for a, b in ((h.store.flat_params, g.store.flat_params), (h.store.flat_state, g.store.flat_state),
                     (h.adam_m, g.adam_m), (h.adam_v, g.adam_v), (h.batch, g.batch), (h.beta1_power, g.beta1_power),
                     (h.beta2_power, g.beta2_power), (h.bn_decay, g.bn_decay)):
            assert torch.equal(a, b)
