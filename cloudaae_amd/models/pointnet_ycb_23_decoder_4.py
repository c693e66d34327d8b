"""Model builders -- mirror of the reference's models/pointnet_ycb_23_decoder_4.py.

Same function names, arguments, variable scopes and return tuples; tensors are torch
CUDA tensors, `is_training*` are Python bools.  The encoder is composed from the fused
HIP blocks of utils/tf_util.py (kNN without the N^2 matrix, edge-conv without the
k-fold edge tensor, agg conv + BN + ReLU + pool in one pass) instead of the
reference's op-by-op graph; the arithmetic definition is the reference's
(SURVEY.md Appendix A/B).

  get_model_dgcnn_mean_6d   models/...:327-455   (the variant both reference scripts call)
  get_model_dgcnn_max_6d    models/...:592-723   (reduce_max instead of reduce_mean)
  get_model_pn              models/...:23-89     (PointNet encoder)
"""
import torch

from ..utils import tf_util


def _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, pool):
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    end_points = {}
    k = k_neighbor

    # net1..net4 are written as adjacent channel slices of ONE [B,N,320] buffer: that is the
    # tf.concat([net1, net2, net3, net4], axis=-1) of :410 without a copy
    widths = (64, 64, 64, 128)
    concat = torch.empty((batch_size, num_point, sum(widths)), dtype=torch.float32, device=point_cloud.device)

    net = point_cloud                      # [B,N,3+classes]; kNN metric = xyz slice (tf_util.py:608)
    nets, off = [], 0
    for i, width in enumerate(widths):
        adj_matrix = tf_util.pairwise_xyz_distance(net)
        nn_idx = tf_util.knn(adj_matrix, k=k)
        net = tf_util.edge_conv(net, nn_idx, width, scope='dgcnn%d' % (i + 1), pool=pool,
                                bn_decay=bn_decay, is_training=is_training_pl_encoder,
                                out_slot=(concat, off))      # [B,N,1,width]
        nets.append(net)
        off += width

    embedding, before = tf_util.conv2d_concat(nets, 1024, scope='dgcnn_agg', bn_decay=bn_decay,
                                              is_training=is_training_pl_encoder, pool=pool)
    end_points['layer_before_embedding'] = before          # lazy [B,N,1,1024] (see LazyActivation)
    end_points['embedding'] = embedding                    # [B,1024]

    net, _, _ = tf_util.fully_connected(embedding, 1024, bn=True, is_training=is_training,
                                        scope='dgcnn_fc1', bn_decay=bn_decay)
    net, _, _ = tf_util.fully_connected(net, 1024, bn=True, is_training=is_training,
                                        scope='dgcnn_fc2', bn_decay=bn_decay)
    net, out_weight, out_biases = tf_util.fully_connected(net, num_point * 3 * 4, activation_fn=None,
                                                          scope='dgcnn_output')
    net_recon = net.reshape(batch_size, num_point * 4, 3)

    # 6d pose
    net_rot, _, _ = tf_util.fully_connected(embedding, 512, bn=True, is_training=is_training,
                                            scope='dgcnn_rot_fc1', bn_decay=bn_decay)
    net_rot, _, _ = tf_util.fully_connected(net_rot, 256, bn=True, is_training=is_training,
                                            scope='dgcnn_rot_fc2', bn_decay=bn_decay)
    net_rot, _, _ = tf_util.fully_connected(net_rot, 3, activation_fn=None, scope='dgcnn_output_rot')

    net_trans, _, _ = tf_util.fully_connected(embedding, 512, bn=True, is_training=is_training,
                                              scope='dgcnn_trans_fc1', bn_decay=bn_decay)
    net_trans, _, _ = tf_util.fully_connected(net_trans, 256, bn=True, is_training=is_training,
                                              scope='dgcnn_trans_fc2', bn_decay=bn_decay)
    net_trans, _, _ = tf_util.fully_connected(net_trans, 3, activation_fn=None, scope='dgcnn_output_trans')

    return net_recon, net_rot, net_trans, end_points


def get_model_dgcnn_mean_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay=None):
    """DGCNN encoder (mean pooling) + FC decoder + rot/trans heads (models/...:327-455).
    point_cloud: BxNxC (xyz + one-hot class); returns
    (net_recon [B,4N,3], net_rot [B,3], net_trans [B,3], end_points)."""
    return _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, 'mean')


def get_model_dgcnn_max_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay=None):
    """Same with reduce_max in place of every reduce_mean (models/...:592-723)."""
    return _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, 'max')


def get_model_pn(point_cloud, is_training, bn_decay=None):
    """ Autoencoder for point clouds, PointNet encoder (models/...:23-89).
    Input:
        point_cloud: tensor BxNxC
        is_training: boolean
        bn_decay: float between 0 and 1
    Output:
        (net_recon [B,4N,3], net_rot, net_trans, end_points)
    """
    batch_size, num_point, point_dim = point_cloud.shape
    end_points = {}
    input_image = point_cloud.unsqueeze(-1)
    # Encoder
    net = tf_util.conv2d(input_image, 64, [1, point_dim], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='pn_conv1_encoder', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True, is_training=is_training,
                         scope='pn_conv2_encoder', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True, is_training=is_training,
                         scope='pn_conv3_encoder', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True, is_training=is_training,
                         scope='pn_conv4_encoder', bn_decay=bn_decay)
    # conv5 + BN + ReLU + max_pool2d([num_point,1]) fused (models/...:55-60)
    embedding, _ = tf_util.conv2d_concat([net], 1024, scope='pn_conv5_encoder', bn_decay=bn_decay,
                                         is_training=is_training, pool='max')
    end_points['embedding'] = embedding

    # FC Decoder
    net, _, _ = tf_util.fully_connected(embedding, 1024, bn=True, is_training=is_training,
                                        scope='pn_fc1_decoder', bn_decay=bn_decay)
    net, _, _ = tf_util.fully_connected(net, 1024, bn=True, is_training=is_training,
                                        scope='pn_fc2_decoder', bn_decay=bn_decay)
    net, out_weight, out_biases = tf_util.fully_connected(net, num_point * 3 * 4, activation_fn=None,
                                                          scope='pn_output')
    net_recon = net.reshape(batch_size, num_point * 4, 3)

    # 6d pose
    net_rot, _, _ = tf_util.fully_connected(embedding, 512, bn=True, is_training=is_training,
                                            scope='pn_rot_fc1', bn_decay=bn_decay)
    net_rot, _, _ = tf_util.fully_connected(net_rot, 256, bn=True, is_training=is_training,
                                            scope='pn_rot_fc2', bn_decay=bn_decay)
    net_rot, _, _ = tf_util.fully_connected(net_rot, 3, activation_fn=None, scope='pn_output_rot')

    net_trans, _, _ = tf_util.fully_connected(embedding, 512, bn=True, is_training=is_training,
                                              scope='pn_trans_fc1', bn_decay=bn_decay)
    net_trans, _, _ = tf_util.fully_connected(net_trans, 256, bn=True, is_training=is_training,
                                              scope='pn_trans_fc2', bn_decay=bn_decay)
    net_trans, _, _ = tf_util.fully_connected(net_trans, 3, activation_fn=None, scope='pn_output_trans')

    return net_recon, net_rot, net_trans, end_points
