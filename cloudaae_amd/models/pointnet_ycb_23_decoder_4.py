"""Model builders -- mirror of the reference's models/pointnet_ycb_23_decoder_4.py.

Same function names, arguments, variable scopes and return tuples; tensors are torch
CUDA tensors, `is_training*` are Python bools.  The encoder is composed from the fused
HIP blocks of utils/tf_util.py (kNN without the N^2 matrix, edge-conv without the
k-fold edge tensor, agg conv + BN + ReLU + pool in one pass) instead of the
reference's op-by-op graph; the arithmetic definition is the reference's
(SURVEY.md Appendix A/B).

  get_model_pn                 models/...:23-89     (PointNet encoder)
  get_model_dgcnn              models/...:93-207    (max pooling, reconstruction only)
  get_model_dgcnn_mean         models/...:210-324
  get_model_dgcnn_mean_6d      models/...:327-455   (the variant both reference scripts call)
  get_model_dgcnn_mean_6d_hand models/...:458-589   (N x 5 decoder output)
  get_model_dgcnn_max_6d       models/...:592-723   (reduce_max instead of reduce_mean)
  get_model_dgcnn_mean_6d_2    models/...:726-856   (variables under 'model2/')
  get_model_dgcnn_mean_vae     models/...:859-984
"""
import torch

from .. import _lib
from ..utils import _functions as F
from ..utils import tf_util


def _dgcnn_encoder(point_cloud, is_training_pl_encoder, k, bn_decay, pool, prefix='', nn_out=None):
    """The shared DGCNN encoder (models/...:337-426 and its copies): 4 x { kNN -> edge conv ->
    pool over k }, concat, conv 320->1024, pool over the points.  Returns (embedding [B,1024],
    lazy [B,N,1,1024] activation)."""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    # net1..net4 are written as adjacent channel slices of ONE [B,N,320] buffer: that is the
    # tf.concat([net1, net2, net3, net4], axis=-1) of :410 without a copy
    widths = (64, 64, 64, 128)
    slot = F.ConcatSlot(_lib.empty((batch_size, num_point, sum(widths)), dtype=torch.float32,
                                   device=point_cloud.device))
    net = point_cloud                      # [B,N,C]; kNN metric = xyz slice (tf_util.py:608)
    nets, off = [], 0
    nn_idx = None
    for i, width in enumerate(widths):
        adj_matrix = tf_util.pairwise_xyz_distance(net)
        # (the lists of the layer before go along as a hint: they bound this layer's k-th distance, the result is the same)
        nn_idx = tf_util.knn(adj_matrix, k=k, hint=nn_idx)
        if nn_out is not None:
            nn_out.append(nn_idx)
        net = tf_util.edge_conv(net, nn_idx, width, scope='%sdgcnn%d' % (prefix, i + 1), pool=pool,
                                bn_decay=bn_decay, is_training=is_training_pl_encoder,
                                out_slot=(slot, off),         # [B,N,1,width]
                                in_slot=(slot, off - widths[i - 1]) if i > 0 else None)
        nets.append(net)
        off += width
    return tf_util.conv2d_concat(nets, 1024, scope=prefix + 'dgcnn_agg', bn_decay=bn_decay,
                                 is_training=is_training_pl_encoder, pool=pool, slot=slot)


def _decoder_chain(out_units, prefix='', stem='dgcnn'):
    # FC decoder (models/...:413-421): 1024 -> 1024 -> num_point*4*3, batch norm + ReLU on the first two
    return [(prefix + stem + '_fc1', 1024, True), (prefix + stem + '_fc2', 1024, True),
            (prefix + stem + '_output', out_units, False)]


def _pose_chains(prefix='', stem='dgcnn'):
    # rotation and translation heads (models/...:424-441): 512 -> 256 -> 3 each
    return [[(prefix + '%s_%s_fc1' % (stem, h), 512, True), (prefix + '%s_%s_fc2' % (stem, h), 256, True),
             (prefix + '%s_output_%s' % (stem, h), 3, False)] for h in ('rot', 'trans')]


def _decoder(net, out_units, is_training, bn_decay, prefix=''):
    return tf_util.fully_connected_chains(net, [_decoder_chain(out_units, prefix)], bn_decay=bn_decay,
                                          is_training=is_training)[0]


def _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, pool, prefix='',
              point_out=(4, 3)):
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    end_points = {}
    nn = []
    embedding, before = _dgcnn_encoder(point_cloud, is_training_pl_encoder, k_neighbor, bn_decay, pool, prefix, nn)
    for i, idx in enumerate(nn):
        end_points['nn_idx%d' % (i + 1)] = idx              # [B,N,k] int32 grouping indices (extra key)
    end_points['layer_before_embedding'] = before          # lazy [B,N,1,1024] (see LazyActivation)
    end_points['embedding'] = embedding                    # [B,1024]
    mult, dim = point_out
    # decoder and both heads read the embedding: three chains, evaluated depth by depth
    net, net_rot, net_trans = tf_util.fully_connected_chains(
        embedding, [_decoder_chain(num_point * mult * dim, prefix)] + _pose_chains(prefix), bn_decay=bn_decay,
        is_training=is_training, point_outputs={0: dim, 2: 3})
    net_recon = net.reshape(batch_size, num_point * mult, dim)
    return net_recon, net_rot, net_trans, end_points


def get_model_dgcnn_mean_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay=None):
    """DGCNN encoder (mean pooling) + FC decoder + rot/trans heads (models/...:327-455).
    point_cloud: BxNxC (xyz + one-hot class); returns
    (net_recon [B,4N,3], net_rot [B,3], net_trans [B,3], end_points)."""
    return _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, 'mean')


def get_model_dgcnn_max_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay=None):
    """Same with reduce_max in place of every reduce_mean (models/...:592-723)."""
    return _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, 'max')


def get_model_dgcnn_mean_6d_hand(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay=None):
    """models/...:458-589: as mean_6d, but the decoder emits num_point x (3+2) values."""
    return _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, 'mean',
                     point_out=(1, 5))


def get_model_dgcnn_mean_6d_2(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay=None):
    """models/...:726-856: a second copy of mean_6d whose variables live under 'model2/'."""
    return _dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, bn_decay, 'mean',
                     prefix='model2/')


def get_model_dgcnn(point_cloud, is_training, bn_decay=None):
    """models/...:93-207: max-pooling DGCNN auto-encoder, k = 10, reconstruction only:
    returns (net [B,4N,3], end_points)."""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    end_points = {}
    embedding, _ = _dgcnn_encoder(point_cloud, is_training, 10, bn_decay, 'max')
    net = _decoder(embedding, num_point * 3 * 4, is_training, bn_decay)
    return net.reshape(batch_size, num_point * 4, 3), end_points


def get_model_dgcnn_mean(point_cloud, is_training, bn_decay=None):
    """models/...:210-324: the same with mean pooling."""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    end_points = {}
    embedding, _ = _dgcnn_encoder(point_cloud, is_training, 10, bn_decay, 'mean')
    net = _decoder(embedding, num_point * 3 * 4, is_training, bn_decay)
    return net.reshape(batch_size, num_point * 4, 3), end_points


def get_model_dgcnn_mean_vae(point_cloud, is_training, bn_decay=None, noise=None):
    """models/...:859-984: mean-pooling encoder, z = z_mean + z_std * N(0,1) (:953), FC decoder.
    Returns (net [B,4N,3], z_mean, z_std, end_points).  `noise` ([B,1024]) may be passed for
    reproducibility; otherwise it is drawn on the device like tf.random_normal."""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    end_points = {}
    pooled, _ = _dgcnn_encoder(point_cloud, is_training, 10, bn_decay, 'mean')
    z_mean, _, _ = tf_util.fully_connected(pooled, 1024, bn=True, is_training=is_training,
                                           scope='dgcnn_z_mean', bn_decay=bn_decay)
    z_std, _, _ = tf_util.fully_connected(pooled, 1024, bn=True, is_training=is_training,
                                          scope='dgcnn_z_std', bn_decay=bn_decay)
    if noise is None:
        noise = torch.randn(z_mean.shape, dtype=torch.float32, device=z_mean.device)
    net = tf_util.mul_add(z_mean, z_std, noise)
    end_points['embedding'] = net
    net = _decoder(net, num_point * 3 * 4, is_training, bn_decay)
    return net.reshape(batch_size, num_point * 4, 3), z_mean, z_std, end_points


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

    # FC decoder (models/...:63-70) and the 6d pose heads (:73-86): three chains over the embedding,
    # evaluated depth by depth
    net, net_rot, net_trans = tf_util.fully_connected_chains(
        embedding,
        [[('pn_fc1_decoder', 1024, True), ('pn_fc2_decoder', 1024, True), ('pn_output', num_point * 3 * 4, False)],
         [('pn_rot_fc1', 512, True), ('pn_rot_fc2', 256, True), ('pn_output_rot', 3, False)],
         [('pn_trans_fc1', 512, True), ('pn_trans_fc2', 256, True), ('pn_output_trans', 3, False)]],
        bn_decay=bn_decay, is_training=is_training, point_outputs={0: 3, 2: 3})
    net_recon = net.reshape(batch_size, num_point * 4, 3)

    return net_recon, net_rot, net_trans, end_points
