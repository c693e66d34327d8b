"""Layer library -- mirror of the reference's utils/tf_util.py for the functions the
CloudAAE scripts call (conv2d :111-179, fully_connected :321-365,
batch_norm_* :473-570, pairwise_xyz_distance :597-618, knn :621-632,
get_edge_feature :635-669), hosted on torch tensors and backed by libcloudaae_hip.so.

Same names, argument meaning and return values as the reference; differences that a
torch host forces:
  * `is_training` is a Python bool (TF: a bool placeholder); `bn_decay` is a float or
    a 1-element device tensor (TF: float or float tensor);
  * variables live in a VariableStore (utils/variables.py) under the reference's
    scoped names instead of a TF graph collection;
  * `activation_fn` is `tf_util.relu` or None.
`edge_conv` is the fused form of get_edge_feature -> conv2d(bn) -> reduce_mean/max that
the model builders use; the unfused functions remain available.
"""
import os

import torch

from .. import _lib
from .._lib import ptr, require, stream
from . import _functions as F
from .variables import VariableStore, default_store, reset_default_store, set_default_store  # noqa: F401

relu = "relu"   # stands for tf.nn.relu in `activation_fn=`


def variable_scope(name):
    return default_store().variable_scope(name)


def _variable_on_cpu(name, shape, initializer, use_fp16=False, trainable=None):
    """tf_util.py:10-22.  (The reference pins variables to /cpu:0 and re-uploads them
    every step; here they live in HBM, in the store's flat buffer.)"""
    require(not use_fp16, "fp16 variables are not supported")
    return default_store().get_variable(name, shape, initializer, trainable=(trainable is not False))


def _variable_with_weight_decay(name, shape, stddev, wd, use_xavier=True, trainable=None, fan=None):
    """tf_util.py:24-50.  Xavier-uniform (tf.contrib.layers.xavier_initializer) or
    truncated normal.  The reference always passes weight_decay=0.0, which only adds a
    constant-zero term to an unused collection; a non-zero value is rejected."""
    if use_xavier:
        fan_in, fan_out = fan
        init = VariableStore.xavier_uniform(fan_in, fan_out)
    else:
        init = VariableStore.truncated_normal(stddev)
    require(not wd, "weight decay is not used by the CloudAAE scripts and is not supported")
    return _variable_on_cpu(name, shape, init, trainable=trainable)


def _decay_tensor(bn_decay):
    store = default_store()
    if bn_decay is None:
        return store.scalar(0.9)          # tf_util.py:493
    if torch.is_tensor(bn_decay):
        return bn_decay.reshape(1)
    return store.scalar(float(bn_decay))


def _bn_variables(num_channels):
    """beta/gamma (tf_util.py:488-491) and the EMA shadows of the batch moments
    (tf_util.py:493-500; TF names them .../bn/moments/Squeeze[_1]/ExponentialMovingAverage)."""
    beta = _variable_on_cpu("beta", [num_channels], VariableStore.constant(0.0))
    gamma = _variable_on_cpu("gamma", [num_channels], VariableStore.constant(1.0))
    ema_mean = _variable_on_cpu("moments/Squeeze/ExponentialMovingAverage", [num_channels],
                                VariableStore.constant(0.0), trainable=False)
    ema_var = _variable_on_cpu("moments/Squeeze_1/ExponentialMovingAverage", [num_channels],
                               VariableStore.constant(0.0), trainable=False)
    return beta, gamma, ema_mean, ema_var


def batch_norm_template(inputs, is_training, scope, moments_dims, bn_decay, _relu=False):
    """ Batch normalization on convolutional maps and beyond... (tf_util.py:473-511)

    Args:
        inputs:        Tensor, k-D input ... x C could be BC or BHWC or BDHWC
        is_training:   bool, true indicates training phase
        scope:         string, variable scope
        moments_dims:  a list of ints, indicating dimensions for moments calculation
        bn_decay:      float or float tensor variable, controling moving average weight
    Return:
        normed:        batch-normalized maps
    """
    require(list(moments_dims) == list(range(inputs.dim() - 1)),
            "moments_dims must be all axes but the last (as every reference call site has)")
    C = inputs.shape[-1]
    with variable_scope(scope):
        beta, gamma, ema_mean, ema_var = _bn_variables(C)
    x2 = inputs.reshape(-1, C)
    out, _, _ = F.BatchNormFn.apply(x2, gamma.data, beta.data, ema_mean.data, ema_var.data,
                                    _decay_tensor(bn_decay), bool(is_training), bool(_relu), 0, 0, True)
    return out.reshape(inputs.shape)


def batch_norm_for_fc(inputs, is_training, bn_decay, scope):
    """tf_util.py:514-525"""
    return batch_norm_template(inputs, is_training, scope, [0, ], bn_decay)


def batch_norm_for_conv1d(inputs, is_training, bn_decay, scope):
    """tf_util.py:528-539"""
    return batch_norm_template(inputs, is_training, scope, [0, 1], bn_decay)


def batch_norm_for_conv2d(inputs, is_training, bn_decay, scope):
    """tf_util.py:544-555"""
    return batch_norm_template(inputs, is_training, scope, [0, 1, 2], bn_decay)


def _check_activation(activation_fn):
    require(activation_fn in (relu, None), "activation_fn must be tf_util.relu or None")
    return activation_fn == relu


def conv2d(inputs,
           num_output_channels,
           kernel_size,
           scope,
           stride=[1, 1],
           padding='SAME',
           use_xavier=True,
           stddev=1e-3,
           weight_decay=0.0,
           activation_fn=relu,
           bn=False,
           bn_decay=None,
           is_training=None,
           trainable=None):
    """ 2D convolution with non-linear operation (tf_util.py:111-179).

    Args:
      inputs: 4-D tensor variable BxHxWxC
      num_output_channels: int
      kernel_size: a list of 2 ints -- [1,1] (every DGCNN call site) or [1,W] with
                   padding 'VALID' on a BxHxWx1 input (PointNet's first layer,
                   models/pointnet_ycb_23_decoder_4.py:39); both are a GEMM over rows
      scope: string
      ...
    Returns:
      Variable tensor BxHx1xCout / BxHxWxCout
    """
    require(inputs.dim() == 4, "conv2d: inputs must be BxHxWxC")
    require(list(stride) == [1, 1], "conv2d: only stride [1,1] is used by CloudAAE")
    kernel_h, kernel_w = kernel_size
    B, H, W, Cin = inputs.shape
    act = _check_activation(activation_fn)
    with variable_scope(scope):
        kernel = _variable_with_weight_decay('weights', [kernel_h, kernel_w, Cin, num_output_channels],
                                             stddev=stddev, wd=weight_decay, use_xavier=use_xavier,
                                             trainable=trainable,
                                             fan=(kernel_h * kernel_w * Cin, kernel_h * kernel_w * num_output_channels))
        biases = _variable_on_cpu('biases', [num_output_channels], VariableStore.constant(0.0),
                                  trainable=trainable)
        if kernel_h == 1 and kernel_w == 1:
            rows = inputs.reshape(B * H * W, Cin)
            out_shape = (B, H, W, num_output_channels)
        else:
            require(kernel_h == 1 and kernel_w == W and padding == 'VALID',
                    "conv2d: only 1x1 and full-width [1,W] VALID kernels are supported")
            rows = inputs.reshape(B * H, W * Cin)
            out_shape = (B, H, 1, num_output_channels)
        w2 = kernel.data.reshape(-1, num_output_channels)
        w2._cloudaae_var = kernel
        outputs = F.LinearFn.apply(rows, w2, biases.data, bool(bn))
        if bn:
            with variable_scope('bn'):
                beta, gamma, ema_mean, ema_var = _bn_variables(num_output_channels)
            outputs, _, _ = F.BatchNormFn.apply(outputs, gamma.data, beta.data, ema_mean.data, ema_var.data,
                                                _decay_tensor(bn_decay), bool(is_training), act, 0, 0, True,
                                                biases.data)
        elif act:
            outputs = _relu_rows(outputs)
    return outputs.reshape(out_shape)


def fully_connected(inputs,
                    num_outputs,
                    scope,
                    use_xavier=True,
                    stddev=1e-3,
                    weight_decay=0.0,
                    activation_fn=relu,
                    bn=False,
                    bn_decay=None,
                    is_training=None,
                    trainable=None):
    """ Fully connected layer with non-linear operation (tf_util.py:321-365).

    Args:
      inputs: 2-D tensor BxN
      num_outputs: int

    Returns:
      (outputs B x num_outputs, weights, biases)
    """
    require(inputs.dim() == 2, "fully_connected: inputs must be BxN")
    act = _check_activation(activation_fn)
    weights, biases, gamma, beta, ema_mean, ema_var = _fc_variables(
        scope, inputs.shape[-1], num_outputs, bn, use_xavier=use_xavier, stddev=stddev, weight_decay=weight_decay,
        trainable=trainable)
    # a batch of <= 128 clouds (cloudaae_fc_max_rows): the whole layer is one launch (not with SyncBN in training mode: the moments
    # leave for the other ranks between the product and the normalisation)
    if F.fc_fits(inputs.shape[0]) and not (bn and is_training and F.BN_SYNC is not None):
        require(bn or not act, "fully_connected: ReLU without batch norm does not occur in CloudAAE")
        outputs = F.FcFn.apply(inputs, weights, biases, gamma, beta, ema_mean, ema_var,
                               _decay_tensor(bn_decay) if bn else None, bool(is_training), act)
        return outputs, weights, biases
    outputs = F.LinearFn.apply(inputs, weights, biases, bool(bn), False)   # FC stack: always fp32
    if bn:
        outputs, _, _ = F.BatchNormFn.apply(outputs, gamma, beta, ema_mean, ema_var, _decay_tensor(bn_decay),
                                            bool(is_training), act, 0, 0, True, biases)
    elif act:
        outputs = _relu_rows(outputs)
    return outputs, weights, biases


def _fc_variables(scope, num_inputs, num_outputs, bn, use_xavier=True, stddev=1e-3, weight_decay=0.0,
                  trainable=None):
    """The variables of one fully_connected layer, created (first call) in the reference's order
    (tf_util.py:345-355): weights, biases, then bn/beta, bn/gamma and the two EMA shadows."""
    gamma = beta = ema_mean = ema_var = None
    with variable_scope(scope):
        weights = _variable_with_weight_decay('weights', [num_inputs, num_outputs], stddev=stddev,
                                              wd=weight_decay, use_xavier=use_xavier, trainable=trainable,
                                              fan=(num_inputs, num_outputs))
        biases = _variable_on_cpu('biases', [num_outputs], VariableStore.constant(0.0), trainable=trainable)
        if bn:
            with variable_scope('bn'):
                beta, gamma, ema_mean, ema_var = _bn_variables(num_outputs)
            beta.tag = gamma.tag = 'fc'
            beta, gamma, ema_mean, ema_var = beta.data, gamma.data, ema_mean.data, ema_var.data
    # (the data-parallel exchange reduces the fully connected stack -- 97 % of the parameters, and the
    # first gradients backward produces -- as one early piece)
    weights.tag = biases.tag = 'fc'
    return weights.data, biases.data, gamma, beta, ema_mean, ema_var


def fully_connected_chains(inputs, chains, bn_decay=None, is_training=None, point_outputs=None):
    """Several independent chains of fully_connected layers over one input -- the decoder and the two
    pose heads of models/pointnet_ycb_23_decoder_4.py:413-455 -- evaluated depth by depth: with a batch of
    <= 128 clouds the layers of one depth share a launch per direction (F.FcGroupFn), otherwise each layer
    runs as fully_connected() does.  chains: list of chains; a chain is a list of
    (scope, num_outputs, bn) with ReLU exactly on the bn layers.  Variables are created chain by chain,
    i.e. in the order separate fully_connected() calls would create them.  Returns the chain outputs.
    point_outputs: {chain index: d} -- the MODEL's statement that this chain's output is a list of d-vectors in
    camera coordinates (reconstruction [B, rows * d], translation [B, d]); only such chains may take the training
    step's offer F.FC_OUT_ADD (a [B, d] row vector added in the output layer's epilogue, train...:232-233)."""
    require(inputs.dim() == 2, "fully_connected_chains: inputs must be BxN")
    if not F.fc_fits(inputs.shape[0]) or len(chains) > F.fc_max_group() or (is_training and F.BN_SYNC is not None):
        outs = []
        branches = F.FanOutFn.apply(inputs, len(chains)) if len(chains) > 1 else (inputs,)
        for net, chain in zip(branches, chains):
            for scope, num_outputs, bn in chain:
                net, _, _ = fully_connected(net, num_outputs, scope, bn=bn, is_training=is_training,
                                            bn_decay=bn_decay, activation_fn=relu if bn else None)
                if FC_TAP is not None:
                    FC_TAP[scope] = net
            outs.append(net)
        return outs
    variables = []
    for chain in chains:
        cin, row = inputs.shape[-1], []
        for scope, num_outputs, bn in chain:
            row.append(_fc_variables(scope, cin, num_outputs, bn))
            cin = num_outputs
        variables.append(row)
    decay = _decay_tensor(bn_decay) if any(bn for chain in chains for _, _, bn in chain) else None
    depth = max(len(c) for c in chains)
    # F.FC_OUT_ADD: the [B, d] row vector the step wants added to the model's point outputs.  Taken only when the model
    # declared which chains those are (point_outputs), every one of them has width d per point (a decoder of 5-vectors
    # does not take a 3-vector) and ends in a layer without batch norm (the kernel adds to a plain output only);
    # otherwise the offer stays and the caller adds the vector itself (F.AddRowVecFn, which checks the shapes)
    vec = F.FC_OUT_ADD
    offer = None
    if vec is not None and point_outputs and vec.dim() == 2 and vec.shape[0] == inputs.shape[0] and \
            all(0 <= i < len(chains) and not chains[i][-1][2] and int(d) == vec.shape[1] and chains[i][-1][1] % int(d) == 0
                for i, d in point_outputs.items()):
        offer = (vec, tuple(sorted(point_outputs)))
        F.FC_OUT_ADD = None
    nets = [None] * len(chains)
    for d in range(depth):
        members = [i for i, c in enumerate(chains) if d < len(c)]
        if d == 0:
            xs, x_index = [inputs], [0] * len(members)      # one input: the consumers' gradients add up in place
        else:
            xs, x_index = [nets[i] for i in members], list(range(len(members)))
        flat = []
        for i in members:
            flat.extend(variables[i][d])
        cfg = (len(xs), tuple(x_index), bool(is_training), tuple(bool(chains[i][d][2]) for i in members))
        if offer is not None and any(d == len(chains[i]) - 1 and i in offer[1] for i in members):
            # the output layers of the chains named by F.FC_OUT_ADD add its row vector in their epilogue
            cfg = cfg + (tuple(offer[0] if (d == len(chains[i]) - 1 and i in offer[1]) else None for i in members),)
        outs = F.FcGroupFn.apply(cfg, decay, *(xs + flat))
        for i, o in zip(members, outs):
            nets[i] = o
            if FC_TAP is not None:
                FC_TAP[chains[i][d][0]] = o
    return nets


# Debugging / test aid: a dict that fully_connected_chains fills with scope -> output of every layer it evaluates (the
# tensors the next layer reads: arena buffers of a recorded step stay valid afterwards).  tests/test_03_configs_gpu.py
# takes the activation pattern of the fully connected stack from it; None (the default) = nothing is kept.
FC_TAP = None
KNN_TAP = None      # a list: cloudaae_knn's C = 64 inputs are appended (development)


def _relu_rows(x):
    # ReLU without batch norm does not occur in the CloudAAE graphs (every activated layer
    # has bn=True); route it through the BN kernel with identity statistics.
    C = x.shape[-1]
    dev = x.device
    one = torch.ones(C, dtype=torch.float32, device=dev)
    zero = torch.zeros(C, dtype=torch.float32, device=dev)
    var = torch.full((C,), 1.0 - 1e-3, dtype=torch.float32, device=dev)   # rsqrt(var + eps) == 1
    out, _, _ = F.BatchNormFn.apply(x.reshape(-1, C), one, zero, zero, var, None, False, True, 0, 0, True)
    return out.reshape(x.shape)


class PairwiseDistance(object):
    """What `pairwise_xyz_distance` returns: the [B,N,N] matrix of tf_util.py:618
    in LAZY form.  The reference materialises it only to feed `knn`; here `knn`
    consumes the lazy form with the fused kernel (cloudaae_knn) and the matrix is
    never written."""

    def __init__(self, points, channels):
        self.points = points      # [B, N, ld] fp32 whose rows are contiguous
        self.channels = channels  # leading channels that form the metric

    @property
    def shape(self):
        b, n, _ = self.points.shape
        return (b, n, n)


def pairwise_xyz_distance(point_cloud):
    """Compute pairwise distance of a point cloud (tf_util.py:597-618).

    Args:
      point_cloud: tensor (batch_size, num_points, num_dims)  or
                   (batch_size, num_points, 1, num_dims)
    Returns:
      pairwise distance: (batch_size, num_points, num_points)  [lazy]

    tf_util.py:608 slices `[:, :, 0:3]`: on a 3-D input that keeps xyz only; on the
    4-D `[B,N,1,C]` tensors the later layers pass it hits the size-1 axis and keeps
    all C channels.  Both behaviours are reproduced.
    """
    require(point_cloud.dtype == torch.float32, "pairwise_xyz_distance: float32 expected")
    if point_cloud.dim() == 4:
        require(point_cloud.shape[2] == 1, "pairwise_xyz_distance: expected [B,N,1,C]")
        pts = point_cloud[:, :, 0, :]
        channels = pts.shape[2]
    else:
        require(point_cloud.dim() == 3, "pairwise_xyz_distance: expected [B,N,C]")
        pts = point_cloud
        channels = min(3, pts.shape[2])
    pts = pts.detach()
    B, N, _ = pts.shape
    if not (pts.stride(2) == 1 and (B == 1 or pts.stride(0) == N * pts.stride(1))):
        pts = pts.contiguous()
    return PairwiseDistance(pts, channels)


# hint=...: which layers may take the neighbour lists of the layer before as a bound (cloudaae_knn_hinted).  The RESULT
# does not depend on it; it pays where the bound pass of the plain kernel is long (k = 20 at 4096 points: BASELINE
# configs[4]) and costs a launch where it is short.  None = by shape (KNN_HINT_MIN_WORK), True / False = always / never.
KNN_HINT = {"0": False, "1": True}.get(os.environ.get("CLOUDAAE_KNN_HINT", ""), None)
KNN_HINT_MIN_WORK = 4096 * 20


def knn(adj_matrix, k=9, hint=None):
    """Get KNN based on the pairwise distance (tf_util.py:621-632).
    Args:
      pairwise distance: (batch_size, num_points, num_points)
      k: int
      hint (not in the reference): (batch_size, num_points, k) int32, k distinct indices per point that are likely to be
        near it -- the result of this function on the previous layer's features.  Same result with or without.

    Returns:
      nearest neighbors: (batch_size, num_points, k)   int32, ascending distance,
      ties -> lower index (tf.nn.top_k of the negated matrix, tf_util.py:630-631)
    """
    require(isinstance(adj_matrix, PairwiseDistance),
            "knn expects the result of pairwise_xyz_distance")
    x = adj_matrix.points
    b, n, _ = x.shape
    ld = x.stride(1)
    nn_idx = _lib.empty((b, n, int(k)), dtype=torch.int32, device=x.device)
    # bench.py times a launch over 64 feature channels live (F.TIMED_SITES["knn64"]): one of every three -- layers
    # 2-4 launch the same shape, and every event pair costs the step a few microseconds of stream markers
    if KNN_TAP is not None and adj_matrix.channels == 64:       # (development: tools/dev/knn_step_data.py looks at the features)
        KNN_TAP.append(x.detach().clone())
    rec = F.TIMED_SITES.get("knn64") if adj_matrix.channels == 64 else None
    if rec is not None:
        F.KNN64_SEEN += 1
        if F.KNN64_SEEN % 3 != 1:
            rec = None
    if rec is not None:
        _lib.host(F._mark, rec)
    use_hint = (hint is not None and adj_matrix.channels == 64 and tuple(hint.shape) == (b, n, int(k)) and
                (KNN_HINT if KNN_HINT is not None else n * int(k) >= KNN_HINT_MIN_WORK))
    if use_hint:
        tau = _lib.empty((b, n), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().cloudaae_knn_hinted(b, n, adj_matrix.channels, ld, int(k), x.data_ptr(), ptr(hint), ptr(tau),
                                                  ptr(nn_idx), stream()), "cloudaae_knn_hinted")
    else:
        _lib.check(_lib.lib().cloudaae_knn(b, n, adj_matrix.channels, ld, int(k), x.data_ptr(), ptr(nn_idx),
                                           stream()), "cloudaae_knn")
    if rec is not None:
        _lib.host(F._mark, rec)
    return nn_idx


def edge_conv(point_cloud, nn_idx, num_output_channels, scope, pool='mean', bn_decay=None, is_training=None,
              out_slot=None, in_slot=None):
    """Fused  get_edge_feature(point_cloud, nn_idx, k)            (tf_util.py:635-669)
              -> conv2d(., num_output_channels, [1,1], bn=True)   (tf_util.py:111-179)
              -> tf.reduce_mean / tf.reduce_max(axis=-2, keep_dims=True)
    i.e. one DGCNN block of models/pointnet_ycb_23_decoder_4.py:337-350 (mean) / :605-615 (max).
    Variables are created under `scope` exactly as conv2d would ('weights' [1,1,2C,Cout],
    'biases', 'bn/beta', 'bn/gamma').

    Args:
      point_cloud: (batch_size, num_points, num_dims) or (batch_size, num_points, 1, num_dims)
      nn_idx: (batch_size, num_points, k) int32
      pool: 'mean' or 'max'
      out_slot: optional (buffer [B,N,Ctot] or ConcatSlot, channel offset) to write the result into
      in_slot: optional (ConcatSlot, channel offset) saying point_cloud IS that slice of the slot
    Returns:
      (batch_size, num_points, 1, num_output_channels)
    """
    require(pool in ('mean', 'max'), "pool must be 'mean' or 'max'")
    x = point_cloud[:, :, 0, :] if point_cloud.dim() == 4 else point_cloud
    B, N, C = x.shape
    if not (x.stride(2) == 1 and (B == 1 or x.stride(0) == N * x.stride(1))):
        x = x.contiguous()
    with variable_scope(scope):
        kernel = _variable_with_weight_decay('weights', [1, 1, 2 * C, num_output_channels], stddev=1e-3, wd=0.0,
                                             use_xavier=True, fan=(2 * C, num_output_channels))
        biases = _variable_on_cpu('biases', [num_output_channels], VariableStore.constant(0.0))
        with variable_scope('bn'):
            beta, gamma, ema_mean, ema_var = _bn_variables(num_output_channels)
    w2 = kernel.data.reshape(2 * C, num_output_channels)
    w2._cloudaae_var = kernel
    out = F.EdgeConvFn.apply(x, nn_idx, w2, biases.data, gamma.data, beta.data, ema_mean.data, ema_var.data,
                             _decay_tensor(bn_decay), bool(is_training), 1 if pool == 'mean' else 2, out_slot,
                             in_slot)
    return out.unsqueeze(2)


def conv2d_concat(inputs_list, num_output_channels, scope, bn_decay=None, is_training=None, pool=None, slot=None):
    """conv2d(tf.concat(inputs_list, axis=-1), C, [1,1], bn=True) followed, when `pool` is
    'mean'/'max', by the reduction over the point axis (models/...:410-419 / :675-684) --
    without the concat copy when the inputs are adjacent slices of one buffer, and without
    writing the [B,N,1,C] activation when only the pooled embedding is consumed.
    Returns (pooled [B,C] or None, lazy activation)."""
    B, N = inputs_list[0].shape[0], inputs_list[0].shape[1]
    rows = [t.reshape(B * N, t.shape[-1]) for t in inputs_list]
    cin = sum(r.shape[1] for r in rows)
    with variable_scope(scope):
        kernel = _variable_with_weight_decay('weights', [1, 1, cin, num_output_channels], stddev=1e-3, wd=0.0,
                                             use_xavier=True, fan=(cin, num_output_channels))
        biases = _variable_on_cpu('biases', [num_output_channels], VariableStore.constant(0.0))
        with variable_scope('bn'):
            beta, gamma, ema_mean, ema_var = _bn_variables(num_output_channels)
    w2 = kernel.data.reshape(cin, num_output_channels)
    w2._cloudaae_var = kernel
    # (flag 1: the batch norm writes the bias gradient; 2: it runs in training mode right after, so the
    # product leaves it the column sums of y)
    mode = {None: 0, 'mean': 1, 'max': 2}[pool]
    # (flag 4: the consumer is the training-mode batch norm + ReLU + MEAN pool -- with bf16 operands the product may
    # then keep y as bfloat16, F.ACT_BF16; csrc/bn16.hip pools groups that are multiples of 64 rows)
    y = F.ConcatLinearFn.apply(slot, w2, biases.data, (3 if is_training else 1) | (4 if (is_training and mode == 1 and N % 64 == 0) else 0),
                               *rows)
    if mode == 0:
        act, mean, var = F.BatchNormFn.apply(y, gamma.data, beta.data, ema_mean.data, ema_var.data,
                                             _decay_tensor(bn_decay), bool(is_training), True, 0, 0, True,
                                             biases.data)
        return None, act.reshape(B, N, 1, num_output_channels)
    pooled, mean, var = F.BatchNormFn.apply(y, gamma.data, beta.data, ema_mean.data, ema_var.data,
                                            _decay_tensor(bn_decay), bool(is_training), True, N, mode, False,
                                            biases.data)
    return pooled, LazyActivation(y, mean, var, gamma.data, beta.data, (B, N, 1, num_output_channels))


class LazyActivation(object):
    """relu(batch_norm(y)) that is only written out when somebody asks for it
    (end_points['layer_before_embedding'], models/...:418; nothing in the training graph
    reads it, and at B=32, N=1024 it is a 134 MB tensor)."""

    def __init__(self, y, mean, var, gamma, beta, shape):
        self._y, self._mean, self._var, self._gamma, self._beta = y.detach(), mean, var, gamma.detach(), beta.detach()
        self.shape = tuple(shape)

    def tensor(self):
        if self._y.dtype == torch.bfloat16:     # y was stored as bfloat16 (F.ACT_BF16): the fp32 kernel reads a widened copy
            self._y = self._y.to(torch.float32)
        out, _, _ = F.BatchNormFn.apply(self._y, self._gamma, self._beta, self._mean, self._var, None, False, True,
                                        0, 0, True)
        return out.reshape(self.shape)


def _points3(point_cloud):
    x = point_cloud[:, :, 0, :] if point_cloud.dim() == 4 else point_cloud      # tf.squeeze, :645
    B, N, _ = x.shape
    if not (x.stride(2) == 1 and (B == 1 or x.stride(0) == N * x.stride(1))):
        x = x.contiguous()
    return x


def get_edge_feature(point_cloud, nn_idx, k):
    """Construct edge feature for each point (tf_util.py:635-669)
    Args:
      point_cloud: (batch_size, num_points, num_dims)  [or (batch_size, num_points, 1, num_dims)]
      nn_idx: (batch_size, num_points, k)
      k: int
    Returns:
      edge features: (batch_size, num_points, k, 2*num_dims) = concat(central, neighbours - central)
    Unfused form (materialises the k-fold tensor); the model builders use `edge_conv`."""
    require(nn_idx.shape[2] == k, "get_edge_feature: nn_idx last dim != k")
    return F.EdgeFeatureFn.apply(_points3(point_cloud), nn_idx, True)


def get_edge_feature_wo_center(point_cloud, nn_idx, k):
    """tf_util.py:672-706: (batch_size, num_points, k, num_dims) = neighbours - central."""
    require(nn_idx.shape[2] == k, "get_edge_feature_wo_center: nn_idx last dim != k")
    return F.EdgeFeatureFn.apply(_points3(point_cloud), nn_idx, False)


def mul_add(a, b, c):
    """a + b * c with `c` a constant tensor (VAE reparameterisation, models/...:953)."""
    return F.MulAddFn.apply(a, b, c)


def _reduce(x, axis, keep_dims, mode):
    nd = x.dim()
    axis = axis % nd
    require(all(x.shape[d] == 1 for d in range(axis + 1, nd - 1)),
            "reduce: only an axis whose trailing axes (before the channels) have size 1 is supported")
    C = x.shape[-1]
    rows = x.shape[axis]
    out = F.PoolRowsFn.apply(x.reshape(-1, C), rows, mode)
    shape = list(x.shape)
    if keep_dims:
        shape[axis] = 1
    else:
        del shape[axis]
    return out.reshape(shape)


def reduce_mean(x, axis, keep_dims=False):
    """tf.reduce_mean over one axis (models/...:350, :419)."""
    return _reduce(x, axis, keep_dims, 1)


def reduce_max(x, axis, keep_dims=False):
    """tf.reduce_max over one axis (models/...:615, :684); the gradient is shared among equal maxima."""
    return _reduce(x, axis, keep_dims, 2)
