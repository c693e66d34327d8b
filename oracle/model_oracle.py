"""CPU restatement of the CloudAAE model, losses and training step -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Plain torch-CPU fp32 ops with autograd (fp64 for the rotation loss), written op by op
the way the reference graph is (no fusion, the k-fold edge tensor and the per-layer
[B,N,N] semantics are kept), each function citing the reference lines it restates.

PARITY UNPINNED BY THE REFERENCE: TensorFlow cannot be imported here and the reference
ships no golden vector for any of these functions, so this file
is pinned only by its citations and by property tests.  The kNN indices come from the C
oracle (oracle_knn: the documented summation order), everything differentiable is torch.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import native as O

BN_EPS = 1e-3  # utils/tf_util.py:510


# --------------------------------------------------------------------------------------
# variables: names/order/shapes/initial values of the reference graph
# --------------------------------------------------------------------------------------
class Vars(object):
    """name -> tensor, created on first use in graph-construction order
    (utils/tf_util.py:10-50, 488-500).  Trainables require grad; EMA shadows do not."""

    def __init__(self, seed=0):
        self.p = OrderedDict()      # trainable
        self.s = OrderedDict()      # BN EMA shadows
        self.gen = torch.Generator().manual_seed(seed)

    def get(self, name, shape, kind, fan=None):
        if name in self.p:
            return self.p[name]
        if kind == "xavier":   # tf.contrib.layers.xavier_initializer(): U(+-sqrt(6/(fan_in+fan_out)))
            lim = math.sqrt(6.0 / (fan[0] + fan[1]))
            t = (torch.rand(shape, generator=self.gen) * 2 - 1) * lim
        elif kind == "zeros":
            t = torch.zeros(shape)
        elif kind == "ones":
            t = torch.ones(shape)
        else:
            raise ValueError(kind)
        t = t.float().requires_grad_(True)
        self.p[name] = t
        return t

    def shadow(self, name, shape):
        if name not in self.s:
            self.s[name] = torch.zeros(shape)     # EMA shadows start at 0, no zero-debias
        return self.s[name]

    def state_dict(self):
        d = OrderedDict((k, v.detach().clone()) for k, v in self.p.items())
        d.update((k, v.clone()) for k, v in self.s.items())
        return d


# --------------------------------------------------------------------------------------
# utils/tf_util.py
# --------------------------------------------------------------------------------------
def batch_norm(x, scope, V, is_training, bn_decay, stats_from=None):
    """batch_norm_template, utils/tf_util.py:473-511, moments over all axes but the last.
    stats_from: take the batch moments from this tensor instead of x (ACT_BF16: x is the bfloat16-stored copy of it)."""
    C = x.shape[-1]
    beta = V.get(scope + "/beta", (C,), "zeros")
    gamma = V.get(scope + "/gamma", (C,), "ones")
    sm = V.shadow(scope + "/moments/Squeeze/ExponentialMovingAverage", (C,))
    sv = V.shadow(scope + "/moments/Squeeze_1/ExponentialMovingAverage", (C,))
    flat = (x if stats_from is None else stats_from).reshape(-1, C)
    if is_training:
        mean = flat.mean(0)                                        # tf.nn.moments
        var = ((flat - mean.detach()) ** 2).mean(0)                # biased; stop_gradient(mean) inside
        # the derivative of `var` w.r.t. the mean path vanishes identically, so autograd gives
        # the same gradient as TF's moments op
        decay = np.float32(0.9 if bn_decay is None else bn_decay)  # :493
        om = np.float32(1.0) - decay
        with torch.no_grad():                                      # assign_moving_average
            sm -= (sm - mean.detach()) * float(om)
            sv -= (sv - var.detach()) * float(om)
    else:
        mean, var = sm, sv                                         # :507-509
    inv = gamma * torch.rsqrt(var + BN_EPS)                        # tf.nn.batch_normalization
    return x * inv + (beta - mean * inv)


# ---- BASELINE config 3: dense layers with bf16 operands ------------------------------------------
# GEMM_BF16 = True: every per-point product (conv2d 1x1, edge convolution, dgcnn_agg; NOT the fully
# connected stack) rounds BOTH operands to bfloat16 (round to nearest
# even) and accumulates in fp32, in the forward product AND in the two gradient products
# (dx = bf16(dy) bf16(W)^T, dW = bf16(x)^T bf16(dy)); everything else stays fp32.  The reference has
# no such mode (it is BASELINE.json's config 3), so this IS the definition; the edge convolution is
# then evaluated in the split form y_ij = (x_i Wc - x_i Wn + b) + x_j Wn, because rounding x_j - x_i
# is not rounding x_j and x_i.
GEMM_BF16 = False
# ACT_BF16 = True (with GEMM_BF16, training mode): the output y of dgcnn_agg is STORED as bfloat16 -- the batch norm
# takes its moments from the fp32 y and normalises the rounded copy, the gradient passes the rounding unchanged.
# (The other bf16-stored tensors of that mode -- the concatenated features, W, dy -- are operands of products
# that round them anyway, so they move no rounding point.)
ACT_BF16 = False


class _StoreBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _MatmulBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _bf(x) @ _bf(w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = _bf(dy)
        x2 = _bf(x).reshape(-1, x.shape[-1])
        return (dyb @ _bf(w).t()).reshape(x.shape), x2.t() @ dyb.reshape(-1, dy.shape[-1])


def mm(x, w):
    """x[..., cin] @ w[cin, cout] -- fp32, or with bf16 operands when GEMM_BF16 is set."""
    return _MatmulBf16.apply(x, w) if GEMM_BF16 else x @ w


def edge_conv_split(x, nn_idx, cout, scope, V, is_training, bn_decay):
    """The DGCNN block in the algebraically equal split form (used when GEMM_BF16 is set):
    conv([x_i, x_j - x_i]) = x_i Wc + (x_j - x_i) Wn + b = (x_i Wc - x_i Wn + b) + x_j Wn."""
    B, N, C = x.shape
    W = V.get(scope + "/weights", (1, 1, 2 * C, cout), "xavier", fan=(2 * C, cout)).reshape(2 * C, cout)
    b = V.get(scope + "/biases", (cout,), "zeros")
    P = mm(x, W[:C])
    Q = mm(x, W[C:])
    idx = nn_idx.long() + (torch.arange(B) * N).view(B, 1, 1)
    y = ((P - Q) + b).unsqueeze(2) + Q.reshape(B * N, cout)[idx]          # [B,N,k,cout]
    y = batch_norm(y, scope + "/bn", V, is_training, bn_decay)
    return torch.relu(y)


def conv2d_1x1(x, cout, scope, V, bn, is_training, bn_decay, relu=True, store_bf16=False):
    """conv2d with a [1,1] kernel, utils/tf_util.py:111-179: matmul over the last axis + bias
    (+BN, +ReLU)."""
    cin = x.shape[-1]
    W = V.get(scope + "/weights", (1, 1, cin, cout), "xavier", fan=(cin, cout))
    b = V.get(scope + "/biases", (cout,), "zeros")
    y = mm(x, W.reshape(cin, cout)) + b
    if bn and store_bf16 and GEMM_BF16 and ACT_BF16 and is_training:
        y = batch_norm(_StoreBf16.apply(y), scope + "/bn", V, is_training, bn_decay, stats_from=y)
    elif bn:
        y = batch_norm(y, scope + "/bn", V, is_training, bn_decay)
    return torch.relu(y) if relu else y


def conv2d_full_width(x, cout, scope, V, is_training, bn_decay):
    """conv2d(input_image [B,N,D,1], cout, [1,D], 'VALID'), models/...:39."""
    B, N, D, _ = x.shape
    W = V.get(scope + "/weights", (1, D, 1, cout), "xavier", fan=(D, D * cout))
    b = V.get(scope + "/biases", (cout,), "zeros")
    y = x.reshape(B, N, D) @ W.reshape(D, cout) + b
    y = batch_norm(y.reshape(B, N, 1, cout), scope + "/bn", V, is_training, bn_decay)
    return torch.relu(y)


# Test infrastructure (tests/test_03_configs_gpu.py): {scope: bool mask [B, C]} -- the activation pattern ANOTHER
# implementation found for a batch-normalised fully connected layer.  The layers of the fully connected stack have B rows:
# ONE unit whose normalised value lies within fp32 round-off of the ReLU's corner is a coin toss between two correct
# implementations, and its term is 1 / B of its column's gradient (tests/test_oracle_conditioning.py).  With an override
# the units within RELU_TIE of the corner take the other implementation's side (value and derivative); every other unit
# must agree with it, which is asserted -- the analogue of nn_override for the neighbour sets.
RELU_OVERRIDE = None
RELU_TIE = 1e-4
RELU_REPORT = None      # a list: (scope, ambiguous units, units that took the other side, the largest |value| among those)


def fully_connected(x, cout, scope, V, bn=False, is_training=None, bn_decay=None, relu=True):
    """utils/tf_util.py:321-365."""
    cin = x.shape[-1]
    W = V.get(scope + "/weights", (cin, cout), "xavier", fan=(cin, cout))
    b = V.get(scope + "/biases", (cout,), "zeros")
    y = x @ W + b                      # the FC stack stays fp32 in the bf16 mode too
    if bn:
        y = batch_norm(y, scope + "/bn", V, is_training, bn_decay)
    if not relu:
        return y
    out = torch.relu(y)
    if RELU_OVERRIDE is not None and scope in RELU_OVERRIDE:
        other = RELU_OVERRIDE[scope].to(torch.bool)
        mine = y.detach() > 0
        tie = y.detach().abs() < RELU_TIE
        if bool(((mine != other) & ~tie).any()):
            raise AssertionError("%s: activation patterns differ away from the ReLU corner" % scope)
        out = torch.where(tie & other, y, torch.where(tie, torch.zeros_like(y), out))
        if RELU_REPORT is not None:
            flipped = tie & (mine != other)
            RELU_REPORT.append((scope, int(tie.sum()), int(flipped.sum()),
                                float(y.detach().abs()[flipped].max()) if bool(flipped.any()) else 0.0))
    return out


def knn_indices(point_cloud, k):
    """pairwise_xyz_distance + knn, utils/tf_util.py:597-632.  3-D input: xyz slice (:608);
    4-D [B,N,1,C] input: all C channels (the slice hits the size-1 axis)."""
    if point_cloud.dim() == 4:
        pts = point_cloud[:, :, 0, :].detach().numpy()
        c = pts.shape[2]
    else:
        pts = point_cloud.detach().numpy()
        c = min(3, pts.shape[2])
    return torch.from_numpy(O.knn(np.ascontiguousarray(pts), k, channels=c, threads=O.max_threads())).long()


def get_edge_feature(point_cloud, nn_idx, k):
    """utils/tf_util.py:635-669: concat(central tiled k times, neighbours - central)."""
    pc = point_cloud[:, :, 0, :] if point_cloud.dim() == 4 else point_cloud
    B, N, C = pc.shape
    flat = pc.reshape(B * N, C)
    idx = nn_idx + (torch.arange(B) * N).view(B, 1, 1)
    neighbors = flat[idx.reshape(-1)].reshape(B, N, k, C)
    central = pc.unsqueeze(2).expand(B, N, k, C)
    return torch.cat([central, neighbors - central], dim=-1)


# --------------------------------------------------------------------------------------
# models/pointnet_ycb_23_decoder_4.py
# --------------------------------------------------------------------------------------
def _decoder_and_heads(embedding, num_point, V, is_training, bn_decay, prefix, fc_suffix="", point_out=(4, 3),
                       heads=True):
    B = embedding.shape[0]
    net = fully_connected(embedding, 1024, prefix + "_fc1" + fc_suffix, V, True, is_training, bn_decay)
    net = fully_connected(net, 1024, prefix + "_fc2" + fc_suffix, V, True, is_training, bn_decay)
    net = fully_connected(net, num_point * point_out[0] * point_out[1], prefix + "_output", V, relu=False)
    recon = net.reshape(B, num_point * point_out[0], point_out[1])
    if not heads:
        return recon, None, None
    rot = fully_connected(embedding, 512, prefix + "_rot_fc1", V, True, is_training, bn_decay)
    rot = fully_connected(rot, 256, prefix + "_rot_fc2", V, True, is_training, bn_decay)
    rot = fully_connected(rot, 3, prefix + "_output_rot", V, relu=False)
    tr = fully_connected(embedding, 512, prefix + "_trans_fc1", V, True, is_training, bn_decay)
    tr = fully_connected(tr, 256, prefix + "_trans_fc2", V, True, is_training, bn_decay)
    tr = fully_connected(tr, 3, prefix + "_output_trans", V, relu=False)
    return recon, rot, tr


def get_model_dgcnn_6d(point_cloud, is_training_pl_encoder, is_training, k_neighbor, V, bn_decay=None,
                       pool="mean", prefix="", point_out=(4, 3), heads=True, vae_noise=None, nn_override=None):
    """get_model_dgcnn_mean_6d (models/...:327-455) / _max_6d (:592-723) / _mean_6d_hand (:458-589,
    point_out=(1,5)) / _mean_6d_2 (:726-856, prefix='model2/') / get_model_dgcnn[_mean] (:93-324,
    heads=False) / _mean_vae (:859-984, vae_noise given)."""
    B, N, _ = point_cloud.shape
    k = k_neighbor
    red = (lambda t, d: t.mean(d, keepdim=True)) if pool == "mean" else (lambda t, d: t.max(d, keepdim=True)[0])
    end_points = {}
    net = point_cloud
    nets = []
    for i, cout in enumerate((64, 64, 64, 128)):
        # nn_override: grouping indices given by the caller (tests feed the GPU's, so that a
        # round-off-level k-th/(k+1)-th near-tie cannot send the two networks down different paths)
        nn_idx = knn_indices(net, k) if nn_override is None else nn_override[i].long()
        if GEMM_BF16:
            net = edge_conv_split(net.reshape(B, N, -1), nn_idx, cout, "%sdgcnn%d" % (prefix, i + 1), V,
                                  is_training_pl_encoder, bn_decay)
        else:
            edge = get_edge_feature(net, nn_idx, k)
            net = conv2d_1x1(edge, cout, "%sdgcnn%d" % (prefix, i + 1), V, True, is_training_pl_encoder, bn_decay)
        net = red(net, -2)                                  # [B,N,1,cout]
        nets.append(net)
        end_points["nn_idx%d" % (i + 1)] = nn_idx
    net = conv2d_1x1(torch.cat(nets, dim=-1), 1024, prefix + "dgcnn_agg", V, True, is_training_pl_encoder,
                     bn_decay, store_bf16=True)
    end_points["layer_before_embedding"] = net
    net = red(net, 1)
    embedding = net.reshape(B, -1)
    end_points["nets"] = nets
    if vae_noise is not None:       # :945-953
        z_mean = fully_connected(embedding, 1024, "dgcnn_z_mean", V, True, is_training, bn_decay)
        z_std = fully_connected(embedding, 1024, "dgcnn_z_std", V, True, is_training, bn_decay)
        embedding = z_mean + z_std * vae_noise
        end_points["z_mean"], end_points["z_std"] = z_mean, z_std
    end_points["embedding"] = embedding
    recon, rot, tr = _decoder_and_heads(embedding, N, V, is_training, bn_decay, prefix + "dgcnn",
                                        point_out=point_out, heads=heads)
    return recon, rot, tr, end_points


def get_model_pn(point_cloud, is_training, V, bn_decay=None):
    """models/...:23-89."""
    B, N, D = point_cloud.shape
    end_points = {}
    net = conv2d_full_width(point_cloud.unsqueeze(-1), 64, "pn_conv1_encoder", V, is_training, bn_decay)
    for i, c in ((2, 64), (3, 64), (4, 128), (5, 1024)):
        net = conv2d_1x1(net, c, "pn_conv%d_encoder" % i, V, True, is_training, bn_decay)
    net = net.max(1, keepdim=True)[0]                       # max_pool2d [num_point,1]
    embedding = net.reshape(B, -1)
    end_points["embedding"] = embedding
    recon, rot, tr = _decoder_and_heads(embedding, N, V, is_training, bn_decay, "pn", fc_suffix="_decoder")
    return recon, rot, tr, end_points


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
class _NnDistance(torch.autograd.Function):
    """tf_ops/nn_distance: forward tf_nndistance.cpp:21-43, gradient :126-163 (C oracle)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        a, b = xyz1.detach().numpy(), xyz2.detach().numpy()
        d1, i1, d2, i2 = O.nn_distance(a, b, threads=O.max_threads())
        ctx.save_for_backward(xyz1, xyz2)
        ctx.idx = (i1, i2)
        return torch.from_numpy(d1), torch.from_numpy(i1), torch.from_numpy(d2), torch.from_numpy(i2)

    @staticmethod
    def backward(ctx, g1, gi1, g2, gi2):
        xyz1, xyz2 = ctx.saved_tensors
        gx1, gx2 = O.nn_distance_grad(xyz1.detach().numpy(), xyz2.detach().numpy(), g1.numpy(), ctx.idx[0],
                                      g2.numpy(), ctx.idx[1], threads=O.max_threads())
        return torch.from_numpy(gx1), torch.from_numpy(gx2)


def chamfer_loss(pred, label):
    """losses/chamfer_loss.py:8-14."""
    d1, _, d2, _ = _NnDistance.apply(pred, label)
    per = d1 + d2
    return per.mean(), per


def translation_error(pred, label):
    """losses/trans_distance.py:4-9."""
    per = torch.sqrt(((label - pred) ** 2).sum(1))
    return per.mean(), per


def skew_symmetric(v):
    """losses/angular_distance_taylor.py:6-27."""
    z = torch.zeros_like(v[:, 0])
    return torch.stack([torch.stack([z, -v[:, 2], v[:, 1]], 1),
                        torch.stack([v[:, 2], z, -v[:, 0]], 1),
                        torch.stack([-v[:, 1], v[:, 0], z], 1)], 1)


def exponential_map(axag, EPS=1e-2):
    """losses/angular_distance_taylor.py:30-66 (float64).  tf.where evaluates both branches;
    the unselected one gets a zero gradient, which torch.where reproduces as long as it is
    finite -- the denominators are made safe on the unselected side only."""
    ss = skew_symmetric(axag)
    theta_sq = (axag ** 2).sum(1)
    small = theta_sq < EPS
    safe_sq = torch.where(small, torch.ones_like(theta_sq), theta_sq)
    theta = torch.sqrt(safe_sq)
    p4 = theta_sq * theta_sq
    p6 = theta_sq * theta_sq * theta_sq
    p8 = theta_sq * theta_sq * theta_sq * theta_sq
    t1 = torch.where(small, 1 - (theta_sq / 6) + (p4 / 120) - (p6 / 5040) + (p8 / 362880),
                     torch.sin(theta) / theta)
    t2 = torch.where(small, 0.5 - (theta_sq / 24) + (p4 / 720) - (p6 / 40320) + (p8 / 3628800),
                     (1 - torch.cos(theta)) / safe_sq)
    eye = torch.eye(3, dtype=axag.dtype).expand(axag.shape[0], 3, 3)
    return eye + t1[:, None, None] * ss + t2[:, None, None] * (ss @ ss)


def rotation_error(pred, label):
    """losses/angular_distance_taylor.py:69-116: theta = acos(clip((tr(R_l R_p^T) - 1)/2))."""
    Rp = exponential_map(pred)
    Rl = exponential_map(label)
    R = Rl @ Rp.transpose(1, 2)
    t = (torch.diagonal(R, dim1=1, dim2=2).sum(1) - 1) / 2
    t = torch.clamp(t, -0.9999999, 0.9999999)   # clip_by_value: zero gradient outside, as torch.clamp
    theta = torch.acos(t)
    return theta.mean(), theta


# --------------------------------------------------------------------------------------
# train_cloudAAE_ycbv.py:194-273
# --------------------------------------------------------------------------------------
def bn_decay_schedule(step, batch_size):
    """:166-169,194-202: min(0.99, 1 - 0.5 * 0.5^floor(step*B/40)) in fp32."""
    p = np.floor(np.float32(step) * np.float32(batch_size) / np.float32(40.0))
    mom = np.float32(0.5) * np.float32(0.5) ** np.float32(p)
    return float(min(np.float32(0.99), np.float32(1.0) - np.float32(mom)))


def assemble_input(visible, noise, class_id, num_point, num_class=21):
    """:206-226."""
    B = visible.shape[0]
    onehot = torch.zeros(B, num_class)
    valid = (class_id >= 0) & (class_id < num_class)
    onehot[torch.arange(B)[valid], class_id[valid]] = 1.0
    tile = onehot.unsqueeze(1).expand(B, num_point, num_class)
    v = visible[:, :num_point, :]
    if noise is not None:
        v = v + noise
    mean = v.mean(1)
    return torch.cat([v - mean.unsqueeze(1), tile], dim=2), mean, v


def forward_losses(batch, V, num_point, is_training=True, bn_decay=None, k=10, model="dgcnn_mean_6d",
                   nn_override=None):
    """:206-268 -> dict with the tensors the reference fetches (:350-367)."""
    pc, mean, noisy = assemble_input(batch["visiblePoints"], batch.get("noise"), batch["class_id"], num_point)
    target = batch["visiblePoints_org"][:, :num_point * 4, :]
    if model == "pn":
        recon_res, rot_pred, trans_res, ep = get_model_pn(pc, is_training, V, bn_decay)
    else:
        recon_res, rot_pred, trans_res, ep = get_model_dgcnn_6d(
            pc, is_training, is_training, k, V, bn_decay, pool="max" if model == "dgcnn_max_6d" else "mean",
            nn_override=nn_override)
    xyz_recon = recon_res + mean.unsqueeze(1)
    trans_pred = trans_res + mean
    xyz_loss, xyz_per = chamfer_loss(xyz_recon, target)
    trans_loss, trans_per = translation_error(trans_pred, batch["translation"])
    axag_loss64, axag_per = rotation_error(rot_pred.double(), batch["axisangle"].double())
    axag_loss = axag_loss64.float()
    total = 1000 * xyz_loss + 10 * trans_loss + axag_loss
    return dict(total_loss=total, xyz_loss=xyz_loss, trans_loss=trans_loss, axag_loss=axag_loss,
                xyz_recon=xyz_recon, xyz_loss_per_sample=xyz_per, trans_loss_perSample=trans_per,
                axag_loss_perSample=axag_per, rot_pred=rot_pred, trans_pred=trans_pred, input_pc=pc,
                element_mean=mean, end_points=ep)


class AdamTF(object):
    """tf.train.AdamOptimizer(lr) (:263-273), TF-1.x ApplyAdam update."""

    def __init__(self, lr=0.0008, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = (np.float32(x) for x in (lr, beta1, beta2, eps))
        self.b1p, self.b2p = np.float32(beta1), np.float32(beta2)
        self.m, self.v = {}, {}

    def apply(self, params, grads):
        lr_t = self.lr * np.sqrt(np.float32(1) - self.b2p) / (np.float32(1) - self.b1p)
        with torch.no_grad():
            for name, p in params.items():
                g = grads[name]
                m = self.m.setdefault(name, torch.zeros_like(p))
                v = self.v.setdefault(name, torch.zeros_like(p))
                m += (g - m) * float(np.float32(1) - self.b1)
                v += (g * g - v) * float(np.float32(1) - self.b2)
                p -= (m * float(lr_t)) / (torch.sqrt(v) + float(self.eps))
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)


def train_step(batch, V, opt, step, num_point, batch_size, k=10, model="dgcnn_mean_6d", nn_override=None):
    """One iteration of the loop at :344-368: bn_decay(step) -> forward -> losses ->
    gradients of total_loss w.r.t. every trainable -> Adam.  Returns (outputs, grads)."""
    decay = bn_decay_schedule(step, batch_size)
    out = forward_losses(batch, V, num_point, True, decay, k, model, nn_override)
    names = list(V.p.keys())
    gs = torch.autograd.grad(out["total_loss"], [V.p[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(V.p[n])) for n, g in zip(names, gs)}
    opt.apply(V.p, grads)
    out["bn_decay"] = decay
    return out, grads


def synthetic_batch(batch_size, num_point, seed=123456789, num_class=21, single_class=None):
    """Synthetic inputs of SURVEY.md section 8d (the reference cannot supply 4N target points
    for N > 512): object-scale points + translation, class ids, axis-angle labels."""
    g = torch.Generator().manual_seed(seed)
    B, N = batch_size, num_point
    t = torch.empty(B, 3)
    t[:, :2] = torch.rand(B, 2, generator=g) * 0.5 - 0.25
    t[:, 2] = torch.rand(B, generator=g) + 0.5
    vis = torch.randn(B, N, 3, generator=g) * 0.05 + t[:, None, :]
    org = torch.randn(B, 4 * N, 3, generator=g) * 0.05 + t[:, None, :]
    noise = torch.randn(B, N, 3, generator=g) * (0.004 / 3.0)
    cls = torch.randint(0, num_class, (B,), generator=g)
    if single_class is not None:
        cls[:] = single_class
    axis = torch.randn(B, 3, generator=g, dtype=torch.float64)
    axis = axis / axis.norm(dim=1, keepdim=True)
    angle = (torch.rand(B, generator=g, dtype=torch.float64) * 2 - 1) * math.pi
    return dict(visiblePoints=vis, visiblePoints_org=org, noise=noise, class_id=cls, translation=t.clone(),
                axisangle=axis * angle[:, None])
