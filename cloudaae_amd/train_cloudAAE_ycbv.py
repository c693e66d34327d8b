"""Training step -- mirror of the reference's train_cloudAAE_ycbv.py (setup_graph :137-329,
train_graph :332-437, CLI :440-484), MI355X-native.

What the reference assembles as a TF graph (:194-273) is `TrainGraph.forward` here; one
`TrainGraph.train_step(batch)` is one iteration of the reference's session loop (:350-368):
bn_decay schedule -> input assembly -> model -> three losses -> gradients -> Adam.
Everything between the input batch and the updated weights runs in libcloudaae_hip.so on
the GPU; step counter, Adam beta powers and the BN decay are device scalars, so a step
issues no host synchronisation.

Data parallelism (new; the reference is single-GPU): one process per GPU, the batch is
sharded across ranks, gradients are averaged with RCCL all-reduce over the store's flat
gradient buffer -- the 12.6 M-element dgcnn_output weight gradient, produced first in
backward, is reduced asynchronously while the encoder backward still runs.  Batch-norm
statistics are per-rank (local BN).
"""
import argparse
import importlib
import math
import os
import time

import torch
import torch.distributed as dist

from . import _lib
from ._lib import ptr, require, stream
from .losses import angular_distance_taylor, chamfer_loss, trans_distance
from .utils import _functions as F
from .utils import tf_util
from .utils.grad_exchange import GradExchange
from .utils.variables import reset_default_store

NUM_CLASS = 21                      # train_cloudAAE_ycbv.py:29
BN_INIT_DECAY = 0.5                 # :166-169
BN_DECAY_DECAY_RATE = 0.5
BN_DECAY_DECAY_STEP = float(40)
BN_DECAY_CLIP = 0.99
NOISE_STDDEV = 0.004 / 3.           # :217
LOSS_WEIGHTS = (1000.0, 10.0, 1.0)  # :268
K_NEIGHBOR = 10                     # :230


def get_training_argparser():
    """Same flags as train_cloudAAE_ycbv.py:440-467, plus the data-parallel/benchmark ones."""
    parser = argparse.ArgumentParser()
    general = parser.add_argument_group('general')
    general.add_argument('--gpu', type=int, default=0, help='GPU to use [default: GPU 0]')
    general.add_argument('--model', default='pointnet_ycb_23_decoder_4', help='Model module name')
    general.add_argument('--log_dir', default='log', help='Log dir [default: log]')
    general.add_argument('--num_point', type=int, default=256, help='Point Number [256/512/1024] [default: 256]')
    general.add_argument('--total_num_point', type=int, default=512, help='Dataset Point Number')
    train_opts = parser.add_argument_group('training_options')
    train_opts.add_argument('--max_epoch', type=int, default=500, help='Epoch to run')
    train_opts.add_argument('--optimizer', default='adam', help='adam or gd [default: adam]')
    hyper = parser.add_argument_group('hyperparameters')
    hyper.add_argument('--batch_size', type=int, default=128, help='Batch Size during training [default: 128]')
    hyper.add_argument('--learning_rate', type=float, default=0.0008, help='Initial learning rate')
    hyper.add_argument('--momentum', type=float, default=0.9)
    hyper.add_argument('--decay_step', type=int, default=30000)
    hyper.add_argument('--decay_rate', type=float, default=0.7)
    hyper.add_argument('--trans_tol', type=float, default=0.1)
    extra = parser.add_argument_group('mi355x')
    extra.add_argument('--model_fn', default='get_model_dgcnn_mean_6d')
    extra.add_argument('--k', type=int, default=K_NEIGHBOR)
    extra.add_argument('--steps', type=int, default=100, help='synthetic steps to run')
    return parser


def parse_arg_groups(parser, argv=None):
    """train_cloudAAE_ycbv.py:470-475."""
    args = parser.parse_args(argv)
    arg_groups = {}
    for group in parser._action_groups:
        arg_groups[group.title] = {a.dest: getattr(args, a.dest, None) for a in group._group_actions}
    return arg_groups


class TrainGraph(object):
    """The counterpart of setup_graph(): owns the variables, the optimiser slots and the
    device-side bookkeeping scalars; `forward` is the graph of :206-268."""

    def __init__(self, general_opts=None, train_opts=None, hyperparameters=None, device=None,
                 model_fn='get_model_dgcnn_mean_6d', k_neighbor=K_NEIGHBOR, process_group=None, seed=123456789):
        general_opts = dict(general_opts or {})
        train_opts = dict(train_opts or {})
        hyperparameters = dict(hyperparameters or {})
        self.NUM_POINT = int(general_opts.get('num_point', 256))
        self.BATCH_SIZE = int(hyperparameters.get('batch_size', 128))          # GLOBAL batch
        self.BASE_LEARNING_RATE = float(hyperparameters.get('learning_rate', 0.0008))
        self.OPTIMIZER = train_opts.get('optimizer', 'adam')
        require(self.OPTIMIZER in ('adam', 'gd'), "optimizer must be adam or gd")
        self.device = torch.device(device if device is not None else 'cuda:%d' % int(general_opts.get('gpu', 0)))
        self.k = int(k_neighbor)
        # process_group: None = the default group when torch.distributed is initialised;
        # False = single-process even then; or an explicit group
        solo = process_group is False or not (dist.is_available() and dist.is_initialized())
        self.pg = None if solo else process_group
        self.world = 1 if solo else dist.get_world_size(self.pg)
        self.rank = 0 if solo else dist.get_rank(self.pg)
        require(self.BATCH_SIZE % self.world == 0, "global batch must divide by the number of ranks")
        self.local_batch = self.BATCH_SIZE // self.world
        # MODEL = importlib.import_module(general_opts['model'])   (:147) -- the plugin seam
        self.MODEL = importlib.import_module('cloudaae_amd.models.' +
                                             general_opts.get('model', 'pointnet_ycb_23_decoder_4'))
        self.model_fn = getattr(self.MODEL, model_fn)
        self.is_pn = model_fn == 'get_model_pn'

        torch.cuda.set_device(self.device)
        self.store = reset_default_store(device=self.device, seed=seed)   # tf.set_random_seed(123456789), :160
        dev = self.device
        self.batch = torch.zeros(1, dtype=torch.float32, device=dev)         # `batch = tf.Variable(0.)`, :192
        self.bn_decay = torch.full((1,), 0.5, dtype=torch.float32, device=dev)
        self.beta1_power = torch.full((1,), 0.9, dtype=torch.float32, device=dev)
        self.beta2_power = torch.full((1,), 0.999, dtype=torch.float32, device=dev)
        self._build()

    # -- graph construction: create every variable once, then pack them ---------------------
    def _build(self):
        B, N = max(2, min(self.local_batch, 2)), self.NUM_POINT
        dummy = torch.zeros((B, N, 3 + NUM_CLASS), dtype=torch.float32, device=self.device)
        dummy[:, :, :3] = torch.rand((B, N, 3), device=self.device)
        with torch.no_grad():
            self._call_model(dummy, False)
        self.store.flatten()
        n = self.store.flat_params.numel()
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
        # overlap bucket: the decoder output weights (12*N*1024 floats), first gradient of backward
        early = None
        for name in ('dgcnn_output/weights', 'pn_output/weights'):
            v = self.store.vars.get(name)
            if v is not None:
                o = self.store.offsets[name]
                early = (o, o + v.data.numel())
        self.exchange = GradExchange(self.store.flat_grads, early, self.pg, world=self.world)
        self.exchange.broadcast_params(self.store.flat_params)     # identical initial weights on every rank
        if early is not None and self.exchange.active:
            for name in ('dgcnn_output/weights', 'pn_output/weights'):
                if name in self.store.vars:
                    self.store.vars[name].on_ready = self.exchange.early_ready

    def _call_model(self, pc, is_training):
        if self.is_pn:
            return self.model_fn(pc, is_training, bn_decay=self.bn_decay)
        return self.model_fn(pc, is_training, is_training, self.k, bn_decay=self.bn_decay)

    # -- :206-268 -----------------------------------------------------------------------------
    def forward(self, element, is_training=True):
        """element: dict like the reference's `next_element` (device tensors):
        visiblePoints [B,>=N,3], visiblePoints_org [B,>=4N,3], translation [B,3],
        axisangle [B,3], class_id [B] int64, optional noise [B,N,3]."""
        N = self.NUM_POINT
        vis = element['visiblePoints'].contiguous()
        B, P, _ = vis.shape
        require(P >= N, "visiblePoints has fewer rows than num_point")
        noise = element.get('noise')
        if noise is None and is_training:
            # tf.random.normal(shape, stddev=0.004/3), :217
            noise = torch.randn((B, N, 3), dtype=torch.float32, device=vis.device) * NOISE_STDDEV
        pc = torch.empty((B, N, 3 + NUM_CLASS), dtype=torch.float32, device=vis.device)
        element_mean = torch.empty((B, 3), dtype=torch.float32, device=vis.device)
        noisy = torch.empty((B, N, 3), dtype=torch.float32, device=vis.device)
        cls = element['class_id'].to(torch.int64).contiguous()
        _lib.check(_lib.lib().cloudaae_input_assemble(B, P, N, NUM_CLASS, ptr(vis),
                                                      ptr(noise.contiguous()) if noise is not None else None,
                                                      ptr(cls), ptr(pc), ptr(element_mean), ptr(noisy), stream()),
                   "cloudaae_input_assemble")
        org = element['visiblePoints_org']
        require(org.shape[1] >= 4 * N, "visiblePoints_org has fewer than 4*num_point rows "
                                        "(the reference silently truncates here, train...:211-214)")
        visiblePoints_org_final = org[:, 0:N * 4, :].contiguous()

        xyz_recon_res, rot_pred, trans_pred_res, endpoint = self._call_model(pc, is_training)
        xyz_recon = F.AddRowVecFn.apply(xyz_recon_res, element_mean)                          # :232
        trans_pred = F.AddRowVecFn.apply(trans_pred_res.unsqueeze(1), element_mean).squeeze(1)  # :233
        xyz_loss, xyz_loss_per_sample = chamfer_loss.get_loss(xyz_recon, visiblePoints_org_final)  # :236
        trans_loss, trans_loss_perSample = trans_distance.get_translation_error(
            trans_pred, element['translation'].to(torch.float32))                                # :241
        axag_loss, axag_loss_perSample = angular_distance_taylor.get_rotation_error(
            rot_pred, element['axisangle'])                                                      # :249-253
        total_loss = F.LossMixFn.apply(xyz_loss, trans_loss, axag_loss, *LOSS_WEIGHTS)          # :268
        return dict(total_loss=total_loss, xyz_loss=xyz_loss, trans_loss=trans_loss, axag_loss=axag_loss,
                    xyz_recon=xyz_recon, xyz_loss_per_sample=xyz_loss_per_sample,
                    trans_loss_perSample=trans_loss_perSample, axag_loss_perSample=axag_loss_perSample,
                    rot_pred=rot_pred, trans_pred=trans_pred, visiblePoints_final=noisy,
                    visiblePoints_org_final=visiblePoints_org_final, class_id=cls, input_pc=pc,
                    element_mean=element_mean, end_points=endpoint)

    # -- one iteration of the loop at :344-368 ------------------------------------------------
    def train_step(self, element):
        L = _lib.lib()
        s = stream()
        self.store.begin_step()
        # bn_decay = min(0.99, 1 - 0.5 * 0.5^floor(batch*BATCH_SIZE/40)), :194-202
        _lib.check(L.cloudaae_bn_decay_schedule(ptr(self.batch), float(self.BATCH_SIZE), BN_INIT_DECAY,
                                                BN_DECAY_DECAY_STEP, BN_DECAY_DECAY_RATE, BN_DECAY_CLIP,
                                                ptr(self.bn_decay), s), "cloudaae_bn_decay_schedule")
        out = self.forward(element, is_training=True)
        out['total_loss'].backward()
        self.exchange.finish()            # RCCL all-reduce of the flat gradient buffer (no-op for 1 rank)
        n = self.store.flat_params.numel()
        scale = self.exchange.scale
        if self.OPTIMIZER == 'adam':      # tf.train.AdamOptimizer(learning_rate), :266
            _lib.check(L.cloudaae_adam_tf(n, ptr(self.store.flat_params), ptr(self.store.flat_grads),
                                          ptr(self.adam_m), ptr(self.adam_v), self.BASE_LEARNING_RATE, 0.9,
                                          0.999, 1e-8, ptr(self.beta1_power), ptr(self.beta2_power), scale, 1,
                                          stream()), "cloudaae_adam_tf")
        else:                             # GradientDescentOptimizer(learning_rate*10), :264
            _lib.check(L.cloudaae_sgd(n, ptr(self.store.flat_params), ptr(self.store.flat_grads),
                                      self.BASE_LEARNING_RATE * 10, scale, stream()), "cloudaae_sgd")
        _lib.check(L.cloudaae_increment(ptr(self.batch), 1.0, stream()), "cloudaae_increment")  # global_step
        return out

    def eval_step(self, element):
        with torch.no_grad():
            return self.forward(element, is_training=False)


# ---- the data pipeline of train_cloudAAE_ycbv.py:42-117, batched on the GPU --------------------
def get_object_model(x, obj_models):
    """:68-76  x['obj_batch'] = obj_models[class_id]  (obj_models [21,2048,6] device tensor); the
    gather itself happens inside transform_object_model."""
    x['obj_model'] = obj_models
    return x


def get_rotation_matrix(x):
    """:79-85  rot_mat = float32(exponential_map(float64(axisangle)))."""
    x['axisangle'] = x['axisangle'].to(torch.float64)
    x['rot_mat64'] = angular_distance_taylor.exponential_map(x['axisangle'])
    x['rot_mat'] = x['rot_mat64'].to(torch.float32)
    return x


def transform_object_model(x):
    """:88-93  model_xyz_rot_trans = obj_batch[:, :, 0:3] R^T + translation."""
    models = x['obj_model'].to(torch.float32).contiguous()
    nmodels, npts, _ = models.shape
    t = x['translation'].to(torch.float32).contiguous()
    B = t.shape[0]
    out = torch.empty((B, npts, 3), dtype=torch.float32, device=t.device)
    cls = x['class_id'].to(torch.int64).contiguous()
    _lib.check(_lib.lib().cloudaae_transform_object_model(B, npts, nmodels, ptr(models), ptr(cls),
                                                          ptr(x['rot_mat64'].contiguous()), ptr(t), ptr(out),
                                                          stream()), "cloudaae_transform_object_model")
    x['model_xyz_rot_trans'] = out
    return x


def get_small_data(records, obj_models, seed=0):
    """:96-117 for one batch: records = dict of device tensors translation [B,3], axisangle [B,3],
    class_id [B] (e.g. from tfrecord_io.PoseRecords.epoch); returns the reference's element dict:
    visiblePoints [B,2449,3], visiblePoints_org [B,2049,3], occluder, model_xyz_rot_trans, ..."""
    from .utils import generate_occluder, hidden_point_removal as hpr
    x = dict(records)
    x = get_object_model(x, obj_models)
    x = get_rotation_matrix(x)
    x = transform_object_model(x)
    x = generate_occluder.get_random_spherical_occluder(x, 'ycbv', seed=seed)
    x = hpr.sphericalFlip(x, None, 0.8 * math.pi)            # center = zeros_like(translation), :103
    x = hpr.hidden_point_removal(x, seed=seed)
    x = hpr.sphericalFlip_org(x, None, 0.8 * math.pi)
    x = hpr.hidden_point_removal_org(x, seed=seed)
    return x


def synthetic_element(local_batch, num_point, device, seed=123456789, rank=0, single_class=None):
    """Synthetic `next_element` of SURVEY.md section 8d, generated on the device: object-scale
    points N(0, 0.05^2) + translation (t_xy ~ U(+-0.25), t_z ~ U(0.5,1.5)), class ids U{0..20},
    axis-angle = uniform axis x U(-pi,pi); the target cloud has 4N points of the same law."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + rank)
    B, N = local_batch, num_point
    t = torch.empty((B, 3), device=device)
    t[:, :2] = torch.rand((B, 2), generator=g, device=device) * 0.5 - 0.25
    t[:, 2] = torch.rand((B,), generator=g, device=device) + 0.5
    vis = torch.randn((B, N, 3), generator=g, device=device) * 0.05 + t[:, None, :]
    org = torch.randn((B, 4 * N, 3), generator=g, device=device) * 0.05 + t[:, None, :]
    cls = torch.randint(0, NUM_CLASS, (B,), generator=g, device=device)
    if single_class is not None:
        cls[:] = single_class
    axis = torch.randn((B, 3), generator=g, device=device, dtype=torch.float64)
    axis = axis / axis.norm(dim=1, keepdim=True)
    angle = (torch.rand((B,), generator=g, device=device, dtype=torch.float64) * 2 - 1) * math.pi
    return dict(visiblePoints=vis, visiblePoints_org=org, class_id=cls, translation=t.clone(),
                axisangle=axis * angle[:, None])


def main(argv=None):
    parser = get_training_argparser()
    groups = parse_arg_groups(parser, argv)
    general, topts, hyper, extra = groups['general'], groups['training_options'], groups['hyperparameters'], groups['mi355x']
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl')
        general['gpu'] = local
    graph = TrainGraph(general, topts, hyper, model_fn=extra['model_fn'], k_neighbor=extra['k'])
    el = synthetic_element(graph.local_batch, graph.NUM_POINT, graph.device, rank=graph.rank)
    t0 = time.time()
    for i in range(extra['steps']):
        out = graph.train_step(el)
        if i % 10 == 0 and graph.rank == 0:
            print("step %d xyz_loss %f trans_loss %f axag_loss %f" %
                  (i, float(out['xyz_loss']), float(out['trans_loss']), float(out['axag_loss'])))
    torch.cuda.synchronize()
    if graph.rank == 0:
        print("%.1f clouds/s" % (extra['steps'] * graph.BATCH_SIZE / (time.time() - t0)))


if __name__ == "__main__":
    main()
