"""TF-1.x style variable store for the torch-hosted mirror of utils/tf_util.py.

The reference builds a graph once: `tf.get_variable` inside nested
`tf.variable_scope`s creates `dgcnn1/weights`, `dgcnn1/biases`, `dgcnn1/bn/beta`,
`dgcnn1/bn/gamma`, ... (utils/tf_util.py:20-22, 42-46, 488-491) and the EMA shadows of
the batch-norm moments (tf_util.py:493-500).  Here `get_model_*` runs every step, so
the store hands back the SAME variable on every call with the same scoped name
(AUTO_REUSE semantics) and keeps the reference's names so that a TF checkpoint could
be mapped on.

MI355X-first layout: after the first (building) call, `flatten()` packs all trainable
variables into ONE fp32 buffer, their gradients into a second and the Adam slots into
two more (16.4 M floats = 65.5 MB each at N=1024).  Layers receive views; gradients
are written by the HIP kernels straight into the flat gradient buffer, which is what
the RCCL all-reduce and the fused Adam kernel consume without any gather/scatter.
"""
import contextlib
import math
from collections import OrderedDict

import torch


class Variable(object):
    __slots__ = ("name", "shape", "trainable", "data", "grad", "fresh", "zeroed", "on_ready", "tag")

    def __init__(self, name, data, trainable):
        self.name = name
        self.shape = tuple(data.shape)
        self.trainable = trainable
        self.data = data      # leaf tensor (requires_grad for trainables)
        self.grad = None      # view into the flat gradient buffer (trainables)
        self.fresh = True     # first gradient write of a backward pass stores, later ones add
        self.zeroed = False   # begin_step(zero_grads=True) cleared the gradient slot as part of one big fill
        self.on_ready = None  # called right after the gradient kernel is enqueued (comm overlap)
        self.tag = None       # 'fc' for the fully connected stack (the early gradient bucket)


class VariableStore(object):
    def __init__(self, device=None, seed=123456789):
        # seed: tf.set_random_seed(123456789), train_cloudAAE_ycbv.py:160
        self.device = torch.device(device if device is not None else "cuda")
        self.vars = OrderedDict()
        self._scope = []
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(seed)
        self.flat_params = None
        self.flat_grads = None
        self.flat_state = None
        self.offsets = None
        self._scalars = {}

    # ---- scopes -----------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name):
        self._scope.append(name)
        try:
            yield "/".join(self._scope)
        finally:
            self._scope.pop()

    def scoped(self, name):
        return "/".join(self._scope + [name])

    # ---- creation ---------------------------------------------------------------
    def get_variable(self, name, shape, initializer, trainable=True):
        full = self.scoped(name)
        v = self.vars.get(full)
        if v is not None:
            if tuple(shape) != v.shape:
                raise ValueError("variable %s exists with shape %s, requested %s" % (full, v.shape, tuple(shape)))
            return v
        if self.flat_params is not None:
            raise RuntimeError("variable %s requested after flatten(); build the model first" % full)
        host = initializer(tuple(shape), self._gen)
        data = host.to(self.device, dtype=torch.float32).contiguous()
        if trainable:
            data.requires_grad_(True)
        v = Variable(full, data, trainable)
        data._cloudaae_var = v
        self.vars[full] = v
        return v

    # ---- initialisers (host side, fp32) --------------------------------------------
    @staticmethod
    def xavier_uniform(fan_in, fan_out):
        # tf.contrib.layers.xavier_initializer() (uniform=True): U(-l, l), l = sqrt(6/(fan_in+fan_out))
        limit = math.sqrt(6.0 / (fan_in + fan_out))

        def init(shape, gen):
            return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * limit
        return init

    @staticmethod
    def truncated_normal(stddev):
        def init(shape, gen):
            x = torch.randn(shape, generator=gen, dtype=torch.float32)
            bad = x.abs() > 2.0
            while bad.any():
                x[bad] = torch.randn(int(bad.sum()), generator=gen, dtype=torch.float32)
                bad = x.abs() > 2.0
            return x * stddev
        return init

    @staticmethod
    def constant(value):
        def init(shape, gen):
            return torch.full(shape, float(value), dtype=torch.float32)
        return init

    # ---- flat layout ----------------------------------------------------------------
    def trainable_variables(self):
        return [v for v in self.vars.values() if v.trainable]

    def state_variables(self):
        return [v for v in self.vars.values() if not v.trainable]

    @property
    def num_params(self):
        return sum(v.data.numel() for v in self.trainable_variables())

    def flatten(self, last=()):
        """Pack trainables (and, separately, non-trainable state) into flat buffers;
        every Variable.data becomes a view.  Offsets are 16-byte aligned.  `last`: names of (or a
        predicate selecting) variables to place at the END of the flat buffer (the data-parallel exchange reduces the
        decoder output weights on their own, early; with them last the rest is ONE contiguous piece)."""
        if self.flat_params is not None:
            return

        def pack(vs, need_grad):
            offs, total = [], 0
            for v in vs:
                offs.append(total)
                total += (v.data.numel() + 3) // 4 * 4
            flat = torch.zeros(max(total, 4), dtype=torch.float32, device=self.device)
            for v, o in zip(vs, offs):
                n = v.data.numel()
                flat[o:o + n].copy_(v.data.detach().reshape(-1))
                view = flat[o:o + n].view(v.shape)
                if need_grad:
                    view.requires_grad_(True)
                view._cloudaae_var = v
                v.data = view
            return flat, offs

        tv = self.trainable_variables()
        is_last = last if callable(last) else (lambda v: v.name in last)
        tv = [x for x in tv if not is_last(x)] + [x for x in tv if is_last(x)]
        self.flat_params, offs = pack(tv, True)
        self.flat_grads = torch.zeros_like(self.flat_params)
        self._zero = torch.zeros(1, dtype=torch.float32, device=self.device)
        for v, o in zip(tv, offs):
            v.grad = self.flat_grads[o:o + v.data.numel()].view(v.shape)
            v.data.grad = v.grad          # `.grad` is visible in the usual place
        self.flat_state, _ = pack(self.state_variables(), False)
        self.offsets = OrderedDict((v.name, o) for v, o in zip(tv, offs))

    def begin_step(self, zero_grads=False, zero_limit=None):
        """Marks every gradient slot unwritten.  zero_grads: clear the flat gradient buffer with ONE
        fill, so that the split-K weight-gradient products of the step can add into it directly
        instead of each clearing its own output first (a dozen tiny fills per step).  zero_limit: clear
        only the first zero_limit elements -- the slots beyond belong to layers whose gradient kernels
        store every element (the fully connected stack at batch <= 128: 63 of the 65 MB)."""
        zeroed = bool(zero_grads) and self.flat_grads is not None
        limit = self.flat_grads.numel() if (zeroed and zero_limit is None) else (int(zero_limit) if zeroed else 0)
        if zeroed and limit > 0:
            from .. import _lib
            plan = _lib.recording()
            fill = _lib.lib().cloudaae_fill_scaled
            if plan is not None and (limit * 4) % 16 == 0 and self.flat_grads.data_ptr() % 16 == 0:
                # a recorded step clears these slots with the launch that clears its zero zones (one launch
                # less per replay); the recording pass itself clears them now
                plan.clear_at_replay(self.flat_grads, limit * 4)
                fill = _lib.lib()._cdll.cloudaae_fill_scaled
            _lib.check(fill(limit, _lib.ptr(self._zero), 1.0, None, _lib.ptr(self.flat_grads), _lib.stream()),
                       "cloudaae_fill_scaled")
        for v in self.vars.values():
            v.fresh = True
            o = self.offsets.get(v.name) if self.offsets is not None else None
            v.zeroed = zeroed and o is not None and o + v.data.numel() <= limit

    # ---- checkpoint-style access with the reference's variable names -------------------
    def state_dict(self):
        return OrderedDict((n, v.data.detach().clone()) for n, v in self.vars.items())

    def load_state_dict(self, sd, strict=True):
        for n, t in sd.items():
            if n not in self.vars:
                if strict:
                    raise KeyError(n)
                continue
            with torch.no_grad():
                self.vars[n].data.copy_(torch.as_tensor(t, dtype=torch.float32).reshape(self.vars[n].shape))
        if strict:
            missing = [n for n in self.vars if n not in sd]
            if missing:
                raise KeyError("missing variables: %s" % missing[:5])

    # ---- cached device scalars (bn_decay given as a Python float) ----------------------
    def scalar(self, value):
        key = float(value)
        t = self._scalars.get(key)
        if t is None:
            if len(self._scalars) > 256:
                self._scalars.clear()
            t = torch.full((1,), key, dtype=torch.float32, device=self.device)
            self._scalars[key] = t
        return t


_default = None


def default_store():
    """The process-wide store (the analogue of TF's default graph)."""
    global _default
    if _default is None:
        _default = VariableStore()
    return _default


def set_default_store(store):
    global _default
    _default = store
    return store


def reset_default_store(device=None, seed=123456789):
    """tf.reset_default_graph() (train_cloudAAE_ycbv.py:138)."""
    return set_default_store(VariableStore(device=device, seed=seed))
