"""Update of the training step on flat buffers: gradient-norm clipping + AdamW as two launches
(csrc/glx_optim.hip) and the gradient exchange of the data-parallel step on the same flat gradient buffer.

Our counterpart of tools/train_utils/train_utils.py:38-39 (`clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP)`,
`optimizer.step()`) with the `adam_onecycle` optimiser (tools/train_utils/optimization/__init__.py:29-53: Adam with
true weight decay, betas (0.9, 0.99); OneCycle drives the learning rate and beta1).

Layout: every parameter becomes a view into ONE flat fp32 buffer (`p.data` is re-pointed once, at construction;
16-byte aligned slices), gradients are gathered into a flat buffer of the same layout after backward (a multi-tensor
copy), both moments are flat.  Consequences:
  * the update is two launches moving 7 x 4 B per element once instead of ~300 launches of per-tensor ops;
  * the data-parallel exchange is one all-reduce ON the flat gradient buffer: no pack / unpack copies;
  * learning rate, beta1 and the step count are device scalars, so a recorded HIP graph replays the update.
The kernel writes parameters through raw pointers: torch's version counters do not move.  Caches of tensors derived
from weights (packed sparse-conv images, folded BatchNorms, recorded inference graphs) key on
`_lib.weights_epoch()` as well, which step() -- and the step() of a recorded training step, whose replay runs no
Python -- advances; in training mode the sparse convs re-pack inside the step anyway
(spconv.core.SparseConvolution._packed_weight)."""
import ctypes

import torch

from . import _lib


class FlatGrads:
    """The gradient half of FlatAdamW for a step whose optimizer belongs to the CALLER (dropin.record): one flat fp32 gradient
    buffer, every parameter's slice of it as a view with the parameter's own strides (`grad_views`, lent to the kernels that can
    write a gradient in place through `p._glx_grad_view`), and pack_grads() gathering whatever backward left in `.grad`.  The
    parameters themselves are NOT touched (no flat parameter buffer): a torch optimizer steps them as it finds them."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGrads: no trainable parameters")
        if any(not p.is_cuda for p in self.params):
            raise _lib.GlxError("FlatGrads: parameters must be device tensors (HIP); there is no CPU path")
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:                       # 16-byte aligned slices
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.n = n
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self._gen = [0]
        self.grad_views = []
        for p, o in zip(self.params, self.offsets):
            self.grad_views.append(FlatAdamW._view(self.flat_grad, o, p))
            p._glx_grad_view = self.grad_views[-1]
            p._glx_grad_gen = self._gen

    def pack_grads(self):
        return FlatAdamW.pack_grads(self)

    def release(self):
        """Take the lending marks off the parameters again (the recorder is being dropped)."""
        for p in self.params:
            for a in ("_glx_grad_view", "_glx_grad_gen", "_glx_grad_lent"):
                if hasattr(p, a):
                    delattr(p, a)


class FlatAdamW:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = self.params[0].device
        if any(not p.is_cuda for p in self.params):
            raise _lib.GlxError("FlatAdamW: parameters must be device tensors (HIP); there is no CPU path")
        if any(p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdamW: fp32 parameters only")
        self.offsets, n = [], 0
        for p in self.params:                       # 16-byte aligned slices
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.n = n
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad_views = []
        self._gen = [0]             # this optimizer's lending counter (_lib.grad_buffer): bumped by ITS pack_grads only
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self._view(self.flat_param, o, p)
                view.copy_(p.data)
                p.data = view
                self.grad_views.append(self._view(self.flat_grad, o, p))
                p._glx_grad_view = self.grad_views[-1]      # producers that can write a gradient in place (pack_grads skips it)
                p._glx_grad_gen = self._gen
        self.hyper = torch.tensor([float(lr), float(betas[0])], dtype=torch.float32, device=dev)
        self.beta2, self.eps, self.weight_decay = float(betas[1]), float(eps), float(weight_decay)
        self.max_norm = float(max_norm) if max_norm else 0.0
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.grad_scale = 1.0          # the update reads grads * grad_scale: 1 / world_size after a SUM all-reduce
        self._ws = torch.empty(_lib.query("glx_adamw_workspace_bytes"), dtype=torch.uint8, device=dev)
        self._zeros = None
        self.after_pack = []

    @staticmethod
    def _view(flat, offset, p):
        """The slice of `flat` seen with the parameter's own strides: a dense parameter keeps its memory format
        (channels-last convolution weights stay channels-last, so MIOpen's NHWC kernels take them -- and hand back
        their gradients -- without a layout copy per call); anything else becomes row-major."""
        t = p.data
        keeps = ((t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
                 or (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d)))
        if t.is_contiguous() or not keeps:
            return flat[offset:offset + p.numel()].view_as(p)
        return torch.as_strided(flat, p.shape, p.stride(), offset)

    def set_lr(self, lr, beta1=None):
        """Device scalars the (recorded) update reads; two fills, no synchronisation."""
        self.hyper[0:1].fill_(float(lr))
        if beta1 is not None:
            self.hyper[1:2].fill_(float(beta1))

    def pack_grads(self, only=None, bump=True):
        """Gather the .grad tensors into the flat gradient buffer (a parameter without a gradient contributes
        zeros, as an optimizer that skips it would leave it -- except for weight decay, which torch skips too
        for such parameters; the training step gives every parameter a gradient).
        only: indices into `params` (a bucket of the step's exchange: the rest is packed by a later call);
        bump=False: more of this step's gradients follow (the lending generation moves with the last call)."""
        # (a gradient that was computed INTO its view -- dense_path.run_deferred_fc_wgrads does that for the 21 MB first RoI
        # Linear, 70 % of this copy -- needs none)
        have, missing = [], []
        pairs = zip(self.grad_views, self.params) if only is None else ((self.grad_views[i], self.params[i]) for i in only)
        for v, p in pairs:
            if p.grad is None:
                missing.append(v)           # genuinely no gradient this step (in-place gradients are NOT missing)
            elif not (p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == v.stride()):
                have.append((v, p.grad))
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([h[0] for h in have], [h[1] for h in have])
        if bump:
            self._gen[0] += 1             # the views may be lent to the next backward pass (_lib.grad_buffer)
            _lib.next_grad_generation()   # (parameters without an owning optimizer follow the process-wide counter)
        return self.flat_grad

    def buckets(self, late_params):
        """The flat buffer as two buckets of a step whose gradients become final in two instalments: `late_params` (one
        contiguous run of `params`: the sparse backbone, whose backward ends the step) and everything else.
        -> (early element ranges [(lo, hi), ...], late range (lo, hi), early parameter indices, late parameter indices)."""
        ids = {id(p) for p in late_params}
        late = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not late:
            raise ValueError("buckets: none of the late parameters belongs to this optimizer")
        if late != list(range(late[0], late[-1] + 1)):
            raise ValueError("buckets: the late parameters are not one contiguous run of the flat buffer")
        lo = self.offsets[late[0]]
        hi = self.offsets[late[-1] + 1] if late[-1] + 1 < len(self.params) else self.n
        early = [r for r in ((0, lo), (hi, self.n)) if r[1] > r[0]]
        keep = set(late)
        return early, (lo, hi), [i for i in range(len(self.params)) if i not in keep], late

    def allreduce_ranges_(self, ranges, average=True, async_op=False):
        """SUM all-reduce of element ranges of the flat gradient buffer (a bucket of the exchange).  async_op (RCCL): the
        collectives are queued behind what the current stream holds and the handles are returned -- wait() on them makes
        the current stream wait; over gloo with device tensors the exchange goes through host copies and is complete
        on return."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return []
        if dist.get_world_size() == 1 and not getattr(self, "exchange_alone", False):
            return []
        self.grad_scale = 1.0 / dist.get_world_size() if average else 1.0
        handles = []
        for lo, hi in ranges:
            buf = self.flat_grad[lo:hi]
            if dist.get_backend() == "gloo" and buf.is_cuda:
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                buf.copy_(host)
            elif async_op:
                handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True))
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return handles

    def allreduce_(self, average=True):
        """Data-parallel exchange: one SUM all-reduce on the flat gradient buffer (RCCL; gloo for the CPU-side
        plumbing tests goes through a host copy).  average: the 1 / world_size of DistributedDataParallel is folded
        into the update (`grad_scale`, read by glx_adamw_clip_step_scaled) instead of a pass over the buffer."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size() == 1 and not getattr(self, "exchange_alone", False):
            return
        self.grad_scale = 1.0 / dist.get_world_size() if average else 1.0
        buf = self.flat_grad
        if dist.get_backend() == "gloo" and buf.is_cuda:
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    def broadcast_state_(self, src=0):
        """Every rank starts from rank `src`'s parameters, moments and step count (DistributedDataParallel
        broadcasts the module state at construction, tools/train.py:144-145; without it ranks that were seeded or
        restored differently would average gradients of different models)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        for buf in (self.flat_param, self.exp_avg, self.exp_avg_sq, self.step_count, self.hyper):
            if dist.get_backend() == "gloo" and buf.is_cuda:
                host = buf.cpu()
                dist.broadcast(host, src)
                buf.copy_(host)
            else:
                dist.broadcast(buf, src)
        _lib.bump_weights_epoch(self.params)

    def l2_norm_sum(self, params, scale=1.0):
        """scale * (sum of the 2-norms of `params`, tensors of this optimizer) from the flat parameter buffer, two launches, no
        autograd (cvae_uncertainty/model.py:20-28); `add_l2_norm_grad` adds the gradient of the last value to flat_grad."""
        key = tuple(id(p) for p in params)
        if getattr(self, "_l2_key", None) != key:
            index = {id(p): i for i, p in enumerate(self.params)}
            missing = [i for i, p in enumerate(params) if id(p) not in index]
            if missing:
                raise ValueError("FlatAdamW.l2_norm_sum: %d tensors are not parameters of this optimizer" % len(missing))
            order = sorted({index[id(p)] for p in params})
            segs = [[self.offsets[i], self.params[i].numel()] for i in order]
            self._l2_segs = torch.tensor(segs, dtype=torch.int64, device=self.flat_param.device)
            self._l2_norms = torch.zeros(len(segs), dtype=torch.float32, device=self.flat_param.device)
            self._l2_total = torch.zeros(1, dtype=torch.float32, device=self.flat_param.device)
            self._l2_ws = torch.empty(_lib.query("glx_flat_l2_workspace_bytes", len(segs)), dtype=torch.uint8, device=self.flat_param.device)
            self._l2_key = key
        self._l2_scale = float(scale)
        _lib.call("glx_flat_l2_norms", self.flat_param, self._l2_segs, int(self._l2_segs.shape[0]), ctypes.c_float(self._l2_scale),
                  self._l2_norms, self._l2_total, self._l2_ws, _lib.size_arg(self._l2_ws.numel()))
        return self._l2_total[0]

    def add_l2_norm_grad(self, coef=None):
        """flat_grad += coef * d(l2_norm_sum's last value) / d params (coef: a device scalar, None = 1)."""
        _lib.call("glx_flat_l2_norm_grad_add", self.flat_param, self._l2_segs, int(self._l2_segs.shape[0]), self._l2_norms, coef,
                  ctypes.c_float(self._l2_scale), self.flat_grad, ctypes.c_int64(self.n))

    def step(self, packed=False):
        """clip + AdamW.  packed=True: flat_grad already holds this step's (exchanged) gradients."""
        if not packed:
            self.pack_grads()
            for fn in self.after_pack:          # gradient terms that are cheaper to add to the flat buffer than to hand to autograd
                fn()
        _lib.call("glx_adamw_clip_step_scaled", self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq,
                  ctypes.c_int64(self.n), self.hyper, ctypes.c_float(self.beta2), ctypes.c_float(self.eps),
                  ctypes.c_float(self.weight_decay), ctypes.c_float(self.max_norm), ctypes.c_float(self.grad_scale),
                  self.step_count, self.grad_norm, self._ws, _lib.size_arg(self._ws.numel()))
        _lib.bump_weights_epoch(self.params)      # changed through raw pointers: version-keyed caches of THESE are stale

    def bump_versions(self):
        """Tell torch the parameters changed (eager loops that rely on version-keyed caches)."""
        torch.autograd.graph.increment_version(self.params)

    def state_dict(self):
        return dict(exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), step=self.step_count.clone(),
                    hyper=self.hyper.clone())

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count.copy_(sd["step"])
        self.hyper.copy_(sd["hyper"])
