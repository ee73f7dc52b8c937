"""3x3 / stride-1 / pad-1 convolutions of the BEV backbone on our own kernels (csrc/glx_conv2d.hip).

BaseBEVBackbone's blocks (pcdet/models/backbones_2d/base_bev_backbone.py:30-49) are 47 % of the training step and all
fp32 3x3 convolutions; the fp32 MFMA of CDNA4 is 1/16 of its bf16 rate, so the kernels here compute the fp32 products as
six bf16 products of three-way split operands with fp32 accumulation (fp32 accuracy, 6/16 of the matrix time).
`conv3x3(x, weight)` is `F.conv2d(x, weight, None, 1, 1)` for channels-last maps; its backward runs the same kernel on
the flipped pack for the input gradient and leaves the weight gradient on the step's weight-gradient stream."""
import contextlib
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import call, query


def supported(x, weight, stride, padding, dilation, groups, bias):
    """The layers these kernels cover: 3x3, stride 1, zero padding 1, no bias, channel counts that tile."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and bias is None
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0
            and x.shape[1] == weight.shape[1] and x.is_contiguous(memory_format=torch.channels_last)
            # the kernels index a map with 32-bit element offsets (GLX_REQUIRE in glx_conv3x3_forward): larger batches
            # take the caller's fallback instead of an error
            and x.shape[0] * x.shape[2] * x.shape[3] * max(int(weight.shape[0]), int(weight.shape[1])) < (1 << 31))


def bn_state_available():
    from .spconv import core
    return core.USE_BN_STATE


_packs = {}
# Piece images packed ahead for the current training step by prepack(): {weight data_ptr: (fwd, bwd)}; the caller
# resets it to None after the step (the images are valid only while the weights do not change).
STEP_PACKS = None


# A weight whose forward image the strided layer's kernel reads (csrc/glx_deconv2d.hip: bf16x3 whatever the process arithmetic
# is) carries the attribute _glx_strided_pack (set at its first strided call); prepack() packs it accordingly.


def prepack(weights):
    """Both piece images of every (Cout, Cin, 3, 3) weight in `weights` with ONE launch (glx_conv3x3_pack_multi_arith),
    parked in STEP_PACKS for packs() to hand out."""
    global STEP_PACKS
    STEP_PACKS = {}
    jobs = []
    for w in weights:
        kind = "bf16x3" if getattr(w, "_glx_strided_pack", False) else arithmetic()
        w = w.detach()
        cout, cin = int(w.shape[0]), int(w.shape[1])
        if not (w.is_cuda and w.dtype == torch.float32 and tuple(w.shape[2:]) == (3, 3) and cout % 64 == 0 and cin % 64 == 0):
            continue
        n = query("glx_conv3x3_packed_bytes", cin, cout)
        jobs.append((w, cin, cout, torch.empty(n, dtype=torch.uint8, device=w.device),
                     torch.empty(n, dtype=torch.uint8, device=w.device), kind))
    if not jobs:
        return
    n = len(jobs)
    ptrs = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
    strides = (ctypes.c_longlong * (4 * n))(*[int(v) for j in jobs for v in j[0].stride()])
    cins = (ctypes.c_int32 * n)(*[j[1] for j in jobs])
    couts = (ctypes.c_int32 * n)(*[j[2] for j in jobs])
    fwd = (ctypes.c_void_p * n)(*[j[3].data_ptr() for j in jobs])
    bwd = (ctypes.c_void_p * n)(*[j[4].data_ptr() for j in jobs])
    kinds = (ctypes.c_int32 * n)(*[1 if j[5] == "f16x2" else 0 for j in jobs])
    call("glx_conv3x3_pack_multi_arith", n, ptrs, strides, cins, couts, fwd, bwd, kinds)
    for j in jobs:
        STEP_PACKS[(j[0].data_ptr(), tuple(j[0].stride()))] = (j[3], j[4], j[5])


def arithmetic():
    """'f16x2' (default) or 'bf16x3': how the forward / input-gradient kernel forms its fp32 products (csrc/glx_conv2d.hip;
    env GLX_CONV3X3_ARITH)."""
    return "f16x2" if query("glx_conv3x3_get_arith") else "bf16x3"


def set_arithmetic(name):
    """Switch the process to 'f16x2' or 'bf16x3'; the cached packs are dropped (they carry the layout of the arithmetic they
    were made under).  Not while a recorded step that holds packs is alive: its replays would read the old layout."""
    if name not in ("f16x2", "bf16x3"):
        raise ValueError("conv3x3 arithmetic is 'f16x2' or 'bf16x3', got %r" % (name,))
    old = arithmetic()
    if name != old:
        _lib.load().glx_conv3x3_set_arith(1 if name == "f16x2" else 0)
        _packs.clear()
        _lib.bump_weights_epoch()
    return old


def packs(weight, strided=False):
    """(fwd, bwd) piece images of a (Cout, Cin, 3, 3) weight; rebuilt when the weights epoch or the tensor's version
    moves (one launch writes both).  strided: the images for glx_conv3x3s2_forward* (the bf16x3 layout whatever the process
    arithmetic is)."""
    kind = "bf16x3" if strided else arithmetic()
    if strided and not getattr(weight, "_glx_strided_pack", False):
        try:
            weight._glx_strided_pack = True
        except AttributeError:
            pass
    if STEP_PACKS is not None:
        hit = STEP_PACKS.get((weight.data_ptr(), tuple(weight.stride())))
        if hit is not None and hit[2] == kind:
            return hit[0], hit[1]
    key = (weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()), kind)
    tag = (_lib.weights_epoch(weight), weight._version)
    hit = _packs.get(key)
    if hit is not None and hit[3]() is not weight:
        hit = None                         # another tensor that happens to live where a freed weight did
    if hit is not None and hit[0] == tag and not torch.cuda.is_current_stream_capturing():
        return hit[1], hit[2]
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    if hit is not None:
        fwd, bwd = hit[1], hit[2]          # same storage every step: a recorded step rewrites it in place
    else:
        n = query("glx_conv3x3_packed_bytes", cin, cout)
        fwd = torch.empty(n, dtype=torch.uint8, device=weight.device)
        bwd = torch.empty(n, dtype=torch.uint8, device=weight.device)
    s = weight.stride()
    ll = ctypes.c_longlong
    call("glx_conv3x3_pack_arith", weight.detach(), ll(s[0]), ll(s[1]), ll(s[2]), ll(s[3]), cin, cout, fwd, bwd,
         1 if kind == "f16x2" else 0)
    _packs[key] = (tag, fwd, bwd, weakref.ref(weight))
    return fwd, bwd


def _run(x, pack, cout, bn=None, epi=None, pre=None, bwd=None):
    """bn: a training-mode BatchNorm2d that follows the conv -- its batch statistics are taken in the kernel's epilogue
    (glx_conv_opts.bn) and the call returns (y, coef, save_mean, save_invstd).  epi: an _lib.Epilogue (inference).
    pre: (coef (2 * Cin: scale, shift), relu) -- the input is transformed on load (glx_conv_opts.prologue).
    bwd: (y_prev, coef_prev, mean, invstd, gamma) -- the call is an input-gradient convolution whose output is the gradient of
    relu(bn(y_prev)): the epilogue masks it and takes the BatchNorm backward's sums (glx_conv_opts.bn_bwd); returns
    (dz, coef (3 C), dgamma, dbeta)."""
    b, c, h, w = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.GlxError("conv3x3 expects a float32 channels-last device map, got %s strides %s on %s"
                            % (x.dtype, tuple(x.stride()), x.device))
    y = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    stats = opts = None
    prologue = None
    if pre is not None:
        coef, relu = pre
        prologue = ctypes.pointer(_lib.epilogue(coef[:c], coef[c:], relu))
    if bn is not None:
        from .spconv import core
        stats = tuple(torch.empty(n, dtype=torch.float32, device=x.device) for n in (2 * cout, cout, cout))
        st = _lib.bn_stats(core._bn_state(x.device), bn, *stats)
        opts = _lib.ConvOpts(ctypes.pointer(st), None, prologue)
    elif epi is not None:
        opts = _lib.ConvOpts(None, ctypes.pointer(epi), prologue)
    elif prologue is not None:
        opts = _lib.ConvOpts(None, None, prologue)
    if bwd is not None:
        from .spconv import core
        y_prev, coef_prev, mean, invstd, gamma = bwd
        stats = tuple(torch.empty(n, dtype=torch.float32, device=x.device) for n in (3 * cout, cout, cout))
        st = _lib.BnBwdStats(*[_lib._p(t) for t in (core._bn_state(x.device), y_prev, coef_prev, mean, invstd, gamma) + stats])
        opts = _lib.ConvOpts(None, None, None, ctypes.pointer(st))
    call("glx_conv3x3_forward_ex", x, b, h, w, c, pack, cout, y, ctypes.byref(opts) if opts is not None else None)
    if bn is not None:
        if bn.track_running_stats:
            _lib.bump_weights_epoch((bn.running_mean, bn.running_var))     # moved behind torch's back
        return (y,) + stats
    if bwd is not None:
        return (y,) + stats
    return y


def wgrad(x, gy, weight, pre=None):
    """dW of conv3x3 for a weight of `weight`'s shape and strides (a fresh tensor); pre as in _run (the input is
    transformed on load).  The split-K workspace is the
    (device, CURRENT stream, pipeline scope) buffer of _lib.workspace: weight gradients launched on different streams
    (the staged backward moves them between the main and the side stream; two pipelines) never share it."""
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    b, _, h, w = x.shape
    n = query("glx_conv3x3_wgrad_workspace_bytes", cin, cout)
    ll = ctypes.c_longlong
    prologue = None
    if pre is not None:
        coef, relu = pre
        prologue = ctypes.byref(_lib.epilogue(coef[:cin], coef[cin:], relu))
    gw = _lib.grad_buffer(weight)       # the optimizer's flat gradient buffer when the step has one (no gather copy)
    s = gw.stride()
    if DEFERRED_WGRAD_REDUCES is not None and _lib.is_lent(weight, gw):
        # the blocks' partial sums now, into a buffer of this layer's own; their sum with every other layer's in ONE launch when
        # the step calls run_deferred_wgrad_reduces() (autograd only keeps `gw`, the optimizer's view, and reads nothing)
        ws = torch.empty(n, dtype=torch.uint8, device=x.device)
        call("glx_conv3x3_wgrad_ex", x, gy, b, h, w, cin, cout, None, ll(s[0]), ll(s[1]), ll(s[2]), ll(s[3]), prologue, ws,
             _lib.size_arg(n))
        # an ALIAS of gw in the job: AccumulateGrad keeps a gradient as it is only while nobody else holds the tensor object
        DEFERRED_WGRAD_REDUCES.append((cin, cout, gw.detach(), tuple(s), ws, torch.cuda.current_stream(x.device), weight))
        return gw
    ws = _lib.workspace.get(n, x.device)
    call("glx_conv3x3_wgrad_ex", x, gy, b, h, w, cin, cout, gw, ll(s[0]), ll(s[1]), ll(s[2]), ll(s[3]), prologue, ws,
         _lib.size_arg(n))
    return gw


# A training step that sets this to a list gets the 3x3 layers' weight gradients in two parts: each layer's partial sums where the
# backward pass reaches it, the sums of ALL layers in one launch from run_deferred_wgrad_reduces() (ten ~8 us launches on the
# step's main chain become one).  None (the default): every weight gradient is complete when its call returns.
DEFERRED_WGRAD_REDUCES = None


def run_deferred_wgrad_reduces(jobs):
    """Finish the weight gradients `jobs` (the list DEFERRED_WGRAD_REDUCES was) on the current stream, which waits for the
    streams the partial sums were written on."""
    if not jobs:
        return
    cur = torch.cuda.current_stream(jobs[0][2].device)
    for st in {j[5] for j in jobs}:
        if st != cur:
            cur.wait_stream(st)
    n = len(jobs)
    i32 = ctypes.c_int32 * n
    strides = (ctypes.c_longlong * (4 * n))(*[v for j in jobs for v in j[3]])
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    call("glx_conv3x3_wgrad_reduce_multi", n, i32(*[j[0] for j in jobs]), i32(*[j[1] for j in jobs]), ptrs([j[2] for j in jobs]),
         strides, ptrs([j[4] for j in jobs]), (ctypes.c_size_t * n)(*[j[4].numel() for j in jobs]))
    for j in jobs:
        j[4].record_stream(cur)
        _lib.settle_lent_grad(j[6], j[2])      # param.grad IS the filled view, or gets its contents


OWN_WGRAD = True
BN_BWD_IN_DGRAD = True     # _ConvPre3x3: BatchNorm backward sums in the dgrad epilogue


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bn=None):
        fwd, bwd = packs(weight)
        ctx.save_for_backward(x, weight)
        ctx.bwd_pack = bwd
        out = _run(x.detach(), fwd, int(weight.shape[0]), bn)
        if bn is not None:
            ctx.mark_non_differentiable(*out[1:])
            ctx.set_materialize_grads(False)       # no zero tensors (3 fill launches) for the statistics' "gradients"
        return out

    @staticmethod
    def backward(ctx, gy, *_):
        from .spconv import core
        x, weight = ctx.saved_tensors
        if gy is None:
            return None, None, None
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        if ctx.needs_input_grad[1]:
            side = core.WGRAD_STREAM
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(x.device))
                for t in (x, gy, weight):
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                if OWN_WGRAD:
                    gw = wgrad(x, gy, weight)
                else:
                    gw = torch.ops.aten.convolution_backward(gy, x, weight, None, (1, 1), (1, 1), (1, 1), False,
                                                             (0, 0), 1, [False, True, False])[1]
        if ctx.needs_input_grad[0]:
            gx = _run(gy, ctx.bwd_pack, int(weight.shape[1]))
        return gx, gw, None


class _ConvPre3x3(torch.autograd.Function):
    """conv3x3(relu(bn_prev(y_prev)), weight) (+ the statistics of `bn` in the epilogue) WITHOUT the normalised map: the
    convolution and its weight gradient transform y_prev on load (glx_conv_opts.prologue, coef_prev = the scale / shift the
    previous convolution's epilogue left), and this node's backward carries the previous BatchNorm's backward
    (glx_bn_relu_backward on y_prev and the input gradient): one pass over the map less per layer forward.
    Inputs (y_prev, coef_prev, mean_prev, invstd_prev, gamma_prev, beta_prev, weight, bn) -> (y, coef, mean, invstd)."""

    @staticmethod
    def forward(ctx, y_prev, coef_prev, mean_prev, invstd_prev, gamma_prev, beta_prev, weight, bn):
        fwd, bwd = packs(weight)
        ctx.save_for_backward(y_prev, coef_prev, mean_prev, invstd_prev, gamma_prev, beta_prev, weight)
        ctx.bwd_pack = bwd
        out = _run(y_prev.detach(), fwd, int(weight.shape[0]), bn, pre=(coef_prev, True))
        ctx.mark_non_differentiable(*out[1:])
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gy, *_):
        from .spconv import core
        y_prev, coef_prev, mean_prev, invstd_prev, gamma_prev, beta_prev, weight = ctx.saved_tensors
        if gy is None:
            return (None,) * 8
        gy = gy.contiguous(memory_format=torch.channels_last)
        gw = None
        if ctx.needs_input_grad[6]:
            side = core.WGRAD_STREAM
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(y_prev.device))
                for t in (y_prev, gy, weight, coef_prev):
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                gw = wgrad(y_prev, gy, weight, pre=(coef_prev, True))
        b, c, h, w = y_prev.shape
        n = b * h * w
        rows = y_prev.permute(0, 2, 3, 1).reshape(n, c)
        dx = torch.empty_like(rows)
        if BN_BWD_IN_DGRAD:
            # the input-gradient convolution masks its output with the ReLU and takes the BatchNorm backward's sums in its
            # epilogue (no statistics pass over the two maps); what is left is the transform
            dz, coef3, dgamma, dbeta = _run(gy, ctx.bwd_pack, int(weight.shape[1]),
                                            bwd=(y_prev, coef_prev, mean_prev, invstd_prev, gamma_prev))
            call("glx_bn_backward_apply", rows, dz.permute(0, 2, 3, 1).reshape(n, c), coef3, mean_prev, invstd_prev, n, c, None, dx)
        else:
            gh = _run(gy, ctx.bwd_pack, int(weight.shape[1]))              # gradient of relu(bn_prev(y_prev))
            grows = gh.permute(0, 2, 3, 1).reshape(n, c)
            dgamma = torch.empty(c, dtype=torch.float32, device=rows.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=rows.device)
            ws = core.workspace.get(query("glx_bn_workspace_bytes", c), rows.device)
            call("glx_bn_relu_backward", rows, grows, None, n, c, gamma_prev, beta_prev, mean_prev, invstd_prev, 1, dx,
                 dgamma, dbeta, None, ws, _lib.size_arg(ws.numel()), core._bn_state(rows.device), 0)
        return dx.view(b, h, w, c).permute(0, 3, 1, 2), None, None, None, dgamma, dbeta, gw, None


def conv3x3(x, weight):
    return _Conv3x3.apply(x, weight)


def _count_batch(bn):
    from .spconv import core
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if core.DEFERRED_COUNTERS is not None:
            core.DEFERRED_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


def conv3x3_bn_raw(x, weight, bn, pending=None):
    """The convolution in front of a training-mode BatchNorm2d with the statistics in its epilogue, the transform NOT
    applied: returns (y, coef, mean, invstd, bn) for bn_apply() or for the next layer's conv3x3_bn_raw(pending=...), which
    then reads y through the transform (x is ignored)."""
    if pending is not None:
        y_prev, coef_prev, mean_prev, invstd_prev, bn_prev = pending
        out = _ConvPre3x3.apply(y_prev, coef_prev, mean_prev, invstd_prev, bn_prev.weight, bn_prev.bias, weight, bn)
    else:
        out = _Conv3x3.apply(x, weight, bn)
    _count_batch(bn)
    return out + (bn,)


def bn_apply(pending, relu):
    """relu?(bn(y)) of a conv3x3_bn_raw() result as a map (csrc/glx_bn.hip row kernels)."""
    from .spconv import core
    y, coef, mean, invstd, bn = pending
    b, c, h, w = y.shape
    rows = y.permute(0, 2, 3, 1).reshape(b * h * w, c)              # a view of channels-last memory
    out = core.FusedBNApply.apply(rows, coef, mean, invstd, bn.weight, bn.bias, relu)
    return out.view(b, h, w, c).permute(0, 3, 1, 2)


def conv3x3_bn(x, weight, bn, relu):
    """relu?(bn(conv3x3(x, weight))) for a training-mode nn.BatchNorm2d: statistics in the conv's epilogue, the
    transform (and the whole backward of the BatchNorm) on the fused row kernels of csrc/glx_bn.hip."""
    return bn_apply(conv3x3_bn_raw(x, weight, bn), relu)


# ------------------------------------------------------------------------------------------------ transposed convolutions
def deconv_supported(x, weight, stride, padding, output_padding, dilation, groups, bias):
    """ConvTranspose2d(c, cu, u, stride=u, bias=False), u in {1, 2} (BaseBEVBackbone's deblocks) on channels-last maps."""
    u = int(weight.shape[2])
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and bias is None
            and u in (1, 2) and int(weight.shape[3]) == u and tuple(stride) == (u, u) and tuple(padding) == (0, 0)
            and tuple(output_padding) == (0, 0) and tuple(dilation) == (1, 1) and groups == 1
            and weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0 and x.shape[1] == weight.shape[0]
            and x.shape[3] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last))


def _deconv_packs(weight):
    """(fwd, bwd) piece images of a (Cin, Cout, u, u) weight, packed at every call (two per training step)."""
    cin, cout, u = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
    n = query("glx_deconv_packed_bytes", cin, cout, u)
    fwd = torch.empty(n, dtype=torch.uint8, device=weight.device)
    bwd = torch.empty(n, dtype=torch.uint8, device=weight.device)
    s = weight.stride()
    ll = ctypes.c_longlong
    call("glx_deconv_pack", weight.detach(), ll(s[0]), ll(s[1]), ll(s[2]), ll(s[3]), cin, cout, u, fwd, bwd)
    return fwd, bwd


class _Deconv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bn=None):
        """bn: the training-mode BatchNorm2d behind the transposed convolution -- its statistics ride in the kernel's
        epilogue (glx_deconv_forward_bn) and the call returns (y, coef, save_mean, save_invstd)."""
        cin, cout, u = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        fwd, bwd = _deconv_packs(weight)
        b, _, h, w = x.shape
        x = x.detach()
        y = torch.empty((b, cout, h * u, w * u), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        ctx.save_for_backward(x, weight)
        ctx.bwd_pack = bwd
        if bn is None:
            call("glx_deconv_forward", x, b, h, w, cin, fwd, cout, u, y)
            return y
        from .spconv import core
        stats = tuple(torch.empty(n, dtype=torch.float32, device=x.device) for n in (2 * cout, cout, cout))
        st = _lib.bn_stats(core._bn_state(x.device), bn, *stats)
        call("glx_deconv_forward_bn", x, b, h, w, cin, fwd, cout, u, y, ctypes.byref(st))
        if bn.track_running_stats:
            _lib.bump_weights_epoch((bn.running_mean, bn.running_var))
        ctx.mark_non_differentiable(*stats)
        ctx.set_materialize_grads(False)
        return (y,) + stats

    @staticmethod
    def backward(ctx, gy, *_):
        from .spconv import core
        if gy is None:
            return None, None, None
        x, weight = ctx.saved_tensors
        cin, cout, u = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        b, _, h, w = x.shape
        gy = gy.contiguous(memory_format=torch.channels_last)
        gx = gw = None
        if ctx.needs_input_grad[1]:
            side = core.WGRAD_STREAM
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(x.device))
                for t in (x, gy, weight):
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                n = query("glx_deconv_wgrad_workspace_bytes", cin, cout, u)
                ws = _lib.workspace.get(n, x.device)          # per (device, current stream): see wgrad()
                gw = torch.empty_like(weight)
                s = gw.stride()
                ll = ctypes.c_longlong
                call("glx_deconv_wgrad", x, gy, b, h, w, cin, cout, u, gw, ll(s[0]), ll(s[1]), ll(s[2]), ll(s[3]), ws,
                     _lib.size_arg(n))
        if ctx.needs_input_grad[0]:
            gx = torch.empty((b, cin, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            call("glx_deconv_input_grad", gy, b, h, w, cin, ctx.bwd_pack, cout, u, gx)
        return gx, gw, None


def deconv_bn_raw(x, weight, bn):
    """ConvTranspose2d in front of a training-mode BatchNorm2d with the statistics in its epilogue, the transform not
    applied: (y, coef, mean, invstd) for spconv.core.FusedBNApplyCat."""
    out = _Deconv.apply(x, weight, bn)
    _count_batch(bn)
    return out


def deconv(x, weight):
    """F.conv_transpose2d(x, weight, None, stride=u) for a (Cin, Cout, u, u) weight, u in {1, 2}."""
    return _Deconv.apply(x, weight)


# ------------------------------------------------------------------------------------------------ inference (folded BatchNorm)
_deconv_pack_cache = {}


def _deconv_packs_cached(weight):
    key = (weight.data_ptr(), tuple(weight.shape), tuple(weight.stride()))
    tag = (_lib.weights_epoch(weight), weight._version)
    hit = _deconv_pack_cache.get(key)
    if hit is not None and hit[3]() is weight and hit[0] == tag and not torch.cuda.is_current_stream_capturing():
        return hit[1], hit[2]
    fwd, bwd = _deconv_packs(weight)
    _deconv_pack_cache[key] = (tag, fwd, bwd, weakref.ref(weight))
    return fwd, bwd


def conv3x3_affine(x, weight, scale, shift, relu):
    """relu?(conv3x3(x, weight) * scale[c] + shift[c]) in one launch (no autograd): an eval-mode BatchNorm2d folded into
    the convolution's epilogue."""
    fwd, _ = packs(weight)
    return _run(x, fwd, int(weight.shape[0]), epi=_lib.epilogue(scale, shift, relu))


def conv3x3s2_affine(x, weight, scale, shift, relu):
    """The same for the strided layer (3x3, stride 2, zero padding 1; even maps)."""
    fwd, _ = packs(weight, strided=True)
    b, c, h, w = x.shape
    cout = int(weight.shape[0])
    y = torch.empty((b, cout, h // 2, w // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    call("glx_conv3x3s2_forward_ex", x, b, h, w, c, fwd, cout, y, ctypes.byref(_lib.epilogue(scale, shift, relu)))
    return y


def deconv_affine(x, weight, scale, shift, relu, out=None, channel_offset=0):
    """relu?(conv_transpose2d(x, weight, stride=u) * scale[c] + shift[c]); with `out` (a channels-last (B, Ctot, uH, uW)
    map) the result goes straight into channels [channel_offset, channel_offset + Cout) of it."""
    cin, cout, u = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
    fwd, _ = _deconv_packs_cached(weight)
    b, _, h, w = x.shape
    if out is None:
        out = torch.empty((b, cout, h * u, w * u), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    ldc = int(out.shape[1])
    epi = _lib.epilogue(scale, shift, relu, ldc if ldc != cout else 0, channel_offset)
    call("glx_deconv_forward_ex", x, b, h, w, cin, fwd, cout, u, out, ctypes.byref(epi))
    return out
