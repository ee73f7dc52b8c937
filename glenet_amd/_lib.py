"""ctypes binding of libglenet_hip.so (the C ABI declared in include/glenet_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this
module raises.  torch is used only as the owner of device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLX_HIP_LIB: another build of the same sources (tools/build_variant.sh: kernel experiments A/B'd on one box)
LIB_PATH = os.environ.get("GLX_HIP_LIB") or os.path.join(_HERE, "csrc", "libglenet_hip.so")

_lib = None

c_void_p, c_int, c_float, c_size_t, c_int64 = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64)


class GlxError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GlxError(
            "libglenet_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.glx_last_error.restype = ctypes.c_char_p
    lib.glx_abi_version.restype = c_int
    for name in ("glx_index_workspace_bytes", "glx_sconv_workspace_bytes",
                 "glx_sconv_wgrad_workspace_bytes", "glx_voxelize_hard_workspace_bytes",
                 "glx_voxelize_dynamic_workspace_bytes", "glx_nms_workspace_bytes",
                 "glx_roiaware_pool3d_workspace_bytes", "glx_roipoint_pool3d_workspace_bytes",
                 "glx_assign_targets_workspace_bytes", "glx_rpn_loss_workspace_bytes",
                 "glx_group_points_grad_workspace_bytes", "glx_adamw_workspace_bytes", "glx_bn_workspace_bytes",
                 "glx_pos_pool_workspace_bytes", "glx_vector_pool_workspace_bytes",
                 "glx_sconv_tile_map_workspace_bytes", "glx_sconv_packed_bytes", "glx_mask_shuffle_workspace_bytes",
                 "glx_conv3x3_packed_bytes", "glx_conv3x3_wgrad_workspace_bytes",
                 "glx_deconv_packed_bytes", "glx_deconv_wgrad_workspace_bytes", "glx_pair_lists_bytes",
                 "glx_sconv_wgrad_pairs_workspace_bytes", "glx_pointmax_wsum_workspace_bytes", "glx_rows128_moments_workspace_bytes", "glx_rows_bwd_64_128_workspace_bytes",
                 "glx_head1x1_wgrad_workspace_bytes", "glx_topk_workspace_bytes", "glx_fc_tower_scratch_bytes",
                 "glx_bn_cm_workspace_bytes", "glx_rows_linear_workspace_bytes", "glx_narrowfeat_workspace_bytes", "glx_point_layer1_workspace_bytes", "glx_flat_l2_workspace_bytes"):
        if hasattr(lib, name):
            getattr(lib, name).restype = c_size_t
    lib.glx_index_words.restype = c_int64
    _lib = lib
    return lib


def _arg(a):
    """torch tensor -> device pointer; python scalars pass through; None -> NULL."""
    if a is None:
        return c_void_p(0)
    if isinstance(a, torch.Tensor):
        return c_void_p(a.data_ptr())
    if isinstance(a, float):
        return c_float(a)
    if isinstance(a, int):
        return c_int(a)
    return a


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current torch stream of the current device as a hipStream_t (every entry point's last argument).  The raw getter
    skips building a torch.cuda.Stream object per call: 925 of them per eager training step were 0.5 ms of host time."""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    """Call an int-returning entry point with the current torch stream appended."""
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*[_arg(a) for a in args], stream_ptr())
    if rc != 0:
        raise GlxError("%s failed (%d): %s" % (name, rc, lib.glx_last_error().decode()))
    return rc


def call_nostream(name, *args):
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*[_arg(a) for a in args])
    if rc != 0:
        raise GlxError("%s failed (%d): %s" % (name, rc, lib.glx_last_error().decode()))
    return rc


def query(name, *args):
    """Call a size_t/int64-returning query (no stream)."""
    lib = load()
    return int(getattr(lib, name)(*[_arg(a) for a in args]))


CONV_GRADS_IN_PLACE = True
_grad_generation = [0]      # bumped by FlatAdamW.pack_grads: one lending of a parameter's gradient view per optimizer step


def grad_generation_of(param):
    """The lending counter that governs `param`'s gradient view: its optimizer's own (`_glx_grad_gen`, a one-element list
    FlatAdamW shares with its parameters), else the process-wide one."""
    own = getattr(param, "_glx_grad_gen", None)
    return own[0] if own is not None else _grad_generation[0]


def next_grad_generation():
    _grad_generation[0] += 1


def grad_buffer(param, shape=None):
    """Where a weight-gradient kernel writes d loss / d `param`: a fresh alias of the optimizer's flat gradient buffer
    (FlatAdamW leaves `param._glx_grad_view`) when the parameter has no gradient yet -- autograd's AccumulateGrad takes the
    tensor as it is (no other owner, the parameter's layout), `FlatAdamW.pack_grads` finds it in place and copies nothing --
    else a new tensor.  shape: the kernel's view of the weight (same element order), default the parameter's."""
    view = getattr(param, "_glx_grad_view", None) if (CONV_GRADS_IN_PLACE and param is not None) else None
    # lent ONCE per optimizer step: a weight that two layers share gets two gradients in one backward pass, the second of
    # which must not land on the first (autograd adds it to .grad, which is then the view: still no gather).  The step counter
    # is the OWNING optimizer's (FlatAdamW stamps its parameters with its own counter): a second optimizer's pack_grads does
    # not re-open the lending of this one's views in the middle of an accumulation.
    # NOTE: a gradient obtained through torch.autograd.grad() (no AccumulateGrad) may therefore ALIAS the optimizer's flat
    # gradient buffer and is overwritten by the next step -- clone it if it has to outlive the step.
    gen = grad_generation_of(param)
    if (view is not None and param.grad is None and getattr(param, "_glx_grad_lent", -1) != gen
            and view.shape == param.shape and view.stride() == param.stride()
            and view.dtype == param.dtype and (shape is None or view.is_contiguous())):
        param._glx_grad_lent = gen
        g = view.detach()       # the parameter's own strides (channels-last filters keep theirs): the kernels write through them
        return g if shape is None else g.view(shape)
    like = param if shape is None else param.reshape(shape)
    return torch.empty_like(like, memory_format=torch.contiguous_format) if shape is not None else torch.empty_like(like)


def is_lent(param, g):
    """True when `g` (what grad_buffer returned) is the optimizer's own view of `param`'s gradient: autograd then keeps the tensor
    as it is and reads nothing, so a kernel may fill it later in the step (deferred weight-gradient sums)."""
    view = getattr(param, "_glx_grad_view", None) if param is not None else None
    if view is None or g.data_ptr() != view.data_ptr() or param.grad is not None:
        return False
    # AccumulateGrad steals the tensor untouched only in the plain case: no double backward (create_graph reads the gradient),
    # no tensor / post-accumulate hooks on the leaf (a DDP reducer is one) -- otherwise the gradient is complete when its call
    # returns
    if torch.is_grad_enabled() or getattr(param, "_backward_hooks", None) or getattr(param, "_post_accumulate_grad_hooks", None):
        return False
    return True


def settle_lent_grad(param, view):
    """After a deferred weight-gradient sum has filled `view` (the optimizer's lent view of `param`'s gradient): `param.grad`
    must BE that memory.  If autograd cloned the gradient at accumulate time (it then holds what the view held before the sum),
    the finished sum is copied over the clone; a gradient that is missing altogether is an error."""
    view = getattr(param, "_glx_grad_view", view)      # the parameter's own shape / strides (a kernel's view may be reshaped)
    g = param.grad
    if g is None:
        raise GlxError("deferred weight-gradient sum: the parameter has no .grad after the backward pass")
    if g.data_ptr() != view.data_ptr() or g.stride() != view.stride():
        g.copy_(view)


def size_arg(n):
    return c_size_t(int(n))


def double_arg(x):
    return ctypes.c_double(float(x))


# ---- per-call option structs of the *_ex entry points (include/glenet_hip.h: glx_bn_stats, glx_epilogue,
# glx_sconv_opts, glx_conv_opts): explicit arguments, no "next call" state anywhere
def _p(t):
    return None if t is None else (t.data_ptr() if isinstance(t, torch.Tensor) else t)


class BnStats(ctypes.Structure):
    _fields_ = [("state", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("momentum", c_float),
                ("coef", c_void_p), ("save_mean", c_void_p), ("save_invstd", c_void_p), ("running_mean", c_void_p),
                ("running_var", c_void_p), ("count", ctypes.c_int64)]


class Epilogue(ctypes.Structure):
    _fields_ = [("scale", c_void_p), ("shift", c_void_p), ("relu", c_int), ("ldc", c_int), ("coff", c_int)]


class BnBwdStats(ctypes.Structure):
    _fields_ = [("state", c_void_p), ("y", c_void_p), ("coef_fwd", c_void_p), ("mean", c_void_p), ("invstd", c_void_p),
                ("gamma", c_void_p), ("coef", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p)]


class SconvOpts(ctypes.Structure):
    _fields_ = [("tile_map", c_void_p), ("bn", ctypes.POINTER(BnStats)), ("profile_start", c_void_p),
                ("profile_stop", c_void_p), ("bn_bwd", ctypes.POINTER(BnBwdStats)), ("prologue", ctypes.POINTER(Epilogue))]


class ConvOpts(ctypes.Structure):
    _fields_ = [("bn", ctypes.POINTER(BnStats)), ("epilogue", ctypes.POINTER(Epilogue)), ("prologue", ctypes.POINTER(Epilogue)),
                ("bn_bwd", ctypes.POINTER(BnBwdStats))]


class RoiQuery(ctypes.Structure):
    """glx_roi_query (include/glenet_hip.h)."""
    _fields_ = [("Z", c_int), ("Y", c_int), ("X", c_int), ("nsample", c_int), ("z_range", c_int), ("y_range", c_int),
                ("x_range", c_int), ("stride", c_int), ("radius", c_float), ("indices", c_void_p), ("bitmap", c_void_p),
                ("prefix", c_void_p), ("rank_to_row", c_void_p), ("idx", c_void_p)]


class RoiHeadLossesArgs(ctypes.Structure):
    """glx_roi_head_losses_args (include/glenet_hip.h)."""
    _fields_ = [("ori_cls", c_void_p), ("std_logit", c_void_p), ("cls_labels", c_void_p), ("rcnn_reg", c_void_p),
                ("rcnn_reg_std", c_void_p), ("rois", c_void_p), ("gt_ct", c_void_p), ("gt_ct_ld", c_int),
                ("gt_src", c_void_p), ("gt_src_ld", c_int), ("label_var", c_void_p), ("reg_valid", c_void_p), ("R", c_int),
                ("code_weights", c_float * 7), ("beta", c_float), ("w_cls", c_float), ("w_reg", c_float),
                ("w_corner", c_float), ("rcnn_cls", c_void_p), ("out", c_void_p), ("grad_ori", c_void_p),
                ("grad_std_logit", c_void_p), ("grad_reg", c_void_p), ("grad_reg_std", c_void_p)]


class FcBn(ctypes.Structure):
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("save_mean", c_void_p), ("save_invstd", c_void_p), ("eps", c_float), ("momentum", c_float)]


class FcTower(ctypes.Structure):
    """glx_fc_tower (include/glenet_hip.h)."""
    _fields_ = [("R", c_int), ("drop_p", c_float), ("drop_u", c_void_p), ("z0", c_void_p), ("w", c_void_p * 6), ("bn", FcBn * 6),
                ("z", c_void_p * 6), ("h", c_void_p * 6), ("w_cls", c_void_p), ("b_cls", c_void_p), ("w_reg", c_void_p),
                ("b_reg", c_void_p), ("w_std", c_void_p), ("b_std", c_void_p), ("bn_s7", FcBn), ("w_fc1", c_void_p),
                ("b_fc1", c_void_p), ("bn_s64", FcBn), ("w_fc2", c_void_p), ("b_fc2", c_void_p), ("ori_cls", c_void_p),
                ("rcnn_reg", c_void_p), ("rcnn_reg_std", c_void_p), ("std_logit", c_void_p), ("scratch", c_void_p),
                ("barrier", c_void_p), ("cooperative", c_int)]


class FcTowerGrads(ctypes.Structure):
    """glx_fc_tower_grads (include/glenet_hip.h)."""
    _fields_ = [("g_cls", c_void_p), ("g_logit", c_void_p), ("g_reg", c_void_p), ("g_std", c_void_p), ("dz", c_void_p * 6),
                ("dgamma", c_void_p * 6), ("dbeta", c_void_p * 6), ("dw_cls", c_void_p), ("db_cls", c_void_p),
                ("dw_reg", c_void_p), ("db_reg", c_void_p), ("dw_std", c_void_p), ("db_std", c_void_p), ("dgamma7", c_void_p),
                ("dbeta7", c_void_p), ("dw_fc1", c_void_p), ("db_fc1", c_void_p), ("dgamma64", c_void_p), ("dbeta64", c_void_p),
                ("dw_fc2", c_void_p), ("db_fc2", c_void_p), ("scratch", c_void_p)]


def bn_stats(state, bn, coef, save_mean, save_invstd, count=0):
    """glx_bn_stats of a training-mode torch BatchNorm module (statistics of the convolution in front of it)."""
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return BnStats(_p(state), _p(bn.weight), _p(bn.bias), float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.1),
                   _p(coef), _p(save_mean), _p(save_invstd), _p(rm), _p(rv), int(count))


def epilogue(scale, shift, relu, ldc=0, coff=0):
    return Epilogue(_p(scale), _p(shift), 1 if relu else 0, int(ldc), int(coff))


def check_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise GlxError("expected a device tensor (HIP), got %s; there is no CPU path" % t.device)
        if not t.is_contiguous():
            raise GlxError("expected a contiguous tensor")


# ---------------------------------------------------------------------------------------------------------------
# Weights epoch.  The fused optimizer and the fused training BatchNorm write parameters / running statistics through
# raw pointers, and a replayed HIP graph runs no Python at all: torch's per-tensor version counters do not move.
# Every cache of tensors DERIVED from weights (packed sparse-conv images, eval-mode BatchNorm affines, folded
# BatchNorm + linear / conv weights, the weights tag of recorded inference graphs) therefore keys on
# (tensor versions, data pointers, weights_epoch()), and whatever updates weights behind torch's back calls
# bump_weights_epoch(): FlatAdamW.step, the fused training BatchNorm entry points, and the step() / replay() of the
# recorded training steps.  (train -> eval on the same module ran the first eval's packed weights.)
# The epoch is SCOPED to the tensors it protects (a process-global epoch made an eval pipeline re-record itself
# whenever ANY model trained): a writer names the tensors it wrote -- each then carries its own `_glx_epoch` -- and a reader
# asks for the epoch of the tensors its cache derives from.  A writer that cannot name them bumps the global part, which
# every reader includes (the conservative behaviour of before).
_weights_epoch = 0          # global part
_epoch_counter = 0


def weights_epoch(*tensors):
    """Epoch of the caches derived from `tensors` (None entries ignored): the global part + the newest per-tensor stamp."""
    e = _weights_epoch
    for t in tensors:
        if t is not None:
            te = t.__dict__.get("_glx_epoch", 0)
            if te > e:
                e = te
    return e


def tensors_epoch_sum(tensors):
    """Sum of the per-tensor stamps (they only grow): the tag of a cache that derives from many tensors."""
    return sum(t.__dict__.get("_glx_epoch", 0) for t in tensors)


def bump_weights_epoch(tensors=None):
    """tensors: the parameters / buffers that were written behind torch's back (their stamps move, nobody else's caches
    are invalidated); None: unknown -- everything is stale."""
    global _weights_epoch, _epoch_counter
    _epoch_counter += 1
    if tensors is None:
        _weights_epoch = _epoch_counter
    else:
        for t in tensors:
            if t is not None:
                t.__dict__["_glx_epoch"] = _epoch_counter
    return _epoch_counter


# ---------------------------------------------------------------------------------------------------------------
# hipGraphExec objects are never destroyed while the process lives (ROCm 7.2 workaround).  Destroying the exec of a
# graph with parallel branches leaves the runtime with a dangling hip::Stream: a LATER hipGraphLaunch of another graph
# segfaults in hip::Graph::UpdateStreams <- hip::GraphExec::Run (rocgdb backtrace: profiles/r03_graph_exec_destroy_
# crash.txt; reproduced by tests/test_backbone_gpu.py + tests/test_train_step_gpu.py in one process, gone with this
# list).  A pipeline that records itself again (weights changed) retires its old graph here instead of dropping it;
# the cost is that graph's private memory pool until exit -- 288 GB of HBM make that the cheaper side of the trade.
# GLX_KEEP_GRAPH_EXECS=0 restores normal lifetimes.
KEEP_GRAPH_EXECS = os.environ.get("GLX_KEEP_GRAPH_EXECS", "1") != "0"
_graph_execs = []


# ---------------------------------------------------------------------------------------------------------------
# Memset nodes.  On ROCm 7.2 a hipMemsetAsync RECORDED into a graph fills its buffer with the requested value on the first
# launch of the graph and with a stale 16-byte pattern (host pointers by the look of them) on every later launch
# (tools/graph_memset_repro.py, 40 lines, torch + ctypes only; profiles/r04_graph_memset_repro.txt).  Whatever zeroes scratch
# memory with a memset in front of a kernel is therefore wrong from the second replay on -- this is what turned the recorded
# CVAE training step's gradients into NaN (torch's multi-block reductions zero their semaphores that way,
# tools/graph_reduce_repro.py).  Own kernels never use memsets (glx_fill_multi is a kernel); library calls inside a recorded
# step may.  Every recorded pipeline therefore ends its capture with finish_graph() (below), which replaces the memset nodes
# by fill-kernel nodes and RAISES when it cannot (the pipelines store the count as `memsets_replaced`).  AUDIT_GRAPHS only
# gates the diagnostic `audit_graph` (node-type census of what was recorded).
AUDIT_GRAPHS = os.environ.get("GLX_AUDIT_GRAPHS", "1") != "0"
_HIP_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
                   8: "ext_semaphore_signal", 9: "ext_semaphore_wait", 10: "mem_alloc", 11: "mem_free", 12: "memcpy_from_symbol",
                   13: "memcpy_to_symbol"}


class _MemsetParams(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("elementSize", ctypes.c_uint), ("height", c_size_t), ("pitch", c_size_t),
                ("value", ctypes.c_uint), ("width", c_size_t)]


MAX_RETIRED_GRAPHS = int(os.environ.get("GLX_MAX_RETIRED_GRAPHS", "256"))
_retired_warned = [False]


ALLOW_UNFIXED_MEMSETS = os.environ.get("GLX_ALLOW_UNFIXED_MEMSETS", "0") == "1"


def new_graph():
    """torch.cuda.CUDAGraph() whose exec outlives its owner (see above).  The hipGraph_t is ALWAYS kept (keep_graph=True):
    finish_graph() needs it for the memset surgery, which is a correctness fix and not a diagnostic -- it does not hang on
    GLX_AUDIT_GRAPHS.  A torch without keep_graph cannot run recorded pipelines on this runtime: that raises here."""
    try:
        g = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError as e:                              # a torch without keep_graph
        if not ALLOW_UNFIXED_MEMSETS:
            raise GlxError("torch.cuda.CUDAGraph(keep_graph=True) is not available in this torch: recorded pipelines need the "
                       "raw hipGraph_t to replace memset nodes (ROCm 7.2 replays them with a stale pattern -> NaN "
                       "gradients from the second replay on).  Run the pipelines eagerly (enqueue()/step() without "
                       "capture()) or set GLX_ALLOW_UNFIXED_MEMSETS=1 to record anyway.") from e
        import warnings
        warnings.warn("GLX_ALLOW_UNFIXED_MEMSETS=1: recording without the raw hipGraph_t -- memset nodes stay as they are "
                      "(ROCm 7.2 replays them with a stale pattern)", RuntimeWarning, stacklevel=2)
        g = torch.cuda.CUDAGraph()
    if KEEP_GRAPH_EXECS:
        _graph_execs.append(g)
        if len(_graph_execs) > MAX_RETIRED_GRAPHS and not _retired_warned[0]:
            _retired_warned[0] = True
            import warnings
            warnings.warn("%d recorded graphs are being kept alive (ROCm 7.2: destroying a hipGraphExec can crash a later "
                          "launch, profiles/r03_graph_exec_destroy_crash.txt).  Each retired exec holds host memory and its "
                          "kernel-argument buffers; pipelines reuse the retired graph's memory pool, so device memory does "
                          "not grow with it (tests/test_graph_memset_gpu.py::test_recaptures_do_not_grow_reserved_memory). "
                          "Re-recording this often usually means weights change between replays of an inference "
                          "pipeline: record once per evaluation phase." % len(_graph_execs), RuntimeWarning, stacklevel=2)
    return g


def retired_graph_count():
    """Graphs created through new_graph() and kept alive for the life of the process."""
    return len(_graph_execs)




def finish_graph(graph):
    """Call right after a capture into a graph from new_graph(): replaces the memset nodes of what was recorded (child
    graphs included) by fill-kernel nodes (ROCm 7.2 replays memset nodes with a stale pattern, csrc/glx_graph.hip) and
    instantiates the graph.  Returns the number of memset nodes replaced.  Raises GlxError when the raw graph cannot be
    had (the recorded step would be wrong from its second replay on) unless GLX_ALLOW_UNFIXED_MEMSETS=1, in which case
    it warns, leaves the graph as torch instantiated it, and returns None."""
    try:
        raw = graph.raw_cuda_graph()
    except Exception as e:
        if not ALLOW_UNFIXED_MEMSETS:
            raise GlxError("finish_graph: the raw hipGraph_t of the recorded step is not available (%s: %s); memset nodes "
                           "would replay a stale pattern on ROCm 7.2.  Create graphs with _lib.new_graph(), or set "
                           "GLX_ALLOW_UNFIXED_MEMSETS=1 to accept the recorded graph as it is." % (type(e).__name__, e)) from e
        import warnings
        warnings.warn("finish_graph: raw graph not available, memset nodes (if any) left in place", RuntimeWarning, stacklevel=2)
        return None
    n = c_int(0)
    call_nostream("glx_graph_replace_memsets", c_void_p(raw), ctypes.byref(n))
    graph.instantiate()
    return n.value


def audit_graph(graph):
    """{node type: count} of a captured torch.cuda.CUDAGraph created by new_graph(), plus "memset_bytes": the sizes of its
    memset nodes; None when the raw graph is not available (AUDIT_GRAPHS off)."""
    try:
        raw = graph.raw_cuda_graph()
    except Exception:
        return None
    hip = ctypes.CDLL("libamdhip64.so")
    n = c_size_t(0)
    if hip.hipGraphGetNodes(c_void_p(raw), None, ctypes.byref(n)) != 0 or n.value == 0:
        return {} if n.value == 0 else None
    nodes = (c_void_p * n.value)()
    if hip.hipGraphGetNodes(c_void_p(raw), nodes, ctypes.byref(n)) != 0:
        return None
    out, sizes = {}, []
    for node in nodes:
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(c_void_p(node), ctypes.byref(t))
        name = _HIP_NODE_TYPES.get(t.value, "type_%d" % t.value)
        out[name] = out.get(name, 0) + 1
        if name == "memset":
            p = _MemsetParams()
            if hip.hipGraphMemsetNodeGetParams(c_void_p(node), ctypes.byref(p)) == 0:
                sizes.append(int(p.width) * max(int(p.height), 1) * max(int(p.elementSize), 1) if p.height > 1 else int(p.width) * max(int(p.elementSize), 1))
    if sizes:
        out["memset_bytes"] = sorted(sizes)
    return out


class Workspace:
    """Grow-only device scratch buffer per (device, stream): avoids per-call allocation and
    keeps stream-ordered reuse safe."""

    def __init__(self):
        self._buf = {}
        self._scope = None

    def scoped(self, tag):
        """Context manager: buffers handed out inside are private to `tag` (e.g. one frame
        pipeline whose recorded launches may run concurrently with another pipeline's)."""
        ws = self

        class _Scope:
            def __enter__(self):
                self.prev, ws._scope = ws._scope, tag

            def __exit__(self, *exc):
                ws._scope = self.prev

        return _Scope()

    def get(self, nbytes, device):
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, self._scope)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


workspace = Workspace()
