"""Sparse 3-D backbones of the hot path, built on glenet_amd.spconv.

Our counterpart of the callers in pcdet/models/backbones_3d/spconv_backbone.py: the same
layer lists, channels, strides, paddings and indice_key sharing (VoxelBackBone8x :77-117,
VoxelResBackBone8x :191-232; BatchNorm1d eps=1e-3 momentum=0.01 :73), written table-driven.
The reference modules themselves also run unmodified on glenet_amd.spconv through
glenet_amd.dropin; this module is what bench.py and the tests drive.
"""
import contextlib
import gc
import os
from collections import deque
from functools import partial

import torch
from torch import nn

from . import _lib, spconv
from . import voxelize as gv

_norm = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)


def _conv_bn_relu(cin, cout, ksize, key, kind="subm", stride=1, padding=0):
    if kind == "subm":
        conv = spconv.SubMConv3d(cin, cout, ksize, padding=padding, bias=False, indice_key=key)
    else:
        conv = spconv.SparseConv3d(cin, cout, ksize, stride=stride, padding=padding, bias=False,
                                   indice_key=key)
    return spconv.SparseSequential(conv, _norm(cout), nn.ReLU())


FUSED_RESIDUAL_TAIL = True      # ResidualBlock (training): relu(bn2(.) + identity) as one launch


class ResidualBlock(spconv.SparseModule):
    """SparseBasicBlock (spconv_backbone.py:30-64): two biased SubM convs + identity."""

    def __init__(self, channels, key):
        super().__init__()
        self.conv1 = spconv.SubMConv3d(channels, channels, 3, padding=1, bias=True, indice_key=key)
        self.bn1 = _norm(channels)
        self.conv2 = spconv.SubMConv3d(channels, channels, 3, padding=1, bias=True, indice_key=key)
        self.bn2 = _norm(channels)
        self.relu = nn.ReLU()

    def forward(self, x):
        if spconv.core.can_fuse_bn(self.bn1) and spconv.core.can_fuse_bn(self.bn2):
            out = self.conv1(x, fused_bn=self.bn1, fused_relu=True)
            out = self.conv2(out, fused_bn=self.bn2)
        else:
            core = spconv.core
            if core.conv_bn_fusable(self.conv1, self.bn1, x):
                # training, as SparseSequential runs conv + BatchNorm + ReLU: the statistics ride in conv1's epilogue and conv2
                # reads relu(bn1(.)) by transforming the raw rows on load (round 5; the 128 -> 128 blocks, whose convolutions
                # run as two column halves, keep the separate statistics launch below)
                out = self.conv1(x, train_bn=self.bn1, train_relu=True)
            else:
                out = self.conv1(x)
                if core.can_fuse_train_bn(self.bn1, out.features):     # training: fused BN(+ReLU) kernels
                    out = out.replace_feature(core.fused_train_bn(self.bn1, out.features, True, out.count))
                else:
                    out = out.replace_feature(self.relu(self.bn1(out.features)))
            if core.conv_bn_fusable(self.conv2, self.bn2, out):
                if FUSED_RESIDUAL_TAIL and self.bn2.affine:
                    # bn2's transform, the identity branch and the ReLU in one launch (three; two maps of traffic less)
                    return self.conv2(out, train_bn=self.bn2, train_relu=False, residual=x.features)
                out = self.conv2(out, train_bn=self.bn2, train_relu=False)
            else:
                out = self.conv2(out)
                if core.can_fuse_train_bn(self.bn2, out.features):
                    out = out.replace_feature(core.fused_train_bn(self.bn2, out.features, False, out.count))
                else:
                    out = out.replace_feature(self.bn2(out.features))
        return out.replace_feature(self.relu(out.features + x.features))


# stage tables: (cin, cout, padding of the stride-2 conv)
_PLAIN = dict(stem=16, stages=[(16, 32, 1), (32, 64, 1), (64, 64, (0, 1, 1))], out_in=64)
_RES = dict(stem=16, stages=[(16, 32, 1), (32, 64, 1), (64, 128, (0, 1, 1))], out_in=128)


class SparseBackbone8x(nn.Module):
    """residual=False -> VoxelBackBone8x (12 sparse convs, 8 rule tables);
    residual=True  -> VoxelResBackBone8x (17 SubM + 4 strided + conv_input, 9 rule tables)."""

    def __init__(self, input_channels, grid_size, residual=False, last_pad=0):
        super().__init__()
        cfg = _RES if residual else _PLAIN
        gx, gy, gz = [int(g) for g in grid_size]
        self.sparse_shape = [gz + 1, gy, gx]                       # spconv_backbone.py:75
        self.residual = residual
        self.conv_input = _conv_bn_relu(input_channels, cfg["stem"], 3, "subm1", padding=1)
        self.conv_input.leave_pending = True       # its only reader is conv1's first convolution (spconv.core.BN_ON_LOAD)

        def body(c, key):
            if residual:
                return [ResidualBlock(c, key), ResidualBlock(c, key)]
            return [_conv_bn_relu(c, c, 3, key, padding=1)]

        if residual:
            self.conv1 = spconv.SparseSequential(*body(16, "res1"))
        else:
            self.conv1 = spconv.SparseSequential(*body(16, "subm1"))
        stages = []
        for i, (cin, cout, pad) in enumerate(cfg["stages"], start=2):
            down = _conv_bn_relu(cin, cout, 3, "spconv%d" % i, kind="spconv", stride=2, padding=pad)
            if residual:
                rest = body(cout, "res%d" % i)
            else:
                rest = [_conv_bn_relu(cout, cout, 3, "subm%d" % i, padding=1),
                        _conv_bn_relu(cout, cout, 3, "subm%d" % i, padding=1)]
            stages.append(spconv.SparseSequential(down, *rest))
        self.conv2, self.conv3, self.conv4 = stages
        self.conv_out = _conv_bn_relu(cfg["out_in"], 128, (3, 1, 1), "spconv_down2", kind="spconv",
                                      stride=(2, 1, 1), padding=last_pad)
        self.num_point_features = 128
        self.stage_cuts = False      # set by a training step whose loss runs the staged backward
        self.backbone_channels = {"x_conv1": 16, "x_conv2": 32, "x_conv3": 64,
                                  "x_conv4": cfg["stages"][-1][1]}

    def plan(self, voxel_coords, batch_size, index=None, capacities=None, events=False, pair_lists=False):
        """Build every rule table of the backbone from the coordinates alone (they do not depend
        on features), so the convolutions afterwards run back to back without host syncs.
        Returns the indice_dict to pass as batch_dict["rule_plan"].  With a shape-static index
        (voxelize_batch(..., static=True)) the plan itself is free of host syncs too."""
        count = index.count if index is not None else None
        return spconv.core.plan_rules(voxel_coords.int(), self.sparse_shape, batch_size,
                                      list(self.sparse_convs()) + list(self.extra_plan), index=index, count=count,
                                      capacities=capacities, events=events, pair_lists=pair_lists)

    def forward(self, batch_dict):
        index = batch_dict.get("voxel_index")
        x = spconv.SparseConvTensor(batch_dict["voxel_features"], batch_dict["voxel_coords"].int(),
                                    self.sparse_shape, batch_dict["batch_size"],
                                    indice_dict=batch_dict.get("rule_plan"),
                                    count=index.count if index is not None else None)
        x._index = index
        x = self.conv_input(x)
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        cuts = {}
        if self.stage_cuts and torch.is_grad_enabled() and c2.features.is_cuda and c2.features.requires_grad:
            # staged backward (glenet_vr.StagedLoss): conv3 / conv4 read detached leaf copies of x_conv2 / x_conv3, so the
            # backward of the levels above a cut can run before the RoI branch has delivered the gradients of the
            # levels below it; batch_dict["stage_cuts"][name] = (the level's feature tensor, the leaf the next block read)
            def cut(name, st):
                leaf = st.replace_feature(st.features.detach().requires_grad_(True))
                cuts[name] = (st.features, leaf.features)
                return leaf
            c3 = self.conv3(cut("x_conv2", c2))
            c4 = self.conv4(cut("x_conv3", c3))
        else:
            c3 = self.conv3(c2)
            c4 = self.conv4(c3)
        out = self.conv_out(c4)
        batch_dict["stage_cuts"] = cuts
        batch_dict.update(encoded_spconv_tensor=out, encoded_spconv_tensor_stride=8,
                          multi_scale_3d_features=dict(x_conv1=c1, x_conv2=c2, x_conv3=c3, x_conv4=c4),
                          multi_scale_3d_strides=dict(x_conv1=1, x_conv2=2, x_conv3=4, x_conv4=8))
        return batch_dict

    def sparse_convs(self):
        return [m for m in self.modules() if isinstance(m, spconv.SparseConvolution)]

    # geometry of sparse convs BEHIND the stack whose rule tables plan() should build with the others
    # (spconv.core.PlannedConv; the BEV backbone's first layer, dense_path.BEVBackbone._first_layer_sparse)
    extra_plan = ()


def VoxelBackBone8x(input_channels, grid_size, **kw):
    return SparseBackbone8x(input_channels, grid_size, residual=False, **kw)


def VoxelResBackBone8x(input_channels, grid_size, **kw):
    return SparseBackbone8x(input_channels, grid_size, residual=True, **kw)


class MeanVFE(nn.Module):
    """mean_vfe.py:14-31 on device."""

    def forward(self, batch_dict):
        index = batch_dict.get("voxel_index")
        batch_dict["voxel_features"] = gv.mean_vfe(batch_dict["voxels"], batch_dict["voxel_num_points"],
                                                   count=index.count if index is not None else None)
        return batch_dict


def _reset_conv_packs():
    from . import conv2d as own_conv
    own_conv.STEP_PACKS = None


class HeightCompression(nn.Module):
    """height_compression.py:10-26: dense() then fold depth into channels."""

    channels_last = False       # True: the BEV map is produced in channels-last memory (SparseConvTensor.dense_bev)
    defer = False               # True: no map here -- the consumer (dense_path.BEVBackbone) runs its first layer on the
                                # sparse tensor, or builds the map itself when it cannot

    def forward(self, batch_dict):
        if self.defer and self.channels_last:
            batch_dict["spatial_features"] = None
            batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
            return batch_dict
        if self.channels_last:
            batch_dict["spatial_features"] = batch_dict["encoded_spconv_tensor"].dense_bev()
            batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
            return batch_dict
        dense = batch_dict["encoded_spconv_tensor"].dense()
        n, c, d, h, w = dense.shape
        batch_dict["spatial_features"] = dense.view(n, c * d, h, w)
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict


def voxelize_batch(points_list_or_stacked, batch_idx, batch_size, cfg, train=True, static=False):
    """Device-side replacement of DataProcessor.transform_points_to_voxels + collate_batch
    (data_processor.py:117-152, dataset.py:192-197): stacked points -> batch_dict entries.  The
    voxelizer's cell bitmap doubles as the sparse tensor's cell index (grid depth gz + 1)."""
    max_voxels = cfg["max_voxels_train"] if train else cfg["max_voxels_test"]
    gz = gv.grid_size_of(cfg["point_cloud_range"], cfg["voxel_size"])[2]
    v, c, n, offs, index = gv.hard_voxelize(points_list_or_stacked, cfg["voxel_size"],
                                            cfg["point_cloud_range"], cfg["max_points"], max_voxels,
                                            batch_idx=batch_idx, batch_size=batch_size,
                                            index_depth=gz + 1, static=static)
    return dict(voxels=v, voxel_coords=c, voxel_num_points=n, batch_size=batch_size,
                voxel_offset=offs, voxel_index=index)


class StaticFramePipeline:
    """One batch of the hot path (hard voxelize -> MeanVFE -> rule tables -> sparse backbone ->
    dense()) as a shape-static, host-sync-free sequence, optionally frozen into a HIP graph.

    Buffers are sized by capacity (B * max_voxels rows; 288 GB of HBM make that free), the live
    row counts stay on the device, so the ~120 launches of a frame are enqueued without a single
    read-back; `capture()` records them once (torch.cuda.CUDAGraph = hipGraph) and `replay()`
    re-issues the whole frame with one call.  Inference only.  `check()` is the after-the-fact
    validation of the capacities (one host sync; call it whenever the verdict is needed)."""

    def __init__(self, model, cfg, batch_size, num_points, num_features, train_voxel_cap=True,
                 capacities=None, device=None):
        self.model, self.cfg, self.B = model, cfg, int(batch_size)
        self.train_cap, self.capacities = train_voxel_cap, capacities
        dev = device if device is not None else next(model.parameters()).device
        self.points = torch.zeros((int(num_points), int(num_features)), dtype=torch.float32, device=dev)
        # frame id B = "no frame": padding points are dropped by the voxelizer
        self.batch_idx = torch.full((int(num_points),), self.B, dtype=torch.int32, device=dev)
        self.vfe, self.hc = MeanVFE(), HeightCompression()
        self.graph = None
        self._tag = None
        self.out = None
        self.max_in_flight = 4
        self._inflight = deque()
        # rule tables depend on coordinates only: they are built on a second stream and overlap
        # the convolutions of the levels above them (parallel branches of the HIP graph)
        self.plan_stream = torch.cuda.Stream(dev)
        self.overlap_plan = os.environ.get("GLX_OVERLAP_PLAN", "1") != "0"

    def calibrate(self, points, batch_idx, headroom=1.3):
        """Size the strided convs' output sets from a representative batch: one pass of the exact
        (host-synchronising) path, capacities = headroom x the observed row counts."""
        with torch.no_grad():
            bd = voxelize_batch(points, batch_idx, self.B, self.cfg, train=self.train_cap)
            plan = self.model.plan(bd["voxel_coords"], self.B, index=bd["voxel_index"])
        # multiples of 128 rows: row-wise products on these tensors can then be split along K (roi_grid)
        self.capacities = {key: (int(rs.N_out * headroom) + 64 + 127) // 128 * 128 for key, rs in plan.items()
                           if not rs.subm}
        return self.capacities

    def load(self, points, batch_idx):
        """Copy a stacked batch (P <= num_points rows) into the static input buffers (async)."""
        n = points.shape[0]
        if n > self.points.shape[0]:
            raise ValueError("batch has %d points, pipeline was sized for %d" % (n, self.points.shape[0]))
        if self._load_fused(points, batch_idx, ()):
            return
        self.points[:n].copy_(points, non_blocking=True)
        self.batch_idx[:n].copy_(batch_idx, non_blocking=True)
        if n < self.points.shape[0]:
            self.batch_idx[n:].fill_(self.B)

    def _load_fused(self, points, batch_idx, extra):
        """The copies and padding fills of load() as ONE launch (glx_copy_fill_multi) when every source is a
        contiguous device tensor of the buffers' dtypes; `extra`: (dst, src or None) pairs of (B, cap, C) buffers whose
        first src.shape[1] rows per frame come from src, the rest zeros.  False: the caller falls back to tensor ops."""
        import ctypes
        from . import _lib
        dev = self.points.device
        srcs = [points, batch_idx] + [s for _, s in extra if s is not None]
        if not all(t.is_cuda and t.device == dev and t.is_contiguous() for t in srcs):
            return False
        if points.dtype != self.points.dtype or batch_idx.dtype != self.batch_idx.dtype or points.shape[1:] != self.points.shape[1:]:
            return False
        n = points.shape[0]
        regions = [(self.points.data_ptr(), points.data_ptr(), points.numel(), points.numel(), 0),
                   (self.batch_idx.data_ptr(), batch_idx.data_ptr(), n, self.batch_idx.numel(), int(self.B))]
        for dst, src in extra:
            b, cap, c = dst.shape
            if src is None:
                regions.append((dst.data_ptr(), 0, 0, dst.numel(), 0))
                continue
            if src.dtype != dst.dtype or src.shape[0] != b or src.shape[2] != c or dst.element_size() != 4:
                return False
            g = src.shape[1]
            for f in range(b):
                regions.append((dst.data_ptr() + f * cap * c * 4, src.data_ptr() + f * g * c * 4, g * c, cap * c, 0))
        m = len(regions)
        if self.points.element_size() != 4 or self.batch_idx.element_size() != 4 or m > 48:
            return False
        arr = [(ctypes.c_void_p * m)(*[r[0] for r in regions]), (ctypes.c_void_p * m)(*[r[1] for r in regions]),
               (ctypes.c_uint32 * m)(*[r[2] for r in regions]), (ctypes.c_uint32 * m)(*[r[3] for r in regions]),
               (ctypes.c_uint32 * m)(*[r[4] for r in regions])]
        _lib.call("glx_copy_fill_multi", m, *arr)
        return True

    def enqueue(self):
        """Launch one frame on the current stream; returns the batch_dict (static buffers)."""
        from ._lib import workspace
        with torch.no_grad(), workspace.scoped(id(self)):
            bd = voxelize_batch(self.points, self.batch_idx, self.B, self.cfg, train=self.train_cap,
                                static=True)
            if self.overlap_plan:
                cur = torch.cuda.current_stream(self.points.device)
                self.plan_stream.wait_stream(cur)            # fork after the voxelizer
                with torch.cuda.stream(self.plan_stream):
                    plan = self.model.plan(bd["voxel_coords"], self.B, index=bd["voxel_index"],
                                           capacities=self.capacities, events=True)
                bd = self.vfe(bd)
                bd["rule_plan"] = plan
                bd = self.model(bd)                          # each conv waits for its rule set
                bd = self.hc(bd)
                cur.wait_stream(self.plan_stream)            # join
            else:
                bd = self.vfe(bd)
                bd["rule_plan"] = self.model.plan(bd["voxel_coords"], self.B,
                                                  index=bd["voxel_index"], capacities=self.capacities)
                bd = self.model(bd)
                bd = self.hc(bd)
        self.out = bd
        return bd

    def capture(self, warmup=2):
        """Warm up (one-time packing / attribute calls must not land in the graph), then record."""
        # Drop dead autograd graphs that only a reference cycle keeps alive: as long as one of them lives, the parameters'
        # AccumulateGrad nodes keep the stream they were created under (the default stream of an earlier eager pass), the
        # recorded backward then makes THAT stream wait on a capturing event -- and capture_end() fails with "capturing
        # stream has unjoined work", or not, depending on when the collector last ran.
        gc.collect()
        # ONE capture stream per pipeline: the caching allocator keeps freed blocks per stream, so a fresh side stream per
        # (re-)capture would strand the warm-up pass's buffers of every earlier capture (55 MB per re-capture measured on
        # a small inference pipeline: tests/test_graph_memset_gpu.py::test_recaptures_do_not_grow_reserved_memory)
        side = self.__dict__.get("_capture_stream")
        if side is None:
            side = self._capture_stream = torch.cuda.Stream(self.points.device)
        side.wait_stream(torch.cuda.current_stream(self.points.device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.enqueue()
        torch.cuda.current_stream(self.points.device).wait_stream(side)
        torch.cuda.synchronize(self.points.device)
        self.check()
        gc.collect()
        # A pipeline that records itself AGAIN (its weights changed) retires the old exec (never destroyed, _lib.new_graph)
        # but records into the old graph's private memory pool: the retired graph is never launched again and its
        # buffers die with the old `out`, so repeated re-captures do not grow the pool.
        retired = self.graph
        pool = retired.pool() if (retired is not None and REUSE_GRAPH_POOL) else None
        self.graph = _lib.new_graph()
        with torch.cuda.graph(self.graph, stream=side, pool=pool), no_gc():
            self.enqueue()
        self.memsets_replaced = _lib.finish_graph(self.graph)       # ROCm 7.2: memset nodes replay a stale pattern
        self._tag = self._weights_tag()
        return self

    def _tagged_modules(self):
        return (self.model,)

    def _weights_tag(self):
        """Inference graphs read packed weights / folded BatchNorms that are cached by tensor version outside the
        graph: a load_state_dict() or an in-place edit after capture() would leave the replays on the old copies.
        The sum of the version counters (they only grow) tells; replay() then records the frame again."""
        tag = stamps = 0
        for m in self._tagged_modules():
            for t in m.parameters():
                tag += t._version
                stamps += t.__dict__.get("_glx_epoch", 0)
            for t in m.buffers():
                tag += t._version
                stamps += t.__dict__.get("_glx_epoch", 0)
        # + the weights epoch of THESE tensors: fused optimizer / BatchNorm updates and replayed training graphs move no
        # version counter.  Scoped: training an unrelated model leaves this pipeline's tag alone.
        return (tag, stamps, _lib.weights_epoch())

    def replay(self):
        """Launch the recorded frame.  At most `max_in_flight` frames are queued: the host waits
        for frame i - max_in_flight before launching frame i (the usual multi-buffering bound;
        deeper hipGraph queues also proved unreliable on ROCm 7.2)."""
        if len(self._inflight) >= self.max_in_flight:
            self._inflight.popleft().synchronize()
        if self._tag is not None and self._weights_tag() != self._tag:
            torch.cuda.current_stream(self.points.device).synchronize()
            self.capture()                     # weights changed since the capture: record again
        self.graph.replay()
        ev = torch.cuda.Event()
        ev.record()
        self._inflight.append(ev)
        return self.out

    def run_checked(self, points, batch_idx):
        """load + replay + verdict for one batch; when a capacity was exceeded (or max_voxels
        dropped cells) the batch is recomputed on the exact-shape path, so the result is always
        right.  Costs one host synchronisation per batch -- the throughput loop uses replay()
        and check() every so often instead."""
        self.load(points, batch_idx)
        out = self.replay() if self.graph is not None else self.enqueue()
        torch.cuda.current_stream(self.points.device).synchronize()
        try:
            self.check()
            return out
        except RuntimeError:
            with torch.no_grad():
                bd = voxelize_batch(points, batch_idx, self.B, self.cfg, train=self.train_cap)
                return self.hc(self.model(self.vfe(bd)))

    def check(self):
        spconv.core.check_static(self.out["rule_plan"], self.out["voxel_index"])

    def live(self, st):
        """Trim a shape-static SparseConvTensor to its live rows (host sync) for inspection."""
        n = int(st.count.item())
        return st.features[:n], st.indices[:n]


REUSE_GRAPH_POOL = os.environ.get("GLX_REUSE_GRAPH_POOL", "1") != "0"


@contextlib.contextmanager
def no_gc():
    """No cyclic garbage collection while a stream is being captured.  A collection that happens to run in the middle
    of a capture may finalise an OLDER pipeline (pipelines are reference cycles: they only ever die in the collector),
    and destroying its hipGraphExec / releasing its memory pool is not allowed while another capture is open -- the
    process aborts (seen once in ~10 runs of the GPU tests, always inside capture()).  torch.cuda.graph collects
    on entry; with the collector off nothing but reference counts frees objects until the capture has ended."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


DEFER_WGRAD_REDUCES = os.environ.get("GLX_DEFER_WGRAD_REDUCES", "1") != "0"    # see StaticTrainStep.step
PREPACK_WEIGHTS = True


class StaticTrainPipeline(StaticFramePipeline):
    """Forward + backward (+ optimizer step) of the sparse backbone as one shape-static launch
    sequence / HIP graph.  The autograd Functions take the device row counts of the shape-static
    rule sets (dgrad = the forward kernels with n_live, k_wgrad_mfma and the fused BatchNorm kernels
    read *n_live, dense()'s adjoint is glx_dense_gather), so the ~600 launches of a training step
    are enqueued without a read-back and replayed with one call.  Rows past a live count carry
    undefined values in activations and gradients alike; no kernel reads them.

    loss_fn(batch_dict) -> scalar tensor; default mean(spatial_features^2) (a stand-in for the
    dense head's loss).  Parameter gradients are left in `.grad` (rewritten by every replay)."""

    def __init__(self, model, cfg, batch_size, num_points, num_features, loss_fn=None, optimizer=None,
                 train_voxel_cap=True, capacities=None, device=None, extra_modules=()):
        """extra_modules: further nn.Modules whose parameters loss_fn trains (e.g. the BEV backbone
        and dense head behind the sparse backbone); their gradients are reset with the backbone's."""
        super().__init__(model, cfg, batch_size, num_points, num_features,
                         train_voxel_cap=train_voxel_cap, capacities=capacities, device=device)
        self.extra_modules = tuple(extra_modules)
        self.loss_fn = loss_fn if loss_fn is not None else (lambda bd: bd["spatial_features"].square().mean())
        self.optimizer = optimizer
        self.loss = None
        self.overlap_wgrad = os.environ.get("GLX_OVERLAP_WGRAD", "1") != "0"
        self.mark = None            # optional callable(stage_name): bench.py records an event per stage
        self.data_step = None       # optional data_pipeline.DeviceDataProcessor: mask + shuffle inside the step

    def _weights_tag(self):
        return None      # training steps pack weights inside the step (spconv.core._packed_weight): nothing cached

    def replay(self):
        """A replayed training step updates parameters and running statistics without running any Python: tell the
        version-keyed caches of everything that shares these weights (eval-mode modules, inference graphs)."""
        out = super().replay()
        _lib.bump_weights_epoch(self._written_tensors())
        return out

    def _written_tensors(self):
        """Everything a replayed step writes behind torch's back: parameters (a recorded optimizer) and BatchNorm
        running statistics of the modules the step runs."""
        hit = self.__dict__.get("_glx_written")
        if hit is None:
            mods = (self.model,) + tuple(self.extra_modules)
            hit = [t for m in mods for t in list(m.parameters()) + list(m.buffers())]
            self.__dict__["_glx_written"] = hit
        return hit

    def enqueue(self):
        from ._lib import workspace
        dev = self.points.device
        # release the previous step's autograd graph first: its AccumulateGrad nodes remember the
        # stream they were created on, and a node of an earlier eager step (default stream) that is
        # still alive makes backward synchronise with that stream -- inside a capture this tears
        # the capture down (a segfault in hipStreamEndCapture on ROCm 7.2).  Callers that keep
        # their own reference to an earlier loss / batch_dict must drop it before capture().
        self.out = self.loss = None
        if self.mark:
            self.mark("start")
        with workspace.scoped(id(self)):
            with torch.no_grad():
                pts, bidx = self.points, self.batch_idx
                if self.data_step is not None:      # range mask + per-frame shuffle on the device
                    pts, bidx = self.data_step.static_step(pts, bidx, self.B)
                    if self.mark:
                        self.mark("device data step (range mask + shuffle)")
                bd = voxelize_batch(pts, bidx, self.B, self.cfg, train=self.train_cap,
                                    static=True)
                cur = torch.cuda.current_stream(dev)
                if self.overlap_plan:
                    self.plan_stream.wait_stream(cur)
                    with torch.cuda.stream(self.plan_stream):
                        plan = self.model.plan(bd["voxel_coords"], self.B, index=bd["voxel_index"],
                                               capacities=self.capacities, events=True, pair_lists=True)
                else:
                    plan = self.model.plan(bd["voxel_coords"], self.B, index=bd["voxel_index"],
                                           capacities=self.capacities, pair_lists=True)
                bd = self.vfe(bd)
            if self.mark:
                self.mark("voxelize + MeanVFE")
            bd["rule_plan"] = plan
            self.model.zero_grad(set_to_none=True)      # .grad tensors are (re)created by backward
            for m in self.extra_modules:
                m.zero_grad(set_to_none=True)
            spconv.core.DEFERRED_COUNTERS = counters = []
            if PREPACK_WEIGHTS:        # every layer's forward + adjoint weight image in one launch
                spconv.core.prepack([m for m in self.model.modules() if isinstance(m, spconv.core.SparseConvolution)])
                from . import conv2d as own_conv       # ... and the BEV backbone's 3x3 filters (split-bf16 pieces) in another
                own_conv.prepack([m.weight for em in self.extra_modules for m in em.modules()
                                  if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride in ((1, 1), (2, 2))
                                  and m.bias is None])
            try:
                with torch.enable_grad():
                    bd = self.hc(self.model(bd))
                    if self.mark:
                        self.mark("sparse backbone fwd + dense()")
                    loss = self.loss_fn(bd)
            except BaseException:
                spconv.core.STEP_PACKS = None
                _reset_conv_packs()
                raise
            finally:
                spconv.core.DEFERRED_COUNTERS = None
            if counters:
                torch._foreach_add_(counters, 1)
            if self.overlap_plan:
                cur.wait_stream(self.plan_stream)
            if self.overlap_wgrad:     # weight gradients next to the input-gradient chain
                spconv.core.WGRAD_STREAM = self.plan_stream
            # a plain loss: the layers' weight-gradient sums in one launch per kind behind the backward pass (a staged loss --
            # glenet_vr.StagedLoss -- does the same inside its own backward)
            from . import conv2d as c2
            defer = DEFER_WGRAD_REDUCES and torch.is_tensor(loss)
            sparse_sums = spconv.core.DEFERRED_WGRAD_REDUCES = [] if defer else None
            bev_sums = c2.DEFERRED_WGRAD_REDUCES = [] if defer else None
            try:
                loss.backward()
            finally:
                spconv.core.WGRAD_STREAM = None
                spconv.core.STEP_PACKS = None
                spconv.core.DEFERRED_WGRAD_REDUCES = c2.DEFERRED_WGRAD_REDUCES = None
                _reset_conv_packs()
            if self.overlap_wgrad:
                cur.wait_stream(self.plan_stream)
            if defer:
                c2.run_deferred_wgrad_reduces(bev_sums)
                spconv.core.run_deferred_wgrad_reduces(sparse_sums)
            if not torch.is_tensor(loss):      # a staged backward (glenet_vr.StagedLoss): its scalar exists now
                loss = loss.detach()
            if self.mark:
                self.mark("backward")
            if self.optimizer is not None:
                self.optimizer.step()
        self.out, self.loss = bd, loss
        return bd
