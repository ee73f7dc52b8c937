"""Host-side mirror of the spconv surface the reference uses.

Mirrors (names, argument meaning, attributes) of the third-party `spconv` package as the
reference calls it:
  * SparseConvTensor(features, indices, spatial_shape, batch_size) with .features .indices
    .spatial_shape .batch_size .dense() .replace_feature()   (spconv_backbone.py:141-146,
    pcdet/utils/spconv_utils.py:28-34, height_compression.py:21)
  * SubMConv3d / SparseConv3d / SparseInverseConv3d (spconv_backbone.py:12-17)
  * SparseSequential / SparseModule (spconv_backbone.py:21,30)
  * conv.SparseConvolution base class with a 5-D .weight (spconv_utils.py:19-21); the layout
    is the spconv-1.x one, (kd, kh, kw, Cin, Cout)  (detector3d_template.py:377-384).

All arithmetic runs in libglenet_hip.so through the C ABI; this file only owns tensors,
caches rule tables per indice_key and wires autograd.

Shape-static mode (an extension, inference only): a SparseConvTensor may carry `count`, a device
int32[1] holding the number of live rows, while `features` / `indices` are allocated at a fixed
capacity.  Every op then takes its row count on the device (the C ABI's n_live arguments), so a
whole frame runs without host synchronisation and can be captured in a HIP graph;
`check_static()` verifies afterwards that no capacity was exceeded.
"""
import contextlib
import ctypes
import weakref
import math
import os
from collections import OrderedDict

import torch
from torch import nn
from torch.autograd import Function
from torch.nn import init

from .. import _lib
from .._lib import call, call_nostream, query, size_arg, workspace


def _triple(v):
    if isinstance(v, (list, tuple)):
        assert len(v) == 3, v
        return tuple(int(x) for x in v)
    return (int(v),) * 3


class CellIndex:
    """Rank dictionary of one active set on a (B, D, H, W) grid (device resident)."""

    __slots__ = ("grid", "bitmap", "flags", "prefix", "rank_to_row", "row_to_rank", "n", "count",
                 "unique")

    def __init__(self, grid, bitmap, flags, prefix, rank_to_row, row_to_rank, n, count=None,
                 unique=None):
        self.grid, self.bitmap, self.flags, self.prefix = grid, bitmap, flags, prefix
        self.rank_to_row, self.row_to_rank, self.n = rank_to_row, row_to_rank, n
        # shape-static mode: n is the capacity, count (device int32[1]) the live rows, unique
        # (device int32[1]) the number of set bits (must equal count: checked by check_static)
        self.count, self.unique = count, unique

    @staticmethod
    def alloc(grid, device):
        words = query("glx_index_words", *grid)
        bitmap = torch.empty(words, dtype=torch.int64, device=device)
        flags = torch.empty((words + 7) // 8, dtype=torch.uint8, device=device)
        prefix = torch.empty(words, dtype=torch.int32, device=device)
        return bitmap, flags, prefix

    @staticmethod
    def build(indices, grid):
        """indices (N,4) int32 [b,z,y,x] in arbitrary row order -> CellIndex."""
        _lib.check_cuda(indices)
        dev = indices.device
        N = indices.shape[0]
        bitmap, flags, prefix = CellIndex.alloc(grid, dev)
        r2row = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        row2r = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        meta = torch.empty(2, dtype=torch.int32, device=dev)  # n_unique, status (both written by the call)
        wsb = query("glx_index_workspace_bytes", *grid)
        ws = workspace.get(wsb, dev)
        call("glx_index_build", indices, N, *grid, bitmap, flags, prefix, r2row, row2r, meta[0:1],
             meta[1:2], ws, size_arg(ws.numel()))
        n_unique, status = meta.tolist()  # one host sync per distinct active set
        if status != 0:
            raise ValueError("SparseConvTensor indices outside spatial_shape/batch_size %s" % (grid,))
        if n_unique != N:
            raise ValueError("SparseConvTensor indices contain duplicates (%d unique of %d)"
                             % (n_unique, N))
        return CellIndex(grid, bitmap, flags, prefix, r2row, row2r, N)


class RuleSet:
    """Rule table of one conv geometry on one input set (what spconv keeps per indice_key)."""

    def __init__(self):
        self.nbr = None          # (N_out, K) int32: input row per output row and offset
        self.nbr_in = None       # (N_in, K) int32 inverse (lazy; strided only)
        self.K = 0
        self.N_in = self.N_out = 0
        self.subm = True
        self.tile_order_out = None   # spatial order of output rows (None = identity)
        self.tile_order_in = None
        self.out_indices = None
        self.out_spatial_shape = None
        self.out_index = None    # CellIndex of the output set
        self.in_index = None
        self.in_indices = None
        self.in_spatial_shape = None
        self.geom = None
        self._pairs_dev = None
        self._pairs = None
        self.count_in = self.count_out = None   # shape-static mode: live rows (device int32[1])
        self.ready = None        # event recorded after the build when it ran on another stream
        self._tile_maps = {}     # rule table (data_ptr) -> work-balanced block -> tile map
        self._pair_lists = {}    # rule table (data_ptr) -> per-offset pair lists (glx_pair_lists_build)

    @property
    def pair_count(self):
        """R = number of (input, output, offset) rules (host sync on first read)."""
        if self._pairs is None:
            n = self.N_out if self.count_out is None else min(self.N_out, int(self.count_out.item()))
            self._pairs = int((self.nbr[:n] >= 0).sum().item()) if n else 0
        return self._pairs

    def tile_map(self, nbr=None, order=None, n_out=None, n_live=None):
        """Work-balanced block -> tile permutation of a rule table of this set (default: the
        forward table) for the 64-row tile kernels (glx_sconv_tile_map); built once, shared by
        every conv that walks the table."""
        if nbr is None:
            nbr, order, n_out, n_live = self.nbr, self.tile_order_out, self.N_out, self.count_out
        if not USE_TILE_MAP or n_out < TILE_MAP_MIN_ROWS or n_out > 64 * 16384:     # beyond: the kernels' built-in map
            return None
        key = nbr.data_ptr()
        m = self._tile_maps.get(key)
        if m is None:
            m = torch.empty((n_out + 63) // 64, dtype=torch.int32, device=nbr.device)
            ws = workspace.get(query("glx_sconv_tile_map_workspace_bytes", n_out), nbr.device)
            call("glx_sconv_tile_map", nbr, order, n_out, self.K, n_live, m, ws, size_arg(ws.numel()))
            self._tile_maps[key] = m
        return m

    def pair_lists(self, nbr, n_out, n_live):
        """Per-offset (input row, output row) pair lists of a rule table of this set -- what the weight gradient contracts
        over (spconv's indice pairs); built on first use (two launches on the caller's stream), shared by every
        convolution that walks the table."""
        key = nbr.data_ptr()
        cur = torch.cuda.current_stream(nbr.device)
        hit = self._pair_lists.get(key)
        if hit is None:
            nbytes = query("glx_pair_lists_bytes", n_out, self.K)
            pl = torch.empty(nbytes, dtype=torch.uint8, device=nbr.device)
            call("glx_pair_lists_build", nbr, n_out, self.K, n_live, pl, size_arg(nbytes))
            ev = torch.cuda.Event()
            ev.record(cur)
            self._pair_lists[key] = (pl, cur, ev)
            return pl
        pl, built_on, ev = hit
        if built_on != cur:          # built by a weight gradient on another stream of this step
            cur.wait_event(ev)
            pl.record_stream(cur)
        return pl

    def inverse_table(self):
        """The table of the transposed rules (input row -> output rows): what the input gradient of a strided convolution
        walks.  Built on first use on the caller's stream -- a training plan builds it ahead, on the plan stream
        (plan_rules(pair_lists=True)): two launches per table that sat in the backward chain of the main stream."""
        cur = torch.cuda.current_stream(self.nbr.device)
        if self.nbr_in is None:
            dev = self.nbr.device
            self.nbr_in = torch.empty((max(self.N_in, 1), self.K), dtype=torch.int32, device=dev)
            call("glx_rules_invert", self.nbr, self.N_out, self.K, self.N_in, self.nbr_in,
                 self.count_out)
            self._nbr_in_built = (cur, torch.cuda.Event())
            self._nbr_in_built[1].record(cur)
            return self.nbr_in
        built = getattr(self, "_nbr_in_built", None)
        if built is not None and built[0] != cur:
            cur.wait_event(built[1])
            self.nbr_in.record_stream(cur)
        return self.nbr_in


def build_subm_rules(x, ksize, dilation=(1, 1, 1)):
    idx = x._ensure_index()
    N = x.indices.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    rs = RuleSet()
    rs.subm, rs.K, rs.N_in, rs.N_out = True, K, N, N
    rs.nbr = torch.empty((max(N, 1), K), dtype=torch.int32, device=x.indices.device)
    call("glx_rules_subm_dilated", x.indices, N, *idx.grid, idx.bitmap, idx.prefix, idx.rank_to_row,
         *ksize, *dilation, rs.nbr, None, x.count)
    rs.count_in = rs.count_out = x.count
    rs.tile_order_out = rs.tile_order_in = idx.rank_to_row
    rs.out_indices, rs.out_spatial_shape, rs.out_index = x.indices, list(x.spatial_shape), idx
    rs.in_index, rs.in_indices, rs.in_spatial_shape = idx, x.indices, list(x.spatial_shape)
    rs.geom = ("subm", ksize) if tuple(dilation) == (1, 1, 1) else ("subm", ksize, tuple(dilation))
    return rs


def build_strided_rules(x, ksize, stride, padding, dilation=(1, 1, 1), out_capacity=None):
    """out_capacity: shape-static mode only -- rows allocated for the output set.  Default: the
    safe bound N_in * prod(ceil(k/s)) (clipped to the grid); calibrated capacities
    (StaticFramePipeline.calibrate) keep the launches tight, check_static() tells if one was
    exceeded."""
    dilation = tuple(int(d) for d in dilation)
    idx = x._ensure_index()
    dev = x.indices.device
    B = x.batch_size
    D, H, W = x.spatial_shape
    out_shape = [(s + 2 * p - d * (k - 1) - 1) // st + 1
                 for s, p, k, st, d in zip((D, H, W), padding, ksize, stride, dilation)]
    if min(out_shape) <= 0:
        raise ValueError("SparseConv3d output shape %s is empty" % (out_shape,))
    ogrid = (B, *out_shape)
    obitmap, oflags, oprefix = CellIndex.alloc(ogrid, dev)
    n_out_dev = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = query("glx_index_workspace_bytes", *ogrid)
    ws = workspace.get(wsb, dev)
    N_in = x.indices.shape[0]
    call("glx_outset_build_dilated", x.indices, N_in, B, D, H, W, idx.rank_to_row, *ksize, *stride, *padding, *dilation,
         *out_shape, obitmap, oflags, oprefix, n_out_dev, x.count, ws, size_arg(ws.numel()))
    static = x.count is not None
    if static:
        cells = B * out_shape[0] * out_shape[1] * out_shape[2]
        reach = 1
        for k, st, d in zip(ksize, stride, dilation):
            reach *= -(-k // st) if d == 1 else k       # dilated taps land on distinct outputs
        N_out = min(int(out_capacity) if out_capacity else N_in * reach, cells)
    else:
        N_out = int(n_out_dev.item())  # host sync: output row count sizes the next tensors
    K = ksize[0] * ksize[1] * ksize[2]
    rs = RuleSet()
    rs.subm, rs.K, rs.N_in, rs.N_out = False, K, N_in, N_out
    rs.out_indices = torch.empty((max(N_out, 1), 4), dtype=torch.int32, device=dev)[:N_out]
    rs.nbr = torch.empty((max(N_out, 1), K), dtype=torch.int32, device=dev)
    if static:
        rs.count_in, rs.count_out = x.count, n_out_dev
    if N_out > 0:
        call("glx_outset_emit", obitmap, oflags, oprefix, *ogrid, N_out, rs.out_indices)
        call("glx_rules_strided_dilated", rs.out_indices, N_out, N_in, B, D, H, W, idx.bitmap, idx.prefix,
             idx.rank_to_row, *ksize, *stride, *padding, *dilation, rs.nbr, None, rs.count_out)
    rs.out_spatial_shape = out_shape
    # rows already sorted: rank == row
    rs.out_index = CellIndex(ogrid, obitmap, oflags, oprefix, None, None, N_out,
                             count=rs.count_out, unique=rs.count_out)
    rs.tile_order_out = None
    rs.tile_order_in = idx.rank_to_row
    rs.in_index, rs.in_indices, rs.in_spatial_shape = idx, x.indices, list(x.spatial_shape)
    rs.geom = ("spconv", ksize, stride, padding) if dilation == (1, 1, 1) else ("spconv", ksize, stride, padding, dilation)
    return rs


class PlannedConv:
    """Geometry of a strided sparse convolution that is not a module of the stack, for plan_rules(): its rule table is
    built with the stack's (on the plan stream) and found under `indice_key` in the tensors' indice_dict."""
    subm = inverse = False
    dilation = (1, 1, 1)

    def __init__(self, indice_key, kernel_size, stride, padding, in_channels, out_channels):
        self.indice_key, self.kernel_size, self.stride, self.padding = indice_key, tuple(kernel_size), tuple(stride), tuple(padding)
        self.in_channels, self.out_channels = in_channels, out_channels


def plan_rules(indices, spatial_shape, batch_size, convs, index=None, count=None,
               capacities=None, events=False, pair_lists=False):
    """Rule tables of a whole conv stack (modules in execution order) from coordinates only:
    returns {indice_key: RuleSet}.  Keys are required (they are how convs find their table).
    count: live rows of `indices` on the device (shape-static mode, no host sync at all);
    capacities: optional {indice_key: rows} for the output sets of the strided convs;
    events: record an event after each rule set (RuleSet.ready) -- for plans built on a side
    stream while the convolutions of the previous levels run on the main one;
    pair_lists: a training step's plan -- the per-offset pair lists the weight gradients contract over
    (RuleSet.pair_lists) are built behind the LAST rule table, so that no convolution of the forward pass waits for them."""
    x = SparseConvTensor(None, indices, spatial_shape, batch_size, count=count)
    x._index = index
    capacities = capacities or {}
    for conv in convs:
        key = conv.indice_key
        if key is None:
            raise ValueError("plan_rules needs an indice_key on every conv")
        if conv.inverse:
            continue
        rs = x.indice_dict.get(key)
        if rs is None:
            rs = (build_subm_rules(x, conv.kernel_size, conv.dilation) if conv.subm else
                  build_strided_rules(x, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                                      out_capacity=capacities.get(key)))
            x.indice_dict[key] = rs
        new_work = rs.ready is None
        if new_work and (conv.subm or not TILE_MAP_SUBM_ONLY) \
                and conv.in_channels * conv.out_channels >= TILE_MAP_MIN_WEIGHTS:
            rs.tile_map()      # on the plan stream too, before the event (see TILE_MAP_MIN_WEIGHTS)
        if events and new_work:
            rs.ready = torch.cuda.Event()
            rs.ready.record()
        if not conv.subm:
            nxt = SparseConvTensor(None, rs.out_indices, rs.out_spatial_shape, batch_size,
                                   indice_dict=x.indice_dict, count=rs.count_out)
            nxt._index = rs.out_index
            x = nxt
    if pair_lists and USE_PAIR_LISTS and PAIR_LISTS_IN_PLAN:
        for rs in x.indice_dict.values():
            if rs.nbr is not None and rs.N_out > 0:
                rs.pair_lists(rs.nbr, rs.N_out, rs.count_out)
    if pair_lists and INVERSE_TABLES_IN_PLAN:          # a training plan: the strided tables' transposes, for the input gradients
        for rs in x.indice_dict.values():
            if rs.nbr is not None and not rs.subm and rs.N_out > 0 and rs.N_in > 0:
                rs.inverse_table()
    return x.indice_dict


def check_static(indice_dict, index=None):
    """Host-side verdict of a shape-static frame (one sync): raises if a live row count exceeded
    its capacity or the voxelizer's cell index held cells that max_voxels dropped."""
    if index is not None and index.count is not None:
        n, u = int(index.count.item()), int(index.unique.item())
        if n > index.n:
            raise RuntimeError("shape-static frame: %d voxels exceed the capacity %d" % (n, index.n))
        if u != n:
            raise RuntimeError("shape-static frame: max_voxels dropped cells (%d of %d kept); "
                               "use the dynamic path or a larger max_voxels" % (n, u))
    for key, rs in indice_dict.items():
        if rs.count_out is not None:
            n = int(rs.count_out.item())
            if n > rs.N_out:
                raise RuntimeError("shape-static frame: rule set %r has %d output rows, capacity %d"
                                   % (key, n, rs.N_out))


_profile_hook = None   # bench.py installs a callable(tag, K, cin, cout, n_out, rules) -> (start, stop) HIP events or None


# Weight images packed ahead for the current training step by prepack(): {(data_ptr, adjoint, flip): packed}; a
# StaticTrainPipeline fills it at the top of the step with ONE launch and clears it at the end (None = no step open).
STEP_PACKS = None


def prepack(convs):
    """Pack the forward and the input-gradient (adjoint) weight image of every SparseConvolution in `convs` with one
    launch (glx_sconv_pack_weights_multi) and park them in STEP_PACKS for pack_weights() to hand out -- the 24 pack
    launches of a VoxelBackBone8x training step become one.  The caller resets STEP_PACKS = None after the step:
    the images are only valid while the weights do not change."""
    global STEP_PACKS
    import ctypes
    jobs = []
    for m in convs:
        w = m.weight.detach()
        if m.inverse or _cin_padding(m.in_channels) or not (w.is_cuda and w.is_contiguous() and w.dtype == torch.float32):
            continue
        wk = w.view(-1, w.shape[-2], w.shape[-1])
        K, cin, cout = wk.shape
        flip = bool(m.subm and not m.inverse)
        for adjoint, fl, ci, co in ((False, False, cin, cout), (True, flip, cout, cin)):
            nbytes = query("glx_sconv_packed_bytes", K, ci, co)
            if nbytes:
                jobs.append((wk, K, ci, co, adjoint, fl, torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)))
    STEP_PACKS = {}
    if not jobs:
        return
    n = len(jobs)
    ptrs = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
    outs = (ctypes.c_void_p * n)(*[j[6].data_ptr() for j in jobs])
    ints = [(ctypes.c_int32 * n)(*[int(j[i]) for j in jobs]) for i in (1, 2, 3, 4, 5)]
    call("glx_sconv_pack_weights_multi", n, ptrs, ints[0], ints[1], ints[2], ints[3], ints[4], outs)
    for j in jobs:
        STEP_PACKS[(j[0].data_ptr(), j[4], j[5])] = j[6]


def pack_weights(weight_kio, adjoint=False, flip=False):
    """MFMA-fragment-ordered copy of (K, Cin, Cout) weights, or None when the channel counts
    run on the scalar kernel.  adjoint: pack the weights of the input-gradient conv (Cout -> Cin)
    straight from the forward weights, with the taps reversed when flip (submanifold)."""
    if STEP_PACKS is not None:
        hit = STEP_PACKS.get((weight_kio.data_ptr(), bool(adjoint), bool(flip)))
        if hit is not None:
            return hit
    K, cin, cout = weight_kio.shape
    if adjoint:
        cin, cout = cout, cin
    nbytes = query("glx_sconv_packed_bytes", K, cin, cout)
    if nbytes == 0:
        return None
    wp = torch.empty(nbytes // 4, dtype=torch.float32, device=weight_kio.device)
    call("glx_sconv_pack_weights_view", weight_kio, K, cin, cout, 1 if adjoint else 0, 1 if flip else 0, wp)
    return wp


_MFMA_CIN = (4, 8, 16, 32, 64, 128)


def _cin_padding(cin):
    """Input channel counts the MFMA kernels do not tile (e.g. Waymo's 5 point features,
    waymo_dataset.yaml:55-59) are zero-padded to the next supported width: exact, differentiable
    (F.pad) and far cheaper than the scalar fallback."""
    target = next((c for c in _MFMA_CIN if c >= cin), None)
    return 0 if target is None else target - cin


def _sconv(features, weight_kio, bias, nbr, tile_order, n_out, packed=None, rules=None, tag="fwd",
           scale=None, shift=None, relu=False, n_live=None, dims=None, bn=None, bwd_bn=None, pre=None):
    """out[j] = relu?((sum_k features[nbr[j,k]] @ weight_kio[k] + bias) * scale + shift).
    weight_kio may be None when `packed` and dims = (K, Cin, Cout) are given.
    bn: a training-mode BatchNorm1d that follows the conv -- its batch statistics are taken in the kernel's epilogue
    (glx_sconv_opts.bn) and the call returns (out, coef, save_mean, save_invstd).
    bwd_bn: (y, coef, mean, invstd, gamma) -- the call is an input-gradient convolution whose output is the gradient of
    relu(bn(y)): the epilogue masks it with the ReLU and takes the BatchNorm backward's two sums (glx_sconv_opts.bn_bwd);
    returns (dz, coef3 (3 * cout), dgamma, dbeta).
    pre: coef (2 * cin: scale | shift) -- the input rows are relu(features * scale + shift) on load (glx_sconv_opts.prologue)."""
    K, cin, cout = dims if dims is not None else weight_kio.shape
    out = torch.empty((n_out, cout), dtype=torch.float32, device=features.device)
    if n_out == 0:
        assert bn is None
        return out
    if packed is None:
        packed = pack_weights(weight_kio)
    ws = workspace.get(256, features.device)
    opts = None           # per-call options travel as an argument (glx_sconv_opts): tile map, BatchNorm statistics, events
    if _profile_hook is not None:
        events = _profile_hook(tag, K, cin, cout, n_out, rules)       # (start, stop) HIP events or None
        if events is not None:
            opts = _lib.SconvOpts(None, None, events[0], events[1])
    if rules is not None and packed is not None and (rules.subm or rules._tile_maps) \
            and cin * cout >= TILE_MAP_MIN_WEIGHTS:
        tmap = rules.tile_map(nbr, tile_order, n_out, n_live)
        if tmap is not None:
            opts = opts or _lib.SconvOpts()
            opts.tile_map = tmap.data_ptr()
    stats = None
    bn_count = 0
    if isinstance(bn, tuple):       # (module, elements per channel): statistics over more than the rows (zeros not stored)
        bn, bn_count = bn
    if bn is not None:
        stats = tuple(torch.empty(n, dtype=torch.float32, device=features.device) for n in (2 * cout, cout, cout))
        st = _lib.bn_stats(_bn_state(features.device), bn, *stats, count=bn_count)
        opts = opts or _lib.SconvOpts()
        opts.bn = ctypes.pointer(st)
    if bwd_bn is not None:
        y_prev, coef_prev, mean_prev, invstd_prev, gamma_prev = bwd_bn
        stats = tuple(torch.empty(n, dtype=torch.float32, device=features.device) for n in (3 * cout, cout, cout))
        stb = _lib.BnBwdStats(*[_lib._p(t) for t in (_bn_state(features.device), y_prev, coef_prev, mean_prev, invstd_prev,
                                                     gamma_prev) + stats])
        opts = opts or _lib.SconvOpts()
        opts.bn_bwd = ctypes.pointer(stb)
    if pre is not None:
        pro = _lib.epilogue(pre[:cin], pre[cin:], True)
        opts = opts or _lib.SconvOpts()
        opts.prologue = ctypes.pointer(pro)
    call("glx_sconv_forward_ex", features, features.shape[0], weight_kio, packed, bias, scale, shift,
         1 if relu else 0, nbr, tile_order, n_out, K, cin, cout, out, n_live, ws,
         size_arg(ws.numel()), ctypes.byref(opts) if opts is not None else None)
    if bn is not None:
        if bn.track_running_stats:
            _lib.bump_weights_epoch((bn.running_mean, bn.running_var))     # moved behind torch's back
        return (out,) + stats
    if bwd_bn is not None:
        return (out,) + stats
    return out


# The BatchNorm backward's statistics pass of an inner layer rides in the epilogue of the NEXT convolution's input-gradient
# launch (glx_sconv_opts.bn_bwd): FusedBNApply.forward leaves a link {y, coef, mean, invstd, gamma, out} on the tensor it
# returns; the SparseConvFunction that consumes exactly that tensor runs its input gradient with the link (ReLU mask, dz,
# sum dz, sum dz xhat, finalize) and parks (dz, coef3, dgamma, dbeta) on it; FusedBNApply.backward, handed that very dz,
# only applies the transform.  Any other gradient (several consumers: autograd hands over a sum) takes the full path --
# the mask is idempotent, so an already masked contribution inside the sum is still right.
BN_BWD_IN_DGRAD = True


def _pre_arg(pre, cin):
    """glx_epilogue* of an input transform (coef = scale | shift) or NULL."""
    if pre is None:
        return None
    return ctypes.byref(_lib.epilogue(pre[0][:cin], pre[0][cin:], True))


# Inner layers of the sparse backbone's blocks: relu(bn(y)) of a convolution is not written at all when the consumer is the
# next convolution -- it transforms y on load (glx_sconv_opts.prologue), its weight gradient does the same
# (glx_sconv_wgrad_pairs_ex) and its backward carries the BatchNorm's.  The producer leaves a `_pending` transform on its
# SparseConvTensor; anything else that reads `.features` materialises it (FusedBNApply) then.  GLX_SCONV_BN_ON_LOAD=0: off.
BN_ON_LOAD = True


class PendingBN:
    """relu(bn(raw)) not yet computed: raw (N, C) = the convolution's output, coef / mean / invstd from its epilogue."""

    def __init__(self, raw, coef, mean, invstd, bn, count, link):
        self.raw, self.coef, self.mean, self.invstd, self.bn, self.count, self.link = raw, coef, mean, invstd, bn, count, link

    def materialise(self):
        return FusedBNApply.apply(self.raw, self.coef, self.mean, self.invstd, self.bn.weight, self.bn.bias, True, self.count,
                                  self.link)


class SparseConvFunction(Function):
    """features (N_in, Cin), weight (K, Cin, Cout), bias (Cout)|None -> (N_out, Cout)."""

    @staticmethod
    def forward(ctx, features, weight, bias, rules, inverse, packed=None, side_ok=False, bn=None, in_link=None, leaf=None,
                pre_coef=None, pre_mean=None, pre_invstd=None, pre_gamma=None, pre_beta=None, pre_count=None):
        """bn: the training-mode BatchNorm1d behind the conv: its statistics ride in the kernel's epilogue and the
        call returns (out, coef, save_mean, save_invstd) for FusedBNApply (the last three non-differentiable).
        in_link: the link FusedBNApply left on `features` (see BN_BWD_IN_DGRAD).
        pre_*: `features` is the RAW output y of the convolution in front and this convolution reads relu(bn(y)) by
        transforming the rows on load (pre_coef = scale | shift its epilogue left, BN_ON_LOAD); this node's backward then
        carries that BatchNorm's backward: it returns d/dy and the gradients of pre_gamma / pre_beta."""
        ctx.pre = pre_coef is not None
        ctx.pre_count = pre_count
        ctx.in_link = in_link if (in_link is not None and in_link.get("out") is not None and in_link["out"]() is features) else None
        # leaf: the parameter `weight` is a plain view of (None: it is not) -- what a deferred weight-gradient sum writes to
        ctx.leaf = leaf if (leaf is not None and leaf.is_leaf and leaf.requires_grad and leaf.numel() == weight.numel()) else None
        features = features.contiguous().float()
        w = weight.contiguous()
        _lib.check_cuda(features, w)
        if inverse:      # SparseInverseConv3d: walk the paired conv's rules backwards
            nbr, order, n_out = rules.inverse_table(), rules.tile_order_in, rules.N_in
        else:
            nbr, order, n_out = rules.nbr, rules.tile_order_out, rules.N_out
        out = _sconv(features, w, bias, nbr, order, n_out, packed=packed, rules=rules,
                     n_live=rules.count_in if inverse else rules.count_out, bn=bn, pre=pre_coef)
        ctx.rules, ctx.inverse, ctx.side_ok = rules, inverse, side_ok
        if ctx.pre:
            ctx.save_for_backward(features, w, pre_coef, pre_mean, pre_invstd, pre_gamma)
        else:
            ctx.save_for_backward(features, w)
        ctx.has_bias = bias is not None
        if bn is not None:
            ctx.mark_non_differentiable(*out[1:])
            ctx.set_materialize_grads(False)       # no zero tensors (3 fill launches) for the statistics' "gradients"
        return out

    @staticmethod
    def backward(ctx, grad_out, *_stats_grads):
        features, w = ctx.saved_tensors[:2]
        pre = ctx.saved_tensors[2:] if ctx.pre else None        # (coef, mean, invstd, gamma) of the BatchNorm in front
        g_pre_gamma = g_pre_beta = None
        rules, inverse = ctx.rules, ctx.inverse
        grad_out = grad_out.contiguous().float()
        K, cin, cout = w.shape
        g_feat = g_w = g_b = None
        # live_fwd / live_bwd: device row counts of a shape-static rule set (None = exact shapes);
        # rows past them hold undefined data in every tensor and are never read by the kernels
        if inverse:
            fwd_nbr, n_fwd_out, live_fwd = rules.inverse_table(), rules.N_in, rules.count_in
            bwd_nbr, bwd_order, n_bwd_out, live_bwd = rules.nbr, rules.tile_order_out, rules.N_out, rules.count_out
            flip = False
        elif rules.subm:
            # nbr_in[i][k] == nbr[i][K-1-k] on a submanifold set: flip the taps instead
            fwd_nbr, n_fwd_out, live_fwd = rules.nbr, rules.N_out, rules.count_out
            bwd_nbr, bwd_order, n_bwd_out, live_bwd = rules.nbr, rules.tile_order_out, rules.N_in, rules.count_in
            flip = True
        else:
            fwd_nbr, n_fwd_out, live_fwd = rules.nbr, rules.N_out, rules.count_out
            bwd_nbr, bwd_order, n_bwd_out, live_bwd = (rules.inverse_table(), rules.tile_order_in, rules.N_in,
                                                       rules.count_in)
            flip = False
        if ctx.needs_input_grad[1]:
            side = WGRAD_STREAM if ctx.side_ok else None
            if side is not None:
                # the weight gradient is a leaf of the backward pass: it runs on a second stream
                # next to the input-gradient chain (whoever set WGRAD_STREAM joins it afterwards)
                cur = torch.cuda.current_stream(w.device)
                side.wait_stream(cur)
                for t in (features, grad_out, fwd_nbr):
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                pairs = USE_PAIR_LISTS and n_fwd_out > 0 and query("glx_sconv_packed_bytes", K, cin, cout)
                if pairs:
                    # written where the optimizer reads it when the parameter is a plain view of `w` (no gather copy)
                    g_w = _lib.grad_buffer(ctx.leaf, w.shape) if ctx.leaf is not None else torch.empty_like(w)
                    pl = rules.pair_lists(fwd_nbr, n_fwd_out, live_fwd)
                    wsb = query("glx_sconv_wgrad_pairs_workspace_bytes", n_fwd_out, K, cin, cout)
                    # bench.py's profiler: torch events around the weight gradient's two launches (pairs + slab sum)
                    wev = _profile_hook.wgrad(K, cin, cout, rules) if hasattr(_profile_hook, "wgrad") else None
                    defer = (DEFERRED_WGRAD_REDUCES is not None and wev is None and ctx.leaf is not None
                             and _lib.is_lent(ctx.leaf, g_w))
                    # deferred: the chunk products now, into a buffer of this layer's own; the sums of ALL layers in one
                    # launch from run_deferred_wgrad_reduces() (autograd keeps `g_w`, the optimizer's view, and reads nothing)
                    ws = torch.empty(wsb, dtype=torch.uint8, device=w.device) if defer else workspace.get(wsb, w.device)
                    if wev is not None:
                        wev[0].record()
                    call("glx_sconv_wgrad_pairs_ex", features, grad_out, pl, n_fwd_out, K, cin, cout, None if defer else g_w,
                         _pre_arg(pre, cin), ws, size_arg(ws.numel()))
                    if defer:
                        # an ALIAS of g_w: AccumulateGrad keeps a gradient as it is only while nobody else holds the tensor object
                        DEFERRED_WGRAD_REDUCES.append((pl, n_fwd_out, K, cin, cout, g_w.detach(), ws, torch.cuda.current_stream(w.device), ctx.leaf))
                    if wev is not None:
                        wev[1].record()
                else:
                    assert pre is None, "the input transform on load needs the pair-list weight gradient"
                    g_w = torch.empty_like(w)
                    wsb = query("glx_sconv_wgrad_workspace_bytes", n_fwd_out, K, cin, cout)
                    ws = workspace.get(wsb, w.device)
                    call("glx_sconv_wgrad", features, features.shape[0], grad_out, fwd_nbr, n_fwd_out, K,
                         cin, cout, g_w, live_fwd, 1 if (rules.subm and not inverse) else 0, ws, size_arg(ws.numel()))
        if ctx.needs_input_grad[0]:
            # input gradient = the same kernels on the adjoint weights (Cout -> Cin, taps flipped on
            # a submanifold set), packed from the forward weights in one launch
            wp_t = pack_weights(w, adjoint=True, flip=flip)
            link = ctx.in_link
            if pre is not None:
                # the input was relu(bn(features)): the input-gradient launch masks with the ReLU and takes the BatchNorm
                # backward's sums in its epilogue, one transform launch turns that into d/d(features)
                coef_p, mean_p, invstd_p, gamma_p = pre
                dz, coef3, g_pre_gamma, g_pre_beta = _sconv(grad_out, None, None, bwd_nbr, bwd_order, n_bwd_out, packed=wp_t,
                                                            rules=rules, tag="dgrad", n_live=live_bwd, dims=(K, cout, cin),
                                                            bwd_bn=(features, coef_p, mean_p, invstd_p, gamma_p))
                g_feat = torch.empty_like(features)
                call("glx_bn_backward_apply", features, dz, coef3, mean_p, invstd_p, features.shape[0], cin, ctx.pre_count, g_feat)
            elif (wp_t is not None and link is not None and BN_BWD_IN_DGRAD and USE_BN_STATE and link["y"].shape == (n_bwd_out, cin)
                    and cin in (16, 32, 64, 128) and not (cin >= 128 and cout >= 128)):
                g_feat, coef3, dgamma, dbeta = _sconv(grad_out, None, None, bwd_nbr, bwd_order, n_bwd_out, packed=wp_t,
                                                      rules=rules, tag="dgrad", n_live=live_bwd, dims=(K, cout, cin),
                                                      bwd_bn=(link["y"], link["coef"], link["mean"], link["invstd"],
                                                              link["gamma"]))
                link["result"] = (g_feat, coef3, dgamma, dbeta)
            elif wp_t is not None:
                g_feat = _sconv(grad_out, None, None, bwd_nbr, bwd_order, n_bwd_out, packed=wp_t, rules=rules,
                                tag="dgrad", n_live=live_bwd, dims=(K, cout, cin))
            else:       # channel counts of the scalar kernel: materialise the adjoint weights
                wt = (w.flip(0) if flip else w).transpose(1, 2).contiguous()
                g_feat = _sconv(grad_out, wt, None, bwd_nbr, bwd_order, n_bwd_out, rules=rules, tag="dgrad",
                                n_live=live_bwd)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if live_fwd is None:
                g_b = grad_out.sum(0)
            else:   # rows past the live count are undefined (possibly NaN): select, do not multiply
                live = torch.arange(grad_out.shape[0], device=grad_out.device) < live_fwd
                g_b = torch.where(live[:, None], grad_out, grad_out.new_zeros(())).sum(0)
        return g_feat, g_w, g_b, None, None, None, None, None, None, None, None, None, None, g_pre_gamma, g_pre_beta, None


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False, count=None):
        self._pending = None    # PendingBN: `features` = relu(bn(raw)) is computed when somebody reads it (BN_ON_LOAD)
        self.features = features
        self.count = count      # shape-static mode: device int32[1] live rows (None = all rows)
        if indices.dtype != torch.int32:
            indices = indices.int()
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self._index = None

    @property
    def features(self):
        if self._features is None and self._pending is not None:
            self._features = self._pending.materialise()
            if self._pending.link is not None:
                self._bn_link = self._pending.link
            self._pending = None
        return self._features

    @features.setter
    def features(self, value):
        self._features = value
        self._pending = None

    def features_meta(self):
        """A tensor with the features' shape / dtype / device (the raw rows while the transform is pending)."""
        return self._features if self._features is not None else self._pending.raw

    # -- spconv 2.x API used by pcdet/utils/spconv_utils.py:28-34
    def replace_feature(self, new_features):
        t = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size,
                             self.grid, self.voxel_num, self.indice_dict, self.benchmark, self.count)
        t._index = self._index
        return t

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def _ensure_index(self):
        if self._index is None:
            if self.count is not None:
                raise ValueError("a shape-static SparseConvTensor needs the cell index of its "
                                 "producer (hard_voxelize(..., static=True))")
            grid = (self.batch_size, *self.spatial_shape)
            self._index = CellIndex.build(self.indices, grid)
        return self._index

    def dense(self, channels_first=True):
        """(B, C, D, H, W) dense tensor (height_compression.py:21-23)."""
        return DenseFunction.apply(self.features, self, channels_first)

    def dense_bev(self):
        """HeightCompression's tensor in one step: logical shape (B, C * D, H, W) (= dense().view(B, C * D, H, W),
        height_compression.py:21-25) held in channels-last memory (B, H, W, C * D), the layout MIOpen's NHWC
        convolution kernels read without transposing.  Needs the cell index (falls back to dense() otherwise)."""
        idx = self._index
        c = self.features.shape[1]
        if (idx is None or self.features.shape[0] == 0 or (c * self.spatial_shape[0]) % 4
                or not (self.count is None or idx.count is self.count)):
            d = self.dense()
            n, c_, dd, h, w = d.shape
            return d.view(n, c_ * dd, h, w)
        return DenseBevFunction.apply(self.features, self)


class DenseFunction(Function):
    @staticmethod
    def forward(ctx, features, st, channels_first):
        f = features.contiguous().float()
        _lib.check_cuda(f, st.indices)
        N, C = f.shape
        D, H, W = st.spatial_shape
        idx = st._index
        if idx is not None and N > 0 and (st.count is None or idx.count is st.count):
            # one pass over the output (value or zero per element) through the cell index
            out = torch.empty((st.batch_size, C, D, H, W), dtype=torch.float32, device=f.device)
            call("glx_dense_from_index", f, N, C, idx.bitmap, idx.prefix, idx.rank_to_row,
                 st.batch_size, D, H, W, out)
        else:
            out = torch.zeros((st.batch_size, C, D, H, W), dtype=torch.float32, device=f.device)
            call("glx_dense_scatter", f, st.indices, N, C, st.batch_size, D, H, W, out, st.count)
        ctx.st, ctx.channels_first = st, channels_first
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        if not ctx.channels_first:
            g = g.permute(0, 4, 1, 2, 3)
        g = g.contiguous().float()
        n, c = st.indices.shape[0], g.shape[1]
        d, h, w = st.spatial_shape
        gf = torch.empty((n, c), dtype=torch.float32, device=g.device)
        call("glx_dense_gather", g, st.indices, n, c, st.batch_size, d, h, w, gf, st.count)
        return gf, None, None


class DenseBevFunction(Function):
    @staticmethod
    def forward(ctx, features, st):
        f = features.contiguous().float()
        _lib.check_cuda(f, st.indices)
        n, c = f.shape
        d, h, w = st.spatial_shape
        idx = st._index
        out = torch.empty((st.batch_size, h, w, c * d), dtype=torch.float32, device=f.device)
        call("glx_dense_from_index_nhwc", f, n, c, idx.bitmap, idx.prefix, idx.rank_to_row, st.batch_size, d, h, w, out)
        ctx.st = st
        return out.permute(0, 3, 1, 2)              # logical NCHW, channels-last strides

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        g = g.permute(0, 2, 3, 1).contiguous().float()      # a view when the gradient is channels-last too
        n, c = st.indices.shape[0], st.features.shape[1]
        d, h, w = st.spatial_shape
        gf = torch.empty((n, c), dtype=torch.float32, device=g.device)
        call("glx_dense_gather_nhwc", g, st.indices, n, c, st.batch_size, d, h, w, gf, st.count)
        return gf, None


class SparseModule(nn.Module):
    """Marker base class: modules that take and return a SparseConvTensor."""
    pass


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0,
                 dilation=1, groups=1, bias=True, subm=False, output_padding=0, transposed=False,
                 inverse=False, indice_key=None, fused_bn=False, use_hash=False, algo=None):
        super().__init__()
        assert ndim == 3, "only 3-D sparse convolution is on the hot path"
        assert groups == 1
        self.ndim, self.in_channels, self.out_channels = ndim, in_channels, out_channels
        self.kernel_size = _triple(kernel_size)
        self.stride, self.padding, self.dilation = _triple(stride), _triple(padding), _triple(dilation)
        self.subm, self.inverse, self.transposed = subm, inverse, transposed
        self.indice_key = indice_key
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # same recipe as torch.nn.ConvNd / spconv: kaiming_uniform(a=sqrt(5)); the receptive
        # fan-in is K*Cin for a (k,k,k,Cin,Cout) weight
        K = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        fan_in = K * self.in_channels
        gain = math.sqrt(2.0 / (1 + 5.0))
        bound = gain * math.sqrt(3.0 / fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                b = 1 / math.sqrt(fan_in)
                self.bias.uniform_(-b, b)

    def extra_repr(self):
        return ("{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, "
                "padding={padding}, subm={subm}, indice_key={indice_key}").format(**self.__dict__)

    def _packed_weight(self, w):
        """Packed copy of the weights, refreshed when the parameter changes (optimizer step,
        load_state_dict, .to())."""
        if torch.is_grad_enabled() and self.weight.requires_grad:
            # training: pack inside the step, every step.  The weights change between steps without the
            # version counter moving (a fused optimizer writes through raw pointers; a replayed HIP graph
            # runs no Python at all), so a version-keyed cache would feed the forward stale weights while
            # the backward packs the adjoint from the live ones.
            with torch.no_grad():
                return pack_weights(w.detach().contiguous())
        tag = (self.weight._version, self.weight.data_ptr(), self.weight.device, _lib.weights_epoch(self.weight))
        cache = self.__dict__.get("_packed_cache")
        if cache is None or cache[0] != tag:
            with torch.no_grad():
                cache = (tag, pack_weights(w.detach().contiguous()))
            self.__dict__["_packed_cache"] = cache
        return cache[1]

    def _rules(self, x):
        key = self.indice_key
        if self.inverse:
            rs = x.find_indice_pair(key)
            if rs is None:
                raise ValueError("SparseInverseConv3d needs the indice_key of a previous "
                                 "SparseConv3d (got %r)" % (key,))
            return rs
        rs = x.find_indice_pair(key)
        if rs is not None:
            if rs.N_in != x.indices.shape[0]:
                raise ValueError("indice_key %r was built for %d inputs, tensor has %d"
                                 % (key, rs.N_in, x.indices.shape[0]))
            return rs
        if self.subm:
            rs = build_subm_rules(x, self.kernel_size, self.dilation)
        else:
            rs = build_strided_rules(x, self.kernel_size, self.stride, self.padding, self.dilation)
        if key is not None:
            x.indice_dict[key] = rs
        return rs

    def forward(self, x, fused_bn=None, fused_relu=False, train_bn=None, train_relu=False, residual=None):
        """fused_bn / fused_relu: inference-only folding of the eval-mode BatchNorm1d (+ReLU)
        that follows this conv into the kernel's epilogue (see SparseSequential).
        train_bn / train_relu: the TRAINING-mode BatchNorm1d (+ReLU) behind the conv: statistics in the conv's
        epilogue, then one transform launch (FusedBNApply).  residual (with train_bn): the output is relu(bn(conv(x)) + residual),
        the tail of a residual block in that same launch (FusedBNApplyAdd)."""
        assert isinstance(x, SparseConvTensor)
        K = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        w = self.weight.reshape(K, self.in_channels, self.out_channels)
        pad = _cin_padding(self.in_channels)
        # the producer left relu(bn(raw)) pending: take the raw rows and transform them on load when this launch can
        pend = x._pending if (x._features is None and not pad and fused_bn is None and not fused_relu
                              and x.indices.shape[0] > 1) else None
        if pend is not None and not (BN_ON_LOAD and BN_BWD_IN_DGRAD and USE_BN_STATE and USE_PAIR_LISTS and torch.is_grad_enabled()
                                     and self.in_channels in (16, 32, 64, 128) and self.out_channels in (16, 32, 64, 128)
                                     and not (self.in_channels >= 128 and self.out_channels >= 128)
                                     and query("glx_sconv_packed_bytes", K, self.in_channels, self.out_channels)):
            pend = None
        x_features = pend.raw if pend is not None else x.features
        if pad:
            w = torch.nn.functional.pad(w, (0, 0, 0, pad))
            x_features = torch.nn.functional.pad(x_features, (0, pad))
        rs = self._rules(x)
        if rs.ready is not None:       # built on another stream: order this stream after it
            torch.cuda.current_stream(x.indices.device).wait_event(rs.ready)
        out_link = None
        if fused_bn is not None or fused_relu:
            scale, shift = _bn_affine(fused_bn) if fused_bn is not None else (None, None)
            if self.inverse:
                nbr, order, n_out = rs.inverse_table(), rs.tile_order_in, rs.N_in
            else:
                nbr, order, n_out = rs.nbr, rs.tile_order_out, rs.N_out
            feats = _sconv(x_features.contiguous().float(), w.detach().contiguous(), self.bias, nbr,
                           order, n_out, packed=self._packed_weight(w), rules=rs, scale=scale,
                           shift=shift, relu=fused_relu,
                           n_live=rs.count_in if self.inverse else rs.count_out)
        else:
            # side_ok: the weight gradient may run on WGRAD_STREAM only when nothing but views
            # separates it from the parameter (a padded weight's backward copies on the main stream)
            in_link = getattr(x, "_bn_link", None) if (not pad and pend is None) else None
            pre_args = (None,) * 6 if pend is None else (pend.coef, pend.mean, pend.invstd, pend.bn.weight, pend.bn.bias, pend.count)
            pending_out = None
            if train_bn is not None:
                feats, coef, mean, invstd = SparseConvFunction.apply(x_features, w, self.bias, rs, self.inverse,
                                                                     self._packed_weight(w), not pad, train_bn, in_link,
                                                                     None if pad else self.weight, *pre_args)
                # relu only: the epilogue re-derives the ReLU mask; a BatchNorm without ReLU keeps the full backward
                out_link = {} if (BN_BWD_IN_DGRAD and train_relu and train_bn.affine) else None
                cnt = rs.count_in if self.inverse else rs.count_out
                if residual is not None:
                    feats = FusedBNApplyAdd.apply(feats, coef, mean, invstd, train_bn.weight, train_bn.bias, residual, cnt)
                    out_link = None
                elif BN_ON_LOAD and out_link is not None and self.out_channels in (16, 32, 64, 128):
                    pending_out = PendingBN(feats, coef, mean, invstd, train_bn, cnt, out_link)      # whoever reads it first
                    feats, out_link = None, None
                else:
                    feats = FusedBNApply.apply(feats, coef, mean, invstd, train_bn.weight, train_bn.bias, train_relu, cnt,
                                               out_link)
                if train_bn.track_running_stats and train_bn.num_batches_tracked is not None:
                    if DEFERRED_COUNTERS is not None:
                        DEFERRED_COUNTERS.append(train_bn.num_batches_tracked)
                    else:
                        train_bn.num_batches_tracked += 1
            else:
                out_link = None
                feats = SparseConvFunction.apply(x_features, w, self.bias, rs, self.inverse,
                                                 self._packed_weight(w), not pad, None, in_link, None if pad else self.weight,
                                                 *pre_args)
        if self.inverse:
            out = SparseConvTensor(feats, rs.in_indices, rs.in_spatial_shape, x.batch_size,
                                   x.grid, x.voxel_num, x.indice_dict, x.benchmark, rs.count_in)
            out._index = rs.in_index
        else:
            out = SparseConvTensor(feats, rs.out_indices, rs.out_spatial_shape, x.batch_size,
                                   x.grid, x.voxel_num, x.indice_dict, x.benchmark, rs.count_out)
            out._index = rs.out_index
        if out_link is not None:
            out._bn_link = out_link
        if not (fused_bn is not None or fused_relu) and pending_out is not None:
            out._pending = pending_out
        return out


def _bn_affine(bn):
    """Eval-mode BatchNorm1d as y = x * scale + shift (cached until its tensors change)."""
    tag = (bn.weight._version if bn.weight is not None else -1,
           bn.bias._version if bn.bias is not None else -1,
           bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(),
           _lib.weights_epoch(bn.weight, bn.bias, bn.running_mean, bn.running_var))
    cache = bn.__dict__.get("_glx_affine")
    if cache is None or cache[0] != tag:
        with torch.no_grad():
            inv = torch.rsqrt(bn.running_var.float() + bn.eps)
            scale = inv * bn.weight.float() if bn.weight is not None else inv
            shift = -bn.running_mean.float() * scale
            if bn.bias is not None:
                shift = shift + bn.bias.float()
            cache = (tag, scale.contiguous(), shift.contiguous())
        bn.__dict__["_glx_affine"] = cache
    return cache[1], cache[2]


# accumulators of the statistics kernels (csrc/glx_bn.hip: k_bn_stats): zero-filled once, self-cleaning, one per
# (device, stream) -- launches that share a buffer must be ordered by their stream.  GLX_BN_STATE=0: the fixed-order
# three-launch scheme.
USE_BN_STATE = True
_BN_STATES = {}


def _bn_state(device):
    if not USE_BN_STATE:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    st = _BN_STATES.get(key)
    if st is None:
        st = _BN_STATES[key] = torch.zeros(query("glx_bn_state_bytes"), dtype=torch.uint8, device=device)
    return st


class FusedBNReLU(Function):
    """Training-mode BatchNorm1d (+ReLU) on (N, C) features in two launches forward and two
    backward (csrc/glx_bn.hip); numerics of nn.BatchNorm1d(eps, momentum) + nn.ReLU."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, count=None):
        x = x.contiguous().float()
        N, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace.get(query("glx_bn_workspace_bytes", C), x.device)
        call("glx_bn_relu_train_forward", x, N, C, weight, bias, ctypes_float(eps), ctypes_float(momentum),
             1 if relu else 0, running_mean, running_var, y, mean, invstd, count, ws, size_arg(ws.numel()),
             _bn_state(x.device), 0)
        if running_mean is not None:
            _lib.bump_weights_epoch((running_mean, running_var))      # running statistics moved behind torch's back
        ctx.save_for_backward(x, weight, bias, mean, invstd)          # not y: backward re-derives the ReLU mask from x
        ctx.relu, ctx.count = relu, count
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous().float()
        N, C = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace.get(query("glx_bn_workspace_bytes", C), x.device)
        call("glx_bn_relu_backward", x, dy, None, N, C, weight, bias, mean, invstd, 1 if ctx.relu else 0, dx,
             dgamma, dbeta, ctx.count, ws, size_arg(ws.numel()), _bn_state(x.device), 0)
        return dx, (dgamma if weight is not None else None), (dbeta if weight is not None else None), \
            None, None, None, None, None, None


class FusedBNApply(Function):
    """The transform half of FusedBNReLU for statistics that were taken in the producing sparse conv's epilogue
    (SparseConvFunction with bn=...): y = relu?(x * scale + shift), one launch; backward = FusedBNReLU's."""

    @staticmethod
    def forward(ctx, x, coef, mean, invstd, weight, bias, relu, count=None, link=None):
        N, C = x.shape
        y = torch.empty_like(x)
        call("glx_bn_apply_forward", x, coef, 1 if relu else 0, N, C, count, y, 0)
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.relu, ctx.count = relu, count
        ctx.link = link
        if link is not None:
            # `out` weakly: the output's grad_fn is this node, whose ctx holds the link -- a strong reference closes a cycle
            # through C++ that Python's collector cannot see, and a step's autograd graph that outlives the step tears
            # the next capture down (see StaticTrainPipeline.enqueue)
            link.update(y=x, coef=coef, mean=mean, invstd=invstd, gamma=weight, out=weakref.ref(y), result=None)
        return y

    @staticmethod
    def backward(ctx, dy):
        link = ctx.link
        res = link.get("result") if link is not None else None
        if res is not None and res[0] is dy:
            # the convolution that consumed this tensor masked its input gradient and took the two sums in its epilogue
            x, weight, bias, mean, invstd = ctx.saved_tensors
            dz, coef3, dgamma, dbeta = res
            link["result"] = None
            dx = torch.empty_like(x)
            call("glx_bn_backward_apply", x, dz, coef3, mean, invstd, x.shape[0], x.shape[1], ctx.count, dx)
            return dx, None, None, None, dgamma, dbeta, None, None, None
        if link is not None:
            link["result"] = None
        dx, dgamma, dbeta = FusedBNReLU.backward(ctx, dy)[:3]
        return dx, None, None, None, dgamma, dbeta, None, None, None


class FusedBNApplyAdd(Function):
    """y = relu(bn(x) + res) for statistics taken in the producing conv's epilogue: the tail of a residual block
    (spconv_backbone.py:30-64) as ONE launch (transform, add and ReLU were three); backward: the ReLU's mask comes from the saved
    output (the BatchNorm-backward kernels take it: glx_bn_relu_backward's `y`), the identity branch gets the masked gradient."""

    @staticmethod
    def forward(ctx, x, coef, mean, invstd, weight, bias, res, count=None):
        N, C = x.shape
        res = res.contiguous()
        y = torch.empty_like(x)
        call("glx_bn_apply_add_forward", x, coef, res, 1, N, C, count, y)
        ctx.save_for_backward(x, weight, bias, mean, invstd, y)
        ctx.count = count
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, invstd, y = ctx.saved_tensors
        dy = dy.contiguous().float()
        N, C = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace.get(query("glx_bn_workspace_bytes", C), x.device)
        call("glx_bn_relu_backward", x, dy, y, N, C, weight, bias, mean, invstd, 1, dx, dgamma, dbeta, ctx.count, ws,
             size_arg(ws.numel()), _bn_state(x.device), 0)
        d_res = torch.ops.aten.threshold_backward(dy, y, 0) if ctx.needs_input_grad[6] else None
        return dx, None, None, None, dgamma, dbeta, d_res, None


class StackedBN(Function):
    """Training-mode BatchNorm of a STACKED tensor -- batch dimension 1, channels next: (1, C, M) / (1, C, M, nsample), the inputs
    of the BatchNorm1d / BatchNorm2d layers in voxel_pool_modules.py:70-130 -- on the channel-major kernels
    (glx_bn_cm_train_forward / _backward, csrc/glx_bn.hip): two launches per direction, nothing prepared per length."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        x = x.contiguous().float()
        c = x.shape[1]
        length = x.numel() // c
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        ws = workspace.get(query("glx_bn_cm_workspace_bytes", c), x.device)
        call("glx_bn_cm_train_forward", x, c, ctypes.c_longlong(length), weight, bias, ctypes_float(eps), ctypes_float(momentum),
             running_mean, running_var, y, mean, invstd, ws, size_arg(ws.numel()))
        if running_mean is not None:
            _lib.bump_weights_epoch((running_mean, running_var))
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous().float()
        c = x.shape[1]
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device) if weight is not None else None
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        ws = workspace.get(query("glx_bn_cm_workspace_bytes", c), x.device)
        call("glx_bn_cm_backward", x, dy, c, ctypes.c_longlong(x.numel() // c), weight, mean, invstd, dx, dgamma, dbeta, ws,
             size_arg(ws.numel()))
        return dx, dgamma, dbeta, None, None, None, None


def stacked_train_bn(bn, x):
    """nn.BatchNorm1d / BatchNorm2d `bn` in training mode on a stacked (1, C, ...) device tensor; None when the module or the
    tensor is not what StackedBN covers (the caller then runs the module's own forward)."""
    if not (bn.training and x.is_cuda and x.dim() >= 3 and x.shape[0] == 1 and x.dtype == torch.float32
            and x.shape[1] == bn.num_features and x.numel() > 0
            and (bn.momentum is not None or not bn.track_running_stats) and (bn.weight is None) == (bn.bias is None)):
        return None
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats and bn.running_mean is not None else (None, None)
    y = StackedBN.apply(x, bn.weight, bn.bias, rm, rv, float(bn.momentum if bn.momentum is not None else 0.0), float(bn.eps))
    if rm is not None and bn.num_batches_tracked is not None:
        if DEFERRED_COUNTERS is not None:
            DEFERRED_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    return y


def conv_bn_fusable(conv, bn, x):
    """A sparse conv whose training-mode BatchNorm statistics can ride in its epilogue (csrc/glx_sconv.hip
    sc_epilogue): MFMA tile kernel in one launch, channel counts the fused BatchNorm kernels cover."""
    cin, cout = conv.in_channels + _cin_padding(conv.in_channels), conv.out_channels
    return (FUSE_BN_STATS_IN_CONV and USE_BN_STATE and torch.is_grad_enabled() and can_fuse_train_bn(bn, x.features_meta())
            and bn.num_features == cout and cout in (16, 32, 64, 128) and cin in (4, 8, 16, 32, 64, 128)
            and not (cin >= 128 and cout >= 128) and not (cin in (4, 8) and cout > 32)
            and x.features_meta().is_cuda and x.indices.shape[0] > 1)


# On by default since round 3's second measurement: 10.18 ms per GLENet-VR training step with the statistics in the conv
# epilogue against 10.29 without (two alternating runs on one box).  The first measurement (11.79 / 11.77 / 11.68 against
# 11.67 / 11.68 / 11.68 ms) had autograd materialising zero "gradients" for the three statistics outputs of every conv --
# 36 fill launches per step that ate the 12 saved statistics launches; set_materialize_grads(False) removed them.
FUSE_BN_STATS_IN_CONV = True


class FusedBNReLUCat(Function):
    """torch.cat([relu(bn_i(x_i)) for i], dim=1) of (N, C_i) matrices without the concatenation copy: every
    BatchNorm's transform writes its column block of the (N, sum C_i) result (y_stride of glx_bn_relu_train_forward),
    backward reads its block of the incoming gradient in place (dy_stride) -- BaseBEVBackbone's concatenation of the
    upsampled maps (base_bev_backbone.py:100-104) on channels-last memory, 144 MB at the KITTI size.
    args: relu, then per part x, weight, bias, running_mean, running_var, momentum, eps."""

    @staticmethod
    def forward(ctx, relu, *args):
        parts = [args[i:i + 7] for i in range(0, len(args), 7)]
        xs = [p[0].contiguous().float() for p in parts]
        N = xs[0].shape[0]
        widths = [x.shape[1] for x in xs]
        total = sum(widths)
        out = torch.empty((N, total), dtype=torch.float32, device=xs[0].device)
        saved, col = [], 0
        for x, (_, w, b, rm, rv, momentum, eps), C in zip(xs, parts, widths):
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            invstd = torch.empty(C, dtype=torch.float32, device=x.device)
            ws = workspace.get(query("glx_bn_workspace_bytes", C), x.device)
            call("glx_bn_relu_train_forward", x, N, C, w, b, ctypes_float(eps), ctypes_float(momentum),
                 1 if relu else 0, rm, rv, out[:, col:], mean, invstd, None, ws, size_arg(ws.numel()),
                 _bn_state(x.device), total)
            if rm is not None:
                _lib.bump_weights_epoch((rm, rv))
            saved += [x, w, b, mean, invstd]
            col += C
        ctx.save_for_backward(*saved)
        ctx.relu, ctx.widths = relu, widths
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous().float()
        total = dout.shape[1]
        grads, col = [None], 0
        for i, C in enumerate(ctx.widths):
            x, w, b, mean, invstd = ctx.saved_tensors[5 * i:5 * i + 5]
            N = x.shape[0]
            dx = torch.empty_like(x)
            dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
            ws = workspace.get(query("glx_bn_workspace_bytes", C), x.device)
            call("glx_bn_relu_backward", x, dout[:, col:], None, N, C, w, b, mean, invstd, 1 if ctx.relu else 0, dx,
                 dgamma, dbeta, None, ws, size_arg(ws.numel()), _bn_state(x.device), total)
            grads += [dx, dgamma, dbeta, None, None, None, None]
            col += C
        return tuple(grads)


class FusedBNApplyCat(Function):
    """FusedBNReLUCat for parts whose statistics were taken elsewhere (a transposed convolution's epilogue,
    conv2d.deconv_bn_raw): every part's transform writes its column block of the concatenated result (one launch per part,
    no statistics pass); backward = FusedBNReLUCat's.  args: relu, then per part x, coef, mean, invstd, weight, bias."""

    @staticmethod
    def forward(ctx, relu, *args):
        parts = [args[i:i + 6] for i in range(0, len(args), 6)]
        N = parts[0][0].shape[0]
        widths = [p[0].shape[1] for p in parts]
        total = sum(widths)
        out = torch.empty((N, total), dtype=torch.float32, device=parts[0][0].device)
        saved, col = [], 0
        for (x, coef, mean, invstd, w, b), C in zip(parts, widths):
            call("glx_bn_apply_forward", x, coef, 1 if relu else 0, N, C, None, out[:, col:], total)
            saved += [x, w, b, mean, invstd]
            col += C
        ctx.save_for_backward(*saved)
        ctx.relu, ctx.widths = relu, widths
        return out

    @staticmethod
    def backward(ctx, dout):
        g = FusedBNReLUCat.backward(ctx, dout)           # (None, then per part dx, dgamma, dbeta, 4 x None)
        grads = [None]
        for i in range(len(ctx.widths)):
            dx, dgamma, dbeta = g[1 + 7 * i:4 + 7 * i]
            grads += [dx, None, None, None, dgamma, dbeta]
        return tuple(grads)


def fused_train_bn_cat(bns, features, relu):
    """relu(bn_i(features_i)) concatenated along the columns (FusedBNReLUCat); every bn as fused_train_bn takes it."""
    args = []
    for bn, f in zip(bns, features):
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        args += [f, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps]
    out = FusedBNReLUCat.apply(relu, *args)
    for bn in bns:
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            if DEFERRED_COUNTERS is not None:
                DEFERRED_COUNTERS.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked += 1
    return out


def ctypes_float(v):
    import ctypes
    return ctypes.c_float(float(v))


USE_FUSED_TRAIN_BN = True      # (False: torch's BatchNorm1d layer by layer; exact-shape eager steps only)
# work-balanced block -> tile maps for the sparse-conv kernels (RuleSet.tile_map); off with GLX_TILE_MAP=0
USE_TILE_MAP = True
USE_PAIR_LISTS = True      # weight gradients over per-offset pair lists
INVERSE_TABLES_IN_PLAN = True
PAIR_LISTS_IN_PLAN = True   # built behind the rule tables (0: by the first weight gradient)
TILE_MAP_MIN_ROWS = 64 * 256       # fewer tiles than CUs: nothing to balance
# building a map costs two small launches (~8 us): worth it for the rule tables of submanifold stacks
# (2-3 convs share one) with at least 32x32 weights per offset -- on the KITTI batch that is
# subm2..subm4, which save 13 / 19 / 7 us per frame; the thin first stack and the single-use strided
# tables would save 1-5 us
TILE_MAP_MIN_WEIGHTS = 32 * 32
TILE_MAP_SUBM_ONLY = True
# stream for the weight-gradient kernels of SparseConvFunction.backward (None = the current one).
# Set by StaticTrainPipeline around its backward pass; the setter waits for it afterwards.
WGRAD_STREAM = None
# A training step that sets this to a list gets the sparse layers' weight gradients in two parts: each layer's chunk products where
# the backward pass reaches it, the sums of ALL layers in one launch from run_deferred_wgrad_reduces() (thirteen ~8 us launches on
# the step's main chain become one).  None (the default): every weight gradient is complete when its call returns.
DEFERRED_WGRAD_REDUCES = None


def run_deferred_wgrad_reduces(jobs):
    """Finish the weight gradients `jobs` (the list DEFERRED_WGRAD_REDUCES was) on the current stream, which waits for the
    streams the chunk products were written on."""
    if not jobs:
        return
    cur = torch.cuda.current_stream(jobs[0][5].device)
    for st in {j[7] for j in jobs}:
        if st != cur:
            cur.wait_stream(st)
    n = len(jobs)
    i32 = ctypes.c_int32 * n
    ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    call("glx_sconv_wgrad_pairs_reduce_multi", n, ptrs([j[0] for j in jobs]), i32(*[j[1] for j in jobs]), i32(*[j[2] for j in jobs]),
         i32(*[j[3] for j in jobs]), i32(*[j[4] for j in jobs]), ptrs([j[5] for j in jobs]), ptrs([j[6] for j in jobs]),
         (ctypes.c_size_t * n)(*[j[6].numel() for j in jobs]))
    for j in jobs:
        j[6].record_stream(cur)
        _lib.settle_lent_grad(j[8], j[5])      # param.grad IS the filled view, or gets its contents


# list collecting the num_batches_tracked buffers of the fused BatchNorms of a step, so that the
# caller bumps them with ONE multi-tensor add instead of a tiny kernel per layer (None = bump at once)
DEFERRED_COUNTERS = None


def can_fuse_train_bn(bn, features):
    """Training-mode BatchNorm1d that the fused kernels cover (else nn.BatchNorm1d runs)."""
    c = bn.num_features
    return (USE_FUSED_TRAIN_BN and isinstance(bn, nn.BatchNorm1d) and bn.training and bn.affine and bn.momentum is not None
            and features.is_cuda and features.shape[0] > 1 and c <= 512
            and ((c % 4 == 0 and 1024 % c == 0) or features.shape[0] <= BN_SMALL_ROWS))


BN_SMALL_ROWS = 4096       # csrc/glx_bn.hip BN_SMALL_N: up to here any channel count runs (one block per channel)


def fused_train_bn(bn, features, relu, count=None):
    """count: device int32 live-row count of a shape-static tensor (statistics over live rows)."""
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    out = FusedBNReLU.apply(features, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, relu, count)
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if DEFERRED_COUNTERS is not None:
            DEFERRED_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    return out


def can_fuse_bn(bn):
    return (isinstance(bn, nn.BatchNorm1d) and not bn.training and bn.track_running_stats
            and not torch.is_grad_enabled())


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, use_hash=False, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, True, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, use_hash=False, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, False, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True,
                 algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                         indice_key=indice_key)


def is_spconv_module(m):
    return isinstance(m, SparseModule)


class SparseSequential(SparseModule):
    """Sequential that applies dense nn.Modules (BatchNorm1d, ReLU, ...) to `.features`
    (spconv_backbone.py:21-25)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, idx):
        mods = list(self._modules.values())
        return mods[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self._modules)), module)

    # A pending relu(bn(raw)) (BN_ON_LOAD) leaves the OUTERMOST SparseSequential materialised -- on the stream and at the place
    # in the launch order where the module ran (a lazy first read on another stream would put the transform, and with it the
    # BatchNorm's backward, on that stream) -- unless the container says its consumer is the next container's convolution.
    leave_pending = False
    _depth = 0

    def forward(self, x):
        SparseSequential._depth += 1
        try:
            x = self._forward(x)
        finally:
            SparseSequential._depth -= 1
        if (SparseSequential._depth == 0 and not self.leave_pending and isinstance(x, SparseConvTensor)
                and x._features is None and x._pending is not None):
            x.features                                   # materialise here
        return x

    def _forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, SparseConvolution) and i + 1 < len(mods) and can_fuse_bn(mods[i + 1])
                    and isinstance(x, SparseConvTensor)):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = m(x, fused_bn=mods[i + 1], fused_relu=relu)
                i += 3 if relu else 2
                continue
            if (isinstance(m, SparseConvolution) and i + 1 < len(mods) and isinstance(x, SparseConvTensor)
                    and isinstance(mods[i + 1], nn.BatchNorm1d) and conv_bn_fusable(m, mods[i + 1], x)):
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                x = m(x, train_bn=mods[i + 1], train_relu=relu)
                x.clean_rows = True        # the transform wrote zeros into the rows past `count`
                i += 3 if relu else 2
                continue
            if (isinstance(x, SparseConvTensor) and isinstance(m, nn.BatchNorm1d)
                    and x.indices.shape[0] > 1 and can_fuse_train_bn(m, x.features)):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = x.replace_feature(fused_train_bn(m, x.features, relu, x.count))
                x.clean_rows = True        # the kernels wrote zeros into the rows past `count`
                i += 2 if relu else 1
                continue
            if is_spconv_module(m):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.count is not None and isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training:
                    raise NotImplementedError("shape-static training needs the fused BatchNorm kernels "
                                              "(torch's batch statistics would include the padding rows)")
                if x.indices.shape[0] != 0:
                    x = x.replace_feature(m(x.features))
            else:
                x = m(x)
            i += 1
        return x
