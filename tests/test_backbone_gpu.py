"""GPU parity of the whole sparse backbone (config 2 shape) against the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle
from glenet_amd import backbone as gb
from glenet_amd import synth
from oracle import backbone as ob

pytestmark = pytest.mark.gpu
K = synth.KITTI


def _drop_graphs(*pipes):
    """Recorded pipelines are reference cycles: left to the cyclic collector their hipGraphExec objects die at an
    arbitrary later moment -- inside another test's capture or replay (ROCm 7.2 does not take that well).
    Tests that record graphs of their own release them here, synchronised."""
    import gc
    torch.cuda.synchronize()
    for p in pipes:
        p.graph = None
        p.out = p.loss = None
    gc.collect()
    torch.cuda.synchronize()


def _condition(model, seed=0):
    """Random-init weights shrink activations ~3x per layer; rescale so every layer's output is
    O(1) and the comparison tolerance means something."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.05)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.02 + 0.01)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


@pytest.mark.parametrize("residual,nframes", [(False, 2), (True, 1)])
def test_backbone_forward_matches_oracle(dev, residual, nframes):
    torch.manual_seed(1)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.SparseBackbone8x(4, grid, residual=residual).eval()
    _condition(model)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    frames = [synth.kitti_frame(10 + i, num_points=8000)[0] for i in range(nframes)]
    v, c, n = oracle.voxelize_hard_batch(frames, K["voxel_size"], K["point_cloud_range"], 5, 16000)
    ref = ob.backbone_forward(sd, oracle.mean_vfe(v, n), c, model.sparse_shape, residual=residual)

    model = model.to(dev)
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    with torch.no_grad():
        bd = gb.voxelize_batch(pts, bidx, nframes, K)
        assert np.array_equal(bd["voxel_coords"].cpu().numpy(), c)
        bd = gb.MeanVFE()(bd)
        bd = model(bd)
        bd = gb.HeightCompression()(bd)
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        t = bd["multi_scale_3d_features"][name]
        assert np.array_equal(t.indices.cpu().numpy(), ref[name].indices), name   # bit exact
        r = ref[name].features
        np.testing.assert_allclose(t.features.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(r).max()))
    out = bd["encoded_spconv_tensor"]
    assert np.array_equal(out.indices.cpu().numpy(), ref["out"].indices)
    r = ref["out"].features
    assert np.abs(r).max() > 1e-2
    np.testing.assert_allclose(out.features.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(r).max()))
    dense_ref = oracle.dense(r, ref["out"].indices, nframes, ref["out"].shape)
    sf = bd["spatial_features"].cpu().numpy()
    assert sf.shape == (nframes, 256, 200, 176)
    np.testing.assert_allclose(sf.reshape(dense_ref.shape), dense_ref, rtol=1e-4,
                               atol=1e-4 * max(1.0, np.abs(r).max()))


def _frames_on(dev, nframes, seed0=20, num_points=9000):
    frames = [synth.kitti_frame(seed0 + i, num_points=num_points)[0] for i in range(nframes)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    return pts, bidx


@pytest.mark.parametrize("residual", [False, True])
def test_static_pipeline_and_graph_match_dynamic_path(dev, residual):
    """The host-sync-free (shape-static) frame and its HIP-graph replay give bit-identical
    results to the dynamic path, also when the graph is replayed on a different batch."""
    torch.manual_seed(2)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.SparseBackbone8x(4, grid, residual=residual).eval().to(dev)
    _condition(model)
    nframes = 2
    pts, bidx = _frames_on(dev, nframes)

    def dynamic(pts, bidx):
        with torch.no_grad():
            bd = gb.voxelize_batch(pts, bidx, nframes, K)
            bd = gb.MeanVFE()(bd)
            bd = model(bd)
            return gb.HeightCompression()(bd)

    ref = dynamic(pts, bidx)
    pipe = gb.StaticFramePipeline(model, K, nframes, pts.shape[0] + 500, 4)   # padded input buffer
    pipe.load(pts, bidx)
    bd = pipe.enqueue()
    torch.cuda.synchronize()
    pipe.check()
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        f, i = pipe.live(bd["multi_scale_3d_features"][name])
        r = ref["multi_scale_3d_features"][name]
        assert torch.equal(i, r.indices), name
        assert torch.equal(f, r.features), name
    assert torch.equal(bd["spatial_features"], ref["spatial_features"])

    pipe.capture()
    pts2, bidx2 = _frames_on(dev, nframes, seed0=40, num_points=8500)
    ref2 = dynamic(pts2, bidx2)
    pipe.load(pts2, bidx2)
    out = pipe.replay()
    torch.cuda.synchronize()
    pipe.check()
    assert torch.equal(out["spatial_features"], ref2["spatial_features"])
    f, i = pipe.live(out["encoded_spconv_tensor"])
    assert torch.equal(i, ref2["encoded_spconv_tensor"].indices)
    assert torch.equal(f, ref2["encoded_spconv_tensor"].features)
    pipe.load(pts, bidx)                      # and back to the first batch
    out = pipe.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["spatial_features"], ref["spatial_features"])


def test_two_pipelines_replay_concurrently(dev):
    """Two captured frame pipelines replayed on their own streams at the same time (what
    bench.py does to overlap consecutive batches) do not disturb each other."""
    torch.manual_seed(3)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.SparseBackbone8x(4, grid).eval().to(dev)
    _condition(model)
    nframes = 2
    batches = [_frames_on(dev, nframes, seed0=60 + 10 * i, num_points=9000 - 500 * i) for i in range(2)]
    refs = []
    with torch.no_grad():
        for pts, bidx in batches:
            bd = gb.voxelize_batch(pts, bidx, nframes, K)
            bd = gb.HeightCompression()(model(gb.MeanVFE()(bd)))
            refs.append(bd["spatial_features"].clone())
    pipes, streams = [], []
    for pts, bidx in batches:
        p = gb.StaticFramePipeline(model, K, nframes, 9000 * nframes, 4)
        p.calibrate(pts, bidx)
        p.load(pts, bidx)
        p.capture()
        pipes.append(p)
        streams.append(torch.cuda.Stream(dev))
    torch.cuda.synchronize()
    for it in range(6):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                pipes[i].load(*batches[(i + it) % 2])
                pipes[i].replay()
        torch.cuda.synchronize()
        for i in (0, 1):
            pipes[i].check()
            assert torch.equal(pipes[i].out["spatial_features"], refs[(i + it) % 2]), (it, i)


def test_run_checked_falls_back_when_a_capacity_is_exceeded(dev):
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.SparseBackbone8x(4, grid).eval().to(dev)
    _condition(model)
    small, _ = _frames_on(dev, 1, seed0=81, num_points=1500), None
    big = _frames_on(dev, 1, seed0=82, num_points=9000)
    pipe = gb.StaticFramePipeline(model, K, 1, 9000, 4)
    pipe.calibrate(*small)                       # capacities sized for the small cloud
    pipe.load(*small)
    pipe.capture()
    with torch.no_grad():
        ref = gb.HeightCompression()(model(gb.MeanVFE()(gb.voxelize_batch(*big, 1, K))))["spatial_features"]
    out = pipe.run_checked(*big)                 # overflows -> exact-shape path
    assert torch.equal(out["spatial_features"], ref)
    with torch.no_grad():
        ref_s = gb.HeightCompression()(model(gb.MeanVFE()(gb.voxelize_batch(*small, 1, K))))["spatial_features"]
    out = pipe.run_checked(*small)               # fits -> graph result
    assert torch.equal(out["spatial_features"], ref_s)


def test_static_pipeline_reports_capacity_overflow(dev):
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.SparseBackbone8x(4, grid).eval().to(dev)
    pts, bidx = _frames_on(dev, 1)
    pipe = gb.StaticFramePipeline(model, K, 1, pts.shape[0], 4, capacities={"spconv2": 64})
    pipe.load(pts, bidx)
    pipe.enqueue()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="capacity"):
        pipe.check()
    tiny = dict(K, max_voxels_train=100)      # max_voxels drops cells -> index unusable, reported
    pipe = gb.StaticFramePipeline(model, tiny, 1, pts.shape[0], 4)
    pipe.load(pts, bidx)
    pipe.enqueue()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="dropped cells"):
        pipe.check()


def test_backbone_state_dict_names_match_reference():
    """The parameter names are the ones GLENet's checkpoints use (spconv_backbone.py:77-117)."""
    m = gb.VoxelBackBone8x(4, [1408, 1600, 40])
    keys = set(m.state_dict().keys())
    for k in ["conv_input.0.weight", "conv_input.1.running_mean", "conv1.0.0.weight",
              "conv2.0.0.weight", "conv2.2.1.bias", "conv3.1.0.weight", "conv4.2.0.weight",
              "conv_out.0.weight", "conv_out.1.weight"]:
        assert k in keys, k
    assert tuple(m.state_dict()["conv4.0.0.weight"].shape) == (3, 3, 3, 64, 64)
    assert tuple(m.state_dict()["conv_out.0.weight"].shape) == (3, 1, 1, 64, 128)
    assert m.sparse_shape == [41, 1600, 1408]


def test_device_data_processor_matches_oracle_on_the_same_point_order(dev):
    """mask -> shuffle -> voxelize on the device: the voxels equal the oracle's voxelization of the
    points in the order the device step produced (first-seen semantics depend on it)."""
    from glenet_amd.data_pipeline import DeviceDataProcessor, mask_points_by_range
    frames = [synth.kitti_frame(90 + i, num_points=7000)[0] for i in range(2)]
    frames[0][:50, 0] = -5.0                                       # points outside the x range
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    proc = DeviceDataProcessor(K, training=True, shuffle=True, seed=5)
    bd = proc(pts, bidx, 2)
    p, b = bd["points"].cpu().numpy(), bd["point_batch_idx"].cpu().numpy()
    assert len(p) == int(mask_points_by_range(pts, K["point_cloud_range"]).sum()) and (np.diff(b) >= 0).all()
    assert not np.array_equal(p[b == 0], frames[0][frames[0][:, 0] >= 0][:len(p[b == 0])])   # shuffled
    v, c, n = oracle.voxelize_hard_batch([p[b == i] for i in range(2)], K["voxel_size"],
                                         K["point_cloud_range"], 5, 16000)
    assert np.array_equal(bd["voxel_coords"].cpu().numpy(), c)
    assert np.array_equal(bd["voxels"].cpu().numpy(), v) and np.array_equal(bd["voxel_num_points"].cpu().numpy(), n)


def test_static_device_data_step_inside_a_recorded_training_step(dev):
    """SURVEY 8f rank 1 as a capturable stage: range mask + per-frame shuffle on capacity-sized buffers (no boolean
    indexing, no read-back), then the voxelizer -- (i) the reference recipe on the SAME permutation: numpy mask
    (common_utils.py:60-63), `points[shuffle_idx]` (data_processor.py:95-105) with the permutation the device drew,
    the oracle's hard voxelizer -> identical voxels; (ii) inside a recorded StaticTrainPipeline the replays draw a new
    permutation every time and the step's own voxels still equal the oracle on that replay's order."""
    from glenet_amd import data_pipeline as dpl
    frames = [synth.kitti_frame(70 + i, num_points=6000)[0] for i in range(2)]
    frames[0][:40, 0] = -5.0                                       # outside the x range
    frames[1][:25, 1] = 41.0                                       # outside the y range
    frames[1][25, 1] = 40.0                                        # ON the closed boundary: kept
    pts = np.concatenate(frames)
    bid = np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])
    cap = len(pts) + 300                                           # padding rows: frame id == batch size
    P = torch.zeros((cap, 4), device=dev)
    Bx = torch.full((cap,), 2, dtype=torch.int32, device=dev)
    P[:len(pts)] = torch.from_numpy(pts).to(dev)
    Bx[:len(pts)] = torch.from_numpy(bid).to(dev)
    rng = K["point_cloud_range"]

    def reference(order):
        """the reference's recipe per frame with the device's permutation"""
        order = order.cpu().numpy()
        out = []
        for f in range(2):
            src = frames[f]
            keep = (src[:, 0] >= rng[0]) & (src[:, 0] <= rng[3]) & (src[:, 1] >= rng[1]) & (src[:, 1] <= rng[4])
            kept_rows = np.flatnonzero(keep) + (0 if f == 0 else len(frames[0]))
            mine = [r for r in order if (r < len(pts)) and bid[r] == f and keep[r - (0 if f == 0 else len(frames[0]))]]
            assert sorted(mine) == sorted(kept_rows.tolist())          # a permutation of exactly the kept points
            masked = src[keep]
            pos = {r: i for i, r in enumerate(kept_rows)}
            shuffle_idx = np.array([pos[r] for r in mine])            # what np.random.permutation would have returned
            out.append(masked[shuffle_idx])
        return out

    state = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
    op, ob_, order = dpl.mask_and_shuffle_static(P, Bx, 2, rng, state)
    assert state.tolist() == [1234, 1]                             # the call counter moved: the next call draws anew
    op2, _, order2 = dpl.mask_and_shuffle_static(P, Bx, 2, rng, state)
    assert not torch.equal(order, order2) and torch.equal(order.sort()[0], order2.sort()[0])
    again = dpl.mask_and_shuffle_static(P, Bx, 2, rng, torch.tensor([1234, 0], dtype=torch.int64, device=dev))[2]
    assert torch.equal(order, again)                               # reproducible from (seed, call number)
    b = ob_.cpu().numpy()
    nkeep = int((b < 2).sum())
    assert (np.diff(b) >= 0).all() and (b[nkeep:] == 2).all() and nkeep == len(pts) - 65
    assert (order[nkeep:] == -1).all() and float(op[nkeep:].abs().max()) == 0.0
    # not a trivial permutation: few fixed points, no long monotone runs
    o0 = order[:nkeep].cpu().numpy()
    f0 = o0[b[:nkeep] == 0]
    assert (f0 == np.sort(f0)).mean() < 0.01 and (np.diff(f0) == 1).mean() < 0.01
    ref_frames = reference(order[:nkeep])
    assert np.array_equal(op.cpu().numpy()[:nkeep], np.concatenate(ref_frames))
    assert not np.array_equal(ref_frames[0][:200], frames[0][40:240])                  # shuffled
    v, c, n = oracle.voxelize_hard_batch(ref_frames, K["voxel_size"], rng, 5, 16000)
    bd = gb.voxelize_batch(op, ob_, 2, K, train=True, static=True)
    nv = int(bd["voxel_index"].count.item())
    assert nv == len(c) and np.array_equal(bd["voxel_coords"].cpu().numpy()[:nv], c)
    assert np.array_equal(bd["voxels"].cpu().numpy()[:nv], v)
    # without shuffling only the mask acts: the order inside a frame is kept
    proc = dpl.DeviceDataProcessor(K, shuffle=False)
    kp, kb = proc.static_step(P, Bx, 2)
    keep_np = (pts[:, 0] >= rng[0]) & (pts[:, 0] <= rng[3]) & (pts[:, 1] >= rng[1]) & (pts[:, 1] <= rng[4])
    assert np.array_equal(kp.cpu().numpy()[:nkeep], pts[keep_np]) and np.array_equal(kb.cpu().numpy()[:nkeep], bid[keep_np])
    # ---- inside a recorded training step
    grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    torch.manual_seed(0)
    model = gb.VoxelBackBone8x(4, grid).to(dev).train()
    pipe = gb.StaticTrainPipeline(model, K, 2, cap, 4)
    pipe.data_step = dpl.DeviceDataProcessor(K, shuffle=True)
    pipe.calibrate(torch.from_numpy(pts).to(dev), torch.from_numpy(bid).to(dev))
    pipe.load(torch.from_numpy(pts).to(dev), torch.from_numpy(bid).to(dev))
    pipe.capture()
    seen = []
    for _ in range(2):
        pipe.replay()
        torch.cuda.synchronize()
        pipe.check()
        nv2 = int(pipe.out["voxel_index"].count.item())
        assert abs(nv2 - nv) <= 0.02 * nv                            # same cells up to the max_points truncation order
        seen.append(pipe.out["voxels"][:nv2].clone())
        assert np.isfinite(float(pipe.loss))
    assert seen[0].shape != seen[1].shape or not torch.equal(seen[0], seen[1])   # a fresh permutation per replay
    _drop_graphs(pipe)


def test_waymo_shaped_residual_backbone_matches_oracle(dev):
    """Config-5 shape: 5 point features (zero-padded to 8 input channels on the MFMA path), Waymo
    grid [41,1504,1504], VoxelResBackBone8x (biased residual blocks), one frame of 30 k points."""
    W = synth.WAYMO
    torch.manual_seed(6)
    grid = oracle.grid_size_of(W["point_cloud_range"], W["voxel_size"])
    model = gb.SparseBackbone8x(5, grid, residual=True).eval()
    _condition(model)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    frame = synth.waymo_frame(3, num_points=30000)[0]
    v, c, n = oracle.voxelize_hard_batch([frame], W["voxel_size"], W["point_cloud_range"], 5, 150000)
    ref = ob.backbone_forward(sd, oracle.mean_vfe(v, n), c, model.sparse_shape, residual=True)
    model = model.to(dev)
    pts = torch.from_numpy(frame).to(dev)
    bidx = torch.zeros(len(frame), dtype=torch.int32, device=dev)
    with torch.no_grad():
        bd = gb.voxelize_batch(pts, bidx, 1, W, train=False)
        assert np.array_equal(bd["voxel_coords"].cpu().numpy(), c)
        bd = model(gb.MeanVFE()(bd))
    out = bd["encoded_spconv_tensor"]
    assert np.array_equal(out.indices.cpu().numpy(), ref["out"].indices)
    r = ref["out"].features
    np.testing.assert_allclose(out.features.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(r).max()))


def test_static_pipeline_handles_empty_frames(dev):
    """A frame without a single in-range point, and a whole empty batch, through the captured graph."""
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.SparseBackbone8x(4, grid).eval().to(dev)
    _condition(model)
    pts, bidx = _frames_on(dev, 2, seed0=95, num_points=4000)
    pipe = gb.StaticFramePipeline(model, K, 2, pts.shape[0], 4)
    pipe.load(pts, bidx)
    pipe.capture()
    half = pts.clone()
    half[bidx == 1, 0] = -100.0                      # frame 1 entirely out of range
    with torch.no_grad():
        ref = gb.HeightCompression()(model(gb.MeanVFE()(gb.voxelize_batch(half, bidx, 2, K))))["spatial_features"]
    pipe.load(half, bidx)
    out = pipe.replay()
    torch.cuda.synchronize()
    pipe.check()
    assert torch.equal(out["spatial_features"], ref) and float(ref[1].abs().max()) == 0.0
    none = pts.clone()
    none[:, 0] = -100.0
    pipe.load(none, bidx)
    out = pipe.replay()
    torch.cuda.synchronize()
    pipe.check()
    assert int(out["voxel_index"].count.item()) == 0 and float(out["spatial_features"].abs().max()) == 0.0
    pipe.load(pts, bidx)                              # and it recovers
    out = pipe.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = gb.HeightCompression()(model(gb.MeanVFE()(gb.voxelize_batch(pts, bidx, 2, K))))["spatial_features"]
    assert torch.equal(out["spatial_features"], ref)


@pytest.mark.parametrize("residual", [False, True])
def test_static_training_step_matches_exact_shape_autograd(dev, residual):
    """StaticTrainPipeline (capacity-sized buffers, device row counts, no read-back) enqueued and
    replayed as a HIP graph: loss, every parameter gradient and the BatchNorm running statistics
    equal the exact-shape autograd path (same kernels; the weight-gradient slices and nothing else
    depend on the buffer size -> 1e-5 relative), on a second batch through the recorded graph too."""
    torch.manual_seed(3)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    B = 2

    def batch(seed, npts):
        frames = [synth.kitti_frame(seed + i, num_points=npts)[0] for i in range(B)]
        pts = torch.from_numpy(np.concatenate(frames)).to(dev)
        bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
        return pts, bidx

    model = gb.SparseBackbone8x(4, grid, residual=residual).to(dev).train()
    _condition(model)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    loss_fn = lambda bd: (bd["spatial_features"] * 3.0).square().mean()   # noqa: E731

    def exact(pts, bidx):
        model.load_state_dict(state0)
        model.zero_grad(set_to_none=True)
        bd = gb.HeightCompression()(model(gb.MeanVFE()(gb.voxelize_batch(pts, bidx, B, K, train=True))))
        loss = loss_fn(bd)
        loss.backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        stats = {n: b.clone() for n, b in model.named_buffers()}
        return loss.detach().clone(), grads, stats

    def compare(pipe, ref):
        loss, grads, stats = ref
        torch.cuda.synchronize()
        pipe.check()
        np.testing.assert_allclose(float(pipe.loss.detach()), float(loss), rtol=1e-5)
        # absolute floor from the largest gradient of the model: a conv bias in front of a
        # BatchNorm has an exactly-zero gradient, what both paths return there is rounding noise
        gmax = max(float(v.abs().max()) for v in grads.values())
        for n, p in model.named_parameters():
            g, w = p.grad.cpu().numpy(), grads[n].cpu().numpy()
            assert np.isfinite(g).all(), n
            np.testing.assert_allclose(g, w, rtol=2e-4, atol=2e-5 * max(gmax * 1e-2, np.abs(w).max()), err_msg=n)
        for n, b in model.named_buffers():
            np.testing.assert_allclose(b.cpu().numpy(), stats[n].cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg=n)

    a, b2 = batch(20, 9000), batch(60, 7000)
    ref_a, ref_b = exact(*a), exact(*b2)
    assert float(ref_a[0]) > 0 and max(float(g.abs().max()) for g in ref_a[1].values()) > 0

    pipe = gb.StaticTrainPipeline(model, K, B, a[0].shape[0] + 1000, 4, loss_fn=loss_fn)
    pipe.calibrate(*a)
    model.load_state_dict(state0)
    pipe.load(*a)
    pipe.enqueue()
    compare(pipe, ref_a)

    model.load_state_dict(state0)
    pipe.capture(warmup=1)
    model.load_state_dict(state0)
    pipe.load(*b2)
    pipe.replay()
    compare(pipe, ref_b)
    model.load_state_dict(state0)
    pipe.load(*a)
    pipe.replay()
    compare(pipe, ref_a)


def test_training_graph_with_optimizer_follows_eager_training(dev):
    """Five Adam steps: exact-shape eager loop vs StaticTrainPipeline with the (capturable, foreach)
    optimizer inside the HIP graph -- same loss trajectory (1e-4 relative), i.e. forward, backward,
    gradient hand-over and parameter update all happen correctly inside the replayed graph."""
    import copy
    torch.manual_seed(0)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    B = 2
    frames = [synth.kitti_frame(20 + i, num_points=9000)[0] for i in range(B)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    base = gb.VoxelBackBone8x(4, grid).to(dev).train()
    steps = 5

    def eager():
        m = copy.deepcopy(base)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        out = []
        for _ in range(steps):
            bd = gb.HeightCompression()(m(gb.MeanVFE()(gb.voxelize_batch(pts, bidx, B, K, train=True))))
            loss = bd["spatial_features"].square().mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
            del bd, loss
        return out

    want = eager()
    assert want[-1] < 0.5 * want[0]                        # it does train
    m = copy.deepcopy(base)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, foreach=True)
    pipe = gb.StaticTrainPipeline(m, K, B, pts.shape[0] + 500, 4, optimizer=opt)
    pipe.calibrate(pts, bidx)
    pipe.load(pts, bidx)
    pipe.capture(warmup=2)                                 # steps 0 and 1 run eagerly while warming up
    got = []
    for _ in range(steps - 2):
        pipe.replay()
        torch.cuda.synchronize()
        got.append(float(pipe.loss.detach()))
    pipe.check()
    np.testing.assert_allclose(got, want[2:], rtol=1e-4)


def test_tile_maps_do_not_change_results(dev):
    """Work-balanced block -> tile maps (RuleSet.tile_map) at the bench size, where they switch on:
    every map is a permutation of the tiles, and the backbone output -- exact-shape path and the
    shape-static path with device row counts -- is bit-identical with and without them."""
    from glenet_amd.spconv import core
    torch.manual_seed(0)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    B = 4
    frames = [synth.kitti_frame(i)[0] for i in range(B)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    model = gb.VoxelBackBone8x(4, grid).to(dev).eval()
    _condition(model)

    def run(static):
        with torch.no_grad():
            if static:
                pipe = gb.StaticFramePipeline(model, K, B, pts.shape[0], 4)
                pipe.calibrate(pts, bidx)
                pipe.load(pts, bidx)
                bd = pipe.enqueue()
                torch.cuda.synchronize()
                pipe.check()
            else:
                bd = gb.HeightCompression()(model(gb.MeanVFE()(gb.voxelize_batch(pts, bidx, B, K))))
        return bd

    assert core.USE_TILE_MAP
    with_maps = run(False)
    maps = [m for rs in with_maps["multi_scale_3d_features"]["x_conv4"].indice_dict.values() for m in rs._tile_maps.values()]
    assert len(maps) >= 3
    for m in maps:
        assert sorted(m.cpu().tolist()) == list(range(m.numel()))
        assert not torch.equal(m.cpu(), torch.arange(m.numel(), dtype=torch.int32))
    static_maps = run(True)
    core.USE_TILE_MAP = False
    try:
        plain, static_plain = run(False), run(True)
    finally:
        core.USE_TILE_MAP = True
    assert torch.equal(with_maps["spatial_features"], plain["spatial_features"])
    assert torch.equal(static_maps["spatial_features"], static_plain["spatial_features"])
    assert torch.equal(static_maps["spatial_features"], plain["spatial_features"])


def test_graphs_follow_weight_changes_made_outside_them(dev):
    """packed sparse-conv weights and folded BatchNorms are caches keyed by tensor version.
    (1) A training graph captured WITHOUT an optimizer, then an eager parameter update between replays: the next
    replay must compute with the new weights in forward AND backward (weights are packed inside the step).
    (2) An inference graph, then load_state_dict(): the next replay records the frame again (version tag)."""
    torch.manual_seed(0)
    frames = [synth.kitti_frame(30 + i, num_points=6000)[0] for i in range(2)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.VoxelBackBone8x(4, grid).to(dev).train()
    pipe = gb.StaticTrainPipeline(model, K, 2, pts.shape[0], 4)
    pipe.calibrate(pts, bidx)
    pipe.load(pts, bidx)
    pipe.capture()
    pipe.replay()
    torch.cuda.synchronize()
    loss0 = float(pipe.loss)
    with torch.no_grad():                               # an eager "optimizer step" outside the graph
        for p in model.parameters():
            if p.dim() == 5:
                p.mul_(1.5)
    pipe.replay()
    torch.cuda.synchronize()
    loss1 = float(pipe.loss)
    g_graph = {n: p.grad.clone() for n, p in model.named_parameters()}
    pipe.graph = None
    pipe.enqueue()                                      # the same step, eager launches, same weights
    torch.cuda.synchronize()
    assert abs(float(pipe.loss) - loss1) <= 1e-6 * abs(loss1) and abs(loss1 - loss0) > 1e-3 * abs(loss0)
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, g_graph[n], rtol=1e-5, atol=1e-7), n
    # ---- inference
    model.eval()
    ipipe = gb.StaticFramePipeline(model, K, 2, pts.shape[0], 4)
    ipipe.calibrate(pts, bidx)
    ipipe.load(pts, bidx)
    ipipe.capture()
    a = ipipe.replay()["spatial_features"].clone()
    other = gb.VoxelBackBone8x(4, grid).to(dev).eval()
    model.load_state_dict(other.state_dict())
    b = ipipe.replay()["spatial_features"].clone()
    with torch.no_grad():
        bd = gb.voxelize_batch(pts, bidx, 2, K, train=True)
        want = gb.HeightCompression()(model(gb.MeanVFE()(bd)))["spatial_features"]
    torch.cuda.synchronize()
    assert not torch.allclose(a, b) and torch.allclose(b, want, rtol=1e-5, atol=1e-6)


def test_eval_after_fused_training_steps_sees_the_new_weights(dev):
    """FlatAdamW and the fused training BatchNorm update parameters / running statistics through
    raw pointers, and a replayed training graph runs no Python -- no tensor version moves.  eval -> train steps ->
    eval on the SAME module (and an inference graph sharing it) must compute with the updated weights: every
    derived-tensor cache also keys on glenet_amd._lib.weights_epoch()."""
    import copy
    from glenet_amd import _lib
    from glenet_amd.optim import FlatAdamW
    torch.manual_seed(0)
    frames = [synth.kitti_frame(40 + i, num_points=6000)[0] for i in range(2)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.VoxelBackBone8x(4, grid).to(dev)

    def evaluate(m):
        m.eval()
        with torch.no_grad():
            bd = gb.voxelize_batch(pts, bidx, 2, K, train=True)
            return gb.HeightCompression()(m(gb.MeanVFE()(bd)))["spatial_features"].clone()

    y0 = evaluate(model)                                   # fills the packed-weight and BatchNorm-affine caches
    ipipe = gb.StaticFramePipeline(model, K, 2, pts.shape[0], 4)
    ipipe.calibrate(pts, bidx)
    ipipe.load(pts, bidx)
    ipipe.capture()
    assert torch.equal(ipipe.replay()["spatial_features"], y0)
    # ---- training steps as one replayed graph with the fused optimizer inside
    model.train()
    opt = FlatAdamW(model.parameters(), lr=5e-3)
    versions = [p._version for p in model.parameters()] + [b._version for b in model.buffers()]
    pipe = gb.StaticTrainPipeline(model, K, 2, pts.shape[0], 4, optimizer=opt)
    pipe.calibrate(pts, bidx)
    pipe.load(pts, bidx)
    pipe.capture()
    e0, g0 = _lib.weights_epoch(*model.parameters()), _lib.weights_epoch()
    # an unrelated model that never trains: its inference graph must NOT record itself again (the epoch is
    # scoped to the tensors that were written, not process-global)
    other = gb.VoxelBackBone8x(4, grid).to(dev).eval()
    opipe = gb.StaticFramePipeline(other, K, 2, pts.shape[0], 4)
    opipe.calibrate(pts, bidx)
    opipe.load(pts, bidx)
    opipe.capture()
    other_graph, other_tag = opipe.graph, opipe._weights_tag()
    for _ in range(3):
        pipe.replay()
    torch.cuda.synchronize()
    assert _lib.weights_epoch(*model.parameters()) >= e0 + 3 and _lib.weights_epoch() == g0
    opipe.replay()
    assert opipe.graph is other_graph and opipe._weights_tag() == other_tag
    # ---- eval again: same module object, same tensors, versions of the PARAMETERS untouched by the replays
    y1 = evaluate(model)
    fresh = gb.VoxelBackBone8x(4, grid).to(dev)
    fresh.load_state_dict(copy.deepcopy(model.state_dict()))
    y_ref = evaluate(fresh)                                # caches built from scratch on the current weights
    assert torch.equal(y1, y_ref)
    assert float((y1 - y0).abs().max()) > 1e-3
    assert torch.equal(ipipe.replay()["spatial_features"], y_ref)     # the inference graph recorded itself again
    assert [p._version for p in model.parameters()] == versions[:len(list(model.parameters()))]
    _drop_graphs(pipe, ipipe, opipe)


@pytest.mark.gpu
@pytest.mark.parametrize("ntiles", [6000, 16384, 17000])
def test_tile_map_of_large_rule_tables(dev, ntiles):
    """glx_sconv_tile_map beyond the 4096 tiles of round 3 (max_voxels Waymo x batch > 2 exceeded it and the
    entry point failed): up to 16384 tiles (1 M output rows) one block sorts them in dynamic LDS -- a permutation of the
    tiles that differs from the identity --, beyond that the identity map; never an error."""
    from glenet_amd import _lib
    n_out, Kk = ntiles * 64, 27
    g = torch.Generator(device=dev).manual_seed(ntiles)
    nbr = torch.randint(0, n_out, (n_out, Kk), device=dev, dtype=torch.int32, generator=g)
    # uneven work: the share of present neighbours grows along the rows
    drop = torch.rand((n_out, Kk), device=dev, generator=g) > torch.linspace(0.05, 0.9, n_out, device=dev)[:, None]
    nbr = torch.where(drop, torch.full_like(nbr, -1), nbr).contiguous()
    tmap = torch.full((ntiles,), -7, dtype=torch.int32, device=dev)
    wsb = _lib.query("glx_sconv_tile_map_workspace_bytes", n_out)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.call("glx_sconv_tile_map", nbr, None, n_out, Kk, None, tmap, ws, _lib.size_arg(wsb))
    got = tmap.cpu().numpy()
    assert np.array_equal(np.sort(got), np.arange(ntiles))
    assert np.array_equal(got, np.arange(ntiles)) == (ntiles > 16384)


def test_batchnorm_on_load_in_the_next_sparse_convolution_changes_nothing(dev):
    """Inner layers of the blocks (spconv_backbone.py:77-117 post_act_block chains): relu(bn(y)) is not written when the
    next convolution consumes it -- that convolution and its weight gradient transform y on load
    (glx_sconv_opts.prologue, glx_sconv_wgrad_pairs_ex) and its backward carries the BatchNorm's.  Same arithmetic: the
    outputs and the running statistics equal the materialising path bit for bit, the gradients to rounding (a layer whose
    BatchNorm backward took the two-launch statistics path there takes the epilogue sums here), and the launches that
    would write the inner activations are gone."""
    from glenet_amd.spconv import core
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    pts, bidx = _frames_on(dev, 2, num_points=12000)

    def run(on_load):
        torch.manual_seed(3)
        model = gb.VoxelBackBone8x(4, grid).to(dev).train()
        _condition(model)
        core.BN_ON_LOAD = on_load
        applied = []
        real = core.FusedBNApply.apply
        core.FusedBNApply.apply = staticmethod(lambda *a, **k: (applied.append(1), real(*a, **k))[1])
        try:
            bd = model(gb.MeanVFE()(gb.voxelize_batch(pts, bidx, 2, K)))
            out = bd["encoded_spconv_tensor"].features
            feats = [bd["multi_scale_3d_features"][k].features for k in ("x_conv1", "x_conv2", "x_conv3", "x_conv4")]
            g = torch.Generator(dev).manual_seed(1)
            loss = sum((f * torch.randn(f.shape, device=dev, generator=g)).sum() for f in feats + [out])
            loss.backward()
        finally:
            core.FusedBNApply.apply = real
            core.BN_ON_LOAD = True
        grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        bufs = {n: b.clone() for n, b in model.named_buffers()}
        return [out] + feats, grads, bufs, len(applied)

    outs1, grads1, bufs1, n1 = run(True)
    outs0, grads0, bufs0, n0 = run(False)
    assert n0 == 12 and n1 == 5, (n0, n1)           # the block outputs (x_conv1..4, conv_out) are still written
    for a, b in zip(outs1, outs0):
        assert torch.equal(a, b)
    for n in grads0:
        scale = float(grads0[n].abs().max()) + 1e-12
        assert float((grads1[n] - grads0[n]).abs().max()) <= 2e-5 * scale, n
    for n in bufs0:
        assert torch.equal(bufs1[n], bufs0[n]), n
