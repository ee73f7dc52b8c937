"""Harness of the two-stage data flow: pure-torch pieces on CPU, the full flow on GPU."""
import numpy as np
import pytest
import torch

from glenet_amd import detector as det
from glenet_amd import synth


def test_anchor_layout_and_box_decode_cpu():
    a = det.generate_anchors([0, -40, -3, 70.4, 40, 1], (176, 200), [[3.9, 1.6, 1.56]], [0, 1.57], [-1.78])
    assert a.shape == (1, 200, 176, 1, 2, 7)                       # 70 400 anchors (GLENet_VR.yaml:63-73)
    np.testing.assert_allclose(a[0, 0, 0, 0, 0].numpy(), [0, -40, -1.78 + 0.78, 3.9, 1.6, 1.56, 0], atol=1e-6)
    np.testing.assert_allclose(a[0, -1, -1, 0, 1, :2].numpy(), [70.4, 40], atol=1e-4)
    assert float(a[0, 0, 1, 0, 0, 0]) == pytest.approx(70.4 / 175)
    enc = torch.zeros(3, 7)
    anc = a.reshape(-1, 7)[:3]
    assert torch.equal(det.decode_boxes(enc, anc), anc)             # zero residual = the anchor
    enc[0] = torch.tensor([0.1, -0.2, 0.5, np.log(2.0), 0.0, np.log(0.5), 0.3])
    d = det.decode_boxes(enc, anc)[0]
    diag = np.sqrt(3.9 ** 2 + 1.6 ** 2)
    np.testing.assert_allclose(d.numpy(), [anc[0, 0] + 0.1 * diag, anc[0, 1] - 0.2 * diag, anc[0, 2] + 0.5 * 1.56,
                                           7.8, 1.6, 0.78, 0.3], rtol=1e-6, atol=1e-6)
    rois = torch.tensor([[[10.0, 2.0, -1.0, 4.0, 2.0, 1.5, np.pi / 2]]])
    out = det.refine_boxes(rois, torch.zeros(1, 7))
    np.testing.assert_allclose(out[0, 0].numpy(), rois[0, 0].numpy(), atol=1e-6)


@pytest.mark.gpu
def test_voxel_rcnn_flow_runs_on_the_kernels(dev):
    torch.manual_seed(0)
    K = synth.KITTI
    frames = [synth.kitti_frame(70 + i, num_points=6000)[0] for i in range(2)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    flow = det.VoxelRCNNFlow(K).to(dev).eval()
    with torch.no_grad():
        out = flow(pts, bidx, 2)
    assert out["spatial_features_2d"].shape == (2, 256, 200, 176)
    assert out["rois"].shape == (2, 100, 7) and out["batch_box_preds"].shape == (2, 100, 7)
    assert out["batch_cls_preds"].shape == (2, 100, 1)
    assert torch.isfinite(out["batch_box_preds"]).all() and torch.isfinite(out["batch_cls_preds"]).all()
    # proposals are padded with zeros behind the kept ones and carry 1-based labels
    n_kept = (out["rois"].abs().sum(-1) > 0).sum(1)
    assert (n_kept > 0).all() and (out["roi_labels"] == 1).all()


@pytest.mark.gpu
def test_batched_proposal_layer_equals_per_frame_loop(dev):
    """The sync-free batched proposal path (one batched NMS launch sequence stopping at
    NMS_POST_MAXSIZE) returns exactly what the per-frame loop does; one frame keeps fewer boxes than
    the padding size."""
    rng = np.random.default_rng(12)
    B, A = 3, 5000
    boxes = np.stack([synth.random_boxes(rng, A, xy_range=[30.0, 30.0, 2.0][b], near_dup=0.5) for b in range(B)])
    logits = rng.normal(size=(B, A, 2)).astype(np.float32) * 3
    bx, lg = torch.from_numpy(boxes).to(dev), torch.from_numpy(logits).to(dev)
    with torch.no_grad():
        det.BATCHED_PROPOSALS = True
        a = det.proposal_layer(bx, lg, 1024, 100, 0.3)
        det.BATCHED_PROPOSALS = False
        try:
            b = det.proposal_layer(bx, lg, 1024, 100, 0.3)
        finally:
            det.BATCHED_PROPOSALS = True
    for u, v in zip(a, b):
        assert u.shape == v.shape and u.dtype == v.dtype and torch.equal(u, v)
    kept = (a[0].abs().sum(-1) > 0).sum(1)
    assert int(kept.min()) < 100 and int(kept.max()) == 100


@pytest.mark.gpu
def test_roi_fc_stack_folded_inference_path(dev):
    """RoIFCStack on the device in eval mode (BatchNorm folded into the Linear layers, first layer
    split along K) vs its own module-by-module path; fp32 sums of 20 736 terms in a different
    order: tolerance 2e-4 of the output scale."""
    from glenet_amd import dense_path as dp
    torch.manual_seed(5)
    fc = dp.RoIFCStack(96, 6).to(dev)
    for m in fc.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    fc.eval()
    x = torch.randn(300, 216, 96, device=dev)
    with torch.no_grad():
        got = fc(x)
        fc.USE_FUSED = False
        want = fc(x)
    for g, w in zip(got, want):
        assert g.shape == w.shape
        scale = float(w.abs().max())
        assert float((g - w).abs().max()) <= 2e-4 * scale


@pytest.mark.gpu
def test_static_detector_pipeline_and_graph_match_the_eager_flow(dev):
    """StaticDetectorPipeline (capacity-sized buffers, no host sync) enqueued directly and replayed
    as a HIP graph reproduces the exact-shape eager flow: proposals identical, refined boxes and
    scores to 1e-4 (same kernels; only buffer sizes differ)."""
    torch.manual_seed(0)
    K = synth.KITTI
    B = 2
    frames = [synth.kitti_frame(40 + i, num_points=8000)[0] for i in range(B)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    flow = det.VoxelRCNNFlow(K).to(dev).eval()
    with torch.no_grad():
        want = flow(pts, bidx, B)
    pipe = det.StaticDetectorPipeline(flow, B, pts.shape[0] + 500)
    # the recorded flow hands the BEV backbone what the eager flow does (channels-last, deferred to the sparse first layer):
    # with the base class's own HeightCompression the map arrived NCHW and the backbone ran on the vendor's kernels
    assert pipe.hc is flow.map_to_bev
    pipe.calibrate(pts, bidx)
    pipe.load(pts, bidx)
    got = pipe.enqueue()
    torch.cuda.synchronize()
    pipe.check()

    def same(a, b):
        assert torch.equal(a["rois"], b["rois"]) and torch.equal(a["roi_labels"], b["roi_labels"])
        for k in ("batch_box_preds", "batch_cls_preds"):
            np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=1e-4, atol=1e-4)

    same(got, want)
    pipe.capture()
    frames2 = [synth.kitti_frame(90 + i, num_points=7000)[0] for i in range(B)]
    pts2 = torch.from_numpy(np.concatenate(frames2)).to(dev)
    bidx2 = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames2)])).to(dev)
    with torch.no_grad():
        want2 = flow(pts2, bidx2, B)
    out = pipe.run_checked(pts2, bidx2)          # other points through the recorded graph
    same(out, want2)
    out = pipe.run_checked(pts, bidx)
    same(out, want)


# ------------------------------------------------------------------ reference-generated goldens
def _glue():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detector_glue_ref.npz"))


def test_anchor_generator_matches_reference_golden():
    """generate_anchors == AnchorGenerator.generate_anchors of the reference (fixture generated
    from /root/reference, tests/golden/make_golden.py), end-to-end and centre-aligned layouts."""
    g = _glue()
    rng = [0, -40.0, -3, 70.4, 40.0, 1]
    a = det.generate_anchors(rng, (22, 25), [[3.9, 1.6, 1.56]], [0, 1.57], [-1.78])
    assert a.shape == g["anchors_car"].shape and np.array_equal(a.numpy(), g["anchors_car"])
    b = det.generate_anchors(rng, (11, 13), [[0.8, 0.6, 1.73]], [0, 1.57], [-0.6], align_center=True)
    assert np.array_equal(b.numpy(), g["anchors_ped_aligned"])
    assert int(g["anchors_per_location"][0]) == 2


def test_box_decode_matches_reference_golden():
    """decode_boxes == ResidualCoder.decode_torch (extra code channel included); predicted_boxes ==
    AnchorHeadTemplate.generate_predicted_boxes' statement sequence (direction classifier);
    refine_boxes == RoIHeadTemplate.generate_predicted_boxes.  north_star tolerance 1e-4 on box
    regressions; the first two are the same arithmetic in the same order -> exact."""
    g = _glue()
    out = det.decode_boxes(torch.from_numpy(g["dec_enc"]), torch.from_numpy(g["dec_anchors"]))
    # the same arithmetic in the same order; torch's CPU exp / sin are vectorised per instruction set (an ulp between machines)
    np.testing.assert_allclose(out.numpy(), g["dec_out"], rtol=1e-6, atol=1e-6)
    anchors = torch.from_numpy(g["anchors_car"])
    cls = torch.zeros(2, 25, 22, 2)
    _, boxes = det.predicted_boxes(cls, torch.from_numpy(g["head_box_preds"]), torch.from_numpy(g["head_dir_preds"]),
                                   anchors)
    np.testing.assert_allclose(boxes.numpy(), g["head_boxes"], rtol=0, atol=1e-6)
    ref = det.refine_boxes(torch.from_numpy(g["roi_rois"]), torch.from_numpy(g["roi_reg"]))
    np.testing.assert_allclose(ref.numpy(), g["roi_boxes"], rtol=1e-6, atol=1e-5)


def test_capturable_voxel_centres_equal_the_reference_formulation_cpu():
    """roi_grid._voxel_centers_capturable (no host->device copies, recordable into a HIP graph) is bit-equal to
    get_voxel_centers (pinned by the reference golden in test_dense_path_cpu.py) at every stride."""
    from glenet_amd import roi_grid as rg
    g = torch.Generator().manual_seed(0)
    for cfg in (synth.KITTI, synth.WAYMO):
        c = torch.randint(0, 1600, (4000, 3), dtype=torch.int32, generator=g)
        for stride in (1, 2, 4, 8):
            a = rg.get_voxel_centers(c, stride, cfg["voxel_size"], cfg["point_cloud_range"])
            b = rg._voxel_centers_capturable(c, stride, cfg["voxel_size"], cfg["point_cloud_range"])
            assert torch.equal(a, b)


@pytest.mark.gpu
def test_voxel_centres_kernel_equals_the_reference_formulation(dev):
    """glx_voxel_centers (one launch, recordable into a HIP graph) is bit-equal to get_voxel_centers
    (common_utils.py:66-82, pinned by the reference golden in test_dense_path_cpu.py) at every stride."""
    import ctypes
    from glenet_amd import _lib, roi_grid as rg
    g = torch.Generator().manual_seed(1)
    f3 = ctypes.c_float * 3
    for cfg in (synth.KITTI, synth.WAYMO):
        ind = torch.randint(0, 1600, (5003, 4), dtype=torch.int32, generator=g)
        for stride in (1, 2, 4, 8):
            want = rg.get_voxel_centers(ind[:, 1:4], stride, cfg["voxel_size"], cfg["point_cloud_range"])
            got = torch.full((len(ind), 3), float("nan"), device=dev)
            _lib.call("glx_voxel_centers", ind.to(dev), len(ind), stride, f3(*cfg["point_cloud_range"][:3]),
                      f3(*cfg["voxel_size"]), got)
            assert torch.equal(got.cpu(), want)


@pytest.mark.gpu
def test_predicted_boxes_kernel_matches_reference_golden_and_tensor_ops(dev):
    """glx_predicted_boxes (decode + direction bin in one launch) against the golden generated by the reference's
    AnchorHeadTemplate.generate_predicted_boxes, and bit for bit against the tensor-op formulation on the device
    (same rounding by construction) at the KITTI head size, without and with the direction classifier."""
    g = _glue()
    anchors = torch.from_numpy(g["anchors_car"]).to(dev)
    cls = torch.zeros(2, 25, 22, 2, device=dev)
    with torch.no_grad():
        _, boxes = det.predicted_boxes(cls, torch.from_numpy(g["head_box_preds"]).to(dev),
                                       torch.from_numpy(g["head_dir_preds"]).to(dev), anchors)
    np.testing.assert_allclose(boxes.cpu().numpy(), g["head_boxes"], rtol=0, atol=1e-6)
    gen = torch.Generator().manual_seed(2)
    a = det.generate_anchors([0, -40, -3, 70.4, 40, 1], (200, 176), [[3.9, 1.6, 1.56]], [0, 1.57], [-1.78]).to(dev)
    n = a.reshape(-1, 7).shape[0]
    bp = (torch.randn(2, n, 7, generator=gen) * 0.6).to(dev)
    dirp = torch.randn(2, n, 2, generator=gen).to(dev)
    dirp[0, :100, 1] = dirp[0, :100, 0]                           # ties -> bin 0, as torch's max
    cls = torch.zeros(2, n, 1, device=dev)
    for d in (None, dirp):
        with torch.no_grad():
            _, got = det.predicted_boxes(cls, bp, d, a)
            det.FUSED_PREDICTED_BOXES = False
            try:
                _, want = det.predicted_boxes(cls, bp, d, a)
            finally:
                det.FUSED_PREDICTED_BOXES = True
        assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("A,K", [(70400, 9000), (70400, 1024), (5000, 4096), (300, 300), (64, 1), (10240, 10240)])
def test_topk_kernel_matches_a_stable_sort(dev, A, K):
    """detector.topk_desc (glx_topk_desc: one block per frame, radix select + stable LSD radix sort in LDS; from 4096
    scores per frame on glx_topk_desc_ws: 32 cooperating blocks per frame select, one block per frame sorts) against
    numpy's stable sort of the negated scores: values, indices, ties by ascending index -- on smooth scores (shared
    high bytes, as sigmoid outputs of one frame have), on heavy ties, with infinities and negative values; a second call
    on other scores finds the cooperative kernels' workspace clean."""
    rng = np.random.default_rng(A + K)
    B = 4
    s = np.empty((B, A), np.float32)
    s[0] = 1 / (1 + np.exp(-rng.normal(-4.6, 0.3, A)))                    # proposal scores of an untrained head
    s[1] = np.round(rng.random(A) * 50) / 50                              # ~51 distinct values: ties everywhere
    s[2] = rng.normal(0, 3, A)                                            # both signs
    s[3] = 0.25                                                           # all equal
    if A > 100:
        s[2, 7], s[2, 11], s[2, 13] = np.inf, -np.inf, -0.0
    from glenet_amd import detector as det
    top, order = det.topk_desc(torch.from_numpy(s).to(dev), K)
    assert top.shape == (B, K) and order.dtype == torch.int64
    want = np.stack([np.argsort(-s[b], kind="stable")[:K] for b in range(B)])
    assert np.array_equal(order.cpu().numpy(), want)
    assert np.array_equal(top.cpu().numpy(), np.take_along_axis(s, want, 1))
    ttop, _ = torch.topk(torch.from_numpy(s).to(dev), K, dim=1)
    assert torch.equal(ttop, top)
    s2 = np.ascontiguousarray(s[::-1, ::-1])
    top2, order2 = det.topk_desc(torch.from_numpy(s2).to(dev), K)
    want2 = np.stack([np.argsort(-s2[b], kind="stable")[:K] for b in range(B)])
    assert np.array_equal(order2.cpu().numpy(), want2)


@pytest.mark.gpu
def test_gather_proposals_kernel_equals_the_tensor_expression(dev):
    """glx_gather_proposals (the proposal layer's tail in one launch) == arange / where / gather / mask on the same
    keep lists, incl. frames with fewer survivors than slots and an empty frame (roi_head_template.py:106-126)."""
    from glenet_amd import _lib
    g = torch.Generator(device=dev).manual_seed(5)
    F, A, K, P, C = 3, 500, 200, 64, 7
    cand = torch.randn((F, K, C), device=dev, generator=g)
    top = torch.rand((F, K), device=dev, generator=g)
    lab = torch.randint(0, 3, (F, A), device=dev, generator=g)
    order = torch.stack([torch.randperm(A, device=dev, generator=g)[:K] for _ in range(F)])
    keep = torch.stack([torch.randperm(K, device=dev, generator=g) for _ in range(F)])
    num = torch.tensor([P + 10, 17, 0], dtype=torch.int32, device=dev)
    rois = torch.empty((F, P, C), device=dev)
    scores = torch.empty((F, P), device=dev)
    labels = torch.empty((F, P), dtype=torch.int64, device=dev)
    _lib.call("glx_gather_proposals", cand, top, lab, order, keep, num, F, A, K, K, P, C, rois, scores, labels)
    valid = torch.arange(P, device=dev)[None, :] < num[:, None]
    sel = torch.where(valid, keep[:, :P], torch.zeros_like(keep[:, :P]))
    want_rois = torch.gather(cand, 1, sel.unsqueeze(-1).expand(F, P, C)) * valid.unsqueeze(-1)
    want_scores = torch.gather(top, 1, sel) * valid
    want_labels = torch.gather(torch.gather(lab, 1, order), 1, sel) * valid + 1
    assert torch.equal(rois, want_rois) and torch.equal(scores, want_scores) and torch.equal(labels, want_labels)
    assert float(rois[2].abs().max()) == 0.0 and int(labels[2].max()) == 1
