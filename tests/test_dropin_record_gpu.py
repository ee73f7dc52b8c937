"""dropin.record(): the reference's training loop at the speed of the recorded step.

tools/train_utils/train_utils.py:45-76 drives `model_func -> loss.backward() -> clip_grad_norm_ -> optimizer.step()` with the
caller's own torch optimizer; record() keeps exactly that contract and replaces what runs inside `model(batch_dict)`: one HIP
graph of forward + backward over the network's own Parameter objects.  On the GPU box the reference does not exist, so the
network handed to record() here is a stand-in with the reference network's surface -- GLENet-VR's state-dict keys
(tests/golden/ref_state_keys.npz pins those to the reference's own build), `model_cfg` = the MODEL block of GLENet_VR.yaml as
the reference's loader leaves it, `dataset.point_cloud_range / voxel_size`; the build container runs record(dry_run=True) on
the reference's real network (tools/ref_dropin_check.py: 272 shared tensors, configuration == the package's constants)."""
import copy
import types

import numpy as np
import pytest
import torch

from glenet_amd import synth

pytestmark = pytest.mark.gpu

# MODEL of tools/cfgs/kitti_models/GLENet_VR.yaml:32-166 (values, not source text), with the smaller proposal / sampling
# counts the other composed-step tests use
MODEL_CFG = dict(
    NAME="VoxelRCNN", VFE=dict(NAME="MeanVFE"), BACKBONE_3D=dict(NAME="VoxelBackBone8x"),
    MAP_TO_BEV=dict(NAME="HeightCompression", NUM_BEV_FEATURES=256),
    BACKBONE_2D=dict(NAME="BaseBEVBackbone", LAYER_NUMS=[5, 5], LAYER_STRIDES=[1, 2], NUM_FILTERS=[64, 128],
                     UPSAMPLE_STRIDES=[1, 2], NUM_UPSAMPLE_FILTERS=[128, 128]),
    DENSE_HEAD=dict(NAME="AnchorHeadSingle", CLASS_AGNOSTIC=False, USE_DIRECTION_CLASSIFIER=True, DIR_OFFSET=0.78539,
                    DIR_LIMIT_OFFSET=0.0, NUM_DIR_BINS=2,
                    ANCHOR_GENERATOR_CONFIG=[dict(class_name="Car", anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57],
                                                  anchor_bottom_heights=[-1.78], align_center=False, feature_map_stride=8,
                                                  matched_threshold=0.6, unmatched_threshold=0.45)],
                    TARGET_ASSIGNER_CONFIG=dict(NAME="AxisAlignedTargetAssigner", POS_FRACTION=-1.0, SAMPLE_SIZE=512,
                                                NORM_BY_NUM_EXAMPLES=False, MATCH_HEIGHT=False, BOX_CODER="ResidualCoder"),
                    LOSS_CONFIG=dict(LOSS_WEIGHTS=dict(cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0] * 7))),
    ROI_HEAD=dict(NAME="VoxelRCNNKLLabelIoUHead", CLASS_AGNOSTIC=True, SHARED_FC=[256, 256], CLS_FC=[256, 256], REG_FC=[256, 256],
                  DP_RATIO=0.0,
                  NMS_CONFIG=dict(TRAIN=dict(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=1024,
                                             NMS_POST_MAXSIZE=128, NMS_THRESH=0.8),
                                  TEST=dict(NMS_TYPE="nms_gpu", MULTI_CLASSES_NMS=False, NMS_PRE_MAXSIZE=2048,
                                            NMS_POST_MAXSIZE=100, NMS_THRESH=0.7)),
                  ROI_GRID_POOL=dict(FEATURES_SOURCE=["x_conv2", "x_conv3", "x_conv4"], PRE_MLP=True, GRID_SIZE=6,
                                     POOL_LAYERS={n: dict(MLPS=[[32, 32]], QUERY_RANGES=[[4, 4, 4]], POOL_RADIUS=[r], NSAMPLE=[16],
                                                          POOL_METHOD="max_pool")
                                                  for n, r in (("x_conv2", 0.4), ("x_conv3", 0.8), ("x_conv4", 1.6))}),
                  TARGET_CONFIG=dict(BOX_CODER="ResidualCoder", ROI_PER_IMAGE=32, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True,
                                     CLS_SCORE_TYPE="roi_iou", CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1,
                                     HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55),
                  LOSS_CONFIG=dict(CLS_LOSS="BinaryCrossEntropy", REG_LOSS="smooth-l1", CORNER_LOSS_REGULARIZATION=True,
                                   GRID_3D_IOU_LOSS=False,
                                   LOSS_WEIGHTS=dict(rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, rcnn_corner_weight=1.0,
                                                     rcnn_iou3d_weight=1.0, code_weights=[1.0] * 7))))
MODEL_CFG["POST_PROCESSING"] = dict(RECALL_THRESH_LIST=[0.3, 0.5, 0.7], SCORE_THRESH=0.3, POST_SCORE_THRESH=0.4, OUTPUT_RAW_SCORE=False,
                                   EVAL_METRIC="kitti",
                                   NMS_CONFIG=dict(MULTI_CLASSES_NMS=False, NMS_TYPE="new_nms_gpu", NMS_THRESH=0.1, NMS_PRE_MAXSIZE=4096,
                                                   NMS_POST_MAXSIZE=500))
JIT = [0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08]


def _collated(ids, num_points, max_gt=16):
    """A batch as pcdet/datasets/dataset.py:170-250 collates it: points (P, 1 + 4) with the frame id in column 0, gt_boxes
    (B, G, 8) zero-padded, gt_uncertaintys (B, G, 7) padded with -1."""
    frames = [synth.kitti_frame(i, num_points=num_points) for i in ids]
    pts = np.concatenate([np.concatenate([np.full((len(f[0]), 1), i, np.float32), f[0]], 1) for i, f in enumerate(frames)])
    gt = np.zeros((len(ids), max_gt, 8), np.float32)
    unc = -np.ones((len(ids), max_gt, 7), np.float32)
    for i, (fid, f) in enumerate(zip(ids, frames)):
        k = len(f[1])
        gt[i, :k, :7] = f[1]
        gt[i, :k, 7] = 1
        unc[i, :k] = synth.gt_uncertainty(fid, k)
    return dict(points=pts, gt_boxes=gt, gt_uncertaintys=unc, batch_size=len(ids))


def _network(dev):
    """The stand-in for build_network's result: GLENet-VR's modules under the reference's names + model_cfg + dataset."""
    from glenet_amd import dropin
    from glenet_amd import glenet_vr as gvr
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    roi_cfg, head_cfg = dropin._translate_cfg(MODEL_CFG)
    net = gvr.GLENetVR(synth.KITTI, roi_cfg=roi_cfg, head_cfg=head_cfg).to(dev).train()
    net.model_cfg = MODEL_CFG
    net.dataset = types.SimpleNamespace(point_cloud_range=np.array(synth.KITTI["point_cloud_range"], np.float32),
                                        voxel_size=synth.KITTI["voxel_size"],
                                        point_feature_encoder=types.SimpleNamespace(num_point_features=4))
    return net


def test_recorded_step_gives_the_eager_step_s_loss_and_gradients_and_trains_under_a_torch_optimizer(dev):
    """ret_dict / tb_dict / disp_dict and `loss.backward()` as the reference's loop uses them; every parameter gradient
    equals the exact-shape eager step's (2e-3 of its scale: the bound the shape-static step is held to elsewhere); the
    gradients are views of one flat buffer on the network's OWN parameters; a torch.optim.AdamW + clip_grad_norm_ loop over
    changing batches follows the same loop driven by eager steps; zero_grad in both forms; a batch that does not fit raises."""
    from glenet_amd import dropin
    net = _network(dev)
    B = 2
    batches = [_collated([60 + 2 * i, 61 + 2 * i], 6000) for i in range(3)]
    R = MODEL_CFG["ROI_HEAD"]["NMS_CONFIG"]["TRAIN"]["NMS_POST_MAXSIZE"]
    P = MODEL_CFG["ROI_HEAD"]["TARGET_CONFIG"]["ROI_PER_IMAGE"]
    gen = torch.Generator(device=dev).manual_seed(5)
    draws = (torch.rand((B, R), device=dev, generator=gen), torch.rand((B, P), device=dev, generator=gen))
    state0 = copy.deepcopy(net.state_dict())
    keys0 = list(net.state_dict().keys())
    ptrs0 = {n: p for n, p in net.named_parameters()}

    step = dropin.record(net, batches, seed_rois_with_gt=JIT)
    step.twin.fixed_draws = draws                      # the RoI sampler's uniform numbers: the same in both paths below
    step.pipe.capture(split=True)                      # (re-recorded with the fixed draws)
    assert list(net.state_dict().keys()) == keys0 and all(ptrs0[n] is p for n, p in net.named_parameters())
    for k, v in state0.items():                        # recording left the training state alone
        assert torch.equal(net.state_dict()[k], v), k

    # ---- one step: the reference loop's statements
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)
    opt.zero_grad()
    ret, tb, disp = step(batches[0])
    loss = ret["loss"].mean()
    assert loss.requires_grad and disp == {} and {"loss_rpn", "rcnn_loss_cls", "rcnn_loss_reg", "rcnn_loss_corner"} <= set(tb)
    loss.backward()
    assert step.check()
    got = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(step.params, step.views))
    got_loss, got_tb = float(loss.detach()), {k: float(v) for k, v in tb.items() if v is not None and torch.is_tensor(v) and v.numel() == 1}

    # ---- the same step eagerly, exact shapes, on a copy
    ref = _network(dev)
    ref.load_state_dict(state0)
    ref.fixed_draws = draws
    pts, bidx, gt, unc = step.tensors(batches[0])
    l, parts = ref.training_step(pts, bidx, B, gt, unc, seed_rois_with_gt=torch.tensor(JIT, device=dev))
    l.backward()
    np.testing.assert_allclose(got_loss, float(l), rtol=2e-4)
    for k in ("loss_rpn", "rcnn_loss_cls", "rcnn_loss_reg", "rcnn_loss_corner"):
        np.testing.assert_allclose(got_tb[k], float(parts[k]), rtol=2e-4, atol=1e-6, err_msg=k)
    for n, p in ref.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        assert float((got[n] - p.grad).abs().max()) <= 2e-3 * scale + 1e-7, n

    # ---- a scaled root gradient scales every gradient (loss * 0.5).backward()
    opt.zero_grad(set_to_none=True)
    ret, _, _ = step(batches[0])
    (ret["loss"] * 0.5).backward()
    for n, p in net.named_parameters():
        # (two replays are not bitwise equal: the RoI pooling's backward adds with atomics)
        assert float((p.grad - got[n] * 0.5).abs().max()) <= 2e-4 * float(got[n].abs().max()) + 1e-9, n

    # ---- three optimizer steps over changing batches: every recorded step == the eager step FROM THE SAME STATE (two free-
    # running Adam trajectories drift by a sign flip of every noise-level gradient element per step: test_train_step_gpu.py)
    lr = 2e-4
    opt = torch.optim.AdamW(net.parameters(), lr=lr, betas=(0.9, 0.99), weight_decay=0.01)
    ref_opt = torch.optim.AdamW(ref.parameters(), lr=lr, betas=(0.9, 0.99), weight_decay=0.01)
    first = None
    for i, bd in enumerate(batches):
        with torch.no_grad():
            ref.load_state_dict(net.state_dict())
        ref_opt.load_state_dict(copy.deepcopy(opt.state_dict()))
        opt.zero_grad(set_to_none=bool(i % 2))
        ret, tb, _ = step(bd)
        ret["loss"].mean().backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
        opt.step()
        ref_opt.zero_grad()
        pts, bidx, gt, unc = step.tensors(bd)
        l, _ = ref.training_step(pts, bidx, B, gt, unc, seed_rois_with_gt=torch.tensor(JIT, device=dev))
        l.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        ref_opt.step()
        np.testing.assert_allclose(float(ret["loss"].detach()), float(l.detach()), rtol=2e-4, err_msg="step %d" % i)
        a = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        b = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
        d = (a - b).abs()
        assert float(d.max()) <= 2.2 * lr, "step %d: an update differs by more than a sign flip" % i
        flipped = float((d > 0.5 * lr).float().mean())
        assert flipped < 2e-3 and float(d.mean()) < 5e-3 * lr, (i, flipped, float(d.mean()) / lr)
        first = first if first is not None else a.clone()
    assert step.check()
    assert float((a - first).abs().max()) > 0                    # the network's own parameters moved

    # ---- evaluation does not go through the recorded step
    net.eval()
    with pytest.raises(RuntimeError):
        step(batches[0])
    net.train()
    # ---- a batch with more points than the recording was sized for is refused, not truncated
    big = _collated([70, 71], 9000)
    with pytest.raises(ValueError):
        step(big)


def test_recorded_inference_pass_returns_the_reference_s_eval_contract(dev):
    """`pred_dicts, recall_dicts = model(batch_dict)` of the eval loop (tools/eval_utils/eval_utils.py:53-66) from ONE recorded graph
    on the network's own parameters: the per-frame dicts equal the exact-shape eager pass (GLENetVR.predict with the translated
    POST_PROCESSING settings), the recall record equals Detector3DTemplate.generate_recall_record's statements evaluated directly,
    a change of the weights is picked up (the pass re-records), training mode is refused."""
    from glenet_amd import detector as det
    from glenet_amd import dropin
    from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_utils
    net = _network(dev).eval()
    batches = [_collated([90 + 2 * i, 91 + 2 * i], 6000) for i in range(2)]
    for b in batches:
        b["batch_size"] = 2
    post_cfg, thresh = dropin._translate_post_cfg(MODEL_CFG)
    assert post_cfg == dict(SCORE_THRESH=0.3, POST_SCORE_THRESH=0.4, NMS_THRESH=0.1, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500)
    # an untrained head scores everything about alike (sigmoid ~ 0.5): thresholds that let boxes through
    post_cfg = dict(post_cfg, SCORE_THRESH=0.0, POST_SCORE_THRESH=None)
    rec = dropin.record_inference(net, batches)
    rec.pipe.post_cfg = post_cfg
    rec.pipe.capture()
    for bd in batches:
        pred, recall = rec(bd)
        assert rec.check()
        pts = torch.as_tensor(bd["points"]).to(dev)
        with torch.no_grad():
            want = net.predict(pts[:, 1:].contiguous(), pts[:, 0].int().contiguous(), 2, post_cfg)
        ref = det.pred_dicts(want["post"])
        assert len(pred) == len(ref) == 2
        for a, b in zip(pred, ref):
            assert a["pred_boxes"].shape == b["pred_boxes"].shape and a["pred_boxes"].shape[0] > 0
            assert torch.equal(a["pred_labels"], b["pred_labels"])
            assert torch.allclose(a["pred_scores"], b["pred_scores"], atol=2e-5)
            assert torch.allclose(a["pred_boxes"], b["pred_boxes"], atol=2e-4)
        # the recall record: detector3d_template.py:318-362 evaluated directly on the pass's own tensors
        gt = torch.as_tensor(bd["gt_boxes"]).to(dev)
        exp = {"gt": 0, **{"%s_%s" % (k, t): 0 for t in thresh for k in ("roi", "rcnn")}}
        for f in range(2):
            g = gt[f][gt[f].abs().sum(1) > 0]
            exp["gt"] += int(g.shape[0])
            i_rcnn = iou3d_nms_utils.boxes_iou3d_gpu(want["batch_box_preds"][f][:, :7].contiguous(), g[:, :7].contiguous())
            i_roi = iou3d_nms_utils.boxes_iou3d_gpu(want["rois"][f][:, :7].contiguous(), g[:, :7].contiguous())
            for t in thresh:
                exp["rcnn_%s" % t] += int((i_rcnn.max(0)[0] > t).sum())
                exp["roi_%s" % t] += int((i_roi.max(0)[0] > t).sum())
        assert recall == exp and recall["gt"] > 0, (recall, exp)
    # new weights: the recorded pass follows them
    before = rec(batches[0])[0][0]["pred_scores"].clone()
    with torch.no_grad():
        net.dense_head.conv_cls.bias.add_(0.5)
        net.roi_head.cls_pred_layer.bias.add_(0.7)
    from glenet_amd import _lib
    _lib.bump_weights_epoch((net.dense_head.conv_cls.bias, net.roi_head.cls_pred_layer.bias))
    after = rec(batches[0])[0][0]["pred_scores"]
    assert after.shape[0] > 0 and (after.shape != before.shape or not torch.allclose(after, before))
    net.train()
    with pytest.raises(RuntimeError):
        rec(batches[0])
