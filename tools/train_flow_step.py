"""First-stage training step of the GLENet-VR flow on one GPU (config 3's front half), 4 frames:
voxelize -> sparse backbone (training mode, fused BatchNorm) -> BEV backbone + anchor head (MIOpen,
training mode) -> device target assignment -> fused dense-head loss -> backward of everything.
Eager launches (the BEV part is MIOpen); ground-truth boxes are the synthetic frames' cars."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import backbone as gb, detector as det, losses, synth, target_assign  # noqa: E402

K = synth.KITTI
dev = torch.device("cuda", 0)
B = 4
frames = [synth.kitti_frame(i) for i in range(B)]
pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
M = max(len(f[1]) for f in frames)
gt = torch.zeros(B, M, 8, device=dev)
for i, f in enumerate(frames):
    gt[i, :len(f[1]), :7] = torch.from_numpy(f[1].astype(np.float32)).to(dev)
    gt[i, :len(f[1]), 7] = 1
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
flow = det.VoxelRCNNFlow(K).to(dev).train()
params = [p for m in (flow.backbone_3d, flow.backbone_2d, flow.dense_head) for p in m.parameters()]
anchors = flow.anchors(dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]


def step(timed=False):
    for p in params:
        p.grad = None
    if timed:
        ev[0].record()
    bd = gb.voxelize_batch(pts, bidx, B, K, train=True)
    bd = flow.map_to_bev(flow.backbone_3d(flow.vfe(bd)))
    if timed:
        ev[1].record()
    bd = flow.dense_head(flow.backbone_2d(bd))
    if timed:
        ev[2].record()
    with torch.no_grad():
        tgt = target_assign.assign_targets([anchors], gt, [1], [0.6], [0.45])
    loss, parts = losses.rpn_loss(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"), tgt["box_cls_labels"],
                                  tgt["box_reg_targets"], anchors)
    if timed:
        ev[3].record()
    loss.backward()
    if timed:
        ev[4].record()
    return loss, parts, tgt


for _ in range(4):
    loss, parts, tgt = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
st = []
for _ in range(5):
    step(True)
    torch.cuda.synchronize()
    st.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)])
st = np.mean(st, 0)
print("first-stage training step, %d frames: %.2f ms/step = %.0f frames/s; loss %.4f (cls %.4f loc %.4f dir %.4f), "
      "positives/frame %s" % (B, dt * 1e3, B / dt, float(loss.detach()), float(parts["rpn_loss_cls"]),
                              float(parts["rpn_loss_loc"]), float(parts["rpn_loss_dir"]),
                              (tgt["box_cls_labels"] > 0).sum(1).tolist()))
print("  stages (ms): voxelize + sparse backbone fwd %.2f | BEV backbone + head fwd %.2f | target assignment + loss %.2f | "
      "backward of everything %.2f" % tuple(st))


# ---- the same step without read-backs, replayed as one HIP graph (StaticTrainPipeline; the BEV part
# and the loss ride along as its loss_fn)
def head_loss(bd):
    bd = flow.dense_head(flow.backbone_2d(bd))
    with torch.no_grad():
        tg = target_assign.assign_targets([anchors], gt, [1], [0.6], [0.45])
    return losses.rpn_loss(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"), tg["box_cls_labels"],
                           tg["box_reg_targets"], anchors)[0]


del loss, parts, tgt
pipe = gb.StaticTrainPipeline(flow.backbone_3d, K, B, pts.shape[0], 4, loss_fn=head_loss,
                              extra_modules=(flow.backbone_2d, flow.dense_head))
pipe.calibrate(pts, bidx)
pipe.load(pts, bidx)
pipe.capture()
for _ in range(3):
    pipe.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    pipe.replay()
torch.cuda.synchronize()
dg = (time.perf_counter() - t0) / n
pipe.check()
print("  shape-static, one HIP graph: %.2f ms/step = %.0f frames/s; loss %.4f" % (dg * 1e3, B / dg, float(pipe.loss.detach())))


# ---- the whole two-stage training step (config 3's shape on one GPU): + proposals (NMS 9000 -> 512), RoI
# targets (roi_targets.ProposalTargetLayer: 512 proposals -> 128 samples per frame, GLENet_VR.yaml:141-153),
# RoI-grid pooling in training mode, FC towers, the three RoI-head losses.
std_layer = torch.nn.Linear(256, 7).to(dev)
torch.nn.init.normal_(std_layer.weight, std=0.001)
roi_params = [p for m in (flow.roi_pool, flow.roi_fc, std_layer) for p in m.parameters()]
R = 128
from glenet_amd import roi_targets  # noqa: E402
target_layer = roi_targets.ProposalTargetLayer(dict(ROI_PER_IMAGE=R, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True,
                                                    CLS_SCORE_TYPE="roi_iou", CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25,
                                                    CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55))
jit = torch.tensor([0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08], device=dev)
ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(6)]


def full_step(timed=False):
    for p in params + roi_params:
        p.grad = None
    if timed:
        ev2[0].record()
    bd = gb.voxelize_batch(pts, bidx, B, K, train=True)
    bd = flow.map_to_bev(flow.backbone_3d(flow.vfe(bd)))
    bd = flow.dense_head(flow.backbone_2d(bd))
    if timed:
        ev2[1].record()
    with torch.no_grad():
        tgt = target_assign.assign_targets([anchors], gt, [1], [0.6], [0.45])
        cls, boxes = det.predicted_boxes(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"), anchors)
        rois, roi_scores, roi_labels = det.proposal_layer(boxes, cls, 9000, 512, 0.8)
        # an untrained head proposes nothing near the ground truth: the first slots take jittered ground-truth
        # boxes (what a trained first stage delivers) so that the regression / corner terms have foreground
        has = (gt[:, :, 7:8] > 0)
        ng = gt.shape[1]
        rois[:, :ng, :7] = torch.where(has, gt[:, :, :7] + jit, rois[:, :ng, :7])
        roi_labels[:, :ng] = torch.where(has[..., 0], gt[:, :, 7].long(), roi_labels[:, :ng])
        td = target_layer({"rois": rois, "roi_scores": roi_scores, "roi_labels": roi_labels, "gt_boxes": gt})
        rois, gt_of = td["rois"].contiguous(), td["gt_of_rois"]
        reg_valid, cls_lab = td["reg_valid_mask"].view(-1), td["rcnn_cls_labels"].view(-1)
        gt_ct = losses.canonical_gt_of_rois(rois, gt_of)
        unc = torch.full((B * R, 7), 0.05, device=dev)
    rpn, _ = losses.rpn_loss(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"), tgt["box_cls_labels"],
                             tgt["box_reg_targets"], anchors)
    if timed:
        ev2[2].record()
    pooled = flow.roi_pool(rois, bd["multi_scale_3d_features"], bd["multi_scale_3d_strides"], B)
    shared = flow.roi_fc.shared_fc_layer(pooled.reshape(pooled.shape[0], -1))
    reg_feat = flow.roi_fc.reg_fc_layers(shared)
    rcnn_reg, rcnn_std = flow.roi_fc.reg_pred_layer(reg_feat), std_layer(reg_feat)
    rcnn_cls = flow.roi_fc.cls_pred_layer(flow.roi_fc.cls_fc_layers(shared))
    if timed:
        ev2[3].record()
    l_cls = losses.rcnn_cls_loss(rcnn_cls, cls_lab)
    l_kl, _ = losses.kl_reg_loss(rcnn_reg, rcnn_std, rois, gt_ct[..., :7], unc, reg_valid)
    l_cor = losses.corner_loss(rcnn_reg, rois, gt_of[..., :7], reg_valid)
    total = rpn + l_cls + l_kl + l_cor
    if timed:
        ev2[4].record()
    total.backward()
    if timed:
        ev2[5].record()
    return total, (rpn, l_cls, l_kl, l_cor), int(reg_valid.sum())


for _ in range(3):
    total, terms, nfg = full_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    full_step()
torch.cuda.synchronize()
df = (time.perf_counter() - t0) / 10
st2 = []
for _ in range(3):
    full_step(True)
    torch.cuda.synchronize()
    st2.append([ev2[i].elapsed_time(ev2[i + 1]) for i in range(5)])
st2 = np.mean(st2, 0)
print("two-stage training step, %d frames x %d RoIs: %.2f ms/step = %.0f frames/s; loss %.4f = rpn %.4f + rcnn cls %.4f "
      "+ KL reg %.4f + corner %.4f; %d foreground RoIs" % ((B, R, df * 1e3, B / df, float(total.detach()))
                                                          + tuple(float(t.detach()) for t in terms) + (nfg,)))
print("  stages (ms): first stage fwd %.2f | targets + proposals + RoI targets + rpn loss %.2f | RoI-grid pool + FC "
      "(training path) %.2f | RoI losses %.2f | backward %.2f" % tuple(st2))
