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
