"""Config-3 shaped inference data flow on one GPU: voxelize -> sparse backbone -> BEV backbone +
anchor head -> proposals (device NMS) -> RoI-grid pool -> FC refine, 4 frames; eager with exact
shapes (stage split by events) and as one shape-static HIP graph (StaticDetectorPipeline)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import detector as det, synth  # noqa: E402

K = synth.KITTI
dev = torch.device("cuda", 0)
B = 4
frames = [synth.kitti_frame(i)[0] for i in range(B)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
flow = det.VoxelRCNNFlow(K).to(dev).eval()


def stage_times():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    from glenet_amd import backbone as gb
    with torch.no_grad():
        ev[0].record()
        bd = gb.voxelize_batch(pts, bidx, B, K, train=False)
        bd = flow.map_to_bev(flow.backbone_3d(flow.vfe(bd)))
        ev[1].record()
        bd = flow.dense_head(flow.backbone_2d(bd))
        ev[2].record()
        cls, boxes = det.predicted_boxes(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"),
                                         flow.anchors(dev))
        rois, _, _ = det.proposal_layer(boxes, cls, *flow.nms)
        ev[3].record()
        pooled = flow.roi_pool(rois, bd["multi_scale_3d_features"], bd["multi_scale_3d_strides"], B)
        flow.roi_fc(pooled)
        ev[4].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]


with torch.no_grad():
    for _ in range(5):
        flow(pts, bidx, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        flow(pts, bidx, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
st = np.mean([stage_times() for _ in range(5)], axis=0)
print("two-stage inference flow, %d frames: %.2f ms/step = %.0f frames/s (eager, exact shapes)" % (B, dt * 1e3, B / dt))
print("  stages (ms): voxelize+sparse backbone+dense %.2f | BEV backbone+head %.2f | "
      "decode+top-k+NMS %.2f | RoI-grid pool+FC %.2f" % tuple(st))

pipe = det.StaticDetectorPipeline(flow, B, pts.shape[0])
pipe.calibrate(pts, bidx)
pipe.load(pts, bidx)
pipe.capture()
for _ in range(5):
    pipe.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    pipe.replay()
torch.cuda.synchronize()
dg = (time.perf_counter() - t0) / n
pipe.check()
with torch.no_grad():
    ref = flow(pts, bidx, B)
ok = torch.equal(ref["rois"], pipe.out["rois"]) and torch.allclose(ref["batch_box_preds"], pipe.out["batch_box_preds"],
                                                                    rtol=1e-4, atol=1e-4)
print("  as one HIP graph (shape-static): %.2f ms/step = %.0f frames/s; matches the eager flow: %s" % (dg * 1e3, B / dg, ok))

# ---- where the recorded flow spends its time: the same frame recorded once more with stamp launches (glx_stamp)
from glenet_amd import _lib, backbone as gb  # noqa: E402
stamps = torch.zeros(16, dtype=torch.int64, device=dev)
names = []


def stamp(name):
    _lib.call("glx_stamp", stamps, len(names))
    names.append(name)


class _Stamped(det.StaticDetectorPipeline):
    def enqueue(self):
        names.clear()
        stamp("start")
        bd = gb.StaticFramePipeline.enqueue(self)
        stamp("voxelize + sparse backbone + dense")
        with torch.no_grad(), _lib.workspace.scoped(id(self)):
            self.flow.__dict__["mark"] = stamp
            try:
                bd = self.flow.second_stage(bd, self.B)
            finally:
                self.flow.__dict__.pop("mark", None)
        self.out = bd
        return bd


try:
    sp = _Stamped(flow, B, pts.shape[0], capacities=pipe.capacities)
    sp.load(pts, bidx)
    sp.capture()
    for _ in range(3):
        sp.replay()
    torch.cuda.synchronize()
    t = stamps[:len(names)].tolist()
    print("  recorded flow, stamps (ms since start): " + " | ".join("%s %.2f" % (n_, (t1 - t[0]) * 1e-5)
                                                                  for n_, t1 in zip(names[1:], t[1:])))
except Exception as e:      # the stamped variant pokes at the flow's modules; the numbers above do not depend on it
    print("  (stamped recording not available: %r)" % (e,))
