"""Node types of the recorded steps (GLENet-VR training step, inference pass, CVAE training step): how many memset nodes
does each contain?  (ROCm 7.2 replays them with a stale pattern: tools/graph_memset_repro.py.)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import _lib, cvae_train as ct, dense_path as dp, detector as det, glenet_vr as gvr, synth  # noqa: E402

dev = torch.device("cuda", 0)
K = synth.KITTI
frames = [synth.kitti_frame(i) for i in range(4)]
pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
gt = torch.zeros(4, 16, 8, device=dev)
unc = torch.full((4, 16, 7), 0.05, device=dev)
for i, f in enumerate(frames):
    gt[i, :len(f[1]), :7] = torch.from_numpy(f[1]).to(dev)
    gt[i, :len(f[1]), 7] = 1
torch.manual_seed(0)
model = gvr.GLENetVR(K).to(dev).train()
pipe = gvr.StaticTrainStep(model, 4, pts.shape[0], 4, max_gt=16, seed_rois_with_gt=[0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08])
pipe.calibrate(pts, bidx)
pipe.load(pts, bidx, gt, unc)
pipe.capture()
print("GLENet-VR training step:", json.dumps(_lib.audit_graph(pipe.graph)), flush=True)
flow = gvr.GLENetVR(K).to(dev).eval()
ip = det.StaticDetectorPipeline(flow, 4, pts.shape[0])
ip.calibrate(pts, bidx)
ip.load(pts, bidx)
ip.capture()
print("GLENet-VR inference pass:", json.dumps(_lib.audit_graph(ip.graph)), flush=True)
p3, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(1024, 2000, 512, with_labels=True))
cv = ct.CVAETrainStep(dp.CVAE(4, 8).to(dev), 1024, 512, lr=3e-4)
cv.load(p3, box8, box7)
cv.capture()
print("CVAE training step (1024 objects):", json.dumps(_lib.audit_graph(cv.graph)), flush=True)
