"""Which parameter gradients of the GLENet-VR training step are written in place into FlatAdamW's flat gradient buffer
(_lib.grad_buffer / dense_path.run_deferred_fc_wgrads) and which ones pack_grads still gathers (name, elements)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glenet_amd import glenet_vr as gvr, synth

dev = torch.device("cuda", 0)
K = synth.KITTI
frames = [synth.kitti_frame(i) for i in range(4)]
pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
gt = torch.zeros(4, 16, 8, device=dev); unc = torch.zeros(4, 16, 7, device=dev)
for i, f in enumerate(frames):
    k = len(f[1]); gt[i, :k, :7] = torch.from_numpy(f[1]).to(dev); gt[i, :k, 7] = 1
    unc[i, :k] = torch.from_numpy(synth.gt_uncertainty(i, k)).to(dev)
torch.manual_seed(0)
model = gvr.GLENetVR(K).to(dev).train()
pipe = gvr.StaticTrainStep(model, 4, pts.shape[0], K["num_features"], max_gt=16, seed_rois_with_gt=[0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08])
pipe.capacities = pipe.calibrate(pts, bidx)
pipe.load(pts, bidx, gt, unc)
pipe.enqueue()          # forward + backward, no update: .grad as backward left it
torch.cuda.synchronize()
inp, rest = 0, []
for n, p in model.named_parameters():
    v = getattr(p, "_glx_grad_view", None)
    if p.grad is not None and v is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == v.stride():
        inp += p.numel()
    else:
        rest.append((p.numel(), n, None if p.grad is None else tuple(p.grad.stride())))
rest.sort(reverse=True)
print("in place: %d elements; gathered by pack_grads: %d elements in %d tensors" % (inp, sum(r[0] for r in rest), len(rest)))
for r in rest[:40]:
    print("  %8d  %s  %s" % r)
