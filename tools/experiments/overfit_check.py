"""End-to-end sanity of the training step's gradients: overfit ONE batch for STEPS recorded steps and print the loss
trajectory -- run once with the own BEV kernels and once with GLX_OWN_CONV3X3=0 GLX_OWN_DECONV=0 GLX_BEV_SPARSE_FIRST=0
(vendor kernels): the curves must agree to the noise of a chaotic trajectory and fall."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from glenet_amd import glenet_vr as gvr
import test_train_step_gpu as helpers

dev = torch.device("cuda", 0)
batch = helpers._batch(dev, [70, 71], 6000)
m = helpers._small_model(dev)
R, P = m.roi_cfg["NMS_TRAIN"][1], m.roi_cfg["TARGET"]["ROI_PER_IMAGE"]
gen = torch.Generator(device=dev).manual_seed(4)
m.fixed_draws = (torch.rand((2, R), device=dev, generator=gen), torch.rand((2, P), device=dev, generator=gen))
pipe = gvr.StaticTrainStep(m, 2, batch[0].shape[0] + 700, max_gt=16, lr=1e-3, seed_rois_with_gt=helpers.JIT)
pipe.calibrate(batch[0], batch[1])
pipe.load(*batch)
pipe.capture()
out = []
for it in range(int(os.environ.get("STEPS", "80"))):
    pipe.load(*batch)
    pipe.step()
    if it % 10 == 0 or it == 79:
        torch.cuda.synchronize()
        out.append("%d:%.4f" % (it, float(pipe.loss)))
pipe.check()
print(" ".join(out))
