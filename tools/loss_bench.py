"""RoI-head loss glue on the GPU box: the reference's tensor-op sequences (glenet_amd.losses.*_torch,
statement-for-statement mirrors) vs the fused kernels, forward + backward, 512 RoIs (4 frames x 128)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import losses  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, N = 4, 128
rois = torch.cat([torch.randn(B, N, 3) * 10, torch.rand(B, N, 3) * 3 + 0.5, torch.rand(B, N, 1) * 6 - 3], -1).to(dev)
gt_src = torch.cat([rois[..., :3].cpu() + torch.randn(B, N, 3) * 0.5, rois[..., 3:6].cpu() * 1.1,
                    rois[..., 6:7].cpu() + 0.2, torch.ones(B, N, 1)], -1).to(dev)
unc = (torch.rand(B, N, 7) * 0.2 + 1e-3).to(dev)
valid = (torch.rand(B * N) > 0.4).long().to(dev)
reg = (torch.randn(B * N, 7) * 0.3).to(dev).requires_grad_(True)
std = torch.randn(B * N, 7).to(dev).requires_grad_(True)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def step(canon, kl, corner):
    reg.grad = std.grad = None
    gt_ct = canon(rois, gt_src)[..., :7]
    l1, _ = kl(reg, std, rois, gt_ct, unc, valid)
    l2 = corner(reg, rois, gt_src[..., :7], valid)
    (l1 + l2).backward()
    return l1, l2


a = step(losses.canonical_gt_of_rois_torch, losses.kl_reg_loss_torch, losses.corner_loss_torch)
b = step(losses.canonical_gt_of_rois, losses.kl_reg_loss, losses.corner_loss)
print("losses (tensor ops) %.6f %.6f | (kernels) %.6f %.6f" % (float(a[0]), float(a[1]), float(b[0]), float(b[1])))
t_ref = timeit(lambda: step(losses.canonical_gt_of_rois_torch, losses.kl_reg_loss_torch, losses.corner_loss_torch))
t_k = timeit(lambda: step(losses.canonical_gt_of_rois, losses.kl_reg_loss, losses.corner_loss))
print("canonical transform + KL reg loss + corner loss, fwd+bwd, %d RoIs: tensor ops %.0f us -> fused kernels %.0f us"
      % (B * N, t_ref, t_k))

# ---- anchor target assignment at the GLENet-VR size: 200 x 176 x 2 car anchors, 4 frames, 20 ground truths
from glenet_amd import detector as det, target_assign  # noqa: E402
anchors = det.generate_anchors([0, -40.0, -3, 70.4, 40.0, 1], (176, 200), [[3.9, 1.6, 1.56]], [0, 1.57], [-1.78], device=dev)
gt = torch.zeros(4, 40, 8, device=dev)
for b in range(4):
    n = 20
    gt[b, :n, 0] = torch.rand(n, device=dev) * 68 + 1
    gt[b, :n, 1] = torch.rand(n, device=dev) * 76 - 38
    gt[b, :n, 2] = -1.0
    gt[b, :n, 3:6] = torch.tensor([3.9, 1.6, 1.56], device=dev) * (1 + torch.randn(n, 3, device=dev) * 0.05)
    gt[b, :n, 6] = torch.rand(n, device=dev) * 6.28 - 3.14
    gt[b, :n, 7] = 1
fn = lambda: target_assign.assign_targets([anchors], gt, [1], [0.6], [0.45])   # noqa: E731
out = fn()
lab = out["box_cls_labels"]
print("anchor target assignment, 4 frames x %d anchors x 20 ground truths: %.0f us; positives/frame %s"
      % (lab.shape[1], timeit(fn), (lab > 0).sum(1).tolist()))

# ---- dense head loss at the same size: focal cls + sin-difference smooth-L1 + direction CE, fwd + bwd
A = lab.shape[1]
cls_p = torch.randn(4, 200, 176, 2, device=dev, requires_grad=True)
box_p = (torch.randn(4, 200, 176, 14, device=dev) * 0.3).requires_grad_(True)
dir_p = torch.randn(4, 200, 176, 4, device=dev, requires_grad=True)


def rpn_step():
    cls_p.grad = box_p.grad = dir_p.grad = None
    l, _ = losses.rpn_loss(cls_p, box_p, dir_p, out["box_cls_labels"], out["box_reg_targets"], anchors)
    l.backward()
    return l


def rpn_step_ref():
    cls_p.grad = box_p.grad = dir_p.grad = None
    l, _ = losses.rpn_loss_torch(cls_p, box_p, dir_p, out["box_cls_labels"], out["box_reg_targets"], anchors)
    l.backward()
    return l


print("dense head loss (3 terms) fwd+bwd, 4 frames x %d anchors: tensor ops %.0f us -> fused %.0f us; loss %.4f / %.4f"
      % (A, timeit(rpn_step_ref), timeit(rpn_step), float(rpn_step_ref().detach()), float(rpn_step().detach())))
