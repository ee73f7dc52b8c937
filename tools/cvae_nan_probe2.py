"""Replicates bench_config3's training-step part with switches, reading back only at the end."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import cvae_train as ct, dense_path as dp, synth
dev = torch.device("cuda", 0)
E = os.environ.get
torch.manual_seed(1)
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(4096, 2000, 512, with_labels=True))
model = dp.CVAE(4, 8).to(dev)
if E("SAMPLER", "1") == "1":
    model.eval()
    gen = torch.Generator(device=dev).manual_seed(7)
    eps = torch.randn((30, 4096, 8), device=dev, generator=gen)
    with torch.no_grad():
        for _ in range(4):
            for s_ in range(30):
                model.sample(pts, eps[s_])
    torch.cuda.synchronize()
step = ct.CVAETrainStep(model, 4096, 512, lr=3e-4)
fixed = torch.randn((4096, 8), device=dev) if E("FIXEPS") == "1" else None
step.load(pts, box8, box7, fixed)
if E("OUTSIDE") == "1":
    step.draw_eps = False
if E("EAGER") != "1":
    step.capture()
for _ in range(2):
    step.step()
torch.cuda.synchronize(dev)
if E("EVENTS", "1") == "1":
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
t0 = time.perf_counter()
for _ in range(10):
    if E("OUTSIDE") == "1":
        step.eps.normal_()          # drawn eagerly, outside the recorded step
    step.step()
if E("EVENTS", "1") == "1":
    e1.record()
torch.cuda.synchronize(dev)
print("ms/step %.2f" % ((time.perf_counter() - t0) * 100), E("TAG", ""), "loss", float(step.loss), "gradnorm", float(step.optimizer.grad_norm), "eps finite", bool(torch.isfinite(step.eps).all()),
      float(step.eps.abs().max()), "nan params", int(torch.isnan(step.optimizer.flat_param).sum()), "terms", [float(t) for t in step.terms], flush=True)
