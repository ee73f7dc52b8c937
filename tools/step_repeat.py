"""Repeatability of the GLENet-VR training step's gradients: the same batch from the same state, several eager passes and
(optionally) replays of the recorded step; prints the largest deviation from the first pass relative to the largest gradient.
Float atomics in the sparse weight gradients and MIOpen's split-K reorder sums by ~1e-7; anything larger is a race."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from glenet_amd import glenet_vr as gvr
import test_train_step_gpu as helpers

dev = torch.device("cuda", 0)
ids = [int(v) for v in os.environ.get("FRAMES", "60,61").split(",")]
batch = helpers._batch(dev, ids, 6000)
npts = batch[0].shape[0] + 700
m = helpers._small_model(dev)
R, P = m.roi_cfg["NMS_TRAIN"][1], m.roi_cfg["TARGET"]["ROI_PER_IMAGE"]
gen = torch.Generator(device=dev).manual_seed(4)
m.fixed_draws = (torch.rand((2, R), device=dev, generator=gen), torch.rand((2, P), device=dev, generator=gen))
probe = gvr.StaticTrainStep(m, 2, npts, max_gt=16, lr=1e-3, seed_rois_with_gt=helpers.JIT)
caps = probe.calibrate(batch[0], batch[1])
del probe
ref = gvr.StaticTrainStep(m, 2, npts, max_gt=16, lr=1e-3, seed_rois_with_gt=helpers.JIT, capacities=caps)
ref.split = True
grads = []
for it in range(int(os.environ.get("N", "6"))):
    ref.load(*batch)
    ref.enqueue()
    torch.cuda.synchronize()
    grads.append(ref.step_optimizer.flat_grad.detach().clone())
    last = ref.net.last
    print("pass %d: loss %.9f fg_rois %s rois checksum %.6f sampled checksum %.6f" % (
        it, float(ref.loss), {k: round(float(v), 6) for k, v in ref.parts.items() if "fg" in k or "cls" in k},
        float(last["rois"].double().abs().sum()), float(last.get("sampled_rois", last["rois"]).double().abs().sum())), flush=True)
scale = float(grads[0].abs().max())
print("scale", scale, "max deviation from the first pass:", [float((g - grads[0]).abs().max()) / scale for g in grads[1:]])
if os.environ.get("GRAPH"):
    # the recorded step (forward + backward graph of a split capture) against the eager passes above
    import copy
    m2 = helpers._small_model(dev)
    m2.fixed_draws = m.fixed_draws
    pipe = gvr.StaticTrainStep(m2, 2, npts, max_gt=16, lr=1e-3, seed_rois_with_gt=helpers.JIT, capacities=caps)
    pipe.load(*batch)
    pipe.capture(split=True)
    dev_list = []
    for it in range(int(os.environ.get("N", "6"))):
        pipe.load(*batch)
        pipe.replay()
        torch.cuda.synchronize()
        dev_list.append(float((pipe.step_optimizer.flat_grad.detach() - grads[0]).abs().max()) / scale)
    print("recorded forward + backward vs the first eager pass:", dev_list)
if os.environ.get("TRACE_CONV"):
    # checksum every conv3x3 launch's input and output in a few more passes: where does the first difference appear ?
    from glenet_amd import conv2d as c2
    real = c2._run
    log = []

    def spy(x, pack, cout, bn=None):
        out = real(x, pack, cout, bn)
        y = out[0] if isinstance(out, tuple) else out
        log.append((tuple(x.shape), cout, bn is not None, x.double().sum().item(), x.permute(0, 2, 3, 1).reshape(-1)[::97].double().abs().sum().item(),
                    y.double().sum().item(), y.permute(0, 2, 3, 1).reshape(-1)[::97].double().abs().sum().item()))
        return out
    c2._run = spy
    logs = []
    for it in range(8):
        log = []
        ref.load(*batch)
        ref.enqueue()
        torch.cuda.synchronize()
        logs.append(log)
    c2._run = real
    for it in range(1, len(logs)):
        for k, (a_, b_) in enumerate(zip(logs[0], logs[it])):
            if a_ != b_:
                print("pass %d: first differing conv launch #%d %s: input sums equal %s, output sums equal %s" % (
                    it, k, a_[:3], a_[3:5] == b_[3:5], a_[5:] == b_[5:]))
                break
        else:
            print("pass %d: all %d conv launches identical" % (it, len(logs[0])))
if os.environ.get("WHERE"):
    # per parameter (forward order): deviation of the most deviating pass from the first, relative to that parameter's gradient
    worst = max(range(1, len(grads)), key=lambda i: float((grads[i] - grads[0]).abs().max()))
    off = 0
    for name, prm in m.named_parameters():
        n = prm.numel()
        d = float((grads[worst][off:off + n] - grads[0][off:off + n]).abs().max())
        s_ = float(grads[0][off:off + n].abs().max())
        off += n
        if d > 1e-5 * max(s_, 1e-12):
            print("%-58s rel dev %.2e (abs %.2e of %.2e)" % (name, d / max(s_, 1e-12), d, s_))
if os.environ.get("BNSTATE"):
    from glenet_amd.spconv import core as spc
    for key, st in spc._BN_STATES.items():
        d = st[:16 * 2 * 512 * 8].view(torch.float64).view(16, 2, 512)
        nz = (d != 0).nonzero()
        print("BnState", key, "non-zero accumulators after the passes:", nz.shape[0],
              nz[:6].tolist(), [float(d[tuple(i)]) for i in nz[:6].tolist()])
if os.environ.get("TRACE_BWD"):
    # keep every conv3x3 launch's input and output of several passes; report launches whose OUTPUT deviates from pass 0 by
    # more than 1e-4 of its scale while its INPUT does not -- and where in the map
    from glenet_amd import conv2d as c2
    real = c2._run
    cur = []

    def spy2(x, pack, cout, bn=None):
        out = real(x, pack, cout, bn)
        y = out[0] if isinstance(out, tuple) else out
        cur.append((x.detach().clone(), y.detach().clone(), bn is not None))
        return out
    c2._run = spy2
    runs = []
    for it in range(int(os.environ.get("NB", "6"))):
        cur = []
        ref.load(*batch)
        ref.enqueue()
        torch.cuda.synchronize()
        runs.append(cur)
    c2._run = real
    for it in range(1, len(runs)):
        msgs = []
        for k, ((x0_, y0_, s0), (x1_, y1_, s1)) in enumerate(zip(runs[0], runs[it])):
            dx_ = float((x0_ - x1_).abs().max()) / max(float(x0_.abs().max()), 1e-30)
            dy_ = float((y0_ - y1_).abs().max()) / max(float(y0_.abs().max()), 1e-30)
            if dy_ > 1e-4:
                bad = ((y0_ - y1_).abs() > 1e-4 * y0_.abs().max()).any(1)          # (B, H, W)
                idx = bad.nonzero()
                msgs.append("launch %d %s stats=%s: input dev %.1e output dev %.1e; %d pixels, b %s y %d..%d x %d..%d" % (
                    k, tuple(x0_.shape), s0, dx_, dy_, idx.shape[0], sorted(set(idx[:, 0].tolist())), int(idx[:, 1].min()),
                    int(idx[:, 1].max()), int(idx[:, 2].min()), int(idx[:, 2].max())))
        print("pass %d:" % it, msgs[:3] if msgs else "no launch deviates")
