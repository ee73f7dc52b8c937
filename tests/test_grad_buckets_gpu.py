"""The data-parallel step's gradient exchange in two buckets on ONE GPU: the forward + backward recorded as two
graphs cut where everything but the sparse backbone's gradients is final must be the same step as the one-graph recording."""
import pytest
import torch

import test_train_step_gpu as helpers

pytestmark = pytest.mark.gpu


def _pipe(dev, caps, npts, buckets):
    from glenet_amd import glenet_vr as gvr
    m = helpers._small_model(dev)
    R, P = m.roi_cfg["NMS_TRAIN"][1], m.roi_cfg["TARGET"]["ROI_PER_IMAGE"]
    gen = torch.Generator(device=dev).manual_seed(4)
    m.fixed_draws = (torch.rand((2, R), device=dev, generator=gen), torch.rand((2, P), device=dev, generator=gen))
    pipe = gvr.StaticTrainStep(m, 2, npts, max_gt=16, lr=1e-3, seed_rois_with_gt=helpers.JIT, capacities=caps)
    return pipe, buckets


def test_two_bucket_recording_is_the_one_graph_step(dev):
    from glenet_amd import glenet_vr as gvr
    batches = [helpers._batch(dev, [60, 61], 6000), helpers._batch(dev, [62, 63], 6000)]
    npts = max(b[0].shape[0] for b in batches) + 700
    probe = gvr.StaticTrainStep(helpers._small_model(dev), 2, npts, max_gt=16, lr=1e-3, seed_rois_with_gt=helpers.JIT)
    caps = {}
    for b in batches:
        for k, v in probe.calibrate(b[0], b[1]).items():
            caps[k] = max(caps.get(k, 0), v)
    del probe
    results = {}
    for buckets in (1, 2):
        pipe, _ = _pipe(dev, caps, npts, buckets)
        pipe.load(*batches[0])
        pipe.capture(split=True, buckets=buckets)
        assert (pipe.graph2 is not None) == (buckets == 2)
        losses, grads = [], None
        for i in range(3):
            pipe.load(*batches[i % 2])
            losses.append(float(pipe.step()))
            if i == 0:
                torch.cuda.synchronize()
                grads = pipe.step_optimizer.flat_grad.detach().clone()
                params = pipe.step_optimizer.flat_param.detach().clone()
        torch.cuda.synchronize()
        pipe.check()
        results[buckets] = (losses, grads, params, pipe)
    (l1, g1, p1, pipe1), (l2, g2, p2, pipe2) = results[1], results[2]
    early, late, ei, li = pipe2._bucket_plan
    n = pipe2.step_optimizer.n
    assert sum(hi - lo for lo, hi in early) + (late[1] - late[0]) == n and len(ei) + len(li) == len(pipe2.step_optimizer.params)
    assert late[1] - late[0] > 0 and early                                   # both buckets exist
    scale = float(g1.abs().max())
    assert float((g1 - g2).abs().max()) <= 2e-4 * scale                      # (float atomics reorder sums by ~1e-7 relative)
    # the first step is the same step; from then on two free-running Adam trajectories (noise-level gradient elements step the
    # other way now and then: the recorded-graph tests of tests/test_train_step_gpu.py see the same)
    for (a, b), tol in zip(zip(l1, l2), (1e-5, 1e-4, 5e-3)):
        assert abs(a - b) <= tol * max(1.0, abs(a)), (l1, l2)
    # the parameters after the FIRST update (Adam's first step moves every element by ~lr sign(g): a noise-level gradient element
    # whose sign differs moves the other way -- 2 lr --, everything else agrees to rounding: the bounds of tests/test_dist_gpu.py)
    assert float((p1 - p2).abs().max()) <= 2.5e-3 and float((p1 - p2).abs().mean()) <= 2e-6
