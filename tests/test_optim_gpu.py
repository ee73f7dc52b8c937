"""glenet_amd.optim.FlatAdamW (csrc/glx_optim.hip: gradient-norm clipping + AdamW on flat buffers, two launches)
against the plain PyTorch fp32 reference of the same update: torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW
(the arithmetic of tools/train_utils/train_utils.py:38-39 with the adam_onecycle optimiser)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(dev):
    torch.manual_seed(0)
    shapes = [(64, 33), (7,), (3, 3, 3, 16, 32), (1,), (256, 20736 // 64), (5, 5)]
    a = [torch.nn.Parameter(torch.randn(s, device=dev) * 0.3) for s in shapes]
    # convolution weights kept in channels-last memory (GLENetVR's 2-D backbone): the flat views keep the strides
    a += [torch.nn.Parameter((torch.randn(s, device=dev) * 0.3).contiguous(memory_format=torch.channels_last))
          for s in [(32, 16, 3, 3), (8, 16, 1, 1)]]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b


@pytest.mark.parametrize("max_norm", [10.0, 0.5, None])
def test_flat_adamw_matches_torch(dev, max_norm):
    from glenet_amd.optim import FlatAdamW
    ours, ref = _models(dev)
    opt = FlatAdamW(ours, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01, max_norm=max_norm)
    topt = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    assert all(p.data_ptr() >= opt.flat_param.data_ptr() for p in ours)          # parameters live in the flat buffer
    assert ours[6].is_contiguous(memory_format=torch.channels_last) and not ours[6].is_contiguous()
    assert opt.grad_views[6].stride() == ours[6].stride()
    g = torch.Generator(device=dev).manual_seed(1)
    for it in range(6):
        lr, b1 = 3e-3 * (1 + it), 0.9 - 0.01 * it                                   # a moving schedule
        opt.set_lr(lr, b1)
        for grp in topt.param_groups:
            grp["lr"], grp["betas"] = lr, (b1, 0.99)
        for p, q in zip(ours, ref):
            gr = torch.randn(p.shape, device=dev, generator=g) * (3.0 if it % 2 else 0.05)
            p.grad, q.grad = gr.clone(), gr.clone()
        want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm) if max_norm else None
        topt.step()
        opt.step()
        if max_norm:
            np.testing.assert_allclose(float(opt.grad_norm), float(want_norm), rtol=1e-6)
        for p, q in zip(ours, ref):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-5, atol=2e-7)
    assert int(opt.step_count) == 6


def test_flat_l2_norm_sum_and_its_gradient_equal_autograd(dev):
    """FlatAdamW.l2_norm_sum / add_l2_norm_grad (csrc/glx_optim.hip: the sum of the parameter tensors' 2-norms from the flat buffer
    and its gradient added into the flat gradient buffer, three launches) == torch's norms and their autograd gradients: a subset
    of the tensors, a channels-last tensor, a ZERO tensor (gradient 0, as torch's norm backward), a one-element tensor, with an
    upstream gradient as a device scalar and on top of gradients that are already in the buffer."""
    from glenet_amd.optim import FlatAdamW
    ours, ref = _models(dev)
    with torch.no_grad():
        ours[1].zero_()
        ref[1].zero_()
    opt = FlatAdamW(ours, lr=1e-3)
    with torch.no_grad():
        opt.flat_param[opt.offsets[1]:opt.offsets[1] + 8] = 0          # (the zero tensor and its alignment padding)
    pick = [0, 1, 3, 4, 6, 7]
    got = opt.l2_norm_sum([ours[i] for i in pick], 1e-4)
    want = 1e-4 * torch.stack([ref[i].norm(2) for i in pick]).sum()
    np.testing.assert_allclose(float(got), float(want.detach()), rtol=2e-6)
    (2.5 * want).backward()
    base = torch.randn(opt.n, device=dev) * 1e-5                       # (the size of the term that is added)
    opt.flat_grad.copy_(base)
    opt.add_l2_norm_grad(torch.tensor([2.5], device=dev))
    for i, (v, q) in enumerate(zip(opt.grad_views, ref)):
        o = opt.offsets[i]
        b = FlatAdamW._view(base, o, ours[i])
        if i in pick:
            np.testing.assert_allclose(v.cpu().numpy(), (b + q.grad).cpu().numpy(), rtol=1e-5, atol=1e-11)
        else:
            assert torch.equal(v, b)
    assert float((opt.grad_views[1] - FlatAdamW._view(base, opt.offsets[1], ours[1])).abs().max()) == 0.0
    with pytest.raises(ValueError):
        opt.l2_norm_sum([torch.nn.Parameter(torch.ones(3, device=dev))])


def test_flat_adamw_replays_in_a_graph_and_handles_missing_grads(dev):
    from glenet_amd.optim import FlatAdamW
    ours, ref = _models(dev)
    opt = FlatAdamW(ours, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0, max_norm=1.0)
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0)
    grads = [torch.randn(p.shape, device=dev) for p in ours]
    grads[1] = None                                     # a parameter the step did not reach
    for p, q, gr in zip(ours, ref, grads):
        p.grad = None if gr is None else gr.clone()
        q.grad = None if gr is None else gr.clone()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        opt.step()                                      # warm-up (step 1)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    from glenet_amd import _lib
    graph = _lib.new_graph()
    with torch.cuda.graph(graph, stream=side):
        opt.step()
    for _ in range(3):
        graph.replay()                                  # steps 2..4 (same gradients)
    torch.cuda.synchronize(dev)
    for _ in range(4):
        for q, gr in zip(ref, grads):
            q.grad = None if gr is None else gr.clone()
        torch.nn.utils.clip_grad_norm_([q for q in ref if q.grad is not None], 1.0)
        topt.step()
    assert int(opt.step_count) == 4
    for i, (p, q) in enumerate(zip(ours, ref)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=5e-5, atol=5e-7,
                                   err_msg="parameter %d" % i)


@pytest.mark.parametrize("shared", [False, True])
def test_conv_weight_gradient_lands_in_the_flat_buffer(dev, shared):
    """_lib.grad_buffer: the own 3x3 convolution writes dW straight into FlatAdamW's gradient view (the parameter's .grad
    IS the view: pack_grads has nothing to gather), every step anew.  The view is lent ONCE per optimizer step: a weight
    that two layers share gets its second gradient in a tensor of its own, the engine sums the two, and the flat buffer
    ends with that sum through the gather."""
    from glenet_amd import conv2d
    from glenet_amd.optim import FlatAdamW
    torch.manual_seed(3)
    w0 = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    x = torch.randn(2, 64, 24, 16, device=dev).contiguous(memory_format=torch.channels_last)

    def loss(a, b):
        return conv2d.conv3x3(torch.relu(conv2d.conv3x3(x, a)), b).square().mean()
    pa, pb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w1.clone())
    loss(pa, pa if shared else pb).backward()
    want = pa.grad.clone()
    oa, ob = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(w1.clone())
    opt = FlatAdamW([oa, ob], lr=0.0, weight_decay=0.0, max_norm=None)
    view = oa._glx_grad_view
    assert view.stride() == oa.stride() and not view.is_contiguous()
    for step in range(2):
        oa.grad = ob.grad = None
        loss(oa, oa if shared else ob).backward()
        np.testing.assert_allclose(oa.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-8)
        if not shared:
            assert oa.grad.data_ptr() == view.data_ptr() and oa.grad.stride() == view.stride(), step
            assert ob.grad.data_ptr() == ob._glx_grad_view.data_ptr(), step
        opt.pack_grads()
        np.testing.assert_allclose(view.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-8)
