"""Checkpoint wire format: sparse-conv weights stored in any of the three spconv layouts load
into glenet_amd modules (the reference's _load_state_dict, detector3d_template.py:366-395, restated for the
direction this build needs), parameter names of GLENetVR are the reference's, and -- on the device -- a backbone
loaded from a spconv-2.x-layout state dict computes exactly what the source model computes."""
import numpy as np
import pytest
import torch

from glenet_amd import backbone as gb, checkpoint as ck, synth


def _reference_1x_to_2x(val, want_shape):
    """detector3d_template.py:377-384 verbatim in behaviour: a 1.x tensor adapted to a 2.x model's shape."""
    native = val.transpose(-1, -2)
    if tuple(native.shape) == tuple(want_shape):
        return native.contiguous()
    implicit = val.permute(4, 0, 1, 2, 3)
    if tuple(implicit.shape) == tuple(want_shape):
        return implicit.contiguous()
    return val


def _as_layout(state, keys, layout):
    out = {}
    for k, v in state.items():
        if k in keys:
            kk, ci, co = v.shape[:3], v.shape[3], v.shape[4]
            want = (*kk, co, ci) if layout == "spconv2_native" else (co, *kk, ci)
            v = _reference_1x_to_2x(v, want) if layout == "spconv2_native" else v.permute(4, 0, 1, 2, 3).contiguous()
        out[k] = v.clone()
    return out


def _model(residual=False, seed=0):
    torch.manual_seed(seed)
    K = synth.KITTI
    grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    return gb.SparseBackbone8x(4, grid, residual=residual)


@pytest.mark.parametrize("layout", ["spconv2_native", "spconv2_implicit", "spconv1"])
@pytest.mark.parametrize("residual", [False, True])
def test_every_spconv_layout_loads_back_to_the_same_weights(layout, residual):
    src, dst = _model(residual, 0), _model(residual, 1)
    keys = ck.find_all_spconv_keys(src)
    assert len(keys) == (21 if residual else 12) and all(k.endswith(".weight") for k in keys)
    state = src.state_dict() if layout == "spconv1" else _as_layout(src.state_dict(), keys, layout)
    if layout == "spconv2_implicit":
        assert tuple(state["conv2.0.0.weight"].shape) == (32, 3, 3, 3, 16)
    loaded, missing = ck.load_params(dst, {"model_state": state, "epoch": 80}, layout=layout)
    assert not missing and set(loaded) == set(src.state_dict())
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # "auto" tells every layout apart that shapes can tell apart; a square 2.x-native weight looks like a 1.x one
    # (the reference's loader has the same blind spot) -- non-square ones are converted
    dst2 = _model(residual, 2)
    ck.load_params(dst2, state, layout="auto")
    for k in keys:
        a, b = src.state_dict()[k], dst2.state_dict()[k]
        square = a.shape[3] == a.shape[4]
        if layout == "spconv2_native" and square:
            assert torch.equal(b, a.transpose(-1, -2))
        else:
            assert torch.equal(a, b), k


def test_glenet_vr_parameter_names_are_the_reference_ones():
    from glenet_amd import glenet_vr as gvr
    m = gvr.GLENetVR(synth.KITTI)
    names = set(m.state_dict())
    for k in ("backbone_3d.conv_input.0.weight", "backbone_3d.conv4.2.1.running_var", "backbone_3d.conv_out.0.weight",
              "backbone_2d.blocks.0.1.weight", "backbone_2d.blocks.1.16.weight", "backbone_2d.deblocks.1.1.bias",
              "dense_head.conv_cls.bias", "dense_head.conv_box.weight", "dense_head.conv_dir_cls.weight",
              "roi_head.roi_grid_pool_layers.0.mlps_in.0.0.weight", "roi_head.roi_grid_pool_layers.2.mlps_pos.0.1.running_mean",
              "roi_head.roi_grid_pool_layers.1.mlps_out.0.1.weight", "roi_head.shared_fc_layer.0.weight",
              "roi_head.shared_fc_layer.4.weight", "roi_head.cls_fc_layers.5.running_mean", "roi_head.cls_pred_layer.bias",
              "roi_head.reg_fc_layers.0.weight", "roi_head.reg_pred_layer.weight", "roi_head.reg_std_layer.weight",
              "roi_head.reg_std_bn.running_var", "roi_head.reg_std_fc1.bias", "roi_head.reg_std_bn1.weight",
              "roi_head.reg_std_fc2.weight"):
        assert k in names, k
    assert tuple(m.state_dict()["roi_head.shared_fc_layer.0.weight"].shape) == (256, 20736)
    assert tuple(m.state_dict()["roi_head.roi_grid_pool_layers.0.mlps_pos.0.0.weight"].shape) == (32, 3, 1, 1)
    # a checkpoint with foreign keys and a wrong-shaped tensor: those are reported, the rest loads
    state = {k: v.clone() for k, v in m.state_dict().items()}
    state["global_step"] = torch.zeros(1)
    state["dense_head.conv_cls.weight"] = torch.zeros(3, 3)
    loaded, missing = ck.load_params(gvr.GLENetVR(synth.KITTI), state)
    assert missing == ["dense_head.conv_cls.weight"] and "global_step" not in loaded


@pytest.mark.gpu
def test_backbone_loaded_from_a_spconv2_checkpoint_computes_the_same(dev):
    K = synth.KITTI
    src = _model(False, 0).to(dev).eval()
    state = _as_layout({k: v.cpu() for k, v in src.state_dict().items()}, ck.find_all_spconv_keys(src), "spconv2_implicit")
    dst = _model(False, 5).to(dev).eval()
    ck.load_params(dst, state)
    frames = [synth.kitti_frame(60 + i, num_points=6000)[0] for i in range(2)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    outs = []
    with torch.no_grad():
        for m in (src, dst):
            bd = gb.voxelize_batch(pts, bidx, 2, K, train=False)
            outs.append(gb.HeightCompression()(m(gb.MeanVFE()(bd)))["spatial_features"])
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0
