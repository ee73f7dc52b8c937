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
    # "auto" resolves the layout ONCE per checkpoint from the non-square weights and applies it to the square
    # ones too (a per-tensor decision loads square 2.x-native weights untransposed)
    assert ck.infer_layout(dst, state) == layout
    dst2 = _model(residual, 2)
    ck.load_params(dst2, state, layout="auto")
    for k in keys:
        assert torch.equal(src.state_dict()[k], dst2.state_dict()[k]), k


def test_auto_layout_refuses_mixed_checkpoints_and_warns_without_evidence():
    src, dst = _model(False, 0), _model(False, 1)
    keys = ck.find_all_spconv_keys(src)
    state = _as_layout(src.state_dict(), keys, "spconv2_native")
    state["conv_out.0.weight"] = src.state_dict()["conv_out.0.weight"].permute(4, 0, 1, 2, 3).contiguous()
    with pytest.raises(ValueError, match="mixes"):
        ck.load_params(dst, state)
    # only square weights in the file: nothing identifies the layout -> 1.x, said aloud
    square = {k: v for k, v in src.state_dict().items() if k not in keys or v.shape[3] == v.shape[4]}
    with pytest.warns(UserWarning, match="identifies"):
        assert ck.infer_layout(dst, square) == "spconv1"
    with pytest.raises(ValueError):
        ck.to_spconv1_layout(src.state_dict()["conv_out.0.weight"], (3, 1, 1, 64, 128), "auto")


def _ref_keys(tag):
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_state_keys.npz"))
    return {str(k): tuple(json.loads(str(s))) for k, s in zip(z[tag + "_keys"], z[tag + "_shapes"])}, \
        [str(m) for m in z[tag + "_modules"]]


def test_glenet_vr_state_dict_equals_the_reference_networks_key_for_key():
    """tests/golden/ref_state_keys.npz = every state-dict key + shape of the network the REFERENCE builds from
    tools/cfgs/kitti_models/GLENet_VR.yaml over our spconv (tools/ref_dropin_check.py --write, run in the build
    container).  Exact set comparison, shapes included; sparse-conv weights are (k,k,k,Cin,Cout) on both sides
    because the reference's modules were built on glenet_amd.spconv."""
    from glenet_amd import glenet_vr as gvr
    ref, modules = _ref_keys("glenet_vr")
    assert modules == ["MeanVFE", "VoxelBackBone8x", "HeightCompression", "BaseBEVBackbone", "AnchorHeadSingle",
                       "VoxelRCNNKLLabelIoUHead"]
    ours = {k: tuple(v.shape) for k, v in gvr.GLENetVR(synth.KITTI).state_dict().items()}
    ref.pop("global_step")                     # Detector3DTemplate's step counter (detector3d_template.py:22), not a weight
    assert set(ours) == set(ref), (sorted(set(ours) - set(ref))[:5], sorted(set(ref) - set(ours))[:5])
    wrong = {k: (ours[k], ref[k]) for k in ref if ours[k] != ref[k]}
    assert not wrong, wrong
    assert len(ref) == 272


def test_sparse_backbones_state_dicts_equal_the_reference_ones():
    """backbone_3d.* keys of the reference's VoxelBackBone8x (GLENet-VR) and VoxelResBackBone8x (Waymo CenterPoint)."""
    for tag, residual, cin in (("glenet_vr", False, 4), ("waymo_centerpoint_res", True, 5)):
        ref, _ = _ref_keys(tag)
        ref = {k[len("backbone_3d."):]: s for k, s in ref.items() if k.startswith("backbone_3d.")}
        torch.manual_seed(0)
        grid = gb.gv.grid_size_of(synth.KITTI["point_cloud_range"], synth.KITTI["voxel_size"])
        ours = {k: tuple(v.shape) for k, v in gb.SparseBackbone8x(cin, grid, residual=residual).state_dict().items()}
        assert ours == ref, (tag, sorted(set(ours) ^ set(ref))[:6])


def test_glenet_vr_loader_reports_foreign_and_misshapen_tensors():
    from glenet_amd import glenet_vr as gvr
    m = gvr.GLENetVR(synth.KITTI)
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
