"""Build-container check: the reference's OWN Python loads and builds its networks on top of the drop-in.

Needs /root/reference (never present on the GPU box, never shipped): `python tools/ref_dropin_check.py [--write]`.

What it does, in one fresh interpreter:
  1. `glenet_amd.dropin.install()` -- spconv / cumm and the six compiled-extension module names of
     setup.py:58-125 resolve to glenet_amd;
  2. imports `pcdet.models`, `pcdet.datasets.processor.data_processor`, `pcdet.utils.spconv_utils` from
     /root/reference UNMODIFIED (tools/train.py's import chain);
  3. reads tools/cfgs/kitti_models/GLENet_VR.yaml (and GLENet_S / GLENet_C, the Waymo CenterPoint res-backbone
     config) with the reference's own `cfg_from_yaml_file` and builds every network with its own
     `build_network` -- VoxelBackBone8x / VoxelResBackBone8x, HeightCompression, BaseBEVBackbone, AnchorHeadSingle /
     AnchorHeadKLLabel*, VoxelRCNNKLLabelIoUHead, CenterHead -- on CPU over OUR spconv classes;
  4. runs what can run without a GPU: `spconv_utils.find_all_spconv_keys`, the reference's checkpoint layout
     conversion `_load_state_dict` on a state dict in the other spconv layout, and the reference's
     `DataProcessor.transform_points_to_voxels` on a synthetic KITTI frame (our host voxelizer behind
     spconv.utils.Point2VoxelCPU3d) against the oracle;
  5. with --write: stores every state-dict key + shape of those networks in tests/golden/ref_state_keys.npz
     (data, not source) -- tests/test_checkpoint.py compares glenet_amd.glenet_vr.GLENetVR against it exactly.

Disclosed placeholders (third-party packages this image lacks, none of them on the path; each is an empty module
whose attributes are do-nothing decorators / Warning classes): SharedArray (imported, never used,
pcdet/utils/common_utils.py:7), numba (+ numba.core.errors; `@numba.jit` on host evaluation helpers), skimage
(image transforms of the camera datasets), easydict (EasyDict = dict with attribute access: restated in 12 lines
below), kornia / torchvision where absent.  `pcdet/__init__.py` is skipped by a bare package object (it imports
a version.py that setup.py generates).  `torch.Tensor.cuda` is a no-op for the duration of the build (anchor
generation calls `.cuda()` in constructors, anchor_head_template.py:44) -- nothing is computed in it.
"""
import importlib
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_state_keys.npz")


class _AnyWarning(Warning):
    pass


class _Placeholder(types.ModuleType):
    """An uninstalled third-party package: every attribute is a Warning class or a pass-through decorator."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name.endswith("Warning"):
            return _AnyWarning

        def passthrough(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        return passthrough


class EasyDict(dict):
    """dict with attribute access, nested dicts converted on the way in (what `easydict.EasyDict` is)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kw):
        for k, v in dict(other or {}, **kw).items():
            self[k] = v


def prepare_imports(install=None):
    """Steps 1-2.  Returns the list of placeholder names that were needed.
    install: callable that registers the extension / spconv module names (default glenet_amd.dropin.install; the
    whole-step golden passes oracle.refshim.install -- the CPU stand-ins backed by the oracle)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    if install is None:
        import glenet_amd.dropin as dropin
        install = dropin.install
    install()
    pk = types.ModuleType("pcdet")
    pk.__path__ = [os.path.join(REF, "pcdet")]
    sys.modules["pcdet"] = pk
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    placeholders = []
    try:
        import easydict  # noqa: F401
    except ModuleNotFoundError:
        sys.modules["easydict"] = ed
        placeholders.append("easydict")
    targets = ("pcdet.models", "pcdet.datasets.processor.data_processor", "pcdet.utils.spconv_utils", "pcdet.config")
    for _ in range(40):
        try:
            for t in targets:
                importlib.import_module(t)
            return placeholders
        except ModuleNotFoundError as e:
            if e.name is None or e.name.startswith("pcdet") or e.name.startswith("glenet_amd") \
                    or e.name.startswith("spconv") or e.name.startswith("cumm"):
                raise           # a module the drop-in should have served: that is the failure this script exists for
            placeholders.append(e.name)
            sys.modules[e.name] = _Placeholder(e.name)
            for k in [k for k in sys.modules if k.startswith("pcdet.") and not k.endswith("_cuda")]:
                del sys.modules[k]
    raise RuntimeError("import did not converge: %s" % placeholders)


class _FakeDataset:
    """The five attributes Detector3DTemplate.build_networks reads (detector3d_template.py:36-45)."""

    def __init__(self, data_cfg, class_names, num_point_features):
        import numpy as np
        self.class_names = class_names
        self.point_feature_encoder = types.SimpleNamespace(num_point_features=num_point_features)
        self.point_cloud_range = np.array(data_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        vox = None
        for p in data_cfg.DATA_PROCESSOR:
            if p.NAME == "transform_points_to_voxels":
                vox = p.VOXEL_SIZE
        self.voxel_size = vox
        gs = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(vox)
        self.grid_size = np.round(gs).astype(np.int64)
        self.depth_downsample_factor = None


def build_reference_network(cfg_rel, num_point_features, edit=None):
    """Step 3: the reference's own config loader and network builder.  edit(cfg): optional change of config VALUES
    (a reduced point-cloud range for the whole-step golden) between loading and building."""
    import torch
    from pcdet.config import cfg_from_yaml_file
    from pcdet.models import build_network
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "tools"))          # _BASE_CONFIG_ paths are relative to tools/
    try:
        cfg = cfg_from_yaml_file(cfg_rel, EasyDict())
    finally:
        os.chdir(cwd)
    if edit is not None:
        edit(cfg)
    ds = _FakeDataset(cfg.DATA_CONFIG, cfg.CLASS_NAMES, num_point_features)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        net = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=ds)
    finally:
        torch.Tensor.cuda = real_cuda
    return cfg, ds, net


CONFIGS = {
    # tag: (config under tools/, point features)
    "glenet_vr": ("cfgs/kitti_models/GLENet_VR.yaml", 4),
    "glenet_s": ("cfgs/kitti_models/GLENet_S.yaml", 4),
    "glenet_c": ("cfgs/kitti_models/GLENet_C.yaml", 4),
    "waymo_centerpoint_res": ("cfgs/waymo_models/centerpoint.yaml", 5),
}


def check_spconv_side(net, report):
    """Step 4a: the reference's spconv helpers over our classes."""
    import torch
    import spconv.pytorch as spconv
    from pcdet.utils import spconv_utils
    import glenet_amd.spconv as ours
    assert spconv is sys.modules["glenet_amd.spconv.pytorch"] and spconv_utils.spconv is spconv
    keys = spconv_utils.find_all_spconv_keys(net)
    convs = [m for m in net.modules() if isinstance(m, ours.conv.SparseConvolution)]
    assert len(keys) == len(convs) > 0, (len(keys), len(convs))
    assert all(net.state_dict()[k].dim() == 5 for k in keys)
    report["spconv_weight_keys"] = len(keys)
    # replace_feature present -> the reference takes its spconv-2.x branch (spconv_utils.py:28-34)
    t = ours.SparseConvTensor.__new__(ours.SparseConvTensor)
    assert "replace_feature" in t.__dir__()
    # the reference's own loader converts a checkpoint written in the OTHER layout (detector3d_template.py:366-400)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k in keys:
        sd[k] = sd[k].permute(4, 0, 1, 2, 3).contiguous()          # spconv 2.x native (Cout, kd, kh, kw, Cin)
    before = {k: net.state_dict()[k].clone() for k in keys}
    state, updated = net._load_state_dict(sd, strict=False)
    assert all(torch.equal(net.state_dict()[k], before[k]) for k in keys)
    report["layout_conversion_by_reference_loader"] = "ok (%d keys)" % len(keys)


def check_data_processor(cfg, report):
    """Step 4b: DataProcessor.transform_points_to_voxels (data_processor.py:117-152) through our
    spconv.utils.Point2VoxelCPU3d (host library) == the oracle's hard voxelizer."""
    import numpy as np
    from pcdet.datasets.processor.data_processor import DataProcessor
    import oracle
    from glenet_amd import synth
    pcr = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, dtype=np.float32)
    dp = DataProcessor(cfg.DATA_CONFIG.DATA_PROCESSOR, point_cloud_range=pcr, training=True, num_point_features=4)
    pts, _ = synth.kitti_frame(0, num_points=20000)
    out = pts
    data = {"points": pts.copy(), "use_lead_xyz": True}
    vcfg = [p for p in cfg.DATA_CONFIG.DATA_PROCESSOR if p.NAME == "transform_points_to_voxels"][0]
    data = dp.transform_points_to_voxels(data_dict=data, config=vcfg)
    v, c, n = oracle.voxelize_hard(out, vcfg.VOXEL_SIZE, pcr, vcfg.MAX_POINTS_PER_VOXEL, vcfg.MAX_NUMBER_OF_VOXELS["train"])
    assert np.array_equal(data["voxel_coords"], c) and np.array_equal(data["voxel_num_points"], n)
    assert np.array_equal(data["voxels"], v)
    assert list(dp.grid_size) == [1408, 1600, 40]
    report["data_processor_voxels"] = int(len(c))


def check_accelerate(net):
    """glenet_amd.dropin.accelerate() on the reference's own network: its BaseBEVBackbone is re-classed to our BEVBackbone,
    the state-dict keys stay, and (on CPU, where both run torch's kernels layer by layer) the output stays."""
    import torch
    from glenet_amd import dropin
    bev = [m for m in net.modules() if type(m).__name__ == "BaseBEVBackbone"][0]
    keys = list(net.state_dict().keys())
    x = torch.randn(1, bev.blocks[0][1].in_channels, 16, 24)
    was = bev.training
    bev.eval()
    with torch.no_grad():
        want = bev({"spatial_features": x})["spatial_features_2d"].clone()
    changed = dropin.accelerate(net)
    with torch.no_grad():
        got = bev({"spatial_features": x})["spatial_features_2d"]
    bev.train(was)
    assert type(bev).__module__ == "glenet_amd.dense_path" and list(net.state_dict().keys()) == keys
    # round 4: the reference's own HeightCompression, NeighborVoxelSAModuleMSG x 3 and ProposalTargetLayer instances are
    # re-classed, VoxelRCNNHead.roi_grid_pool / RoIHeadTemplate.proposal_layer are bound to the device paths
    expect = {"backbone_2d", "map_to_bev_module", "roi_head.proposal_target_layer", "roi_head.roi_grid_pool",
              "roi_head.proposal_layer"} | {"roi_head.roi_grid_pool_layers.%d" % i for i in range(3)}
    assert expect <= set(changed), sorted(expect - set(changed))
    assert type(net.map_to_bev_module).__module__ == "glenet_amd.backbone" and net.map_to_bev_module.defer
    assert type(net.roi_head.proposal_target_layer).__module__ == "glenet_amd.roi_targets"
    assert net.roi_head.__dict__["_glx_pool"].grid_size == 6 and net.roi_head.__dict__["_glx_pool"].num_features == 96
    assert dropin.accelerate(net) == []                          # idempotent
    err = float((got - want).abs().max())
    assert err < 1e-5, err
    return {"modules": changed, "class": type(bev).__module__ + "." + type(bev).__name__, "max_abs_diff_cpu": err}


def check_record(net, ds):
    """glenet_amd.dropin.record(dry_run=True) on the reference's own GLENet-VR network: the configuration the reference's loader
    left in `net.model_cfg` translates to exactly the constants of glenet_amd.glenet_vr (GLENet_VR.yaml), and the twin that
    the recorded step runs holds the network's OWN Parameter and buffer objects under the same state-dict keys."""
    import torch
    from glenet_amd import dropin
    from glenet_amd import glenet_vr as gvr
    net.dataset = ds
    rep = dropin.record(net, dry_run=True)
    assert rep["roi_cfg"] == gvr.ROI_HEAD_CFG, {k: (rep["roi_cfg"][k], gvr.ROI_HEAD_CFG[k]) for k in gvr.ROI_HEAD_CFG if rep["roi_cfg"][k] != gvr.ROI_HEAD_CFG[k]}
    assert rep["head_cfg"] == gvr.DENSE_HEAD_CFG, (rep["head_cfg"], gvr.DENSE_HEAD_CFG)
    twin = rep["twin"]
    mine, theirs = dict(twin.named_parameters()), dict(net.named_parameters())
    assert list(twin.state_dict().keys()) == [k for k in net.state_dict().keys() if k != "global_step"]
    assert all(mine[k] is theirs[k] for k in theirs) and len(mine) == len(theirs)
    bm, bt = dict(twin.named_buffers()), {k: v for k, v in net.named_buffers() if k != "global_step"}
    assert all(bm[k] is bt[k] for k in bt) and len(bm) == len(bt)
    with torch.no_grad():                      # one object: a write through the network is seen by the twin
        p = net.dense_head.conv_cls.bias
        p.add_(1.0)
        assert torch.equal(twin.dense_head.conv_cls.bias, p)
        p.sub_(1.0)
    from glenet_amd import detector as det
    post_cfg, thresh = dropin._translate_post_cfg(net.model_cfg)             # record_inference()'s settings
    assert post_cfg == det.POST_PROCESSING_CFG and thresh == [0.3, 0.5, 0.7], (post_cfg, thresh)
    bad = dict(net.model_cfg)
    try:
        dropin._translate_cfg(EasyDict(dict(net.model_cfg, NAME="PVRCNN")))
        raise AssertionError("a foreign detector was accepted")
    except NotImplementedError:
        pass
    return {"shared_tensors": rep["shared"], "point_cloud_range": rep["cfg"]["point_cloud_range"], "voxel_size": rep["cfg"]["voxel_size"]}


def main(write=False):
    import numpy as np
    import torch  # noqa: F401
    placeholders = prepare_imports()
    report = {"placeholders": placeholders, "networks": {}}
    store = {}
    first_cfg = None
    for tag, (cfg_rel, nfeat) in CONFIGS.items():
        cfg, ds, net = build_reference_network(cfg_rel, nfeat)
        if first_cfg is None:
            first_cfg = cfg
        sd = net.state_dict()
        names = list(sd.keys())
        store[tag + "_keys"] = np.array(names)
        store[tag + "_shapes"] = np.array([json.dumps(list(v.shape)) for v in sd.values()])
        store[tag + "_dtypes"] = np.array([str(v.dtype) for v in sd.values()])
        store[tag + "_modules"] = np.array([type(m).__name__ for m in net.module_list])
        report["networks"][tag] = {"modules": [type(m).__name__ for m in net.module_list], "state_keys": len(names),
                                  "parameters": int(sum(p.numel() for p in net.parameters()))}
        if tag in ("glenet_vr", "waymo_centerpoint_res"):
            check_spconv_side(net, report["networks"][tag])
        if tag == "glenet_vr":
            report["networks"][tag]["record"] = check_record(net, ds)          # before accelerate() re-classes anything
            report["networks"][tag]["accelerate"] = check_accelerate(net)
    check_data_processor(first_cfg, report)
    # the import name every reference file sees is ours
    import pcdet.ops.pointnet2.pointnet2_batch.pointnet2_utils as pb
    assert pb.pointnet2.__name__.startswith("glenet_amd.")
    report["pointnet2_batch_cuda"] = pb.pointnet2.__name__
    if write:
        np.savez_compressed(GOLDEN, **store)
        report["written"] = os.path.relpath(GOLDEN, ROOT)
    report["store_digest"] = {k: int(len(v)) for k, v in store.items() if k.endswith("_keys")}
    print(json.dumps(report))
    return report, store


if __name__ == "__main__":
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to check")
        sys.exit(0)
    main(write="--write" in sys.argv)
