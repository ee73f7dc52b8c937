"""GLENet-VR (Voxel-RCNN with the KL / label-uncertainty RoI head): the composed training step named by
BASELINE config 3 -- voxelize + sparse backbone + BEV head + proposals (NMS 9000 -> 512) + RoI targets +
RoI-grid pooling + FC towers + {dense-head, RoI classification, KL regression, corner} losses, forward and
backward (+ gradient clipping and the AdamW update) -- as a product pipeline.

Our counterpart of the callers (never present at run time):
  tools/train_utils/train_utils.py:11-110            train_one_epoch: forward, loss.backward(), clip_grad_norm_, step
  pcdet/models/detectors/voxel_rcnn.py               module order + get_training_loss = loss_rpn + loss_rcnn
  pcdet/models/roi_heads/voxelrcnn_kl_label_iou_head.py:10-180   the head (reg_std branch, forward, KL loss)
  pcdet/models/roi_heads/voxelrcnn_head.py:8-191     FC towers, roi_grid_pool
  pcdet/models/roi_heads/roi_head_template.py:51-286 proposal_layer, assign_targets, get_loss
  tools/cfgs/kitti_models/GLENet_VR.yaml             every constant below

Module and parameter names equal the reference's (vfe, backbone_3d, map_to_bev_module, backbone_2d,
dense_head, roi_head.{roi_grid_pool_layers, shared_fc_layer, cls_fc_layers, cls_pred_layer, reg_fc_layers,
reg_pred_layer, reg_std_layer, reg_std_bn, reg_std_fc1, reg_std_bn1, reg_std_fc2}), so a GLENet-VR
checkpoint's keys map one to one (glenet_amd.checkpoint).

Two ways to run the same step:
  * GLENetVR.training_step(...)       exact shapes (host read-backs size the sparse tensors), eager launches;
  * StaticTrainStep                   shape-static, no host synchronisation anywhere, recorded once into a HIP
                                      graph and replayed with one call per step (two with a gradient exchange
                                      between backward and the optimizer at world size > 1).
Both produce the same loss terms and gradients (tests/test_train_step_gpu.py)."""
import contextlib
import math
import os

import torch
from torch import nn

from . import backbone as gb
from . import dense_path as dp
from . import detector as det
from . import losses, roi_grid as rg, roi_targets, target_assign

# GLENet_VR.yaml
ROI_HEAD_CFG = dict(
    POOL={n: dict(mlps=[[32, 32]], query_ranges=[[4, 4, 4]], radii=[r], nsamples=[16])
          for n, r in (("x_conv2", 0.4), ("x_conv3", 0.8), ("x_conv4", 1.6))},                # :117-139
    GRID_SIZE=6, SHARED_FC=(256, 256), CLS_FC=(256, 256), REG_FC=(256, 256), DP_RATIO=0.3,      # :95-100
    NMS_TRAIN=(9000, 512, 0.8), NMS_TEST=(2048, 100, 0.7),                                      # :101-115
    TARGET=dict(ROI_PER_IMAGE=128, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou",
                CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8,
                REG_FG_THRESH=0.55),                                                            # :140-153
    LOSS_WEIGHTS=dict(rcnn_cls_weight=1.0, rcnn_reg_weight=1.0, rcnn_corner_weight=1.0,
                      code_weights=[1.0] * 7))                                                  # :155-166
DENSE_HEAD_CFG = dict(anchor_sizes=[[3.9, 1.6, 1.56]], anchor_rotations=[0, 1.57], anchor_bottom_heights=[-1.78],
                      matched_threshold=0.6, unmatched_threshold=0.45,                          # :63-73
                      cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, code_weights=[1.0] * 7)   # :84-90
OPTIM_CFG = dict(LR=0.01, WEIGHT_DECAY=0.01, BETAS=(0.9, 0.99), GRAD_NORM_CLIP=10.0,            # :185-203
                 MOMS=(0.95, 0.85), PCT_START=0.4, DIV_FACTOR=10)


class VoxelRCNNKLHead(rg.RoIGridPool):
    """VoxelRCNNKLLabelIoUHead: RoI-grid pooling + shared / cls / reg FC towers + the log-variance branch
    `reg_std_layer` and the small tower on top of it whose sigmoid output rescales the classification score
    (voxelrcnn_kl_label_iou_head.py:10-37, forward :38-92)."""

    def __init__(self, backbone_channels, voxel_size, point_cloud_range, cfg=None, num_class=1, code_size=7):
        cfg = cfg or ROI_HEAD_CFG
        super().__init__(backbone_channels, cfg["POOL"], cfg["GRID_SIZE"], voxel_size, point_cloud_range)
        self.cfg, self.num_class, self.code_size = cfg, num_class, code_size
        pre = cfg["GRID_SIZE"] ** 3 * self.num_features
        self.shared_fc_layer, pre = dp._fc_tower(pre, cfg["SHARED_FC"], cfg["DP_RATIO"])
        self.cls_fc_layers, c = dp._fc_tower(pre, cfg["CLS_FC"], cfg["DP_RATIO"])
        self.cls_pred_layer = nn.Linear(c, num_class, bias=True)
        self.reg_fc_layers, c = dp._fc_tower(pre, cfg["REG_FC"], cfg["DP_RATIO"])
        self.reg_pred_layer = nn.Linear(c, code_size * num_class, bias=True)
        self.reg_std_layer = nn.Linear(c, code_size * num_class, bias=True)
        self.reg_std_bn = nn.BatchNorm1d(code_size * num_class)
        self.reg_std_fc1 = nn.Linear(code_size * num_class, 64, bias=True)
        self.reg_std_bn1 = nn.BatchNorm1d(64)
        self.reg_std_fc2 = nn.Linear(64, 1, bias=True)
        self.init_weights()

    def init_weights(self):
        """voxelrcnn_head.py:81-93 + voxelrcnn_kl_label_iou_head.py:30-36."""
        for tower in (self.shared_fc_layer, self.cls_fc_layers, self.reg_fc_layers):
            for m in tower.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_normal_(m.weight)
        nn.init.normal_(self.cls_pred_layer.weight, 0, 0.01)
        nn.init.constant_(self.cls_pred_layer.bias, 0)
        nn.init.normal_(self.reg_pred_layer.weight, mean=0, std=0.001)
        nn.init.constant_(self.reg_pred_layer.bias, 0)
        for m in (self.reg_std_layer, self.reg_std_fc1, self.reg_std_fc2):
            nn.init.normal_(m.weight, mean=0, std=0.0001)
            nn.init.constant_(m.bias, 0)

    def heads(self, pooled, raw=False):
        """pooled (R, G^3, C) -> rcnn_cls (R,1) [the rescaled logit], rcnn_reg (R,7), rcnn_reg_std (R,7).
        raw=True: (ori_cls, std_logit, rcnn_reg, rcnn_reg_std) -- the two logits of the rescaling, for the caller
        that fuses it with the classification loss (losses.cls_rescale_loss)."""
        x = pooled.reshape(pooled.shape[0], -1)
        if x.is_cuda and not self.training and not torch.is_grad_enabled() and x.dtype == torch.float32 and self.USE_FOLDED:
            # inference: eval-mode BatchNorm folded into each Linear and the 20736-wide first layer split along K
            # (dense_path.RoIFCStack's folded path -- the towers carry the same attribute names)
            shared_w, cls_w, reg_w = dp.RoIFCStack._folded(self)
            h = x.contiguous()
            for w, b in shared_w:
                h = dp.RoIFCStack._affine_relu(h, w, b)
            c = reg_feat = h
            for w, b in cls_w:
                c = dp.RoIFCStack._affine_relu(c, w, b)
            for w, b in reg_w:
                reg_feat = dp.RoIFCStack._affine_relu(reg_feat, w, b)
            ori_cls = self.cls_pred_layer(c)
        elif dp.fc_tower_usable(self, x):
            # training: the towers behind the first Linear as one launch per direction (csrc/glx_fctower.hip)
            ori_cls, std_logit, rcnn_reg, rcnn_reg_std = dp.fc_towers(self, x)
            if raw:
                return ori_cls, std_logit, rcnn_reg, rcnn_reg_std
            return losses.cls_rescale_torch(ori_cls, std_logit), rcnn_reg, rcnn_reg_std
        else:
            shared = self.shared_fc_layer(x)
            ori_cls = self.cls_pred_layer(self.cls_fc_layers(shared))
            reg_feat = self.reg_fc_layers(shared)
        rcnn_reg = self.reg_pred_layer(reg_feat)
        rcnn_reg_std = self.reg_std_layer(reg_feat)
        s = dp.bn_relu(self.reg_std_bn, rcnn_reg_std.clone())
        s = dp.bn_relu(self.reg_std_bn1, self.reg_std_fc1(s))
        std_logit = self.reg_std_fc2(s)
        if raw:
            return ori_cls, std_logit, rcnn_reg, rcnn_reg_std
        if ori_cls.is_cuda and not torch.is_grad_enabled():
            rcnn_cls = losses.cls_rescale(ori_cls, std_logit)                            # one launch
        else:
            rcnn_cls = losses.cls_rescale_torch(ori_cls, std_logit)                      # :73-75 ("ad hoc")
        return rcnn_cls, rcnn_reg, rcnn_reg_std

    keep_pooled = False       # tests: leave the pooled features of the last forward in `last_pooled`
    USE_FOLDED = True         # inference: BatchNorm folded into the FC towers

    def forward(self, rois, multi_scale_3d_features, multi_scale_3d_strides, batch_size, raw=False):
        pooled = super().forward(rois, multi_scale_3d_features, multi_scale_3d_strides, batch_size)
        if self.keep_pooled:
            self.last_pooled = pooled.detach()
        return self.heads(pooled, raw)


OVERLAP_ROI = os.environ.get("GLX_OVERLAP_ROI", "1") != "0"
STAGE_CUTS = os.environ.get("GLX_STAGE_CUTS", "1") != "0"
DEFER_FC_WGRADS = True
FC_WGRADS_BEHIND_ROI = True      # see StagedLoss.backward
# the slab / partial sums of the sparse and the BEV 3x3 weight gradients: one launch each at the end of the backward pass
# (spconv.core.DEFERRED_WGRAD_REDUCES, conv2d.DEFERRED_WGRAD_REDUCES) instead of one ~8 us launch per layer on the main chain
DEFER_WGRAD_REDUCES = os.environ.get("GLX_DEFER_WGRAD_REDUCES", "1") != "0"
# the RoI head's three loss terms as one launch / one autograd node (losses.roi_head_losses); 0 = the three entry points
ROI_LOSSES_ONE_LAUNCH = True


class StagedLoss:
    """The scalar of a training step whose backward() runs as three partial passes instead of one, so that the RoI
    branch (proposals, RoI targets, RoI-grid pooling, FC towers, RoI losses and their backward: ~350 short launches
    that never fill the chip) runs on a stream of its own BESIDE the dense-head loss and the BEV backbone's backward --
    the two meet again at the sparse backbone's feature tensors:
      A (RoI stream)   d roi_loss / d (x_conv features the RoI grid pools from -- handed over as detached leaves --,
                       RoI-head parameters)
      B (main stream)  d rpn_loss / d (BEV input = the sparse backbone's output -- a detached leaf too --, 2-D parameters)
      C (main stream, after the join)  the sparse backbone's backward from those feature tensors' gradients.
    One backward() from the summed scalar would compute the same gradients but the autograd engine's root gradient
    lives on the caller's stream: the RoI branch would wait for whatever the main stream has queued.  The reference
    has no counterpart (one stream, loss.backward(), tools/train_utils/train_utils.py:47-52).
    `value` (the detached sum) exists after backward()."""

    def __init__(self, rpn, roi, roi_stream, bev_cut, roi_cuts, stage_cuts, mark=None):
        """bev_cut: (sparse backbone's output features, the detached leaf the BEV backbone read);
        roi_cuts {level: (x_conv features, the leaf the RoI head read)};
        stage_cuts {level: (x_conv features, the leaf the backbone's next block read)} (SparseBackbone8x.stage_cuts)."""
        self.rpn, self.roi, self.roi_stream = rpn, roi, roi_stream
        self.bev_cut, self.roi_cuts, self.stage_cuts = bev_cut, roi_cuts, stage_cuts
        self.value, self.mark = None, mark
        # optional callable run between B and C, when every gradient but the sparse backbone's is final (the RoI branch joined,
        # the BEV backbone's weight-gradient sums done): StaticTrainStep's two-bucket gradient exchange starts its first bucket
        # there.  The price is the overlap of the RoI branch's tail with the upper sparse levels.
        self.boundary = None

    LEVELS = ("x_conv4", "x_conv3", "x_conv2", "x_conv1")       # the order their gradients leave the RoI branch

    def backward(self):
        dev = self.rpn.device
        main = torch.cuda.current_stream(dev)
        core = gb.spconv.core
        # an event per level on the RoI stream, recorded when that level's gradient has arrived at its leaf
        ready = {}
        for name, (_, leaf) in self.roi_cuts.items():
            ready[name] = ev = torch.cuda.Event()
            leaf.register_post_accumulate_grad_hook(lambda t, ev=ev: ev.record(torch.cuda.current_stream(t.device)))
        # While the RoI branch is in flight the convolutions' weight gradients stay on the main stream: with a third
        # branch (the weight-gradient stream) in the recorded graph the RoI branch and the BEV backward were executed
        # one after the other (measured with the stage stamps, ROCm 7.2's graph executor); two branches do overlap.
        wgrad_stream = core.WGRAD_STREAM
        # no weight-gradient side stream while the RoI branch is in flight: a third concurrent branch was measured at 8.0 ms
        # per step against 6.44 (round 4, also with two executor queues)
        core.WGRAD_STREAM = None
        from . import conv2d as c2
        sparse_sums = core.DEFERRED_WGRAD_REDUCES = [] if DEFER_WGRAD_REDUCES else None
        bev_sums = c2.DEFERRED_WGRAD_REDUCES = [] if DEFER_WGRAD_REDUCES else None
        try:
            fc_jobs = dp.DEFERRED_FC_WGRADS = [] if DEFER_FC_WGRADS else None
            dp.DEFERRED_FC_SAME_STREAM = FC_WGRADS_BEHIND_ROI
            with torch.cuda.stream(self.roi_stream):      # A: caller stream = RoI stream, nothing of the main stream
                try:
                    torch.autograd.backward(self.roi)     # is waited for; ends at the detached leaves
                finally:
                    dp.DEFERRED_FC_WGRADS = None
                    dp.DEFERRED_FC_SAME_STREAM = False
                if self.mark:
                    self.mark("backward: RoI head (RoI stream)")
                if fc_jobs and FC_WGRADS_BEHIND_ROI:
                    # the FC towers' weight gradients behind the x_conv gradients on the RoI stream: with the fused towers
                    # (csrc/glx_fctower.hip) the RoI branch ends ~0.25 ms before the main stream asks for its last gradient
                    dp.run_deferred_fc_wgrads(fc_jobs)
                    fc_jobs = None
            torch.autograd.backward(self.rpn)             # B: ends at the BEV input's detached leaf
            if fc_jobs:      # ... or in the main stream's wait for the RoI gradients (what paid while the RoI branch was longer)
                dp.run_deferred_fc_wgrads(fc_jobs)
            joined = False
            if self.boundary is not None:
                main.wait_stream(self.roi_stream)         # join: the first bucket holds the RoI head's gradients too
                joined = True
                c2.DEFERRED_WGRAD_REDUCES = None
                c2.run_deferred_wgrad_reduces(bev_sums)
                bev_sums = None
                self.boundary()                           # (a recorded step ends its first graph here and begins the second)
                core.WGRAD_STREAM = wgrad_stream
            # C: the sparse backbone, level by level: the levels above a stage cut run as soon as THEIR RoI gradients are
            # there (x_conv4's leave the RoI branch first, x_conv2's last)
            roots, grads = [self.bev_cut[0]], [self.bev_cut[1].grad]
            self.bev_cut[1].grad = None
            for name in self.LEVELS + tuple(n for n in self.roi_cuts if n not in self.LEVELS):
                if name in self.stage_cuts:
                    torch.autograd.backward(roots, grads)
                    if self.mark:
                        self.mark("backward: sparse backbone above " + name)
                    feat, leaf = self.stage_cuts[name]
                    roots, grads = [feat], [leaf.grad]
                    leaf.grad = None
                if name in self.roi_cuts:
                    feat, leaf = self.roi_cuts[name]
                    if leaf.grad is None:
                        continue
                    if not joined:
                        main.wait_event(ready[name])
                    same = [i for i, t in enumerate(roots) if t is feat]
                    if same:
                        grads[same[0]] = grads[same[0]] + leaf.grad
                    else:
                        roots.append(feat)
                        grads.append(leaf.grad)
                    leaf.grad = None
            if not joined:
                main.wait_stream(self.roi_stream)         # join (parameter gradients of the RoI head)
            self.value = self.rpn.detach() + self.roi.detach()
            self.rpn = self.roi = self.bev_cut = self.roi_cuts = self.stage_cuts = None
            core.WGRAD_STREAM = wgrad_stream              # free again: the RoI branch has been joined
            keep = [(t, g) for t, g in zip(roots, grads) if g is not None]
            torch.autograd.backward([t for t, _ in keep], [g for _, g in keep])
            core.DEFERRED_WGRAD_REDUCES = c2.DEFERRED_WGRAD_REDUCES = None
            if wgrad_stream is not None:
                main.wait_stream(wgrad_stream)            # products written on the side stream (the last sparse layers)
            c2.run_deferred_wgrad_reduces(bev_sums)       # every layer's partial sums -> its gradient: two launches
            core.run_deferred_wgrad_reduces(sparse_sums)
        finally:
            core.WGRAD_STREAM = wgrad_stream
            core.DEFERRED_WGRAD_REDUCES = c2.DEFERRED_WGRAD_REDUCES = None

    def detach(self):
        return self.value


class GLENetVR(nn.Module):
    """Detector3DTemplate module order of GLENet_VR.yaml."""

    def __init__(self, cfg, num_point_features=4, roi_cfg=None, head_cfg=None, bev_channels_last=True):
        """bev_channels_last: the BEV map leaves dense() in channels-last memory and the 2-D backbone runs NHWC
        (MIOpen's implicit-GEMM kernels without their NCHW<->NHWC transposes); values are layout-independent."""
        super().__init__()
        self.cfg = cfg
        self.roi_cfg, self.head_cfg = roi_cfg or ROI_HEAD_CFG, head_cfg or DENSE_HEAD_CFG
        grid = gb.gv.grid_size_of(cfg["point_cloud_range"], cfg["voxel_size"])
        self.vfe = gb.MeanVFE()
        self.backbone_3d = gb.VoxelBackBone8x(num_point_features, grid)
        self.map_to_bev_module = gb.HeightCompression()
        self.map_to_bev_module.channels_last = bool(bev_channels_last)
        # the BEV backbone takes the sparse tensor itself (first layer as a sparse conv, dense_path.BEVBackbone)
        self.map_to_bev_module.defer = bool(bev_channels_last) and dp.SPARSE_FIRST_BEV_LAYER
        self.backbone_2d = dp.BEVBackbone(256)
        self.backbone_2d.head_on_load = bool(bev_channels_last)      # its reader is self.dense_head (dense_path._Head1x1Parts)
        # conv_out's only reader is the BEV backbone's first layer (a sparse convolution): BatchNorm + ReLU on load there
        self.backbone_3d.conv_out.leave_pending = bool(self.map_to_bev_module.defer) and True
        if self.map_to_bev_module.defer:     # its rule table is planned with the sparse backbone's
            d = 256 // self.backbone_3d.num_point_features
            self.backbone_3d.extra_plan = (gb.spconv.core.PlannedConv(dp.BEV_FIRST_KEY, (d, 3, 3), (d, 1, 1), (0, 1, 1),
                                                                       self.backbone_3d.num_point_features, 64),)
        self.dense_head = dp.AnchorHead(self.backbone_2d.num_bev_features, num_class=1, num_anchors_per_location=2)
        self.roi_head = VoxelRCNNKLHead(self.backbone_3d.backbone_channels, cfg["voxel_size"],
                                        cfg["point_cloud_range"], self.roi_cfg)
        self.target_layer = roi_targets.ProposalTargetLayer(self.roi_cfg["TARGET"])
        if bev_channels_last:        # the 2-D convolutions' weights live in the layout the NHWC kernels read
            self.backbone_2d.to(memory_format=torch.channels_last)
            self.dense_head.to(memory_format=torch.channels_last)
        self.feature_map = (grid[0] // 8, grid[1] // 8)
        self._anchors = None
        self.fixed_draws = None      # tests: (key (B,R), pick (B,P)) uniform numbers for the RoI sampler
        self.fixed_proposals = None  # tests: (rois (B,R,7), roi_scores (B,R), roi_labels (B,R)) in place of the own proposals
        self.last = None
        self.mark = None             # optional callable(stage_name), see StaticTrainPipeline.mark
        self.overlap_roi = False     # StaticTrainStep: RoI branch on its own stream, backward in stages (StagedLoss)
        self._roi_streams = {}

    def anchors(self, device):
        if self._anchors is None or self._anchors.device != device:
            h = self.head_cfg
            self._anchors = det.generate_anchors(self.cfg["point_cloud_range"], self.feature_map, h["anchor_sizes"],
                                                 h["anchor_rotations"], h["anchor_bottom_heights"], device=device)
        return self._anchors

    # ------------------------------------------------------------------ stages
    def first_stage(self, bd):
        """voxel features + coordinates -> sparse backbone -> BEV map (the part StaticTrainPipeline records)."""
        return self.map_to_bev_module(self.backbone_3d(self.vfe(bd)))

    def second_stage_losses(self, bd, gt_boxes, gt_uncertaintys, seed_rois_with_gt=None):
        """Everything behind the BEV map of a training step: BEV backbone + anchor head, anchor targets +
        dense-head loss, proposals, RoI targets, RoI-grid pooling, FC towers, the three RoI-head losses.
        gt_boxes (B,G,8) zero-padded [x,y,z,dx,dy,dz,ry,class]; gt_uncertaintys (B,G,7) label variances.
        seed_rois_with_gt: optional (7,) offset -- the first G proposal slots of every frame are overwritten
        with ground truth + offset (what a trained first stage delivers; an untrained one proposes nothing
        near the ground truth, which would leave the regression / corner terms without foreground).
        Returns (loss, parts) with device scalars; free of host synchronisation on shape-static inputs."""
        B = gt_boxes.shape[0]
        h, r = self.head_cfg, self.roi_cfg
        mark = self.mark or (lambda name: None)
        enc = bd.get("encoded_spconv_tensor")
        overlap = bool(self.overlap_roi) and gt_boxes.is_cuda and torch.is_grad_enabled()
        bev_cut, roi_cuts = None, {}
        if overlap:          # cut the autograd graph in front of the BEV backbone (StagedLoss)
            sf = bd.get("spatial_features")
            if torch.is_tensor(sf) and sf.requires_grad:
                bd["spatial_features"] = sf.detach().requires_grad_(True)
                bev_cut = (sf, bd["spatial_features"])
            elif enc is not None and enc._features is None and enc._pending is not None and enc._pending.raw.requires_grad:
                # conv_out left its BatchNorm + ReLU to its reader (the BEV backbone's first layer, a sparse convolution that
                # transforms on load): the cut goes through the RAW rows
                pend = enc._pending
                raw_leaf = pend.raw.detach().requires_grad_(True)
                cut = enc.replace_feature(None)
                cut._pending = gb.spconv.core.PendingBN(raw_leaf, pend.coef, pend.mean, pend.invstd, pend.bn, pend.count, None)
                bd["encoded_spconv_tensor"] = cut
                bev_cut = (pend.raw, raw_leaf)
            elif enc is not None and enc.features.requires_grad:
                bd["encoded_spconv_tensor"] = enc.replace_feature(enc.features.detach().requires_grad_(True))
                bev_cut = (enc.features, bd["encoded_spconv_tensor"].features)
            else:
                overlap = False
        if bd.get("stage_cuts") and not overlap:
            raise RuntimeError("the sparse backbone cut its autograd graph for a staged backward that will not run")
        dev = gt_boxes.device
        msf = bd["multi_scale_3d_features"]
        if overlap:
            # fork #1, in front of the BEV backbone: the RoI stream cuts the autograd graph at the feature tensors the RoI
            # grid pools from.  (Running the pooling scales' first MLP here, beside the BEV forward, was measured in round 4:
            # 7.22 against 7.03 ms per step -- the nine short launches compete with the convolutions that produce the
            # proposals' inputs; that branch is gone.)
            main = torch.cuda.current_stream(dev)
            key = dev.index if dev.index is not None else torch.cuda.current_device()
            if key not in self._roi_streams:
                # default priority: a high-priority stream (like GPU_MAX_HW_QUEUES > 4) doubled the step time
                self._roi_streams[key] = torch.cuda.Stream(dev)
            roi_stream = self._roi_streams[key]
            roi_stream.wait_stream(main)
            with torch.cuda.stream(roi_stream):
                msf = dict(msf)
                for k, st in msf.items():
                    f = getattr(st, "features", None)
                    if torch.is_tensor(f) and f.requires_grad:
                        msf[k] = st.replace_feature(f.detach().requires_grad_(True))
                        if getattr(st, "clean_rows", False):      # same values: still zeros past `count`
                            msf[k].clean_rows = True
                        roi_cuts[k] = (f, msf[k].features)
        bd = self.dense_head(self.backbone_2d(bd))
        mark("BEV backbone + anchor head fwd")
        if self.mark:        # two boundaries inside backward(): gradient hooks run on the stream of the backward pass
            def stamp_when_grad_arrives(t, name):
                if t is not None and t.requires_grad:
                    t.register_hook(lambda g: mark(name))
            head_in = bd.get("spatial_features_2d")
            if head_in is None and bd.get("spatial_features_2d_parts"):       # the head read the deblocks' raw outputs
                head_in = bd["spatial_features_2d_parts"]["parts"][0][0]
            stamp_when_grad_arrives(head_in, "backward: dense-head loss, anchor head" if overlap
                                    else "backward: losses, RoI head, anchor head")
            stamp_when_grad_arrives(bd.get("spatial_features_1x"), "backward: BEV deblocks + block 2")
            stamp_when_grad_arrives(bev_cut[1] if bev_cut else getattr(enc, "features", None), "backward: BEV backbone")
        anchors = self.anchors(gt_boxes.device)
        if overlap:      # fork #2: the rest of the RoI branch needs the head's predictions
            roi_stream.wait_stream(main)
        with (torch.cuda.stream(roi_stream) if overlap else contextlib.nullcontext()):
            with torch.no_grad():
                cls, boxes = det.predicted_boxes(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"), anchors)
                rois, roi_scores, roi_labels = det.proposal_layer(boxes, cls, *r["NMS_TRAIN"])
                own_proposals = (rois, roi_scores, roi_labels)
                if self.fixed_proposals is not None:     # what RoIHeadTemplate.proposal_layer does when `rois` is given
                    rois, roi_scores, roi_labels = self.fixed_proposals                  # (roi_head_template.py:69-70)
                if seed_rois_with_gt is not None:
                    ng = gt_boxes.shape[1]
                    if (rois.is_cuda and rois.is_contiguous() and roi_labels.is_contiguous() and gt_boxes.is_contiguous()
                            and rois.dtype == torch.float32 and roi_labels.dtype == torch.int64 and ng <= rois.shape[1]):
                        gb._lib.call("glx_seed_rois", rois, roi_labels, gt_boxes, seed_rois_with_gt.contiguous().float(),
                                     B, rois.shape[1], rois.shape[2], ng, gt_boxes.shape[2])          # one launch
                    else:
                        has = gt_boxes[:, :, 7:8] > 0
                        rois[:, :ng, :7] = torch.where(has, gt_boxes[:, :, :7] + seed_rois_with_gt, rois[:, :ng, :7])
                        roi_labels[:, :ng] = torch.where(has[..., 0], gt_boxes[:, :, 7].long(), roi_labels[:, :ng])
                key, pick = self.fixed_draws if self.fixed_draws is not None else (None, None)
                td = self.target_layer({"rois": rois, "roi_scores": roi_scores, "roi_labels": roi_labels,
                                        "gt_boxes": gt_boxes, "gt_uncertaintys": gt_uncertaintys}, key, pick)
                rois_s = td["rois"].contiguous()
                gt_src = td["gt_of_rois"]                                                    # roi_head_template.py:137
                gt_ct = losses.canonical_gt_of_rois(rois_s, gt_src)                          # :140-159
                reg_valid, cls_lab = td["reg_valid_mask"].view(-1), td["rcnn_cls_labels"].view(-1)
                unc = td["gt_uncertaintys_of_rois"].reshape(-1, 7)
            mark("proposals (NMS) + RoI targets")
            ori_cls, std_logit, rcnn_reg, rcnn_std = self.roi_head(rois_s, msf, bd["multi_scale_3d_strides"], B, raw=True)
            mark("RoI-grid pooling + FC towers fwd")
            w = r["LOSS_WEIGHTS"]
            if ROI_LOSSES_ONE_LAUNCH and losses.roi_head_losses_supported(ori_cls, rcnn_reg, cls_lab, rois_s, gt_ct, gt_src, unc,
                                                                           reg_valid):
                # the three terms, their sum and both gradients of rcnn_reg in ONE launch / one autograd node
                roi_loss, kl_parts, rcnn_cls = losses.roi_head_losses(
                    ori_cls, std_logit, rcnn_reg, rcnn_std, cls_lab, rois_s, gt_ct, gt_src, unc, reg_valid,
                    code_weights=w["code_weights"], w_cls=w["rcnn_cls_weight"], w_reg=w["rcnn_reg_weight"],
                    w_corner=w["rcnn_corner_weight"])
                l_cls, l_kl, l_cor = kl_parts["cls"], kl_parts["kl"], kl_parts["corner"]
            else:
                if ori_cls.is_cuda:      # score rescaling + classification loss + their backward: one launch
                    l_cls, rcnn_cls = losses.cls_rescale_loss(ori_cls, std_logit, cls_lab, weight=w["rcnn_cls_weight"])
                else:
                    rcnn_cls = losses.cls_rescale_torch(ori_cls, std_logit)
                    l_cls = losses.rcnn_cls_loss_torch(rcnn_cls, cls_lab, weight=w["rcnn_cls_weight"])
                l_kl, kl_parts = losses.kl_reg_loss(rcnn_reg, rcnn_std, rois_s, gt_ct[..., :7], unc, reg_valid,
                                                    code_weights=w["code_weights"], weight=w["rcnn_reg_weight"])
                l_cor = losses.corner_loss(rcnn_reg, rois_s, gt_src[..., :7], reg_valid, weight=w["rcnn_corner_weight"])
                roi_loss = l_cls + l_kl + l_cor
            mark("RoI-head losses")
        with torch.no_grad():
            tgt = target_assign.assign_targets([anchors], gt_boxes, [1], [h["matched_threshold"]],
                                               [h["unmatched_threshold"]])
        rpn, rpn_parts = losses.rpn_loss(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"),
                                         tgt["box_cls_labels"], tgt["box_reg_targets"], anchors,
                                         code_weights=h["code_weights"], cls_weight=h["cls_weight"],
                                         loc_weight=h["loc_weight"], dir_weight=h["dir_weight"])
        mark("anchor targets + dense-head loss")
        if overlap:
            loss = StagedLoss(rpn, roi_loss, roi_stream, bev_cut, roi_cuts, bd.get("stage_cuts") or {}, self.mark)
            loss.boundary = getattr(self, "grad_boundary", None)
        else:
            loss = rpn + roi_loss                                                        # voxel_rcnn.py get_training_loss
        parts = dict(loss_rpn=rpn.detach(), rcnn_loss_cls=l_cls.detach(), rcnn_loss_reg=l_kl.detach(),
                     rcnn_loss_corner=l_cor.detach(), fg_rois=kl_parts["fg"], **rpn_parts)
        # detached views for inspection / tests (a live autograd graph of an earlier step must not survive into
        # the next capture, see StaticTrainPipeline.enqueue)
        self.last = dict(rois=rois_s, rcnn_cls=rcnn_cls.detach(), rcnn_reg=rcnn_reg.detach(),
                         rcnn_reg_std=rcnn_std.detach(), targets=td, proposals=rois, own_proposals=own_proposals,
                         batch_cls_preds=cls.detach(), batch_box_preds=boxes.detach(), gt_of_rois_ct=gt_ct,
                         anchor_targets=tgt, cls_preds=bd["cls_preds"].detach(), box_preds=bd["box_preds"].detach(),
                         dir_cls_preds=bd["dir_cls_preds"].detach() if "dir_cls_preds" in bd else None)
        return loss, parts

    def training_step(self, points, batch_idx, batch_size, gt_boxes, gt_uncertaintys, seed_rois_with_gt=None):
        """Exact-shape forward of one training step (host read-backs size the sparse tensors)."""
        bd = gb.voxelize_batch(points, batch_idx, batch_size, self.cfg, train=True)
        bd = self.first_stage(bd)
        return self.second_stage_losses(bd, gt_boxes, gt_uncertaintys, seed_rois_with_gt)

    @property
    def map_to_bev(self):            # the name detector.StaticDetectorPipeline reads
        return self.map_to_bev_module

    @torch.no_grad()
    def second_stage(self, bd, batch_size, post_cfg=None, post=True):
        """Everything of the inference pass behind the BEV map: BEV backbone + anchor head, proposals (NMS_TEST), RoI-grid
        pooling, FC towers with the score rescaling, box refinement (voxelrcnn_kl_label_iou_head.py:38-85) and -- post --
        Detector3DTemplate.post_processing on the device (det.post_processing).  Free of host synchronisation."""
        mark = self.mark or (lambda name: None)
        bd = self.dense_head(self.backbone_2d(bd))
        mark("BEV backbone + anchor head")
        cls, boxes = det.predicted_boxes(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"),
                                         self.anchors(bd["cls_preds"].device))
        rois, roi_scores, roi_labels = det.proposal_layer(boxes, cls, *self.roi_cfg["NMS_TEST"])
        mark("decode + top-k + NMS")
        rcnn_cls, rcnn_reg, rcnn_std = self.roi_head(rois, bd["multi_scale_3d_features"],
                                                     bd["multi_scale_3d_strides"], batch_size)
        mark("RoI-grid pooling + FC towers")
        bd.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels,
                  batch_cls_preds=rcnn_cls.view(batch_size, -1, rcnn_cls.shape[-1]),
                  batch_box_preds=det.refine_boxes(rois, rcnn_reg),
                  batch_box_std_preds=rcnn_std.view(batch_size, -1, rcnn_std.shape[-1]),
                  cls_preds_normalized=False, has_class_labels=True)
        if post:
            bd["post"] = det.post_processing(bd["batch_cls_preds"], bd["batch_box_preds"], bd["batch_box_std_preds"],
                                             bd["roi_labels"], post_cfg)
            mark("post-processing (variance-voting NMS)")
        return bd

    @torch.no_grad()
    def forward(self, points, batch_idx, batch_size):
        """Inference data flow up to the refined boxes (voxelrcnn_kl_label_iou_head.py:77-85)."""
        bd = gb.voxelize_batch(points, batch_idx, batch_size, self.cfg, train=False)
        return self.second_stage(self.first_stage(bd), batch_size, post=False)

    @torch.no_grad()
    def predict(self, points, batch_idx, batch_size, post_cfg=None):
        """forward + Detector3DTemplate.post_processing (score threshold, top-k, variance-voting NMS with
        variance = exp(batch_box_std_preds), post max size, POST_SCORE_THRESH) on the device:
        bd["post"] = {pred_boxes (B,P,7), pred_scores, pred_labels, pred_index, num}; det.pred_dicts(bd["post"]) gives the
        reference's list of per-frame dicts (one read-back)."""
        bd = self.forward(points, batch_idx, batch_size)
        bd["post"] = det.post_processing(bd["batch_cls_preds"], bd["batch_box_preds"], bd["batch_box_std_preds"],
                                         bd["roi_labels"], post_cfg)
        return bd


def onecycle(step, total_steps, lr_max=OPTIM_CFG["LR"], moms=OPTIM_CFG["MOMS"], div_factor=OPTIM_CFG["DIV_FACTOR"],
             pct_start=OPTIM_CFG["PCT_START"]):
    """fastai-style one-cycle schedule of the reference (tools/train_utils/optimization/learning_schedules_fastai.py
    OneCycle: cosine annealing lr_max/div -> lr_max over pct_start, then -> lr_max/div/1e4; momentum mirrored)."""
    a = int(total_steps * pct_start)
    low = lr_max / div_factor

    def cos(s, e, p):
        return e + (s - e) / 2 * (math.cos(math.pi * p) + 1)
    if step < a:
        p = step / max(a, 1)
        return cos(low, lr_max, p), cos(moms[0], moms[1], p)
    p = (step - a) / max(total_steps - a, 1)
    return cos(lr_max, low / 1e4, p), cos(moms[1], moms[0], p)


class StaticTrainStep(gb.StaticTrainPipeline):
    """The whole GLENet-VR training step as one shape-static launch sequence: the sparse front end of
    StaticTrainPipeline (voxelize, rule tables on a second stream, backbone, dense()) with
    GLENetVR.second_stage_losses as its loss, backward of everything, then gradient-norm clipping and AdamW
    (`torch.optim.AdamW(capturable=True)`: step count and learning rate live on the device).

    capture(split=False): one HIP graph per step.  capture(split=True): two graphs -- forward + backward, and
    clip + update -- so that a gradient exchange (glenet_amd.dist.GradBucket, one flat RCCL all-reduce) runs
    between them: step() = replay fwd/bwd, exchange, replay update.
    Ground truth is part of the static input: gt_boxes (B, G, 8) / gt_uncertaintys (B, G, 7), zero rows = padding."""

    def __init__(self, model, batch_size, num_points, num_features=4, max_gt=32, optimizer=None, lr=None,
                 seed_rois_with_gt=None, grad_clip=OPTIM_CFG["GRAD_NORM_CLIP"], capacities=None, device=None):
        """optimizer: None -> glenet_amd.optim.FlatAdamW (parameters re-pointed into one flat buffer, clip +
        update = two launches); or a torch optimizer built with capturable=True (clip_grad_norm_ + step())."""
        from .optim import FlatAdamW, FlatGrads
        dev = device if device is not None else next(model.parameters()).device
        self.net = model
        self.gt_boxes = torch.zeros((batch_size, max_gt, 8), dtype=torch.float32, device=dev)
        self.gt_unc = torch.zeros((batch_size, max_gt, 7), dtype=torch.float32, device=dev)
        self.seed = (torch.as_tensor(seed_rois_with_gt, dtype=torch.float32, device=dev)
                     if seed_rois_with_gt is not None else None)
        self.parts = None
        self.grad_clip = grad_clip
        self.params = [p for p in model.parameters() if p.requires_grad]
        # optimizer="external": clip + update belong to the caller (dropin.record): the step ends with every gradient in ONE flat
        # buffer (FlatGrads: the in-place weight gradients land there directly), nothing of the update is recorded
        self.external = isinstance(optimizer, str) and optimizer == "external"
        self.flat_grads = FlatGrads(self.params) if self.external else None
        if self.external:
            optimizer = None
        elif optimizer is None:
            optimizer = FlatAdamW(self.params, lr=lr if lr is not None else OPTIM_CFG["LR"], betas=OPTIM_CFG["BETAS"],
                                  weight_decay=OPTIM_CFG["WEIGHT_DECAY"], max_norm=grad_clip)
        self.step_optimizer = optimizer
        self.flat = isinstance(optimizer, FlatAdamW)
        if optimizer is None and not self.external:
            raise ValueError("StaticTrainStep: no optimizer")
        self.exchange = None           # callable run between backward and the update (gradient all-reduce)
        self.update_graph = None
        self.grad_norm = None
        super().__init__(model.backbone_3d, model.cfg, batch_size, num_points, num_features,
                         loss_fn=self._loss, optimizer=None, capacities=capacities, device=dev,
                         extra_modules=(model.backbone_2d, model.dense_head, model.roi_head))
        self.hc = model.map_to_bev_module
        self.split = False
        # the gradient exchange in two buckets (capture(split=True, buckets=2)): everything but the sparse backbone's gradients is
        # final ~1 ms before the step ends -- that bucket's all-reduce runs beside the sparse backward (two recorded graphs)
        self.buckets, self.graph2, self._bucket_capture, self._bucket_plan = 1, None, None, None
        # the staged backward (RoI branch on its own stream, StagedLoss) is a property of THIS pipeline's launch
        # sequence: the model's flags are set around enqueue() only, so the model's eager API (training_losses ->
        # a tensor with .backward()) is what it was before a pipeline was built on it
        self.overlap_roi = OVERLAP_ROI and torch.device(dev).type == "cuda"
        self.stage_cuts = self.overlap_roi and STAGE_CUTS

    def _loss(self, bd):
        loss, self.parts = self.net.second_stage_losses(bd, self.gt_boxes, self.gt_unc, self.seed)
        return loss

    def load(self, points, batch_idx, gt_boxes=None, gt_uncertaintys=None):
        if gt_boxes is not None:
            if points.shape[0] > self.points.shape[0]:
                raise ValueError("batch has %d points, pipeline was sized for %d" % (points.shape[0], self.points.shape[0]))
            if gt_boxes.shape[1] > self.gt_boxes.shape[1]:
                raise ValueError("batch has %d ground-truth rows, pipeline was sized for %d"
                                 % (gt_boxes.shape[1], self.gt_boxes.shape[1]))
            unc = gt_uncertaintys[:, :, :7] if gt_uncertaintys is not None and gt_uncertaintys.shape[2] == 7 else gt_uncertaintys
            if self._load_fused(points, batch_idx, ((self.gt_boxes, gt_boxes), (self.gt_unc, unc))):
                return                       # points, frame ids, ground truth and label variances: one launch
        super().load(points, batch_idx)
        if gt_boxes is not None:
            g = gt_boxes.shape[1]
            if g > self.gt_boxes.shape[1]:
                raise ValueError("batch has %d ground-truth rows, pipeline was sized for %d" % (g, self.gt_boxes.shape[1]))
            self.gt_boxes.zero_()
            self.gt_boxes[:, :g].copy_(gt_boxes, non_blocking=True)
            self.gt_unc.zero_()
            if gt_uncertaintys is not None:
                self.gt_unc[:, :g].copy_(gt_uncertaintys, non_blocking=True)

    def set_lr(self, lr, momentum=None):
        """One-cycle schedule hook: learning rate (and beta1) are device scalars the recorded update reads."""
        if self.flat:
            self.step_optimizer.set_lr(lr, momentum)
            return
        for g in self.step_optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    def data_parallel(self):
        """Install the gradient exchange of the data-parallel step (tools/train.py:144-145 wraps the model in
        DistributedDataParallel): one flat all-reduce between backward and the update."""
        import torch.distributed as dist
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        if self.flat:
            self.exchange = self.step_optimizer.allreduce_
            # SUM all-reduce; the 1 / world of DDP's average is an argument of the update kernel (set BEFORE capture:
            # a recorded update graph keeps the value it was recorded with)
            self.step_optimizer.grad_scale = 1.0 / world
            # every rank starts from rank 0's parameters, optimizer state and BatchNorm buffers (DDP's constructor
            # broadcast, tools/train.py:144-145)
            self.step_optimizer.broadcast_state_(0)
            if world > 1:
                for b in self.net.buffers():
                    if dist.get_backend() == "gloo" and b.is_cuda:
                        host = b.cpu()
                        dist.broadcast(host, 0)
                        b.copy_(host)
                    else:
                        dist.broadcast(b, 0)
        else:
            from .dist import GradBucket
            self.exchange = GradBucket(self.params).allreduce_
        return self

    def update(self):
        """clip_grad_norm_ (train_utils.py:38) + optimizer step; no read-back (the norm stays on the device)."""
        if self.external:
            return                         # the caller's clip_grad_norm_ + optimizer.step()
        if self.flat:
            self.step_optimizer.step(packed=True)
            self.grad_norm = self.step_optimizer.grad_norm
        else:
            if self.grad_clip:
                self.grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip, foreach=True)
            self.step_optimizer.step()
        if self.mark:
            self.mark("grad clip + AdamW")

    def _bucket_boundary(self):
        """StagedLoss calls this when every gradient but the sparse backbone's is final: gather that bucket; a capture in
        progress ends its first graph here and begins the second (same stream, same memory pool)."""
        self.step_optimizer.pack_grads(only=self._bucket_plan[2], bump=False)
        st = self._bucket_capture
        if st is not None and st["phase"] == 1:
            st["g1"].capture_end()
            st["g2"].capture_begin(pool=st["g1"].pool())
            st["phase"] = 2

    def enqueue(self):
        losses.UNIT_ROOT_GRAD = True      # the step's root scalar is the unweighted sum of the loss terms
        net = self.net
        was = (net.overlap_roi, net.backbone_3d.stage_cuts)
        net.overlap_roi, net.backbone_3d.stage_cuts = self.overlap_roi, self.stage_cuts
        bucketed = self.buckets == 2
        if bucketed:
            net.grad_boundary = self._bucket_boundary
        try:
            bd = super().enqueue()
        finally:
            losses.UNIT_ROOT_GRAD = False
            net.overlap_roi, net.backbone_3d.stage_cuts = was
            net.grad_boundary = None
        if bucketed:
            self.step_optimizer.pack_grads(only=self._bucket_plan[3])      # the late bucket; the lending generation moves
        elif self.flat:
            self.step_optimizer.pack_grads()          # part of the forward + backward graph
        elif self.external:
            self.flat_grads.pack_grads()
        if not self.split:
            if self.exchange is not None:
                self.exchange()
            self.update()
        return bd

    def _training_state(self):
        """Every tensor a step mutates besides gradients: parameters, optimizer moments / step count / schedule
        scalars, BatchNorm running statistics and batch counters."""
        opt = self.step_optimizer
        if self.external:
            ts = list(self.params)
        elif self.flat:
            ts = [opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_count, opt.hyper]
        else:
            ts = list(self.params) + [v for st in opt.state.values() for v in st.values() if torch.is_tensor(v)]
        return ts + [b for b in self.net.buffers()]

    def capture(self, warmup=2, split=False, keep_state=True, buckets=1):
        """split=False: fwd + bwd + clip + update in one graph.  split=True: the update is its own graph and
        step() runs `exchange` between the two.  buckets=2 (with split=True, the flat optimizer and the staged loss):
        forward + backward are recorded as TWO graphs cut where everything but the sparse backbone's gradients is final;
        step() starts that bucket's all-reduce behind the first graph, launches the second (the sparse backward) beside it,
        exchanges the sparse backbone's bucket and waits for both before the update graph.
        keep_state (default): the warm-up passes and the capture itself run REAL steps on the loaded batch (that is
        how the allocator pool and the lazily created optimizer state get sized) -- their effect on parameters, Adam
        moments, the step count (bias correction), BatchNorm running statistics and num_batches_tracked is undone
        afterwards by copying a snapshot back in place (pointers recorded in the graph stay valid), so the first
        step() after capture() is training step 1 at the schedule's first learning rate."""
        self.split = bool(split)
        if not split and self.exchange is not None:
            raise ValueError("a gradient exchange needs capture(split=True)")
        if buckets not in (1, 2):
            raise ValueError("capture: buckets is 1 or 2")
        if buckets == 2 and not (split and self.flat and self.overlap_roi and not self.external):
            raise ValueError("capture(buckets=2) needs split=True, the flat optimizer and the staged loss (overlap_roi)")
        self.buckets = buckets
        self._bucket_plan = self.step_optimizer.buckets(list(self.net.backbone_3d.parameters())) if buckets == 2 else None
        before = self._training_state() if keep_state else []
        snap = [t.detach().clone() for t in before]
        try:
            self._capture(warmup, split)
        finally:
            if keep_state:
                torch.cuda.synchronize(self.points.device)
                known = {id(t) for t in before}
                with torch.no_grad():
                    for t, s in zip(before, snap):
                        t.copy_(s)
                    for t in self._training_state():      # optimizer state a torch optimizer created during warm-up
                        if id(t) not in known:
                            t.zero_()
                gb._lib.bump_weights_epoch(self._written_tensors())
        return self

    def _capture(self, warmup, split):
        from . import runtime
        runtime.note_capture()          # one diagnostic if the executor setting of the published step time is not in effect
        if self.buckets == 2:
            self._capture_two_graphs(warmup)
        else:
            super().capture(warmup)
        if split and not self.external:
            dev = self.points.device
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self.update()                              # warm-up of the optimizer state (lazy init)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.update_graph = gb._lib.new_graph()
            with torch.cuda.graph(self.update_graph, stream=side), gb.no_gc():
                self.update()
            self.update_memsets_replaced = gb._lib.finish_graph(self.update_graph)   # raises if it cannot repair
        return self

    def _capture_two_graphs(self, warmup):
        """StaticFramePipeline.capture with the forward + backward in two graphs (the cut: _bucket_boundary)."""
        import gc
        dev = self.points.device
        gc.collect()
        side = self.__dict__.get("_capture_stream")
        if side is None:
            side = self._capture_stream = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.enqueue()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.check()
        gc.collect()
        retired = self.graph
        pool = retired.pool() if (retired is not None and gb.REUSE_GRAPH_POOL) else None
        g1, g2 = gb._lib.new_graph(), gb._lib.new_graph()
        self._bucket_capture = st = dict(g1=g1, g2=g2, phase=1)
        torch.cuda.empty_cache()
        try:
            with torch.cuda.stream(side), gb.no_gc():
                if pool is not None:
                    g1.capture_begin(pool=pool)
                else:
                    g1.capture_begin()
                try:
                    self.enqueue()
                finally:
                    (g2 if st["phase"] == 2 else g1).capture_end()
        finally:
            self._bucket_capture = None
        if st["phase"] != 2:
            raise RuntimeError("capture(buckets=2): the step never reached the bucket boundary (no staged loss?)")
        self.graph, self.graph2 = g1, g2
        self.memsets_replaced = (gb._lib.finish_graph(g1) or 0) + (gb._lib.finish_graph(g2) or 0)
        self._tag = self._weights_tag()

    def enqueue_eager_marked(self):
        """One eager pass of the step's launches with `mark` called at the stage boundaries (bench.py)."""
        self.enqueue()
        if self.split:
            if self.exchange is not None:
                self.exchange()
            self.update()

    def last_rois_shape(self):
        return tuple(self.net.last["rois"].shape)

    def step(self):
        """One training step on the loaded batch."""
        if self.graph is None:
            self.enqueue()
            if self.split:
                if self.exchange is not None:
                    self.exchange()
                self.update()
            return self.loss
        self.replay()
        if self.buckets == 2:
            opt = self.step_optimizer
            early, late, _, _ = self._bucket_plan
            exchange = self.exchange is not None
            pending = opt.allreduce_ranges_(early, async_op=True) if exchange else []     # behind the first graph ...
            self.graph2.replay()                                                          # ... beside the sparse backward
            if exchange:
                opt.allreduce_ranges_([late])
            for h in pending:
                h.wait()
            self.update_graph.replay()
            gb._lib.bump_weights_epoch(self._written_tensors())
        elif self.split and not self.external:
            if self.exchange is not None:
                self.exchange()
            self.update_graph.replay()
            gb._lib.bump_weights_epoch(self._written_tensors())
        return self.loss
