"""GLENet's KL regression loss of the RoI head (the "KL loss" of BASELINE config 3): our counterpart
of VoxelRCNNKLLabelIoUHead.get_box_reg_layer_loss, lines 96-138
(pcdet/models/roi_heads/voxelrcnn_kl_label_iou_head.py), without its corner-loss tail.

On the device the whole expression -- ResidualCoder.encode_torch of the ground truth against the RoI
moved to the origin, code-weighted smooth-L1, the variance terms, the foreground-normalised sum -- and
both gradients are ONE kernel (csrc/glx_loss.hip) and no host read-back (the reference reads
`fg_sum` and four tb_dict scalars back every step).

The `*_torch` functions are statement-by-statement tensor-op mirrors of the reference.  They are NOT a
fallback: the public functions (`kl_reg_loss`, `corner_loss`, `canonical_gt_of_rois`,
`rcnn_cls_loss`, `rpn_loss`) take device tensors only and raise otherwise.  The mirrors exist for the
CPU tests (pinned against fixtures generated from the reference's own code) and as the
"reference formulation" leg of tools/loss_bench.py."""
import ctypes

import torch

from . import _lib


def kl_reg_loss_torch(rcnn_reg, rcnn_reg_std, rois, gt_of_rois, gt_uncertainty, reg_valid_mask,
                      code_weights=None, beta=1.0 / 9.0, weight=1.0):
    """-> (loss, {'src','square','log'}), tensor ops only (reference statement order)."""
    r = rcnn_reg.shape[0]
    anchors = rois.detach().reshape(r, 7).clone()
    anchors[:, 0:3] = 0
    anchors[:, 6] = 0
    boxes = gt_of_rois.reshape(r, 7).clone()
    anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], min=1e-5)          # box_coder_utils.py:22-23
    boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], min=1e-5)
    xa, ya, za, dxa, dya, dza, ra = torch.split(anchors, 1, dim=-1)
    xg, yg, zg, dxg, dyg, dzg, rg = torch.split(boxes, 1, dim=-1)
    diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
    target = torch.cat([(xg - xa) / diagonal, (yg - ya) / diagonal, (zg - za) / dza, torch.log(dxg / dxa),
                        torch.log(dyg / dya), torch.log(dzg / dza), rg - ra], dim=-1)
    target = torch.where(torch.isnan(target), rcnn_reg, target)           # loss_utils.py:123
    diff = rcnn_reg - target
    if code_weights is not None:
        diff = diff * torch.as_tensor(code_weights, dtype=diff.dtype, device=diff.device).view(1, -1)
    n = torch.abs(diff)
    src = n if beta < 1e-5 else torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    label_var_log = torch.log(gt_uncertainty.reshape(r, 7) + 1e-10)
    fg = (reg_valid_mask.reshape(-1) > 0)
    fg_sum = fg.long().sum().clamp(min=1)
    std = torch.where(rcnn_reg_std < -50, torch.full_like(rcnn_reg_std, -50.0).detach(), rcnn_reg_std)
    m = fg.unsqueeze(-1).float()
    l_src = (torch.exp(-std) * src * m).sum() / fg_sum * weight
    l_sq = (torch.exp(label_var_log - std) * m).sum() / fg_sum * weight
    l_log = (-0.5 * (label_var_log - std) * m).sum() / fg_sum * weight
    return l_src + l_sq + l_log, {"src": l_src.detach(), "square": l_sq.detach(), "log": l_log.detach()}


# The loss kernels hand back d loss / d input; backward() scales it by the incoming gradient of the scalar.  A caller
# whose root scalar is the plain sum of these terms (GLENetVR.second_stage_losses under StaticTrainStep: voxel_rcnn.py
# get_training_loss adds them unweighted, and loss.backward() seeds the sum with 1) sets UNIT_ROOT_GRAD for its backward
# pass: the incoming gradient IS 1.0, and seven multiplications by it (three of them over the BEV head maps) are skipped.
UNIT_ROOT_GRAD = False


def _scaled(g, g_loss):
    return g if UNIT_ROOT_GRAD else g * g_loss


class _KLRegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rcnn_reg, rcnn_reg_std, rois, gt_of_rois, gt_uncertainty, fg, code_weights, beta, weight):
        args = [t.contiguous().float() for t in (rcnn_reg, rcnn_reg_std, rois.detach(), gt_of_rois, gt_uncertainty, fg)]
        _lib.check_cuda(*args)
        r = args[0].shape[0]
        out = torch.empty(5, dtype=torch.float32, device=args[0].device)
        g_reg, g_std = torch.empty_like(args[0]), torch.empty_like(args[1])
        cw = (ctypes.c_float * 7)(*[float(v) for v in code_weights]) if code_weights is not None else None
        _lib.call("glx_kl_reg_loss", *args, r, cw, ctypes.c_float(beta), ctypes.c_float(weight), out, g_reg, g_std)
        ctx.save_for_backward(g_reg, g_std)
        return out[0], out[1:5]

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        g_reg, g_std = ctx.saved_tensors
        return _scaled(g_reg, g_loss), _scaled(g_std, g_loss), None, None, None, None, None, None, None


def kl_reg_loss(rcnn_reg, rcnn_reg_std, rois, gt_of_rois, gt_uncertainty, reg_valid_mask, code_weights=None,
                beta=1.0 / 9.0, weight=1.0):
    """rcnn_reg, rcnn_reg_std (R,7); rois (B,N,7) or (R,7); gt_of_rois in the RoI frame; gt_uncertainty
    = label variances; reg_valid_mask (R) -> (loss, parts) with parts['src'|'square'|'log'|'fg']
    (device scalars: no read-back)."""
    _lib.check_cuda(rcnn_reg.contiguous())
    r = rcnn_reg.shape[0]
    fg = (reg_valid_mask.reshape(-1) > 0).float()
    loss, parts = _KLRegLoss.apply(rcnn_reg.reshape(r, 7), rcnn_reg_std.reshape(r, 7), rois.reshape(r, 7),
                                   gt_of_rois.reshape(r, 7), gt_uncertainty.reshape(r, 7), fg, code_weights,
                                   float(beta), float(weight))
    parts = parts.detach()
    return loss, {"src": parts[0], "square": parts[1], "log": parts[2], "fg": parts[3]}


# ------------------------------------------------------------------ corner-loss regularisation
def _corners(boxes):
    """boxes_to_corners_3d (pcdet/utils/box_utils.py:28-53): (N,7) -> (N,8,3)."""
    template = boxes.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                                 [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    c = boxes[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    cs, sn = torch.cos(boxes[:, 6]), torch.sin(boxes[:, 6])
    x = c[..., 0] * cs[:, None] - c[..., 1] * sn[:, None]
    y = c[..., 0] * sn[:, None] + c[..., 1] * cs[:, None]
    return torch.stack([x, y, c[..., 2]], dim=-1) + boxes[:, None, 0:3]


def corner_loss_torch(rcnn_reg, rois, gt_of_rois_src, reg_valid_mask, weight=1.0):
    """The reference's statements (voxelrcnn_kl_label_iou_head.py:148-172) in tensor ops."""
    fg = reg_valid_mask.reshape(-1) > 0
    if int(fg.sum()) == 0:
        return rcnn_reg.sum() * 0.0
    reg = rcnn_reg.reshape(-1, 7)[fg]
    roi = rois.reshape(-1, 7)[fg].detach()
    gt = gt_of_rois_src.reshape(-1, 7)[fg]
    dxa, dya, dza, ra = roi[:, 3], roi[:, 4], roi[:, 5], roi[:, 6]
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    xl, yl, zl = reg[:, 0] * diag, reg[:, 1] * diag, reg[:, 2] * dza
    ca, sa = torch.cos(ra), torch.sin(ra)
    pred = torch.stack([xl * ca - yl * sa + roi[:, 0], xl * sa + yl * ca + roi[:, 1], zl + roi[:, 2],
                        torch.exp(reg[:, 3]) * dxa, torch.exp(reg[:, 4]) * dya, torch.exp(reg[:, 5]) * dza,
                        reg[:, 6] + ra], dim=-1)
    flip = gt.clone()
    flip[:, 6] += 3.141592653589793
    pc = _corners(pred)
    d = torch.min(torch.norm(pc - _corners(gt), dim=2), torch.norm(pc - _corners(flip), dim=2))
    loss = torch.where(d < 1.0, 0.5 * d ** 2, d - 0.5).mean(dim=1)
    return loss.mean() * weight


class _CornerLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rcnn_reg, rois, gt_src, fg, weight):
        args = [t.contiguous().float() for t in (rcnn_reg, rois.detach(), gt_src, fg)]
        _lib.check_cuda(*args)
        out = torch.empty(2, dtype=torch.float32, device=args[0].device)
        g_reg = torch.empty_like(args[0])
        _lib.call("glx_corner_loss", *args, args[0].shape[0], ctypes.c_float(weight), out, g_reg)
        ctx.save_for_backward(g_reg)
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        (g_reg,) = ctx.saved_tensors
        return _scaled(g_reg, g_loss), None, None, None, None


def corner_loss(rcnn_reg, rois, gt_of_rois_src, reg_valid_mask, weight=1.0):
    """Mean corner loss over the foreground RoIs (0 when there are none, where the reference skips
    the term); device tensors: one kernel, no read-back of the foreground count."""
    _lib.check_cuda(rcnn_reg.contiguous())
    r = rcnn_reg.shape[0]
    fg = (reg_valid_mask.reshape(-1) > 0).float()
    return _CornerLoss.apply(rcnn_reg.reshape(r, 7), rois.reshape(r, 7), gt_of_rois_src.reshape(r, 7), fg,
                             float(weight))


# ------------------------------------------------------------------ canonical transformation
def canonical_gt_of_rois_torch(rois, gt_of_rois):
    """RoIHeadTemplate.assign_targets lines 140-159 in tensor ops (rois (B,N,7+), gt (B,N,7+C))."""
    import numpy as np
    gt = gt_of_rois.clone()
    roi_center = rois[:, :, 0:3]
    roi_ry = rois[:, :, 6] % (2 * np.pi)
    gt[:, :, 0:3] = gt[:, :, 0:3] - roi_center
    gt[:, :, 6] = gt[:, :, 6] - roi_ry
    c, s = torch.cos(-roi_ry), torch.sin(-roi_ry)
    x, y = gt[:, :, 0].clone(), gt[:, :, 1].clone()
    gt[:, :, 0] = x * c - y * s
    gt[:, :, 1] = x * s + y * c
    h = gt[:, :, 6] % (2 * np.pi)
    opposite = (h > np.pi * 0.5) & (h < np.pi * 1.5)
    h[opposite] = (h[opposite] + np.pi) % (2 * np.pi)
    flag = h > np.pi
    h[flag] = h[flag] - np.pi * 2
    gt[:, :, 6] = torch.clamp(h, min=-np.pi / 2, max=np.pi / 2)
    return gt


def canonical_gt_of_rois(rois, gt_of_rois):
    """-> gt_of_rois in the RoI frame, same shape; one kernel on the device."""
    r = rois.shape[0] * rois.shape[1]
    a, g = rois.contiguous().float(), gt_of_rois.contiguous().float()
    _lib.check_cuda(a, g)
    out = torch.empty_like(g)
    _lib.call("glx_roi_canonical_gt", a, a.shape[-1], g, g.shape[-1], r, out)
    return out


# ------------------------------------------------------------------ dense (anchor) head loss
class _RpnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, cfg):
        from ._lib import query, size_arg, workspace
        B, A = labels.shape
        cp = cls_preds.reshape(B, A, -1).contiguous().float()
        bp = box_preds.reshape(B, A, 7).contiguous().float()
        dp = dir_preds.reshape(B, A, 2).contiguous().float() if dir_preds is not None else None
        lab = labels.contiguous().int()
        tg = reg_targets.reshape(B, A, 7).contiguous().float()
        an = anchors.reshape(-1, anchors.shape[-1])[:, 0:7].contiguous().float()
        _lib.check_cuda(cp, bp, dp, lab, tg, an)
        out = torch.empty(4, dtype=torch.float32, device=cp.device)
        g_cls, g_box = torch.empty_like(cp), torch.empty_like(bp)
        g_dir = torch.empty_like(dp) if dp is not None else None
        cw = (ctypes.c_float * 7)(*[float(v) for v in cfg["code_weights"]])
        ws = workspace.get(query("glx_rpn_loss_workspace_bytes", B, A), cp.device)
        _lib.call("glx_rpn_loss", cp, bp, dp, lab, tg, an, B, A, cp.shape[-1], 1 if cp.shape[-1] == 1 else 0,
                  ctypes.c_float(cfg["alpha"]), ctypes.c_float(cfg["beta"]), cw, ctypes.c_float(cfg["dir_offset"]),
                  ctypes.c_float(cfg["cls_weight"]), ctypes.c_float(cfg["loc_weight"]), ctypes.c_float(cfg["dir_weight"]),
                  out, g_cls, g_box, g_dir, ws, size_arg(ws.numel()))
        ctx.save_for_backward(g_cls, g_box, g_dir if g_dir is not None else g_box)
        ctx.has_dir = g_dir is not None
        ctx.shapes = (cls_preds.shape, box_preds.shape, dir_preds.shape if dir_preds is not None else None)
        return out[0], out[1:4]

    @staticmethod
    def backward(ctx, g_loss, _g_parts):
        g_cls, g_box, g_dir = ctx.saved_tensors
        s = ctx.shapes
        return (_scaled(g_cls, g_loss).reshape(s[0]), _scaled(g_box, g_loss).reshape(s[1]),
                _scaled(g_dir, g_loss).reshape(s[2]) if ctx.has_dir else None, None, None, None, None)


def rpn_loss(cls_preds, box_preds, dir_preds, box_cls_labels, box_reg_targets, anchors, code_weights=(1.0,) * 7,
             cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, dir_offset=0.78539, alpha=0.25, beta=1.0 / 9.0):
    """AnchorHeadTemplate.get_loss on the device in three launches: cls_preds (B,H,W,A*C), box_preds
    (B,H,W,A*7), dir_preds (B,H,W,A*2) or None, box_cls_labels (B,N) int, box_reg_targets (B,N,7), anchors
    (..., 7) with N entries -> (rpn_loss, {'rpn_loss_cls','rpn_loss_loc','rpn_loss_dir'}) as device scalars
    (no `.item()`), gradients through autograd.  Defaults = GLENet_VR.yaml:84-90, DIR_OFFSET 0.78539,
    2 direction bins."""
    B, N = box_cls_labels.shape
    cfg = dict(code_weights=code_weights, cls_weight=cls_weight, loc_weight=loc_weight, dir_weight=dir_weight,
               dir_offset=dir_offset, alpha=alpha, beta=beta)
    loss, parts = _RpnLoss.apply(cls_preds.reshape(B, N, -1), box_preds.reshape(B, N, 7),
                                 dir_preds.reshape(B, N, 2) if dir_preds is not None else None, box_cls_labels,
                                 box_reg_targets, anchors, cfg)
    parts = parts.detach()
    return loss, {"rpn_loss_cls": parts[0], "rpn_loss_loc": parts[1], "rpn_loss_dir": parts[2]}


# ------------------------------------------------------------------ RoI classification loss
class _RcnnClsLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rcnn_cls, labels, weight):
        x, y = rcnn_cls.reshape(-1).contiguous().float(), labels.reshape(-1).contiguous().float()
        _lib.check_cuda(x, y)
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        g = torch.empty_like(x)
        _lib.call("glx_rcnn_cls_loss", x, y, x.shape[0], ctypes.c_float(weight), out, g)
        ctx.save_for_backward(g)
        ctx.shape = rcnn_cls.shape
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        (g,) = ctx.saved_tensors
        return _scaled(g, g_loss).reshape(ctx.shape), None, None


class _ClsRescaleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ori_cls, std_logit, labels, weight):
        a, b = ori_cls.reshape(-1).contiguous().float(), std_logit.reshape(-1).contiguous().float()
        y = labels.reshape(-1).contiguous().float()
        _lib.check_cuda(a, b, y)
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        z, ga, gb = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        _lib.call("glx_cls_rescale_loss", a, b, y, a.shape[0], ctypes.c_float(weight), z, out, ga, gb)
        ctx.save_for_backward(ga, gb)
        ctx.shapes = (ori_cls.shape, std_logit.shape)
        z = z.reshape(ori_cls.shape)
        ctx.mark_non_differentiable(z)
        return out[0], z

    @staticmethod
    def backward(ctx, g_loss, _g_z):
        ga, gb = ctx.saved_tensors
        return _scaled(ga, g_loss).reshape(ctx.shapes[0]), _scaled(gb, g_loss).reshape(ctx.shapes[1]), None, None


class _RoiHeadLosses(torch.autograd.Function):
    """The RoI head's three terms as one launch (glx_roi_head_losses): see roi_head_losses."""

    @staticmethod
    def forward(ctx, ori_cls, std_logit, rcnn_reg, rcnn_reg_std, cls_labels, rois, gt_ct, gt_src, label_var, reg_valid, cfg):
        r = rcnn_reg.shape[0]
        a, b = ori_cls.reshape(-1).contiguous().float(), std_logit.reshape(-1).contiguous().float()
        reg, std = rcnn_reg.reshape(r, 7).contiguous().float(), rcnn_reg_std.reshape(r, 7).contiguous().float()
        _lib.check_cuda(a, b, reg, std, cls_labels, rois, gt_ct, gt_src, label_var, reg_valid)
        dev = a.device
        out = torch.empty(10, dtype=torch.float32, device=dev)
        z, ga, gb = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        g_reg, g_std = torch.empty_like(reg), torch.empty_like(std)
        t = _lib.RoiHeadLossesArgs()
        t.ori_cls, t.std_logit, t.cls_labels = a.data_ptr(), b.data_ptr(), cls_labels.data_ptr()
        t.rcnn_reg, t.rcnn_reg_std, t.rois = reg.data_ptr(), std.data_ptr(), rois.data_ptr()
        t.gt_ct, t.gt_ct_ld, t.gt_src, t.gt_src_ld = gt_ct.data_ptr(), gt_ct.shape[-1], gt_src.data_ptr(), gt_src.shape[-1]
        t.label_var, t.reg_valid, t.R = label_var.data_ptr(), reg_valid.data_ptr(), r
        cw = cfg["code_weights"]
        for k in range(7):
            t.code_weights[k] = float(cw[k]) if cw is not None else 1.0
        t.beta, t.w_cls, t.w_reg, t.w_corner = cfg["beta"], cfg["w_cls"], cfg["w_reg"], cfg["w_corner"]
        t.rcnn_cls, t.out = z.data_ptr(), out.data_ptr()
        t.grad_ori, t.grad_std_logit, t.grad_reg, t.grad_reg_std = ga.data_ptr(), gb.data_ptr(), g_reg.data_ptr(), g_std.data_ptr()
        _lib.call("glx_roi_head_losses", ctypes.byref(t))
        ctx.save_for_backward(ga, gb, g_reg, g_std)
        ctx.shapes = (ori_cls.shape, std_logit.shape, rcnn_reg.shape, rcnn_reg_std.shape)
        z = z.reshape(ori_cls.shape)
        parts = out[1:]
        ctx.mark_non_differentiable(z, parts)
        return out[0], parts, z

    @staticmethod
    def backward(ctx, g_loss, _g_parts, _g_z):
        s = ctx.shapes
        return tuple(_scaled(g, g_loss).reshape(sh) for g, sh in zip(ctx.saved_tensors, s)) + (None,) * 7


def roi_head_losses_supported(ori_cls, rcnn_reg, cls_labels, rois, gt_ct, gt_src, label_var, reg_valid):
    """The one-launch form takes the target layer's tensors as they are: float32 soft labels, contiguous (.., >= 7)
    ground-truth rows, an int64 mask."""
    ok = ori_cls.is_cuda and ori_cls.dtype == torch.float32 and rcnn_reg.dtype == torch.float32
    for t, dt in ((cls_labels, torch.float32), (rois, torch.float32), (gt_ct, torch.float32), (gt_src, torch.float32),
                  (label_var, torch.float32), (reg_valid, torch.int64)):
        ok = ok and t.is_cuda and t.dtype == dt and t.is_contiguous()
    return bool(ok and rois.shape[-1] == 7 and gt_ct.shape[-1] >= 7 and gt_src.shape[-1] >= 7 and label_var.shape[-1] == 7)


def roi_head_losses(ori_cls, std_logit, rcnn_reg, rcnn_reg_std, cls_labels, rois, gt_ct, gt_src, label_var, reg_valid,
                    code_weights=None, beta=1.0 / 9.0, w_cls=1.0, w_reg=1.0, w_corner=1.0):
    """cls_rescale_loss + kl_reg_loss + corner_loss (the same arithmetic) as ONE launch and one autograd node:
    -> (loss, parts, rcnn_cls); parts = device scalars {cls, kl, src, square, log, fg, corner}; gt_ct / gt_src: the
    (.., >= 7)-column target rows as the target layer leaves them, reg_valid its int64 mask (no compare / cast / slice
    launches), d loss / d rcnn_reg = KL + corner gradient (no accumulation launch in backward)."""
    cfg = dict(code_weights=code_weights, beta=float(beta), w_cls=float(w_cls), w_reg=float(w_reg), w_corner=float(w_corner))
    loss, p, z = _RoiHeadLosses.apply(ori_cls, std_logit, rcnn_reg, rcnn_reg_std, cls_labels.reshape(-1), rois, gt_ct, gt_src,
                                      label_var, reg_valid.reshape(-1), cfg)
    p = p.detach()
    return loss, {"cls": p[0], "kl": p[2], "src": p[3], "square": p[4], "log": p[5], "fg": p[6], "corner": p[7]}, z


def cls_rescale_torch(ori_cls, std_logit):
    """voxelrcnn_kl_label_iou_head.py:70-76 in tensor ops: the logit of sigmoid(ori_cls) * sigmoid(std_logit)."""
    p = torch.sigmoid(ori_cls) * torch.sigmoid(std_logit)
    return torch.log((p + 1e-6) / (1 - p + 1e-6))


def cls_rescale(ori_cls, std_logit):
    """The same as one launch, for callers without autograd (inference)."""
    a, b = ori_cls.reshape(-1).contiguous().float(), std_logit.reshape(-1).contiguous().float()
    _lib.check_cuda(a, b)
    z = torch.empty_like(a)
    _lib.call("glx_cls_rescale_loss", a, b, None, a.shape[0], ctypes.c_float(1.0), z, None, None, None)
    return z.reshape(ori_cls.shape)


def cls_rescale_loss(ori_cls, std_logit, rcnn_cls_labels, weight=1.0):
    """GLENet's score rescaling + get_box_cls_layer_loss + the chain rule to both logits in one launch:
    -> (loss, rcnn_cls); rcnn_cls (the rescaled logit, shape of ori_cls) carries no autograd history -- the
    gradient flows through the loss."""
    return _ClsRescaleLoss.apply(ori_cls, std_logit, rcnn_cls_labels, float(weight))


def rcnn_cls_loss_torch(rcnn_cls, rcnn_cls_labels, weight=1.0):
    """The reference's statements (roi_head_template.py:253-264) in tensor ops."""
    flat, lab = rcnn_cls.view(-1), rcnn_cls_labels.view(-1)
    bl = torch.nn.functional.binary_cross_entropy(torch.sigmoid(flat), lab.float().clamp(min=0), reduction="none")
    valid = (lab >= 0).float()
    return (bl * valid).sum() / torch.clamp(valid.sum(), min=1.0) * weight


def rcnn_cls_loss(rcnn_cls, rcnn_cls_labels, weight=1.0):
    """get_box_cls_layer_loss with CLS_LOSS = BinaryCrossEntropy (device tensors)."""
    return _RcnnClsLoss.apply(rcnn_cls, rcnn_cls_labels, float(weight))


def rpn_loss_torch(cls_preds, box_preds, dir_preds, box_cls_labels, box_reg_targets, anchors, code_weights=(1.0,) * 7,
                   cls_weight=1.0, loc_weight=2.0, dir_weight=0.2, dir_offset=0.78539, alpha=0.25, beta=1.0 / 9.0):
    """AnchorHeadTemplate.get_loss (anchor_head_template.py:108-232) in tensor ops, statement by
    statement (test / baseline mirror, any device)."""
    import numpy as np
    B, N = box_cls_labels.shape
    labels = box_cls_labels.clone().long()
    num_class = cls_preds.reshape(B, N, -1).shape[-1]
    cared, positives, negatives = labels >= 0, labels > 0, labels == 0
    cls_weights = (negatives * 1.0 + 1.0 * positives).float()
    reg_weights = positives.float()
    if num_class == 1:
        labels[positives] = 1
    pos_normalizer = positives.sum(1, keepdim=True).float()
    reg_weights = reg_weights / torch.clamp(pos_normalizer, min=1.0)
    cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
    cls_targets = labels * cared.type_as(labels)
    one_hot = torch.zeros(B, N, num_class + 1, dtype=cls_preds.dtype, device=cls_preds.device)
    one_hot.scatter_(-1, cls_targets.unsqueeze(-1), 1.0)
    one_hot = one_hot[..., 1:]
    x = cls_preds.reshape(B, N, num_class)
    ps = torch.sigmoid(x)
    alpha_w = one_hot * alpha + (1 - one_hot) * (1 - alpha)
    pt = one_hot * (1.0 - ps) + (1.0 - one_hot) * ps
    bce = torch.clamp(x, min=0) - x * one_hot + torch.log1p(torch.exp(-torch.abs(x)))
    cls_loss = (alpha_w * torch.pow(pt, 2.0) * bce * cls_weights.unsqueeze(-1)).sum() / B * cls_weight
    bp, tg = box_preds.reshape(B, N, 7), box_reg_targets.reshape(B, N, 7)
    sin_p = torch.sin(bp[..., 6:7]) * torch.cos(tg[..., 6:7])
    sin_t = torch.cos(bp[..., 6:7]) * torch.sin(tg[..., 6:7])
    bp_s, tg_s = torch.cat([bp[..., :6], sin_p], -1), torch.cat([tg[..., :6], sin_t], -1)
    tg_s = torch.where(torch.isnan(tg_s), bp_s, tg_s)
    diff = (bp_s - tg_s) * torch.as_tensor(code_weights, dtype=bp.dtype, device=bp.device).view(1, 1, -1)
    n = torch.abs(diff)
    loc = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta) * reg_weights.unsqueeze(-1)
    loc_loss = loc.sum() / B * loc_weight
    dir_loss = loc_loss * 0.0
    if dir_preds is not None:
        an = anchors.reshape(1, -1, anchors.shape[-1])[..., :7].to(bp.dtype)
        rot_gt = tg[..., 6] + an[..., 6]
        v = rot_gt - dir_offset
        off = v - torch.floor(v / (2 * np.pi) + 0) * (2 * np.pi)
        bins = torch.clamp(torch.floor(off / (2 * np.pi / 2)).long(), min=0, max=1)
        w = positives.type_as(bp)
        w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
        ce = torch.nn.functional.cross_entropy(dir_preds.reshape(B, N, 2).permute(0, 2, 1), bins, reduction="none")
        dir_loss = (ce * w).sum() / B * dir_weight
    total = cls_loss + loc_loss + dir_loss
    return total, {"rpn_loss_cls": cls_loss.detach(), "rpn_loss_loc": loc_loss.detach(), "rpn_loss_dir": dir_loss.detach()}


# ------------------------------------------------------------------------------------------------------------------
# Single-stage GLENet heads (GLENet-S: AnchorHeadKLLabel, GLENet-C: AnchorHeadKLLabelIoU;
# pcdet/models/dense_heads/anchor_head_kl_label.py).  Tensor-op product code on device tensors (every step is a
# bandwidth-bound map over the (B, A, 7) anchor tensors; the rotated aligned IoU of the IoU branch is the
# glx_iou3d_boxes_aligned_overlap_bev kernel through pcdet_ops.iou3d.iou3d_utils.boxes_aligned_iou3d_gpu).
def _sin_difference(a, b, dim=6):
    """AnchorHeadTemplate.add_sin_difference (anchor_head_template.py:176-184)."""
    sa = torch.sin(a[..., dim:dim + 1]) * torch.cos(b[..., dim:dim + 1])
    sb = torch.cos(a[..., dim:dim + 1]) * torch.sin(b[..., dim:dim + 1])
    return (torch.cat([a[..., :dim], sa, a[..., dim + 1:]], dim=-1),
            torch.cat([b[..., :dim], sb, b[..., dim + 1:]], dim=-1))


def rpn_kl_box_loss(box_preds, box_std_preds, box_reg_targets, box_cls_labels, label_uncertainty, loc_weight=2.0,
                    code_weights=(1.0,) * 7, beta=1.0 / 9.0):
    """AnchorHeadKLLabel.get_box_reg_layer_loss without its direction term (anchor_head_kl_label.py:128-228): the
    KL divergence between the predicted Gaussian (mean box_preds, log-variance box_std_preds) and the label
    distribution (variance = the CVAE's label uncertainty the target assigner attached to every positive anchor),
        exp(-s) * smoothL1 + exp(log var_label - s) * w - 0.5 (log var_label - s) * w,   w = [positive] / #positives,
    summed over anchors, divided by the batch size.  box_* (B, H, W, A_loc * 7) or (B, A, 7); label_uncertainty
    (B, A, 7).  -> (loss, parts).  The log-variance is clamped at -50 in place upstream (a detached floor here)."""
    b = box_preds.shape[0]
    box_preds = box_preds.reshape(b, -1, 7)
    std = box_std_preds.reshape(b, -1, 7)
    std = torch.where(std < -50, torch.full_like(std, -50.0).detach(), std)
    positives = box_cls_labels > 0
    w = positives.float()
    w = w / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
    label_var_log = torch.log(label_uncertainty.reshape(b, -1, 7) + 1e-10)
    p_sin, t_sin = _sin_difference(box_preds, box_reg_targets.reshape(b, -1, 7))
    t_sin = torch.where(torch.isnan(t_sin), p_sin, t_sin)
    cw = _code_weights_t(tuple(code_weights), box_preds.device)
    n = torch.abs((p_sin - t_sin) * cw)
    l1 = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta) * w.unsqueeze(-1)
    src = (torch.exp(-std) * l1).sum() / b
    square = (torch.exp(label_var_log - std) * w.unsqueeze(-1)).sum() / b
    log = (-0.5 * (label_var_log - std) * w.unsqueeze(-1)).sum() / b
    loss = (src + square + log) * loc_weight
    return loss, {"rpn_loss_loc": loss.detach(), "rpn_loss_loc_src": (src * loc_weight).detach(),
                  "rpn_loss_loc_square": (square * loc_weight).detach(), "rpn_loss_loc_log": (log * loc_weight).detach()}


_CW_T = {}


def _code_weights_t(values, device):
    key = (values, str(device))
    if key not in _CW_T:
        _CW_T[key] = torch.tensor(values, dtype=torch.float32, device=device)
    return _CW_T[key]


def decode_residual(box_encodings, anchors):
    """ResidualCoder.decode_torch (box_coder_utils.py:46-71), 7 code channels."""
    xa, ya, za, dxa, dya, dza, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, dxt, dyt, dzt, rt = torch.split(box_encodings, 1, dim=-1)
    diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
    return torch.cat([xt * diagonal + xa, yt * diagonal + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya,
                      torch.exp(dzt) * dza, rt + ra], dim=-1)


def rpn_iou_loss(iou_preds, box_preds, box_reg_targets, box_cls_labels, anchors, aligned_iou3d=None, beta=1.0 / 9.0):
    """AnchorHeadKLLabelIoU.get_box_iou_layer_loss (anchor_head_kl_label.py:392-434): on the positive anchors, the
    predicted IoU (one logit per anchor) regressed with smooth-L1 towards 2 * IoU3D(decoded prediction, decoded
    target) - 1 (the aligned rotated IoU of pcdet.ops.iou3d, detached), weighted by 1 / #positives of the frame,
    summed and divided by the batch size.  aligned_iou3d(a, b) -> (n, 1): defaults to the device kernel."""
    if aligned_iou3d is None:
        from .pcdet_ops.iou3d.iou3d_utils import boxes_aligned_iou3d_gpu as aligned_iou3d
    b = iou_preds.shape[0]
    positives = box_cls_labels > 0
    w = positives.float()
    w = w / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
    mask = w > 0
    a = anchors.reshape(1, -1, 7).expand(b, -1, -1)
    pred_boxes = decode_residual(box_preds.reshape(b, -1, 7), a)
    gt_boxes = decode_residual(box_reg_targets.reshape(b, -1, 7), a)
    target = aligned_iou3d(pred_boxes[mask].detach().contiguous(), gt_boxes[mask].detach().contiguous()).detach()
    target = 2 * target - 1
    diff = torch.abs(iou_preds.reshape(b, -1, 1).float()[mask] - target)
    l1 = torch.where(diff < beta, 0.5 * diff ** 2 / beta, diff - 0.5 * beta) * w[mask].unsqueeze(-1)
    loss = l1.sum() / b
    return loss, {"rpn_loss_iou": loss.detach()}
