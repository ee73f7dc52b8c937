"""Loading reference checkpoints into glenet_amd modules (checkpoint wire format).

Our counterpart of Detector3DTemplate._load_state_dict / load_params_from_file
(pcdet/models/detectors/detector3d_template.py:366-414) and find_all_spconv_keys (pcdet/utils/spconv_utils.py:11-25).
Parameter names of glenet_amd.glenet_vr.GLENetVR equal the reference's, so only the sparse-conv weight LAYOUT needs
care: a checkpoint holds whichever layout the spconv version it was trained with uses,
    spconv 1.x           (k1, k2, k3, C_in, C_out)     <- the layout of glenet_amd.spconv (core.py)
    spconv 2.x native    (k1, k2, k3, C_out, C_in)
    spconv 2.x implicit  (C_out, k1, k2, k3, C_in)
(the reference converts 1.x -> 2.x with transpose(-1, -2) / permute(4, 0, 1, 2, 3), lines 377-384; here the model is
the 1.x side, so the inverse maps are applied).  A square 2.x-native weight (C_in == C_out) has the same shape as a 1.x
one, so the layout is a property of the CHECKPOINT, never of a tensor: layout="auto" infers it ONCE from the weights
whose shapes can tell the layouts apart (every reference backbone has non-square ones: 4->16, 16->32, ...) and applies
it to all of them; checkpoints whose evidence disagrees are refused, checkpoints without any evidence (only square
k x k x k x C x C weights) are taken as 1.x with a warning.  (The reference decides per tensor and silently mixes layouts
in that situation -- its blind spot is not reproduced.)"""
import warnings

import torch

from . import spconv

LAYOUTS = ("auto", "spconv1", "spconv2_native", "spconv2_implicit")


def find_all_spconv_keys(model, prefix=""):
    """Names of the sparse-conv weights of `model` (spconv_utils.py:11-25)."""
    found = set()
    for name, child in model.named_children():
        p = "%s.%s" % (prefix, name) if prefix else name
        if isinstance(child, spconv.conv.SparseConvolution):
            found.add(p + ".weight")
        found |= find_all_spconv_keys(child, p)
    return found


def to_spconv1_layout(val, want_shape, layout="auto"):
    """One sparse-conv weight from a checkpoint -> (k1, k2, k3, C_in, C_out), or None when no layout fits."""
    want_shape = tuple(want_shape)
    if val.dim() != 5:
        return None
    native = val.transpose(-1, -2)                        # (k, k, k, C_out, C_in) -> ours
    implicit = val.permute(1, 2, 3, 4, 0)                 # (C_out, k, k, k, C_in) -> ours
    if layout == "spconv1":
        cands = [val]
    elif layout == "spconv2_native":
        cands = [native]
    elif layout == "spconv2_implicit":
        cands = [implicit]
    else:
        raise ValueError("to_spconv1_layout converts from one named layout; resolve 'auto' with infer_layout")
    for c in cands:
        if tuple(c.shape) == want_shape:
            return c.contiguous()
    return None


def infer_layout(model, state):
    """The spconv layout a checkpoint was written in, from the sparse-conv weights whose shape fits exactly one
    layout.  Raises when two weights point at different layouts; returns "spconv1" (with a warning) when nothing
    in the checkpoint can tell."""
    own = model.state_dict()
    votes = {}
    for key in sorted(find_all_spconv_keys(model)):
        if key not in state or state[key].dim() != 5:
            continue
        fits = [name for name in LAYOUTS[1:] if to_spconv1_layout(state[key], own[key].shape, name) is not None]
        if len(fits) == 1:
            votes.setdefault(fits[0], []).append(key)
    if len(votes) > 1:
        raise ValueError("checkpoint mixes sparse-conv weight layouts: %s"
                         % {k: v[:2] for k, v in votes.items()})
    if votes:
        return next(iter(votes))
    if any(k in state for k in find_all_spconv_keys(model)):
        warnings.warn("no sparse-conv weight of this checkpoint identifies its spconv layout (all are square); "
                      "taking it as spconv 1.x -- pass layout= explicitly if it was written by spconv 2.x")
    return "spconv1"


def adapt_state_dict(model, state, layout="auto"):
    """-> {name: tensor} with every entry of `state` that has a home in `model`, sparse-conv weights converted
    from ONE layout (given, or inferred once per checkpoint by infer_layout)."""
    if layout not in LAYOUTS:
        raise ValueError("layout must be one of %s" % (LAYOUTS,))
    if layout == "auto":
        layout = infer_layout(model, state)
    own = model.state_dict()
    conv_keys = find_all_spconv_keys(model)
    out = {}
    for key, val in state.items():
        if key not in own:
            continue
        if key in conv_keys:
            val = to_spconv1_layout(val, own[key].shape, layout)
            if val is None:
                continue
        if tuple(own[key].shape) == tuple(val.shape):
            out[key] = val
    return out


def load_params(model, checkpoint, layout="auto", strict=False):
    """checkpoint: a path (torch.load), a {'model_state': ...} dict as train_utils.save_checkpoint writes
    (tools/train_utils/train_utils.py:113-146) or a bare state dict.  Returns (loaded_keys, not_updated_keys)."""
    if isinstance(checkpoint, (str, bytes)):
        checkpoint = torch.load(checkpoint, map_location="cpu")
    state = checkpoint.get("model_state", checkpoint)
    update = adapt_state_dict(model, state, layout)
    own = model.state_dict()
    missing = [k for k in own if k not in update]
    if strict and missing:
        raise KeyError("checkpoint lacks %d tensors, e.g. %s" % (len(missing), missing[:5]))
    own.update(update)
    model.load_state_dict(own)
    return sorted(update), missing
