"""CPU oracle of the sparse backbone forward (eval-mode BatchNorm) -- TEST INFRASTRUCTURE ONLY.

Restates the layer list of VoxelBackBone8x.forward (pcdet/models/backbones_3d/
spconv_backbone.py:77-117,128-156) and of VoxelResBackBone8x (:191-232, SparseBasicBlock
:30-64) over the oracle's gather-GEMM-scatter convolution.  Weights come from a state dict
with the reference's parameter names (conv_input.0.weight, conv2.1.0.weight, ...), weight
layout (kd,kh,kw,Cin,Cout).
"""
import time

import numpy as np

from . import build_rules, sconv_forward

EPS = 1e-3  # BatchNorm1d eps, spconv_backbone.py:73

# (name of conv, kind, ksize, stride, padding, indice_key); BN sits at the sibling index + 1
PLAIN = [
    ("conv_input.0", "subm", (3, 3, 3), 1, 1, "subm1"),
    ("conv1.0.0", "subm", (3, 3, 3), 1, 1, "subm1"),
    ("conv2.0.0", "spconv", (3, 3, 3), 2, 1, "spconv2"),
    ("conv2.1.0", "subm", (3, 3, 3), 1, 1, "subm2"),
    ("conv2.2.0", "subm", (3, 3, 3), 1, 1, "subm2"),
    ("conv3.0.0", "spconv", (3, 3, 3), 2, 1, "spconv3"),
    ("conv3.1.0", "subm", (3, 3, 3), 1, 1, "subm3"),
    ("conv3.2.0", "subm", (3, 3, 3), 1, 1, "subm3"),
    ("conv4.0.0", "spconv", (3, 3, 3), 2, (0, 1, 1), "spconv4"),
    ("conv4.1.0", "subm", (3, 3, 3), 1, 1, "subm4"),
    ("conv4.2.0", "subm", (3, 3, 3), 1, 1, "subm4"),
    ("conv_out.0", "spconv", (3, 1, 1), (2, 1, 1), 0, "spconv_down2"),
]
TAPS = {"conv1.0.0": "x_conv1", "conv2.2.0": "x_conv2", "conv3.2.0": "x_conv3", "conv4.2.0": "x_conv4"}


def _bn_name(conv_name):
    head, last = conv_name.rsplit(".", 1)
    return "%s.%d" % (head, int(last) + 1)


def _bn_relu(x, sd, name, relu=True):
    g, b = sd[name + ".weight"], sd[name + ".bias"]
    m, v = sd[name + ".running_mean"], sd[name + ".running_var"]
    y = (x - m) / np.sqrt(v + np.float32(EPS)) * g + b
    return np.maximum(y, 0).astype(np.float32) if relu else y.astype(np.float32)


class State:
    def __init__(self, features, indices, shape):
        self.features, self.indices, self.shape = features, indices, shape


def _conv(st, sd, name, kind, ks, stride, pad, key, rules_cache, stats):
    w = sd[name + ".weight"]
    K = ks[0] * ks[1] * ks[2]
    w = w.reshape(K, w.shape[-2], w.shape[-1])
    t0 = time.perf_counter()
    if key not in rules_cache:
        rules_cache[key] = build_rules(st.indices, st.shape, ks, stride, pad, subm=(kind == "subm"))
    r = rules_cache[key]
    t1 = time.perf_counter()
    bias = sd.get(name + ".bias")
    out = sconv_forward(st.features, w, r, bias=bias)
    t2 = time.perf_counter()
    if stats is not None:
        stats.append(dict(layer=name, key=key, R=r.R, N_in=len(st.indices), N_out=len(r.out_indices),
                          cin=w.shape[1], cout=w.shape[2], K=K, t_rules=t1 - t0, t_conv=t2 - t1))
    return State(out, r.out_indices, r.out_shape)


def backbone_forward(sd, voxel_features, voxel_coords, sparse_shape, residual=False, stats=None):
    """sd: dict name -> float32 ndarray.  Returns dict with 'out' and the x_conv taps, each a
    State(features, indices, shape)."""
    sd = {k: np.asarray(v, np.float32) for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    st = State(np.asarray(voxel_features, np.float32), np.asarray(voxel_coords, np.int32),
               [int(s) for s in sparse_shape])
    rules, taps = {}, {}
    if not residual:
        for name, kind, ks, stride, pad, key in PLAIN:
            st = _conv(st, sd, name, kind, ks, stride, pad, key, rules, stats)
            st.features = _bn_relu(st.features, sd, _bn_name(name))
            if name in TAPS:
                taps[TAPS[name]] = st
        taps["out"] = st
        return taps

    def block(st, prefix, key):      # SparseBasicBlock.forward, spconv_backbone.py:51-68
        ident = st.features
        o = _conv(st, sd, prefix + ".conv1", "subm", (3, 3, 3), 1, 1, key, rules, stats)
        o.features = _bn_relu(o.features, sd, prefix + ".bn1")
        o = _conv(o, sd, prefix + ".conv2", "subm", (3, 3, 3), 1, 1, key, rules, stats)
        o.features = _bn_relu(o.features, sd, prefix + ".bn2", relu=False)
        o.features = np.maximum(o.features + ident, 0).astype(np.float32)
        return o

    st = _conv(st, sd, "conv_input.0", "subm", (3, 3, 3), 1, 1, "subm1", rules, stats)
    st.features = _bn_relu(st.features, sd, "conv_input.1")
    st = block(block(st, "conv1.0", "res1"), "conv1.1", "res1")
    taps["x_conv1"] = st
    for i, pad in ((2, 1), (3, 1), (4, (0, 1, 1))):
        st = _conv(st, sd, "conv%d.0.0" % i, "spconv", (3, 3, 3), 2, pad, "spconv%d" % i, rules, stats)
        st.features = _bn_relu(st.features, sd, "conv%d.0.1" % i)
        st = block(block(st, "conv%d.1" % i, "res%d" % i), "conv%d.2" % i, "res%d" % i)
        taps["x_conv%d" % i] = st
    st = _conv(st, sd, "conv_out.0", "spconv", (3, 1, 1), (2, 1, 1), 0, "spconv_down2", rules, stats)
    st.features = _bn_relu(st.features, sd, "conv_out.1")
    taps["out"] = st
    return taps
