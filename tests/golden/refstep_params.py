"""Deterministic parameters for the whole-step golden (tests/golden/ref_step.npz).

The reference's GLENet-VR has 7.6 M parameters -- 30 MB, too large to commit -- so the fixture stores the SEED, the
state-dict names / shapes and a per-tensor float64 digest, and both sides (make_golden.py refstep, which runs the
reference's own classes on CPU, and tests/test_reference_step_gpu.py, which runs glenet_amd on the device) regenerate
the same tensors from numpy's PCG64 stream here.  The values are chosen so that the step exercises every branch with
an untrained network: the box head starts near zero (proposals = anchors + small residuals, so foreground RoIs exist
for ground-truth cars that are roughly axis aligned), score logits are spread out, the
RoI head's score rescaling lands on both sides of SCORE_THRESH / POST_SCORE_THRESH, BatchNorm running statistics are
non-trivial (eval mode).  Data and arithmetic only: nothing of the reference is restated here."""
import numpy as np

SEED = 20241004

# (substring of the state-dict name, weight std, bias mean) overrides, first match wins
_OVERRIDES = (
    ("dense_head.conv_box", 2e-3, 0.0),
    ("dense_head.conv_cls", 0.08, -2.5),
    ("dense_head.conv_dir_cls", 0.05, 0.0),
    ("roi_head.cls_pred_layer", 0.14, 1.1),
    ("roi_head.reg_pred_layer", 0.015, 0.0),
    ("roi_head.reg_std_layer", 0.04, -1.0),
    ("roi_head.reg_std_fc1", 0.30, 0.0),
    ("roi_head.reg_std_fc2", 0.3, 1.9),
)


def make_params(spec, seed=SEED):
    """spec: iterable of (name, shape, dtype string) in state-dict order -> {name: ndarray}."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, dtype in spec:
        shape = tuple(int(s) for s in shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, np.int64)
            continue
        over = next((o for o in _OVERRIDES if o[0] in name), None)
        if leaf == "running_mean":
            v = rng.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            v = rng.uniform(0.6, 1.4, shape)
        elif leaf == "weight" and len(shape) == 1:                  # BatchNorm gamma
            v = rng.uniform(0.7, 1.3, shape)
        elif leaf == "bias":
            v = rng.normal(over[2] if over else 0.0, 0.05, shape)
        else:
            if len(shape) == 5:                                     # sparse conv (kd, kh, kw, Cin, Cout)
                fan_in = int(np.prod(shape[:4]))
            else:                                                   # Linear / Conv (out, in, ...)
                fan_in = int(np.prod(shape)) // shape[0]
            std = over[1] if over else np.sqrt(2.0 / max(fan_in, 1))
            v = rng.normal(0.0, std, shape)
        out[name] = v.astype(np.dtype(dtype))
    return out


def digest(params):
    """{name: (sum, sum of squares)} in float64: what the fixture stores to prove both sides hold the same tensors."""
    return {k: (float(np.asarray(v, np.float64).sum()), float((np.asarray(v, np.float64) ** 2).sum()))
            for k, v in params.items()}


def canon_ties(scores, *rows):
    """The order of rows with EXACTLY equal scores is unspecified in the reference (torch.topk / sort leave ties in
    whatever order the library's algorithm produces; two of a frame's 3520 float32 anchor scores collide in about every
    second frame): within each run of equal scores of a descending list, order the rows by the first array in `rows`
    (lexicographically).  Returns the permutation; both sides of a comparison are passed through it."""
    scores = np.asarray(scores)
    key = np.asarray(rows[0]).reshape(len(scores), -1)
    perm = np.arange(len(scores))
    i = 0
    while i < len(scores):
        j = i + 1
        while j < len(scores) and scores[j] == scores[i] and scores[i] != 0:
            j += 1
        if j - i > 1:
            sub = perm[i:j]
            perm[i:j] = sub[np.lexsort(key[sub].T[::-1])]
        i = j
    return perm


def sample_rows(a, step):
    """Every `step`-th row of a 2-D+ array (the fixture's way of storing a large activation)."""
    return np.ascontiguousarray(np.asarray(a)[::step])


def grad_sample(g, n=256):
    """<= n entries of a flattened gradient at a fixed stride."""
    f = np.asarray(g).reshape(-1)
    return np.ascontiguousarray(f[::max(1, f.size // n)][:n])
