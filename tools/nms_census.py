"""Census of the NMS predicate on the device against the oracle.

Per seed: one frame of N boxes sorted by score, made of (a) proposal-like boxes (glenet_amd.synth.random_boxes:
independent boxes + jittered duplicates), and (b) a constructed NEAR-THRESHOLD family -- twins of other boxes
shifted along their heading axis by s = L (1 - t) / (1 + t), where IoU is t in real arithmetic and lands within a
few ulps of t in float, for t in the thresholds the reference's configs use (0.8 / 0.7 train / test, 0.1 and 0.01 in
GLENet_S / nms_func; GLENet_VR.yaml:107,115,179), then nudged by 0 / +-1 / +-2 ulps of the centre.
For every pair i < j of the frame:
  * non-candidates (centres further apart than the two circumradii + 0.06, cKDTree): the reference's overlap is
    exactly 0 -- the census checks that the device's IoU matrix holds exactly 0.0 there (count of non-zeros);
  * candidates: oracle IoU (oracle.iou_bev_pairs = the restated iou3d_cpu.cpp, pinned to the reference build by
    tests/golden/nms_pred_ref.npz) vs the device's IoU matrix entry: differing bits, and per threshold the pairs
    that fall on different sides of `iou > thr`;
  * keep lists: glx_nms on the device vs the host sweep on the oracle's relation, per threshold.
Prints one JSON line; `--out FILE` also writes it.  Used by tests/test_libm_gpu.py with a few seeds and run with
200 seeds for profiles/r03_nms_census.json.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

THRESHOLDS = (0.8, 0.7, 0.1, 0.01)


def make_frame(seed, n, twins_per_thr):
    from glenet_amd import synth
    rng = np.random.default_rng(3000 + seed)
    nb = n - twins_per_thr * len(THRESHOLDS)
    base = synth.random_boxes(rng, nb, xy_range=70.0, near_dup=0.5)
    twins = []
    for t in THRESHOLDS:
        src = base[rng.integers(0, nb, twins_per_thr)].copy()
        s = src[:, 3] * np.float32((1.0 - t) / (1.0 + t))
        src[:, 0] += (s * np.cos(src[:, 6])).astype(np.float32)
        src[:, 1] += (s * np.sin(src[:, 6])).astype(np.float32)
        for c in (0, 1):                                    # a few ulps either way
            k = rng.integers(-2, 3, twins_per_thr).astype(np.int32)
            v = np.ascontiguousarray(src[:, c]).view(np.int32) + k
            src[:, c] = v.view(np.float32)
        twins.append(src)
    boxes = np.concatenate([base] + twins).astype(np.float32)
    scores = rng.permutation(len(boxes)).astype(np.float32)
    return boxes[np.argsort(-scores, kind="stable")]


def candidate_pairs(boxes):
    from scipy.spatial import cKDTree
    rad = np.sqrt((boxes[:, 3].astype(np.float64) / 2) ** 2 + (boxes[:, 4].astype(np.float64) / 2) ** 2)
    tree = cKDTree(boxes[:, :2].astype(np.float64))
    pairs = tree.query_pairs(2 * rad.max() + 0.06, output_type="ndarray").astype(np.int32)
    pairs.sort(axis=1)
    d = np.linalg.norm(boxes[pairs[:, 0], :2].astype(np.float64) - boxes[pairs[:, 1], :2].astype(np.float64), axis=1)
    return pairs[d <= rad[pairs[:, 0]] + rad[pairs[:, 1]] + 0.06]


def census(seeds, n=9000, twins_per_thr=400, device="cuda:0", verbose=False):
    import torch
    import oracle
    from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_cuda
    dev = torch.device(device)
    rep = {"frames": 0, "boxes_per_frame": n, "pairs_total": 0, "candidate_pairs": 0, "candidates_iou_gt_0": 0,
           "noncandidate_nonzero_on_device": 0, "iou_bits_differ": 0, "max_abs_iou_diff": 0.0,
           "threshold_side_disagreements": {str(t): 0 for t in THRESHOLDS},
           "near_threshold_pairs_1e-6": {str(t): 0 for t in THRESHOLDS},
           "keep_lists_differ": {str(t): 0 for t in THRESHOLDS}, "keep_lists_compared": 0}
    t0 = time.time()
    for seed in seeds:
        boxes = make_frame(seed, n, twins_per_thr)
        pairs = candidate_pairs(boxes)
        ref = oracle.iou_bev_pairs(boxes, pairs)
        tb = torch.from_numpy(boxes).to(dev)
        m = torch.zeros((n, n), device=dev)
        iou3d_nms_cuda.boxes_iou_bev_gpu(tb, tb, m)
        pi, pj = torch.from_numpy(pairs[:, 0].astype(np.int64)).to(dev), torch.from_numpy(pairs[:, 1].astype(np.int64)).to(dev)
        got = m[pi, pj].cpu().numpy()
        upper_nonzero = int(torch.count_nonzero(torch.triu(m, 1)))
        rep["noncandidate_nonzero_on_device"] += upper_nonzero - int(np.count_nonzero(got))
        rep["frames"] += 1
        rep["pairs_total"] += n * (n - 1) // 2
        rep["candidate_pairs"] += len(pairs)
        rep["candidates_iou_gt_0"] += int((ref > 0).sum())
        rep["iou_bits_differ"] += int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        rep["max_abs_iou_diff"] = max(rep["max_abs_iou_diff"], float(np.abs(got - ref).max()))
        for t in THRESHOLDS:
            tf = np.float32(t)
            rep["threshold_side_disagreements"][str(t)] += int(((got > tf) != (ref > tf)).sum())
            rep["near_threshold_pairs_1e-6"][str(t)] += int((np.abs(ref.astype(np.float64) - t) < 1e-6).sum())
            keep_ref = oracle.nms_from_pairs(n, pairs, ref > tf)
            keep, num = iou3d_nms_cuda.nms_device(tb, float(t))
            keep = keep[:int(num)].cpu().numpy()
            rep["keep_lists_compared"] += 1
            rep["keep_lists_differ"][str(t)] += int(not np.array_equal(keep, keep_ref))
        if verbose:
            print("[census] seed %d: %d candidates, %d bit diffs so far, %.1fs" % (seed, len(pairs), rep["iou_bits_differ"],
                                                                                   time.time() - t0), file=sys.stderr)
    rep["seconds"] = round(time.time() - t0, 1)
    return rep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--boxes", type=int, default=9000)
    ap.add_argument("--twins", type=int, default=400)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import oracle
    oracle.lib().orc_set_threads(max(1, len(os.sched_getaffinity(0))))
    r = census(range(a.seeds), a.boxes, a.twins, verbose=True)
    line = json.dumps(r)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
