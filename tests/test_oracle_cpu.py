"""CPU suite (no GPU): the oracle against the reference's golden vectors and against independent
formulations, host-side logic, and the C-ABI export table."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import oracle
from glenet_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ pinned by the reference
@pytest.mark.parametrize("kind", ["random", "axis", "dup", "degenerate"])
def test_iou3d_oracle_bit_exact_vs_reference_build(kind):
    """oracle restatement == the reference's compiled iou3d_cpu.cpp, bit for bit."""
    g = np.load(os.path.join(GOLD, "iou3d_ref.npz"))
    a, b = g[kind + "_a"], g[kind + "_b"]
    ov = oracle.iou3d_boxes_overlap_bev(a, b)
    iou = oracle.iou3d_boxes_iou_bev(a, b)
    assert np.array_equal(ov, g[kind + "_overlap"])
    assert np.array_equal(iou.view(np.uint32), g[kind + "_iou"].view(np.uint32))   # NaN-safe bits
    assert (g[kind + "_iou"][np.isfinite(g[kind + "_iou"])] > 0).any() or kind == "degenerate"


def test_iou3d_aligned_is_diagonal():
    g = np.load(os.path.join(GOLD, "iou3d_ref.npz"))
    a, b = g["dup_a"][:40], g["dup_b"]
    d = oracle.iou3d_boxes_aligned_overlap_bev(a, b)[:, 0]
    assert np.array_equal(d, np.diagonal(g["dup_overlap"][:40]))


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_nms_func_oracle_matches_reference_python(case):
    """oracle.new_nms_gpu == the reference's iou3d_nms_utils.new_nms_gpu (imported unmodified
    when the fixture was made; see tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "nms_func_ref.npz"))
    boxes, scores = g["c%d_boxes" % case], g["c%d_scores" % case]
    var = g["c%d_var" % case] if ("c%d_var" % case) in g else None
    thr, sthr = g["c%d_params" % case]
    keep, new_boxes = oracle.new_nms_gpu(boxes, scores, thr, sthr, var)
    assert np.array_equal(keep, g["c%d_keep" % case])
    np.testing.assert_allclose(new_boxes, g["c%d_new_boxes" % case], rtol=1e-5, atol=1e-5)


def test_limit_period_matches_reference():
    g = np.load(os.path.join(GOLD, "nms_func_ref.npz"))
    assert np.array_equal(oracle.limit_period(g["limit_period_in"], 0.5, np.pi * 2), g["limit_period_2pi"])
    assert np.array_equal(oracle.limit_period(g["limit_period_in"], 0.5, np.pi), g["limit_period_pi"])


def test_nms_predicate_iou_pinned_to_reference_build():
    """The IoU that decides nms_gpu (iou3d_nms convention) against REFERENCE-EXECUTED values: the reference's
    compiled iou3d/src/iou3d_cpu.cpp run on [x - dx/2, y - dy/2, x + dx/2, y + dy/2, -heading] (fixture
    nms_pred_ref.npz).  Bit-exact on every pair outside the margin band where the two libraries differ by design."""
    import nms_pred_util as u
    g = u.load()
    safe = u.margin_safe(g["a7"], g["b7"])
    assert safe.mean() > 0.99 and ((g["overlap"] > 0) & safe).sum() > 2000 and ((g["iou"] > 0.5) & safe).sum() > 50
    ov, iou = oracle.boxes_overlap_bev(g["a7"], g["b7"]), oracle.boxes_iou_bev(g["a7"], g["b7"])
    assert np.array_equal(ov.view(np.uint32)[safe], g["overlap"].view(np.uint32)[safe])
    assert np.array_equal(iou.view(np.uint32)[safe], g["iou"].view(np.uint32)[safe])
    # the band is real: inside it the two margins do disagree (otherwise the mask would be hiding nothing)
    assert (ov != g["overlap"])[~safe].sum() > 10
    # the host library (boxes_iou_bev_cpu's product implementation) holds the same bits
    from glenet_amd import _host
    assert np.array_equal(_host.boxes_iou_bev(g["a7"], g["b7"]).view(np.uint32)[safe], g["iou"].view(np.uint32)[safe])


def test_device_libm_restatement_matches_the_host_libm(tmp_path):
    """csrc/glx_libm.h compiles for the host unchanged; tools/libm_check.cpp compares it with libm (quick mode:
    every 4099th float for sinf / cosf / atanf, 2e6 pairs for atan2f; the exhaustive run is in DESIGN.md)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "libm_check")
    flags = ["-O2", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(root, "glenet_amd", "csrc")]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")                     # hardware fma(); without it libm's software fma gives the same bits
    r = subprocess.run(["g++"] + flags + [os.path.join(root, "tools", "libm_check.cpp"), "-o", exe, "-lm"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "quick"], capture_output=True, text=True)
    rep = json.loads(r.stdout)
    assert r.returncode == 0 and rep["sinf_diff"] == rep["cosf_diff"] == rep["atanf_diff"] == rep["atan2f_diff"] == 0, rep
    assert rep["values_1d"] > 1000000


# ------------------------------------------------------------------ known answers / properties
def test_rotated_iou_known_answers():
    """SURVEY.md 8c spot values: identical boxes, half-shift of a 2x2 box, 45 degrees."""
    b0 = np.array([[0, 0, 0, 2, 2, 1, 0]], np.float32)
    assert abs(oracle.boxes_iou_bev(b0, b0)[0, 0] - 1.0) < 1e-5
    b1 = np.array([[0.5, 0, 0, 2, 2, 1, 0]], np.float32)
    assert abs(oracle.boxes_iou_bev(b0, b1)[0, 0] - 0.6) < 1e-6
    # a 2x2 box at the origin and a 2x2 box at (1, 1) turned by 45 degrees: the diamond |x-1|+|y-1| <= sqrt(2) cuts the
    # triangle (1,1), (1, 1-sqrt(2)), (1-sqrt(2), 1) out of the square -> area 1, IoU 1 / (4 + 4 - 1) = 0.142857
    v = np.array([[1, 1, 0, 2, 2, 1, np.pi / 4]], np.float32)
    assert abs(oracle.boxes_iou_bev(b0, v)[0, 0] - 1.0 / 7.0) < 1e-5
    assert abs(oracle.boxes_overlap_bev(b0, v)[0, 0] - 1.0) < 1e-5
    # unit boxes whose diamond stays outside the other square: exactly disjoint
    u = np.array([[0.5, 0.5, 0.5, 1, 1, 1, 0]], np.float32)
    w = np.array([[1.5, 1.5, 1.5, 1, 1, 1, np.pi / 4]], np.float32)
    assert oracle.boxes_iou_bev(u, w)[0, 0] == 0.0
    # symmetry and range on random boxes
    rng = np.random.default_rng(0)
    a = synth.random_boxes(rng, 64)
    m = oracle.boxes_iou_bev(a, a)
    assert (m >= 0).all() and (m <= 1.0 + 1e-5).all()
    np.testing.assert_allclose(m, m.T, atol=2e-5)
    assert (np.diagonal(m) > 0.999).all()


def test_nms_properties():
    rng = np.random.default_rng(1)
    boxes = synth.random_boxes(rng, 300, near_dup=0.5)
    scores = rng.permutation(300).astype(np.float32)
    keep = oracle.nms_gpu(boxes, scores, 0.3)
    # kept boxes are mutually below the threshold, every dropped box is covered by a better kept one
    m = oracle.boxes_iou_bev(boxes, boxes)
    kk = m[np.ix_(keep, keep)].copy()
    np.fill_diagonal(kk, 0)
    assert (kk <= 0.3).all()
    dropped = np.setdiff1d(np.arange(300), keep)
    for d in dropped:
        better = keep[scores[keep] > scores[d]]
        assert (m[better, d] > 0.3).any()
    # idempotence: NMS of the survivors keeps them all
    keep2 = oracle.nms_gpu(boxes[keep], scores[keep], 0.3)
    assert len(keep2) == len(keep)
    # axis-aligned variant agrees with a brute-force numpy IoU
    kn = oracle.nms_gpu(boxes, scores, 0.3, normal=True)
    assert len(kn) > 0


def test_sparse_conv_oracle_vs_dense_conv3d():
    """Independent formulation: torch conv3d on the densified grid (SURVEY.md 8c substitute 3)."""
    rng = np.random.default_rng(0)
    B, D, H, W, cin, cout = 2, 9, 14, 12, 5, 7
    occ = rng.random((B, D, H, W)) < 0.15
    idx = np.argwhere(occ).astype(np.int32)
    idx = idx[rng.permutation(len(idx))]
    f = rng.normal(size=(len(idx), cin)).astype(np.float32)
    dense = torch.zeros(B, cin, D, H, W)
    ti = torch.from_numpy(idx).long()
    dense[ti[:, 0], :, ti[:, 1], ti[:, 2], ti[:, 3]] = torch.from_numpy(f)
    # (kernel, stride, padding, submanifold, dilation): the reference backbones' geometries, then dilated ones (spconv's
    # SubMConv3d / SparseConv3d take a dilation; no GLENet config sets it) -- a submanifold conv with dilation d is the dense
    # conv with padding d * (k // 2) read at the active cells
    for ks, st, pd, subm, dl in [((3, 3, 3), 1, 1, True, 1), ((3, 3, 3), 2, 1, False, 1), ((3, 3, 3), 2, (0, 1, 1), False, 1),
                                 ((3, 1, 1), (2, 1, 1), 0, False, 1), ((3, 3, 3), 1, 2, True, 2), ((3, 3, 3), 1, (1, 2, 3), True, (1, 2, 3)),
                                 ((3, 3, 3), 2, 1, False, 2), ((3, 3, 3), 1, (2, 1, 2), False, (2, 1, 2)), ((3, 1, 3), (1, 1, 2), 1, False, (1, 1, 3))]:
        K = ks[0] * ks[1] * ks[2]
        w = rng.normal(size=(K, cin, cout)).astype(np.float32)
        r = oracle.build_rules(idx, [D, H, W], ks, st, 0 if subm else pd, subm=subm, dilation=dl)
        out = oracle.sconv_forward(f, w, r)
        wt = torch.from_numpy(w).reshape(*ks, cin, cout).permute(4, 3, 0, 1, 2)
        ref = torch.nn.functional.conv3d(dense, wt, stride=st, padding=pd, dilation=dl)
        oi = torch.from_numpy(np.asarray(r.out_indices)).long()
        np.testing.assert_allclose(out, ref[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]].numpy(), atol=1e-4)
        if not subm:
            occd = torch.nn.functional.conv3d((dense.abs().sum(1, keepdim=True) > 0).float(),
                                              torch.ones(1, 1, *ks), stride=st, padding=pd, dilation=dl) > 0
            assert np.array_equal(torch.nonzero(occd[:, 0]).numpy(), r.out_indices)
        # backward against autograd of the dense formulation
        g = rng.normal(size=out.shape).astype(np.float32)
        din, dw = oracle.sconv_backward(f, w, g, r)
        d2 = dense.clone().requires_grad_(True)
        w2 = wt.clone().requires_grad_(True)
        o2 = torch.nn.functional.conv3d(d2, w2, stride=st, padding=pd, dilation=dl)
        o2[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]].backward(torch.from_numpy(g))
        np.testing.assert_allclose(din, d2.grad[ti[:, 0], :, ti[:, 1], ti[:, 2], ti[:, 3]].numpy(), atol=1e-4)
        np.testing.assert_allclose(dw, w2.grad.permute(2, 3, 4, 1, 0).reshape(K, cin, cout).numpy(), atol=1e-3)


def test_voxelize_oracle_semantics():
    K = synth.KITTI
    pts, _ = synth.kitti_frame(0, num_points=4000)
    v, c, n = oracle.voxelize_hard(pts, K["voxel_size"], K["point_cloud_range"], 5, 16000)
    # numpy restatement of the cell arithmetic and first-seen order
    cell = np.floor((pts[:, :3] - np.float32(K["point_cloud_range"][:3])) / np.float32(K["voxel_size"])).astype(np.int64)
    lin = (cell[:, 2] * 1600 + cell[:, 1]) * 1408 + cell[:, 0]
    _, first = np.unique(lin, return_index=True)
    order = np.sort(first)
    assert np.array_equal(c, cell[order][:, ::-1])
    assert n.sum() == min(len(pts), np.minimum(np.bincount(np.unique(lin, return_inverse=True)[1]), 5).sum())
    assert np.array_equal(v[:, 0, :], pts[order])
    # max_voxels truncation keeps the first-seen voxels, later points of dropped cells vanish
    v2, c2, n2 = oracle.voxelize_hard(pts, K["voxel_size"], K["point_cloud_range"], 5, 100)
    assert np.array_equal(c2, c[:100]) and np.array_equal(v2, v[:100]) and np.array_equal(n2, n[:100])
    # dynamic voxelization: same cell set, sorted by the x-major key, means of all points
    f, cd = oracle.voxelize_dynamic_mean(pts, np.zeros(len(pts), np.int32), K["voxel_size"], K["point_cloud_range"])
    assert len(cd) == len(c)
    key = (cd[:, 3].astype(np.int64) * 1600 + cd[:, 2]) * 40 + cd[:, 1]
    assert (np.diff(key) > 0).all()


def test_voxel_query_oracle_vs_ball_query_in_window():
    """Independent check: with nsample >= window population, voxel_query returns
    exactly the points of the window that a brute-force radius test accepts, in z,y,x order."""
    rng = np.random.default_rng(3)
    B, Z, Y, X = 1, 6, 12, 12
    occ = rng.random((B, Z, Y, X)) < 0.3
    idx = np.argwhere(occ).astype(np.int32)
    xyz = (idx[:, [3, 2, 1]] + 0.5).astype(np.float32) * 0.5
    v2p = oracle.generate_voxel2pinds(idx, B, [Z, Y, X])
    M = 50
    qc = np.stack([np.zeros(M, np.int32), rng.integers(0, Z, M), rng.integers(0, Y, M), rng.integers(0, X, M)], 1).astype(np.int32)
    q = (qc[:, [3, 2, 1]] + rng.random((M, 3))).astype(np.float32) * 0.5
    got, empty = oracle.voxel_query((1, 2, 2), 0.8, 80, xyz, q, qc, v2p)
    for m in range(M):
        exp = []
        for dz in range(-1, 2):
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    z, y, x = qc[m, 1] + dz, qc[m, 2] + dy, qc[m, 3] + dx
                    if 0 <= z < Z and 0 <= y < Y and 0 <= x < X and v2p[0, z, y, x] >= 0:
                        p = v2p[0, z, y, x]
                        d = xyz[p] - q[m]
                        if np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) <= np.float32(0.8) * np.float32(0.8):
                            exp.append(p)
        if not exp:
            assert empty[m]
        else:
            assert list(got[m, :len(exp)]) == exp and (got[m, len(exp):] == exp[0]).all()


def test_group_points_gradcheck_style():
    rng = np.random.default_rng(4)
    feat = rng.normal(size=(30, 6)).astype(np.float32)
    idx = rng.integers(0, 15, (8, 4)).astype(np.int32)
    fcnt, icnt = np.array([15, 15], np.int32), np.array([4, 4], np.int32)
    out = oracle.group_points(feat, fcnt, idx, icnt)
    assert np.array_equal(out[5, :, 2], feat[15 + idx[5, 2]])
    g = rng.normal(size=out.shape).astype(np.float32)
    gi = oracle.group_points_grad(g, idx, icnt, fcnt, 30)
    ft = torch.from_numpy(feat).requires_grad_(True)
    gather = torch.stack([ft[(0 if m < 4 else 15) + torch.from_numpy(idx[m]).long()].t() for m in range(8)])
    gather.backward(torch.from_numpy(g))
    np.testing.assert_allclose(gi, ft.grad.numpy(), atol=1e-5)


def test_points_in_boxes_margins():
    box = np.array([[0, 0, 0, 2, 2, 2, 0.0]], np.float32)
    pts = np.array([[1.005, 0, 0], [1.02, 0, 0], [0, 0, 1.0], [0, 0, 1.0001], [0.999, 0.999, -1.0]], np.float32)
    assert oracle.points_in_boxes_cpu(pts, box)[0].tolist() == [1, 0, 1, 0, 1]          # MARGIN 1e-2
    assert oracle.points_in_boxes_gpu(pts[None], box[None])[0].tolist() == [-1, -1, 0, -1, 0]   # 1e-5


# ------------------------------------------------------------------ host logic
def test_spconv_mirror_shapes_and_layout():
    from glenet_amd import spconv
    from glenet_amd.backbone import VoxelBackBone8x, VoxelResBackBone8x
    m = VoxelBackBone8x(4, [1408, 1600, 40])
    sd = m.state_dict()
    for k in ["conv_input.0.weight", "conv1.0.0.weight", "conv2.2.1.bias", "conv4.2.0.weight", "conv_out.1.weight"]:
        assert k in sd, k
    assert tuple(sd["conv4.0.0.weight"].shape) == (3, 3, 3, 64, 64)       # spconv-1.x layout
    assert tuple(sd["conv_out.0.weight"].shape) == (3, 1, 1, 64, 128)
    assert m.sparse_shape == [41, 1600, 1408]
    assert len(m.sparse_convs()) == 12
    assert len({c.indice_key for c in m.sparse_convs()}) == 8
    r = VoxelResBackBone8x(5, [1504, 1504, 40])
    assert len(r.sparse_convs()) == 21 and len({c.indice_key for c in r.sparse_convs()}) == 9
    assert isinstance(m.conv_input[0], spconv.conv.SparseConvolution)       # spconv_utils.py:19
    assert "replace_feature" in dir(spconv.SparseConvTensor)                # spconv_utils.py:29


def test_dropin_registers_reference_import_names():
    import sys
    from glenet_amd import dropin
    done = dropin.install()
    import spconv.pytorch as sp          # noqa: F401  (the reference's import line)
    # the parent package `pcdet` is the reference's own and is absent here: check the alias table
    assert hasattr(sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_cuda"], "nms_gpu")
    assert hasattr(sys.modules["pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda"], "voxel_query_wrapper")
    assert hasattr(sp, "SubMConv3d") and hasattr(sp, "SparseConvTensor")
    assert "spconv" in done or "spconv" in sys.modules


def test_product_never_imports_oracle():
    """glenet_amd/ must not reference the oracle (no CPU fallback on the product path)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "glenet_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), os.path.join(dp, f)
                assert "liboracle" not in src


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads here (no GPU needed) and exports exactly what the header declares."""
    lib_path = os.path.join(ROOT, "glenet_amd", "csrc", "libglenet_hip.so")
    if not os.path.exists(lib_path):
        from glenet_amd import build
        build.build()
    import torch  # noqa: F401  (HIP runtime resolution order)
    lib = ctypes.CDLL(lib_path)
    hdr = open(os.path.join(ROOT, "include", "glenet_hip.h")).read()
    names = set(re.findall(r"\b(glx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) > 30
    for n in sorted(names):
        assert hasattr(lib, n), "missing export " + n
    lib.glx_abi_version.restype = ctypes.c_int
    assert lib.glx_abi_version() >= 2
    lib.glx_sconv_packed_bytes.restype = ctypes.c_size_t
    assert lib.glx_sconv_packed_bytes(27, 64, 64) > 27 * 64 * 64 * 4
    assert lib.glx_sconv_packed_bytes(27, 5, 16) == 0


def test_set_abstraction_oracle_vs_independent_numpy():
    """FPS / three_nn / three_interpolate restatements against plain numpy formulations."""
    rng = np.random.default_rng(3)
    xyz = rng.normal(size=(2500, 3)).astype(np.float32)
    cnt = [1500, 1000]
    got = oracle.stack_farthest_point_sample(xyz, cnt, [48, 32])

    def fps(X, m):
        t, o = np.full(len(X), 1e10, np.float32), [0]
        for _ in range(m - 1):
            d = ((X - X[o[-1]]) ** 2).astype(np.float32)
            t = np.minimum(t, (d[:, 0] + d[:, 1]) + d[:, 2])
            o.append(int(t.argmax()))
        return np.array(o)

    assert np.array_equal(got[:48], fps(xyz[:1500], 48)) and np.array_equal(got[48:], fps(xyz[1500:], 32) + 1500)
    d, i = oracle.three_nn(xyz[:100], [60, 40], xyz[100:400], [200, 100])
    D = ((xyz[:60, None] - xyz[None, 100:300]) ** 2).sum(-1)
    j = np.argsort(D, 1, kind="stable")[:, :3]
    assert np.array_equal(i[:60], j)
    np.testing.assert_allclose(d[:60] ** 2, np.take_along_axis(D, j, 1), rtol=1e-5)
    f = rng.normal(size=(300, 5)).astype(np.float32)
    w = rng.random((100, 3)).astype(np.float32)
    out = oracle.three_interpolate(f, i, w)
    np.testing.assert_allclose(out, (f[i] * w[:, :, None]).sum(1), rtol=1e-6, atol=1e-6)
    g = rng.normal(size=(100, 5)).astype(np.float32)
    gf = np.zeros_like(f)
    np.add.at(gf, i.reshape(-1), (g[:, None, :] * w[:, :, None]).reshape(-1, 5))
    np.testing.assert_allclose(oracle.three_interpolate_grad(g, i, w, 300), gf, rtol=1e-5, atol=1e-6)


def test_batch_layout_oracle_vs_the_stacked_restatements():
    """pointnet2_batch restatements cross-checked against the independently written stacked ones on B equal frames
    (different source files upstream, same semantics up to the empty-ball sentinel, index base and FPS block size)."""
    rng = np.random.default_rng(11)
    B, N, m = 3, 1500, 200
    xyz = rng.uniform(-3, 3, (B, N, 3)).astype(np.float32)
    new_xyz = rng.uniform(-3, 3, (B, m, 3)).astype(np.float32)
    new_xyz[1, 0] = 50.0
    cnt, mcnt = np.full(B, N, np.int32), np.full(B, m, np.int32)
    bi = oracle.batch_ball_query(0.5, 12, xyz, new_xyz)
    si, empty = oracle.ball_query(0.5, 12, xyz.reshape(-1, 3), cnt, new_xyz.reshape(-1, 3), mcnt)
    assert np.array_equal(bi.reshape(-1, 12), si) and empty.reshape(B, m)[1, 0] and (bi[1, 0] == 0).all()
    # FPS: N >= 1024 -> block size 1024 on both sides
    bf = oracle.batch_farthest_point_sample(xyz, 64)
    sf = oracle.stack_farthest_point_sample(xyz.reshape(-1, 3), cnt, 64).reshape(B, 64) - (np.arange(B) * N)[:, None]
    assert np.array_equal(bf, sf)
    # N < 1024 with exact ties: the block size opt_n_threads picks (512 for N = 600) decides
    lat = rng.integers(0, 3, (1, 600, 3)).astype(np.float32)
    f600 = oracle.batch_farthest_point_sample(lat, 30)[0]
    d = np.full(600, 1e10, np.float32)
    sel = [0]
    for _ in range(29):                                           # independent statement of the tie rule
        p = lat[0, sel[-1]]
        d = np.minimum(d, ((lat[0] - p) ** 2).sum(-1).astype(np.float32))
        tied = np.flatnonzero(d == d.max())
        sel.append(int(min(tied, key=lambda k: (k % 512, k))))
    assert list(f600) == sel
    # three_nn / interpolate / grouping
    known = rng.uniform(-3, 3, (B, 300, 3)).astype(np.float32)
    bd, bidx = oracle.batch_three_nn(new_xyz, known)
    sd, sidx = oracle.three_nn(new_xyz.reshape(-1, 3), mcnt, known.reshape(-1, 3), np.full(B, 300, np.int32))
    assert np.array_equal(bd.reshape(-1, 3), sd)
    assert np.array_equal(bidx.reshape(-1, 3) + np.repeat(np.arange(B) * 300, m)[:, None], sidx)
    feats = rng.normal(size=(B, 6, 300)).astype(np.float32)
    w = rng.uniform(0, 1, (B, m, 3)).astype(np.float32)
    bo = oracle.batch_three_interpolate(feats, bidx, w)
    so = oracle.three_interpolate(feats.transpose(0, 2, 1).reshape(-1, 6), sidx, w.reshape(-1, 3))
    np.testing.assert_allclose(bo.transpose(0, 2, 1).reshape(-1, 6), so, rtol=1e-6, atol=1e-6)
    g = oracle.batch_group_points(feats, bi % 300)
    assert g.shape == (B, 6, m, 12) and g[2, 4, 7, 3] == feats[2, 4, bi[2, 7, 3] % 300]
    go = rng.normal(size=g.shape).astype(np.float32)
    gg = oracle.batch_group_points_grad(go, bi % 300, 300)
    assert abs(float((gg * feats).sum()) - float((go * g).sum())) < 1e-2          # adjoint identity
    gi = oracle.batch_three_interpolate_grad(bo * 0 + 1, bidx, w, 300)
    np.testing.assert_allclose(gi.sum(-1), np.broadcast_to(w.sum((1, 2))[:, None], (B, 6)), rtol=1e-4)


def test_mean_vfe_and_range_mask_match_reference_golden():
    """oracle.mean_vfe == the reference's MeanVFE.forward (empty voxels: clamp_min(1)) and
    glenet_amd.data_pipeline.mask_points_by_range == common_utils.mask_points_by_range (inclusive
    borders); fixture generated from /root/reference (tests/golden/make_golden.py glue)."""
    import torch
    from glenet_amd import data_pipeline
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detector_glue_ref.npz"))
    got = oracle.mean_vfe(g["vfe_voxels"], g["vfe_num"])
    np.testing.assert_allclose(got, g["vfe_out"], rtol=1e-6, atol=1e-6)
    assert (g["vfe_num"] == 0).any()
    m = data_pipeline.mask_points_by_range(torch.from_numpy(g["mask_points"]), [0, -40.0, -3, 70.4, 40.0, 1])
    assert np.array_equal(m.numpy(), g["mask_out"]) and g["mask_out"][:40].any() and not g["mask_out"].all()


@pytest.mark.parametrize("norm", [False, True])
def test_target_assignment_oracle_matches_reference_golden(norm):
    """oracle.assign (numpy restatement) == the reference's AxisAlignedTargetAssigner run on the same
    inputs (tests/golden/target_assign_ref.npz): labels bit-identical, targets 1e-6, weights exact."""
    from oracle import assign
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "target_assign_ref.npz"))
    lab, tgt, w = assign.assign_targets([g["anchors_car"], g["anchors_cyc"]], g["gt"], [1, 3], [0.6, 0.5],
                                        [0.45, 0.35], norm=norm)
    tag = "norm" if norm else "plain"
    assert np.array_equal(lab, g["labels_" + tag])
    np.testing.assert_allclose(tgt, g["targets_" + tag], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(w, g["weights_" + tag], rtol=1e-7, atol=0)


def test_dense_head_loss_mirror_matches_reference_golden():
    """losses.rpn_loss_torch (tensor-op mirror) == the reference's AnchorHeadTemplate.get_loss called
    unmodified (fixture: make_golden.py assign): loss parts and the gradients of the three maps."""
    import torch
    from glenet_amd import losses
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "target_assign_ref.npz"))
    cls = torch.from_numpy(g["rpn_cls_preds"]).requires_grad_(True)
    box = torch.from_numpy(g["rpn_box_preds"]).requires_grad_(True)
    dr = torch.from_numpy(g["rpn_dir_preds"]).requires_grad_(True)
    loss, parts = losses.rpn_loss_torch(cls, box, dr, torch.from_numpy(g["rpn_labels"]), torch.from_numpy(g["rpn_targets"]),
                                        torch.from_numpy(g["anchors_car"]))
    np.testing.assert_allclose(float(loss.detach()), float(g["rpn_loss"]), rtol=1e-5)
    for k in ("rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir"):
        np.testing.assert_allclose(float(parts[k]), float(g[k]), rtol=1e-5, atol=1e-7)
    loss.backward()
    for t, k in ((cls, "rpn_grad_cls"), (box, "rpn_grad_box"), (dr, "rpn_grad_dir")):
        np.testing.assert_allclose(t.grad.numpy(), g[k], rtol=1e-4, atol=1e-6 * max(1e-3, np.abs(g[k]).max()))


def test_oracle_threads_do_not_change_results():
    """orc_set_threads(n > 1) (OpenMP, used only by bench.py's all-cores cpu_baseline leg) runs loops with
    independent iterations in parallel: sparse-conv forward, input gradient, voxel query, grouping and the NMS
    keep list are bit-identical to the serial oracle; the weight gradient (per-thread partial sums) to 1e-4 of its scale."""
    import oracle
    from glenet_amd import synth
    rng = np.random.default_rng(5)
    K = synth.KITTI
    pts, _ = synth.kitti_frame(3, num_points=6000)
    v, c, n = oracle.voxelize_hard(pts, K["voxel_size"], K["point_cloud_range"], 5, 16000)
    idx = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    f = rng.normal(size=(len(c), 16)).astype(np.float32)
    w = (rng.normal(size=(27, 16, 32)) / 10).astype(np.float32)
    r = oracle.build_rules(idx, [41, 1600, 1408], 3, subm=True)
    g = rng.normal(size=(len(c), 32)).astype(np.float32)
    boxes = synth.random_boxes(rng, 700, near_dup=0.5)
    lib = oracle.lib()

    def run():
        o = oracle.sconv_forward(f, w, r)
        din, dw = oracle.sconv_backward(f, w, g, r)
        return o, din, dw, oracle.nms_sorted(boxes, 0.3)
    assert lib.orc_get_threads() == 1
    a = run()
    lib.orc_set_threads(4)
    try:
        b = run()
    finally:
        lib.orc_set_threads(1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-4, atol=1e-4 * float(np.abs(a[2]).max()))


def test_vector_pool_oracle_vs_independent_numpy_formulation():
    """The restatement of vector_pool_gpu.cu (parity unpinned by the reference: GPU-only code, no tests) against
    an independent vectorised formulation: sub-voxel membership by masks, folded channel sums, three nearest by a
    stable argsort over the local neighbour list; gradient by np.add.at."""
    import oracle
    rng = np.random.default_rng(0)
    n, m = [700, 500], [40, 30]
    sx = rng.uniform(0, 4, (sum(n), 3)).astype(np.float32)
    sf = rng.normal(size=(sum(n), 8)).astype(np.float32)
    nx = rng.uniform(0.5, 3.5, (sum(m), 3)).astype(np.float32)
    r = np.float32(0.6)
    nf, nl, mean, pc, gi = oracle.vector_pool(sx, n, sf, nx, m, (2, 2, 2), 0.6, 4, True, num_mean_points_per_grid=2)
    assert mean == -(-len(gi) // sum(m)) and len(gi) == pc.sum()
    for p in (0, 5, 41, 69):
        b = 0 if p < m[0] else 1
        s0 = 0 if b == 0 else n[0]
        X, F = sx[s0:s0 + n[b]], sf[s0:s0 + n[b]]
        l = X - nx[p]
        ok = ~((np.abs(l) > r).any(1))
        cell = np.floor((l + r) / (r * 2 / 2)).astype(int)
        g = np.clip(cell[:, 0] * 4 + cell[:, 1] * 2 + cell[:, 2], 0, 7)
        for gg in range(8):
            sel = ok & (g == gg)
            assert pc[p, gg] == sel.sum()
            want = (F[sel][:, :4] + F[sel][:, 4:]).sum(0) / max(sel.sum(), 1e-6)
            np.testing.assert_allclose(nf[p, gg * 4:gg * 4 + 4], want, atol=1e-5)
            if sel.sum():
                np.testing.assert_allclose(nl[p, gg * 3:gg * 3 + 3], l[sel].sum(0) / sel.sum(), atol=1e-5)
        rows = gi[gi[:, 1] == p]
        assert np.array_equal(rows[:, 0], s0 + np.nonzero(ok)[0]) and np.array_equal(rows[:, 2], g[ok])
    grad = oracle.vector_pool_grad(np.ones_like(nf), pc, gi, sum(n), 8)
    ref = np.zeros_like(grad)
    np.add.at(ref, gi[:, 0], (1.0 / np.maximum(pc[gi[:, 1], gi[:, 2]], 1))[:, None])
    np.testing.assert_allclose(grad, ref, rtol=1e-6)
    centers = (nx[:, None, :] + rng.uniform(-0.3, 0.3, (sum(m), 8, 3))).astype(np.float32)
    d, idx, _ = oracle.three_nn_for_vector_pool_by_two_step(sx, n, nx, centers, m, 0.6, -1, 0, 3, 8, 2.0)
    for p, gcell in ((3, 2), (50, 7)):
        b = 0 if p < m[0] else 1
        s0 = 0 if b == 0 else n[0]
        X = sx[s0:s0 + n[b]]
        cand = np.nonzero(~((np.abs(X - nx[p]) > np.float32(1.2)).any(1)))[0][:1000]
        dd = ((centers[p, gcell] - X[cand]) ** 2).sum(1)
        o = np.argsort(dd, kind="stable")[:3]
        assert np.array_equal(idx[p, gcell], s0 + cand[o])


def test_split_bf16_pieces_carry_an_fp32_product():
    """The arithmetic of csrc/glx_bf16x3.h restated on the host: three bf16 pieces reproduce an fp32 number to its last
    bit or so, their piece products are exact in fp32, and the six products with i + j <= 4 give x * w to 2^-22."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-30, 30, 200000))).astype(np.float32)
    w = (rng.standard_normal(200000) * np.exp(rng.uniform(-10, 10, 200000))).astype(np.float32)
    a, b, c = oracle.bf16x3_split(x)
    for piece in (a, b, c):                                  # every piece IS a bfloat16: the low 16 bits are zero
        assert not np.any(piece.view(np.uint32) & 0xFFFF)
    rec = a.astype(np.float64) + b.astype(np.float64) + c.astype(np.float64)
    assert np.max(np.abs(rec - x.astype(np.float64)) / np.abs(x.astype(np.float64))) <= 2.0 ** -24
    exact = x.astype(np.float64) * w.astype(np.float64)
    got = oracle.bf16x3_product(x, w)
    assert np.max(np.abs(got - exact) / np.abs(exact)) < 2.0 ** -22
    # a two-piece split (three products) would stop at 2^-15: the third piece is what buys fp32
    two = (oracle.bf16x3_split(w)[0].astype(np.float64) * (a.astype(np.float64) + b) + oracle.bf16x3_split(w)[1].astype(np.float64) * a)
    assert np.max(np.abs(two - exact) / np.abs(exact)) > 2.0 ** -18


def test_scaled_f16_pieces_carry_an_fp32_class_product():
    """The f16x2 arithmetic of csrc/glx_conv2d.hip restated on the host (oracle.f16x2_*): a block of values scaled by the power
    of two of its maximum, two fp16 pieces per value.  An operand keeps 22 bits where its second piece is a normal fp16 number
    (within 2^-18 of the block's maximum), the absolute resolution below that is 2^-25 in the scaled domain (2^-39 of the
    maximum); the three piece products give x * w to 2^-20.4 at worst on such operands; scaling makes the result independent
    of the operands' magnitude."""
    rng = np.random.default_rng(0)
    x = (rng.uniform(1.0, 2.0, 200000) * rng.choice([-1.0, 1.0], 200000)).astype(np.float32)
    w = (rng.uniform(1.0, 2.0, 200000) * rng.choice([-1.0, 1.0], 200000)).astype(np.float32)
    ex, ew = oracle.f16x2_block_exponent(np.abs(x).max()), oracle.f16x2_block_exponent(np.abs(w).max())
    assert ex == 14 and ew == 14 and oracle.f16x2_block_exponent(0.0) == 127
    a, b = oracle.f16x2_split(x, ex)
    assert np.all(np.abs(a) <= 2.0 ** 15) and np.all(np.abs(a) >= 2.0 ** 14)          # scaled into fp16's top binade
    xs = np.ldexp(x.astype(np.float64), ex)
    assert np.max(np.abs(a.astype(np.float64) + b - xs) / np.abs(xs)) <= 2.0 ** -22
    exact = x.astype(np.float64) * w.astype(np.float64)
    err = np.abs(oracle.f16x2_product(x, w, ex, ew) - exact) / np.abs(exact)
    assert np.max(err) < 2.0 ** -20.4 and np.mean(err) < 2.0 ** -23
    # the same values at any common magnitude: bitwise the same relative result (the exponents are exact)
    for kx, kw in ((37, -60), (-90, 20), (0, 0)):
        x2, w2 = np.ldexp(x, kx).astype(np.float32), np.ldexp(w, kw).astype(np.float32)
        e2x, e2w = oracle.f16x2_block_exponent(np.abs(x2).max()), oracle.f16x2_block_exponent(np.abs(w2).max())
        assert (e2x, e2w) == (14 - kx, 14 - kw)
        got = oracle.f16x2_product(x2, w2, e2x, e2w)
        assert np.array_equal(np.ldexp(got, -(kx + kw)), oracle.f16x2_product(x, w, ex, ew))
    # a dim value beside a bright one: full precision down to 2^-18 of the block's maximum, an absolute floor below
    dim = (x * np.float32(2.0 ** -17)).astype(np.float32)
    ad, bd = oracle.f16x2_split(dim, ex)
    sd = np.ldexp(dim.astype(np.float64), ex)
    assert np.max(np.abs(ad.astype(np.float64) + bd - sd) / np.abs(sd)) <= 2.0 ** -21
    dimmer = (x * np.float32(2.0 ** -30)).astype(np.float32)
    ad, bd = oracle.f16x2_split(dimmer, ex)
    assert np.max(np.abs(ad.astype(np.float64) + bd - np.ldexp(dimmer.astype(np.float64), ex))) <= 2.0 ** -25
    # without the second piece (one fp16 product) the product stops at 2^-11
    one = np.ldexp(oracle.f16x2_split(w, ew)[0].astype(np.float64) * a, -(ex + ew))
    assert np.max(np.abs(one - exact) / np.abs(exact)) > 2.0 ** -12


def _exact_sconv(f, w, rules):
    """fp64 sums of the exact products over the rule pairs, and sum |f| |w| (the scale errors are quoted on)."""
    out = np.zeros((len(rules.out_indices), w.shape[2]), np.float64)
    mag = np.zeros_like(out)
    for k in range(w.shape[0]):
        n = int(rules.n_pairs[k])
        i, o = rules.pairs_in[k, :n], rules.pairs_out[k, :n]
        np.add.at(out, o, f[i].astype(np.float64) @ w[k].astype(np.float64))
        np.add.at(mag, o, np.abs(f[i]).astype(np.float64) @ np.abs(w[k]).astype(np.float64))
    return out, mag


def test_sparse_convolution_from_scaled_f16_pieces_is_an_fp32_class_convolution():
    """oracle.sconv_forward_f16x2 (the arithmetic of csrc/glx_sconv.hip's f16x2 block kernel: the filter scaled by one power of
    two, every input row by its own, three piece products) against the exact convolution and beside the fp32 C oracle: rows and
    filters of any magnitude, rows of zeros, an all-zero filter."""
    rng = np.random.default_rng(3)
    shape = (5, 12, 11)
    idx = np.argwhere(rng.random((2,) + shape) < 0.2).astype(np.int32)
    rules = oracle.build_rules(idx, shape, 3, subm=True)
    f0 = rng.normal(size=(len(idx), 32)).astype(np.float32)
    w0 = (rng.normal(size=(27, 32, 64)) / 30).astype(np.float32)
    rows = np.exp2(rng.integers(-30, 31, size=(len(idx), 1))).astype(np.float32)
    chans = np.exp2(rng.integers(-9, 10, size=(1, 32))).astype(np.float32)
    cases = {"plain": (f0, w0), "rows x 2^[-30, 30]": (f0 * rows, w0), "channels x 2^[-9, 9]": (f0 * chans, w0),
             "filter x 2^-40": (f0, w0 * np.float32(2.0 ** -40)), "filter x 2^30": (f0, w0 * np.float32(2.0 ** 30))}
    fz = f0.copy()
    fz[::3] = 0
    cases["every third row zero"] = (fz, w0)
    for name, (f, w) in cases.items():
        exact, mag = _exact_sconv(f, w, rules)
        live = mag > 0
        got = oracle.sconv_forward_f16x2(f, w, rules)
        err = np.abs(got - exact)[live] / mag[live]
        assert err.max() <= 2.0 ** -20.4, (name, np.log2(err.max()))
        # beside the fp32 oracle (exact products, fp32 sums): the same class
        err32 = np.abs(oracle.sconv_forward(f, w, rules) - exact)[live] / mag[live]
        assert err.max() <= 4 * err32.max() + 2.0 ** -24, (name, err.max(), err32.max())
    # scaling a row or the filter by a power of two scales the result bitwise
    base = oracle.sconv_forward_f16x2(f0, w0, rules)
    assert np.array_equal(oracle.sconv_forward_f16x2(f0, w0 * np.float32(2.0 ** 30), rules), base * 2.0 ** 30)
    assert np.array_equal(oracle.sconv_forward_f16x2(f0 * np.float32(2.0 ** -17), w0, rules), base * 2.0 ** -17)
    assert not oracle.sconv_forward_f16x2(f0, np.zeros_like(w0), rules).any()
    b = rng.normal(size=64).astype(np.float32)
    assert np.array_equal(oracle.sconv_forward_f16x2(f0, w0, rules, bias=b), base + b.astype(np.float64))


def test_dense_convolution_restatements_match_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5, 9, 12))
    w = rng.standard_normal((7, 5, 3, 3))
    for s in (1, 2):
        ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, s, 1).numpy()
        np.testing.assert_allclose(oracle.conv2d_3x3(x, w, s), ref, rtol=1e-12, atol=1e-12)
    for u in (1, 2):
        wt = rng.standard_normal((5, 4, u, u))
        ref = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wt), None, stride=u).numpy()
        np.testing.assert_allclose(oracle.conv_transpose2d(x, wt, u), ref, rtol=1e-12, atol=1e-12)
