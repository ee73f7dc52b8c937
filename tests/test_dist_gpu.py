"""Multi-GPU step equivalence without an 8-GPU box: two ranks x 2 frames, one flat all-reduce
between the forward + backward graph and the update graph, against ONE process that averages the two batches'
gradients and applies the same update.  The ranks share this box's GPU and the collective runs over gloo (plumbing;
the arithmetic -- SUM all-reduce, 1 / world folded into glx_adamw_clip_step_scaled, broadcast of rank 0's state at
start-up -- is what the RCCL path executes)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _two_rank_step(backend, buckets=1):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLX_DIST_BACKEND=backend, GLX_TEST_GRAD_BUCKETS=str(buckets))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_step_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = sorted((json.loads(line[len("DPRESULT "):]) for so, _ in outs for line in so.splitlines()
                  if line.startswith("DPRESULT ")), key=lambda d: d["rank"])
    assert [d["rank"] for d in res] == [0, 1]
    for d in res:
        assert d["backend"] == backend and d["world"] == 2
        assert d["buckets"] == buckets and d["graphs"] == 2 + (buckets == 2)
        assert d["ranks_equal"], d                       # rank 1 started from other weights: broadcast + same update
        assert d["grad_scale"] == 0.5 and d["step_count"] == 1
        assert d["grads_differ_between_batches"] > 1e-3 * d["grad_max_abs"]          # the ranks really had different data
        # the averaged gradient the update consumed == the mean of the two batches' gradients (float atomics in the
        # backward reorder sums by ~1e-7 relative)
        assert d["grad_err_max"] <= 2e-4 * d["grad_max_abs"], d
        assert d["grad_err_over_1e4"] <= d["n"] * 1e-5, d
        # Adam's first step moves every element by ~lr * sign(g): a noise-level gradient element whose sign differs
        # moves the other way (2 lr); everything else agrees to rounding
        assert d["param_err_max"] <= 2.5 * d["lr"], d
        assert d["param_err_mean"] <= 2e-3 * d["lr"], d


    return res


def test_two_ranks_make_the_update_of_one_rank_on_both_batches(dev):
    _two_rank_step("gloo")


def test_two_ranks_with_the_gradient_exchange_in_two_buckets(dev):
    """capture(split=True, buckets=2) -- forward + backward as two graphs cut where everything but the sparse
    backbone's gradients is final, that bucket exchanged behind the first graph (beside the sparse backward under RCCL), the sparse
    backbone's behind the second: the same update as one rank on both batches, to the same bounds as the one-bucket step."""
    _two_rank_step("gloo", buckets=2)


def _gpus_visible():
    import torch
    return torch.cuda.device_count()          # counting devices does not initialise the runtime


def test_two_ranks_over_real_rccl_when_two_gpus_are_visible(dev):
    """skips itself on a 1-GPU box and runs the day the suite sees two GPUs.  One process per GPU, the
    flat fp32 SUM all-reduce of the gradient buffer over RCCL (xGMI) between the forward + backward graph and the update
    graph: both ranks end with bit-identical parameters (rank 1 started from other weights: the broadcast), and the
    gradient the update consumed is the mean of the two shards' gradients."""
    if _gpus_visible() < 2:
        pytest.skip("one GPU visible: the 2-GPU RCCL step runs where the box has two")
    res = _two_rank_step("nccl")
    assert sorted(d["device"] for d in res) == [0, 1]
    res = _two_rank_step("nccl", buckets=2)              # ... and with the first bucket's all-reduce beside the sparse backward
    assert sorted(d["device"] for d in res) == [0, 1]


def test_bench_two_gpus_over_rccl_when_two_gpus_are_visible(dev):
    """`python bench.py --gpus 2 --no-extra` as the driver launches it (bench.py spawns one process per GPU before anything
    touches a GPU): ONE JSON line, n_gpus 2, the collective saw two ranks, per-rank diagnostics of both, weak scaling."""
    if _gpus_visible() < 2:
        pytest.skip("one GPU visible: bench.py --gpus 2 runs where the box has two")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GLX_DIST_BACKEND", "GLX_BENCH_FORCE_DP"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--no-config1", "--no-extra", "--no-cpu-baseline", "--no-stages"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["ranks_seen_by_collective"] == 2
    assert len(d["per_rank"]["ms_per_step"]) == 2 and min(d["per_rank"]["exchange_ms"]) >= 0.0


def _syncbn(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_syncbn_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    res = [json.loads(line[len("SYNCBN "):]) for so, _ in outs for line in so.splitlines() if line.startswith("SYNCBN ")]
    assert len(res) == world
    for d in res:
        assert d["world"] == world and d["rows"] > 100 and d["same_indices"]
        assert d["out_err"] <= 2e-5 * d["out_scale"] + 1e-6, d
        assert max(d["grad_err"]) <= 5e-4, d
        assert d["stats_err"] <= 1e-5, d
    return res


def test_sync_bn_conversion_runs_on_the_module_path(dev):
    """The reference's --sync_bn (tools/train.py:119-120: nn.SyncBatchNorm.convert_sync_batchnorm) on a stack of this
    package's spconv classes, process group over RCCL with a world of one: the converted BatchNorms take the module path
    (no conv + BatchNorm fusion) and reproduce the fused kernels' outputs, gradients and running statistics."""
    _syncbn(1)


def test_sync_bn_statistics_span_the_ranks_when_two_gpus_are_visible(dev):
    """... and with one GPU per rank the statistics span both ranks' rows: each rank's rows and the summed weight gradients
    equal one process running the unconverted stack on the union batch.  Skips itself on a 1-GPU box."""
    if _gpus_visible() < 2:
        pytest.skip("one GPU visible: cross-rank BatchNorm statistics need two")
    _syncbn(2)


def test_bench_data_parallel_step_over_rccl_prints_one_json_line(dev):
    """GLX_BENCH_FORCE_DP=1: bench.py's N > 1 code path -- process group over RCCL (`nccl`) with `device_id`, two graphs,
    the flat SUM all-reduce between them, the scaled update, the per-rank diagnostics -- with a world of one process on
    this box's GPU; and what a launcher reads: stdout holds exactly ONE line, the JSON (RCCL's version banner, which the
    C library flushes at exit, goes to stderr with everything else)."""
    env = dict(os.environ, GLX_BENCH_FORCE_DP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GLX_DIST_BACKEND"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-config1",
                        "--no-extra", "--no-cpu-baseline", "--no-stages"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "frames/s"
    assert "2 HIP graph(s)" in d["config"]["mode"] and d["config"]["ranks_seen_by_collective"] == 1
    assert d["per_rank"]["exchange_ms"][0] >= 0.0            # events around the RCCL all-reduce
