#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path (contract: see task brief / DESIGN.md section 5).

Workload at N=1 = BASELINE.json configs[1]: GLENet-VR SECOND sparse backbone
(VoxelBackBone8x as written in the reference: 8 SubMConv3d + 4 SparseConv3d), forward only,
batch 4 synthetic KITTI-shaped frames per GPU.  One step = one pass of the hot path over one
batch whose points are already resident in HBM:
    hard voxelize (4 frames) -> MeanVFE -> rule tables -> 12 sparse convs (+BN/ReLU) -> dense().
Default --mode graph: the frame is shape-static (buffers at capacity, live row counts stay on the
device: glenet_amd.backbone.StaticFramePipeline), so its ~120 launches need no host read-back
and are recorded once into a HIP graph; a step = copy the batch into the graph's input buffers
+ one graph launch.  --mode static enqueues the same launches from Python each step; --mode
dynamic is the exact-shape path the spconv mirror uses by default (host read-backs size every
tensor).  All three compute identical results (tests/test_backbone_gpu.py).
N > 1: one process per GPU (torch.distributed, RCCL), every rank runs its own 4 frames
(weak scaling, frames shard with no data-path collective in a forward pass); the timed
region is bracketed by barrier + synchronize and the max over ranks is reported.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from glenet_amd import backbone as gb  # noqa: E402
from glenet_amd import dist as gdist  # noqa: E402
from glenet_amd import synth  # noqa: E402
from glenet_amd.spconv import core as spcore  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FRAMES_PER_GPU = 4


def kernel_name(cin, cout, K):
    """Name of the kernel glx_sconv_forward dispatches to (launch_mfma in csrc/glx_sconv.hip)."""
    ok = {16, 32, 64, 128}
    if cin in (ok | {4, 8}) and cout in ok and K <= 27:
        if cout >= 64 and cin >= 16:
            return "k_sconv_gemm<%d,%d>" % (cin, cout)
        return "k_sconv_mfma<%d,%d>" % (cin, cout)
    return "k_sconv_generic"


def alg_bytes(R, K, cin, cout):
    """gather read Cin, scatter read-modify-write 2*Cout, two int32
    indices per rule, weights once."""
    return R * (cin + 2 * cout) * 4 + R * 8 + K * cin * cout * 4


class ConvProfiler:
    """Kernel-only duration of every sparse-conv launch: the launcher brackets the kernel
    with two HIP events (hipExtLaunchKernelGGL start/stop) on the stream it is launched on."""

    def __init__(self):
        import ctypes
        from glenet_amd import _lib
        self._lib, self._ct = _lib, ctypes
        self.records = []      # (name, start_evt, stop_evt, rules, K, cin, cout)
        self.enabled = False
        self._pool = []

    def _event(self):
        if self._pool:
            return self._pool.pop()
        e = self._ct.c_void_p()
        self._lib.call_nostream("glx_event_create", self._ct.byref(e))
        return e

    def __call__(self, tag, K, cin, cout, n_out, rules):
        name = kernel_name(cin, cout, K)
        if not self.enabled or name == "k_sconv_generic":
            return
        s, e = self._event(), self._event()
        self._lib.call_nostream("glx_profile_next_sconv", s, e)
        self.records.append((name, s, e, rules, K, cin, cout))

    def summary(self):
        per = {}
        ms = self._ct.c_float()
        for name, s, e, rules, K, cin, cout in self.records:
            self._lib.call_nostream("glx_event_elapsed_ms", s, e, self._ct.byref(ms))
            d = per.setdefault(name, dict(ms=0.0, launches=0, bytes=0))
            d["ms"] += ms.value
            d["launches"] += 1
            d["bytes"] += alg_bytes(rules.pair_count, K, cin, cout)
            self._pool += [s, e]
        self.records = []
        return per


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), if any."""
    p = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return {}


def cpu_baseline(frames_np, model, batches=6):
    """The CPU oracle (oracle/, a scalar C port) over the same workload: `batches` batches of
    synthetic frames (the bench batch first, then further seeds), ~10-15 s of CPU work."""
    import oracle
    from oracle import backbone as ob
    K = synth.KITTI
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    nf = len(frames_np)
    sets = [frames_np] + [[synth.kitti_frame(100 + j * nf + i)[0] for i in range(nf)] for j in range(1, batches)]
    t0 = time.perf_counter()
    for frames in sets:
        v, c, n = oracle.voxelize_hard_batch(frames, K["voxel_size"], K["point_cloud_range"],
                                             K["max_points"], K["max_voxels_train"])
        f = oracle.mean_vfe(v, n)
        taps = ob.backbone_forward(sd, f, c, model.sparse_shape)
        o = taps["out"]
        oracle.dense(o.features, o.indices, nf, o.shape)
    dt = time.perf_counter() - t0
    return dict(value=round(nf * len(sets) / dt, 3), unit="frames/s", cores=1, kind="port",
                sample="%d batches of %d synthetic KITTI-shaped frames (the same workload), one pass "
                       "each, %.1f s; host has %d cores" % (len(sets), nf, dt, os.cpu_count()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("graph", "static", "dynamic"), default="graph")
    ap.add_argument("--no-roofline", action="store_true", help="skip the event-bracketed second pass")
    ap.add_argument("--no-train", action="store_true", help="skip the fwd+bwd side measurement")
    ap.add_argument("--streams", type=int, default=2,
                    help="graph mode: number of frame pipelines replayed round-robin on their own "
                         "HIP streams (consecutive batches overlap on the GPU)")
    ap.add_argument("--serial-plan", action="store_true",
                    help="build the rule tables on the main stream (no overlap with the convolutions)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    rank, local_rank, world = gdist.env_world()
    # one process per GPU; GLX_DIST_BACKEND=gloo + fewer GPUs than ranks is a plumbing test mode
    # (ranks share a device, collectives on the host) -- never used for reported numbers
    backend = os.environ.get("GLX_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    gdist.init(backend, device=dev)

    K = synth.KITTI
    frame_ids = gdist.frames_for_rank(rank, world, FRAMES_PER_GPU)
    frames_np = [synth.kitti_frame(i)[0] for i in frame_ids]
    pts = torch.from_numpy(np.concatenate(frames_np)).to(dev)
    bidx = torch.from_numpy(np.concatenate(
        [np.full(len(f), i, np.int32) for i, f in enumerate(frames_np)])).to(dev)

    torch.manual_seed(0)
    grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    model = gb.VoxelBackBone8x(K["num_features"], grid).to(dev).eval()
    vfe, hc = gb.MeanVFE(), gb.HeightCompression()
    nstreams = max(1, args.streams) if args.mode == "graph" else 1
    pipes = []
    for _ in range(nstreams):
        p_ = gb.StaticFramePipeline(model, K, FRAMES_PER_GPU, pts.shape[0], K["num_features"])
        p_.calibrate(pts, bidx)   # output-set capacities of the strided convs, 1.3x this workload
        p_.overlap_plan = not args.serial_plan
        pipes.append(p_)
    pipe = pipes[0]
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    turn = [0]

    def dynamic_step():
        with torch.no_grad():
            bd = gb.voxelize_batch(pts, bidx, FRAMES_PER_GPU, K, train=True)
            bd = vfe(bd)
            bd["rule_plan"] = model.plan(bd["voxel_coords"], FRAMES_PER_GPU, index=bd["voxel_index"])
            return hc(model(bd))

    def static_step():
        pipe.load(pts, bidx)
        return pipe.enqueue()

    def graph_step():
        """Batch i goes to pipeline i mod S on stream i mod S: consecutive batches overlap."""
        i = turn[0] % nstreams
        turn[0] += 1
        with torch.cuda.stream(streams[i]):
            pipes[i].load(pts, bidx)
            return pipes[i].replay()

    if args.mode == "graph":
        for p_ in pipes:
            p_.load(pts, bidx)
            p_.capture()
    step = dict(graph=graph_step, static=static_step, dynamic=dynamic_step)[args.mode]

    def run(fn, steps):
        out = None
        for _ in range(steps):
            out = fn()
        return out

    run(step, args.warmup)
    torch.cuda.synchronize(dev)

    # ---- headline: exactly K steps between two fences, nothing else in the region
    gdist.fence(dev)
    t0 = time.perf_counter()
    bd = run(step, args.steps)
    gdist.fence(dev)
    dt = gdist.reduce_max(time.perf_counter() - t0, dev)
    if args.mode != "dynamic":
        for p_ in pipes:   # capacities held, voxelizer index valid (one read-back, after the clock)
            p_.check()

    # ---- roofline: K more steps of the same launches, each sparse-conv kernel bracketed by HIP
    # events on its stream (hipExtLaunchKernelGGL start/stop = kernel-only time).  Kept out of the
    # headline loop: event-bracketed launches cannot live in a graph and serialise the queue.
    per = {}
    if not args.no_roofline:
        prof = ConvProfiler()
        spcore._profile_hook = prof
        prof.enabled = True
        run(dynamic_step if args.mode == "dynamic" else static_step, min(args.steps, 50))
        torch.cuda.synchronize(dev)
        prof.enabled = False
        spcore._profile_hook = None
        per = prof.summary()
    prof_steps = min(args.steps, 50)

    # ---- side measurement (not `value`): forward + backward of the same backbone in training
    # mode (BatchNorm batch statistics, autograd through the sparse convs: dgrad = the forward
    # kernels on transposed weights, wgrad = k_wgrad_mfma), loss = mean(out^2), gradients of every
    # parameter; shape-static step replayed as one HIP graph (StaticTrainPipeline) unless
    # --mode dynamic asks for the exact-shape path with its host read-backs
    fwd_bwd = None
    if not args.no_train:
        tmodel = gb.VoxelBackBone8x(K["num_features"], grid).to(dev).train()
        tmodel.load_state_dict(model.state_dict())
        if args.mode == "dynamic":
            def train_step():
                bd_ = gb.voxelize_batch(pts, bidx, FRAMES_PER_GPU, K, train=True)
                bd_ = hc(tmodel(vfe(bd_)))
                tmodel.zero_grad(set_to_none=True)
                bd_["spatial_features"].square().mean().backward()
            tnote = "exact-shape path (host read-backs)"
        else:
            tpipe = gb.StaticTrainPipeline(tmodel, K, FRAMES_PER_GPU, pts.shape[0], K["num_features"])
            tpipe.calibrate(pts, bidx)
            tpipe.load(pts, bidx)
            if args.mode == "graph":
                tpipe.capture()
                train_step = tpipe.replay
                tnote = "shape-static step replayed as one HIP graph"
            else:
                train_step = tpipe.enqueue
                tnote = "shape-static step, eager launches"

        if world > 1:       # data-parallel training: one flat RCCL all-reduce of the gradients per step
            bucket = gdist.GradBucket(tmodel.parameters())
            local_step = train_step

            def train_step():
                local_step()
                bucket.allreduce_()
            tnote += ", + one flat all-reduce of all gradients (RCCL)"

        run(train_step, 6)
        gdist.fence(dev)
        t1 = time.perf_counter()
        run(train_step, 20)
        gdist.fence(dev)
        dtt = gdist.reduce_max(time.perf_counter() - t1, dev)
        if args.mode != "dynamic":
            tpipe.check()
        fwd_bwd = dict(frames_per_s=round(FRAMES_PER_GPU * world * 20 / dtt, 1),
                       ms_per_step=round(dtt / 20 * 1e3, 3), steps=20,
                       note="training-mode backbone fwd+bwd, " + tnote + ", not the headline workload")

    frames_total = FRAMES_PER_GPU * world * args.steps
    dom = max(per, key=lambda k: per[k]["ms"]) if per else None
    traffic = load_traffic()
    roof = None
    if dom:
        d = per[dom]
        achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=dom, achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                    traffic=traffic.get(dom),
                    alg_bytes_per_launch=int(d["bytes"] / d["launches"]),
                    avg_launch_us=round(d["ms"] * 1e3 / d["launches"], 2),
                    launches=d["launches"])
        tot_b = sum(v["bytes"] for v in per.values())
        tot_ms = sum(v["ms"] for v in per.values())
        roof["all_sparse_conv"] = dict(
            achieved=round(tot_b / (tot_ms * 1e-3) / 1e9, 1),
            frac=round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            ms_per_step=round(tot_ms / prof_steps, 4),
            per_kernel={k: dict(GBps=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                us_per_launch=round(v["ms"] * 1e3 / v["launches"], 2),
                                launches_per_step=v["launches"] // prof_steps)
                        for k, v in sorted(per.items())})

    if rank == 0:
        st = bd["encoded_spconv_tensor"]
        n_in = bd["voxel_index"].count.item() if args.mode != "dynamic" else bd["voxel_coords"].shape[0]
        n_out = st.count.item() if st.count is not None else st.indices.shape[0]
        out = dict(metric="LiDAR frames/sec (sparse backbone fwd) on KITTI-shaped clouds",
                   value=round(frames_total / dt, 2), unit="frames/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="configs[1]: VoxelBackBone8x (8 SubMConv3d + 4 SparseConv3d "
                                        "as in spconv_backbone.py:77-117) fwd-only, batch 4 "
                                        "KITTI-shaped frames/GPU, 20000 pts/frame, voxel "
                                        "0.05x0.05x0.1 m; step = voxelize + MeanVFE + rule tables + "
                                        "12 sparse convs + dense()",
                               frames_per_gpu=FRAMES_PER_GPU, points_per_frame=20000,
                               voxels_in=int(n_in), voxels_out=int(n_out),
                               mode={"graph": "shape-static frame replayed as one HIP graph, %d frame "
                                              "pipeline(s) on their own streams" % nstreams,
                                     "static": "shape-static frame, launches enqueued from Python",
                                     "dynamic": "exact shapes, host read-backs"}[args.mode],
                               parallelism="dp%d (frames shard, no data-path collective)" % world),
                   roofline=roof, fwd_bwd=fwd_bwd,
                   baseline_metric="BASELINE.json: 'LiDAR frames/sec (fwd+bwd) on KITTI-shaped clouds at "
                                   "1/2/4/8 MI355X; sparse-conv HBM GB/s' -- `value` is that metric on "
                                   "configs[1] (which is forward-only by its own wording), the fwd+bwd rate "
                                   "of the same backbone is in `fwd_bwd`, the sparse-conv GB/s in `roofline`")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_np, model)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
