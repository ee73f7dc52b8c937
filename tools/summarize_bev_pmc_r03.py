"""gpurun_out/r03_bev_mfma.md from gpurun_out/bev_r03/pmc (tools/pmc_bev_r03.sh): per kernel of the GLENet-VR
TRAINING STEP (bench.py --mode static), the last 3 steps (between occurrences of the once-per-step kernel
k_kl_reg_loss): duration, MFMA instructions, the fraction of SIMD-cycles the matrix pipe was busy -- MIOpen's
dense-convolution kernels (the BEV backbone + anchor head) first, then everything else that issues matrix
instructions."""
import collections
import csv
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(ROOT, "gpurun_out", "bev_r03", "pmc")
cc = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
kt = max(glob.glob(os.path.join(d, "*", "*kernel_trace.csv")), key=os.path.getmtime)
trace = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(trace) if "k_kl_reg_loss" in r["Kernel_Name"]]
assert len(marks) >= 4, "marker kernel seen %d times" % len(marks)
keep = trace[marks[-4] + 1:marks[-1] + 1]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in keep}
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for did, c in vals.items():
    if did not in dur:
        continue
    ns, name = dur[did]
    short = name.split("(")[0].replace("void ", "")[:86]
    a = agg[short]
    a["n"] += 1
    a["ns"] += ns
    for k, v in c.items():
        a[k] += v
SIMDS, GHZ, ITERS = 1024, 2.4, 3
rows = sorted(agg.items(), key=lambda kv: -kv[1]["ns"])
tot_ns = sum(a["ns"] for _, a in rows)
tot_busy = sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for _, a in rows)
mfma_rows = [(n, a) for n, a in rows if a.get("SQ_INSTS_MFMA", 0) > 0]
mfma_ns = sum(a["ns"] for _, a in mfma_rows)
dense = lambda n: n.startswith(("igemm_", "ck::", "_ZN2ck", "miopen", "Cijk_", "naive_conv"))      # noqa: E731
d_ns = sum(a["ns"] for n, a in rows if dense(n) and not n.startswith("Cijk_"))
d_busy = sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for n, a in rows if dense(n) and not n.startswith("Cijk_"))
own = lambda n: n.startswith(("k_conv3x3", "k_pconv"))      # noqa: E731
o_ns = sum(a["ns"] for n, a in rows if own(n))
o_busy = sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for n, a in rows if own(n))
md = ["# GLENet-VR training step: matrix-pipe utilisation per kernel (the BEV head's kernels and the rest)", "",
      "`bash tools/pmc_bev_r03.sh` (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA"
      " SQ_WAVES -- python3 bench.py --mode static --steps 6 --warmup 4 ...): the step's own launches (eager, so that"
      " counters attribute per dispatch), 4 frames (BEV map 4 x 256 x 200 x 176, channels-last fp32), last 3 steps.",
      "", "The own split-bf16 convolution kernels (csrc/glx_conv2d.hip, glx_deconv2d.hip: `k_conv3x3*`, `k_pconv*`): %.2f ms per"
      " step, matrix pipe busy **%.2f** of their SIMD-cycles (bf16 MFMAs; six per fp32-equivalent product tile)."
      % (o_ns / 1e6 / ITERS, o_busy / max(o_ns * GHZ * SIMDS, 1)),
      "", "MIOpen's convolution kernels (what is left on the vendor library: strided layer, head): %.2f ms per step, matrix pipe busy"
      " **%.2f** of their SIMD-cycles." % (d_ns / 1e6 / ITERS, d_busy / max(d_ns * GHZ * SIMDS, 1)),
      "", "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x %.1f GHz x %d SIMDs) (MI355X_MICROARCH.md,"
      " cycle-constants table; the clock under this load is lower than 2.4 GHz, so the true pipe occupancy is higher than"
      " the figure)." % (GHZ, SIMDS), "",
      "| kernel | launches / iteration | ms / iteration | MFMA instructions | MFMA utilisation |", "|---|---|---|---|---|"]
for name, a in [r_ for r_ in rows if r_[1].get("SQ_INSTS_MFMA", 0) > 0][:26]:
    util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (a["ns"] * GHZ * SIMDS) if a["ns"] else 0
    md.append("| `%s` | %.1f | %.3f | %.3g | %.2f |" % (name, a["n"] / ITERS, a["ns"] / 1e6 / ITERS, a.get("SQ_INSTS_MFMA", 0) / ITERS, util))
md += ["", "All kernels: %.2f ms of device time per iteration, matrix pipe busy %.2f of the SIMD-cycles; the kernels that"
       " issue matrix instructions: %.2f ms per iteration, busy %.2f."
       % (tot_ns / 1e6 / ITERS, tot_busy / (tot_ns * GHZ * SIMDS), mfma_ns / 1e6 / ITERS,
          sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for _, a in mfma_rows) / max(mfma_ns * GHZ * SIMDS, 1))]
out = os.path.join(ROOT, "gpurun_out", "r03_bev_mfma.md")
open(out, "w").write("\n".join(md) + "\n")
print("\n".join(md[:16]))
