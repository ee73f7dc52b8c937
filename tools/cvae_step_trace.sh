#!/bin/bash
# Per-kernel table of the recorded CVAE training step (configs[3], 4096 x 512): gpurun_out/<tag>_cvae_kernels.txt
R=$GRAFT_REPO_ROOT; TAG=${1:-t}; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cvae_$TAG -o cvae -- python3 $R/tools/cvae_step_run.py > $R/gpurun_out/${TAG}_cvae_trace.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("/tmp/p_cvae_$TAG/cvae_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = open("$R/gpurun_out/${TAG}_cvae_kernels.txt", "w")
out.write("total kernel time %.1f ms over the run (12 recorded steps + 2 eager)\n" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    out.write("%-110s calls %6s  total %8.2f ms  avg %8.1f us  %5.1f %%\n" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
head -45 $R/gpurun_out/${TAG}_cvae_kernels.txt
