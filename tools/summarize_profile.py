"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh) into the committed evidence under
profiles/: <tag>_kernel_stats.csv, <tag>_bench.json, <tag>_summary.md, traffic_latest.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not fs:
        raise SystemExit("missing " + pattern)
    return fs[-1]


def short(name):
    """k_sconv_gemm<64, 64, 64, 8, 4>(...) -> k_sconv_gemm<64,64> (the key bench.py uses)."""
    for k in ("k_sconv_gemm", "k_sconv_mfma"):
        i = name.find(k + "<")
        if i >= 0:
            a = [x.strip() for x in name[i + len(k) + 1:name.find(">", i)].split(",")]
            return "%s<%s,%s>" % (k, a[0], a[1])
    return None


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(os.path.join(R, d, "*", "*counter_collection.csv")))):
        if r["Counter_Name"] == counter and short(r["Kernel_Name"]):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


stats = newest(os.path.join(R, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(P, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(R, "bench.json"), os.path.join(P, tag + "_bench.json"))
b = json.load(open(os.path.join(R, "bench.json")))
rows = list(csv.DictReader(open(stats)))
fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
hit, miss = per_kernel("pmc_l2", "TCC_HIT_sum"), per_kernel("pmc_l2", "TCC_MISS_sum")
traffic = {k: int(fetch[k] * 1024 * 2 + write.get(k, 0) * 1024) for k in fetch}
json.dump(traffic, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1, sort_keys=True)

roof = b["roofline"]
dom = roof["kernel"]
prof_avg = [float(r["AverageNs"]) / 1e3 for r in rows if short(r["Name"]) == dom]
steps = max(int(r["Calls"]) for r in rows if "k_mean_vfe" in r["Name"])
md = ["# Profile %s (MI355X, `bash tools/profile_round.sh %s`, summarised by tools/summarize_profile.py)" % (tag, tag), "",
      "Headline run (no profiler): `%s_bench.json`.  Kernel-trace run: `rocprofv3 --kernel-trace --stats -- "
      "python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline` -> `%s_kernel_stats.csv`." % (tag, tag), "",
      "## Headline", "",
      "| frames/s | ms/step | mode | dominant kernel | avg launch, HIP events in bench | avg launch, rocprof | "
      "algorithmic GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|---|",
      "| %.0f | %.4f | %s | `%s` | %.1f us | %.1f us | %.0f | %.3f |"
      % (b["value"], b["ms_per_step"], b["config"]["mode"], dom, roof["avg_launch_us"],
         prof_avg[0] if prof_avg else float("nan"), roof["achieved"], roof["frac"]), ""]
if "cpu_baseline" in b:
    md += ["CPU baseline (oracle, scalar C port, %d core): %s frames/s -- %s." % (
        b["cpu_baseline"]["cores"], b["cpu_baseline"]["value"], b["cpu_baseline"]["sample"]), ""]
md += ["All sparse convs of a step: %.3f ms, %.0f GB/s algorithmic (frac %.3f)." % (
    roof["all_sparse_conv"]["ms_per_step"], roof["all_sparse_conv"]["achieved"], roof["all_sparse_conv"]["frac"]), "",
    "| kernel | launches/step | us/launch | algorithmic GB/s |", "|---|---|---|---|"]
for k, v in sorted(roof["all_sparse_conv"]["per_kernel"].items()):
    md.append("| `%s` | %d | %.1f | %.0f |" % (k, v["launches_per_step"], v["us_per_launch"], v["GBps"]))
md += ["", "## HBM traffic of the sparse-conv kernels (separate `--pmc` passes, `--mode static`)", "",
       "`FETCH_SIZE` / `WRITE_SIZE` are KB per dispatch (mean over the launches of the pass).  Correction per",
       "`MI355X_MICROARCH.md` (HBM / rocprofv3 section): on gfx950 FETCH_SIZE counts 16-byte-per-lane reads at half",
       "their size, and every gather / weight load here is `global_load_dwordx4`, so",
       "`traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024` bytes per launch (`traffic_latest.json`, read by bench.py).", "",
       "| kernel | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB/launch | L2 hit rate |", "|---|---|---|---|---|"]
for k in sorted(fetch):
    hr = hit.get(k, 0) / max(hit.get(k, 0) + miss.get(k, 0), 1)
    md.append("| `%s` | %.0f | %.0f | %.1f | %.3f |" % (k, fetch[k], write.get(k, 0), traffic[k] / 1e6, hr))
md += ["", "Dominant kernel: algorithmic bytes/launch (SURVEY 8d) = %.1f MB, measured HBM traffic = %.1f MB: the"
       % (roof["alg_bytes_per_launch"] / 1e6, traffic.get(dom, 0) / 1e6),
       "output-stationary kernel re-reads neighbour rows from L2, not from HBM, and never read-modify-writes its output.", "",
       "## Kernels by device time (%d steps: graph replays + the event-bracketed roofline pass)" % steps, "",
       "| kernel | calls | avg us | us/step | % |", "|---|---|---|---|---|"]
for r in rows[:28]:
    md.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (r["Name"][:72].replace("|", "/"), r["Calls"],
                                                   float(r["AverageNs"]) / 1e3,
                                                   float(r["TotalDurationNs"]) / steps / 1e3, float(r["Percentage"])))
md += ["", "## Sparse-conv kernel variants measured during round 1 (layer 64->64, N = 48 147, R = 386 953)", "",
       "Kernel-only time from HIP events (`tools/sconv_sweep.py`); the variants marked (removed) lived in the tree,",
       "lost, and were deleted (git history has them).", "",
       "| variant | us |", "|---|---|",
       "| block implicit GEMM: LDS-staged gathers, chunk split over 4 waves, 64 rows x 8 waves (**default, Cout >= 64**) | 74 |",
       "| same, 64 rows x 4 waves, split 2 / split 4 | 80 / 81 |",
       "| same, 128 rows x 8 waves / 32 rows x 4 waves / 64 x 8 split 2 (removed) | 92 / 101 / 104 |",
       "| same, 80 / 96 / 112 rows x 8 waves, 96 rows x 12 waves (removed; 2 blocks per CU) | 106 / 85 / 86 / 87 |",
       "| same, software-pipelined loads 2 / 3 / 4 steps ahead, exact `vmcnt(8/7/6)` (removed) | 77 / 77 / 103 |",
       "| same, every LDS read of a chunk (fragments, slot, accumulator row) hoisted above the MFMA chain with `sched_barrier` (removed) | 78 |",
       "| whole-chunk waves, register gathers, 64 rows x 4 waves (**default, Cout <= 32**) | 81-86 |",
       "| same with column split 2 over 8 waves / split 4 over 8 waves (removed) | 80-82 / 115 |",
       "| same, double-buffered W (2 blocks/CU) | 112 |",
       "| same, 128 rows x 4 / x 8 waves, 256 rows x 8 waves, 32 rows, 64 rows x 8 waves (removed) | 100-104 / 89-94 / 103 / 114-121 / 114 |",
       "| wave-private 48-row tiles, W fragments from L2 (removed) / + deep software pipeline (removed) | 107 / 105 |",
       "| register-resident 32-row tiles, one wave per block (removed) | 141 |",
       "| column-owner ring, 512 threads (removed) | 121-146 |",
       "| XCD-contiguous tile ranges on the default kernel (not enabled: XCD imbalance) | +6 % on this layer, -5 % on small ones |",
       "| first version (one wave per 16-row tile, global-atomic-free but no compaction) | 230 |"]
open(os.path.join(P, tag + "_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md[:24]))
