"""Per-step kernel table from a rocprofv3 --kernel-trace CSV: keeps the launches between the (N+1)-th last and
the last occurrence of a once-per-step marker kernel (so warm-up, MIOpen's find-mode trials and the sub-measurements
of bench.py do not pollute it) and aggregates by kernel name.
usage: python tools/trace_window.py <kernel_trace.csv> [--steps 10] [--marker k_kl_reg_loss] [--top 70] [--out file.md]"""
import argparse
import collections
import csv
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--marker", default="k_kl_reg_loss")
ap.add_argument("--top", type=int, default=70)
ap.add_argument("--out")
ap.add_argument("--seq", help="also write the launches of the LAST step in time order (start us, duration us, name)")
ap.add_argument("--seq-name-chars", type=int, default=0, help="--seq: raw kernel names cut to this many characters")
args = ap.parse_args()

rows = []
with open(args.trace) as f:
    rd = csv.DictReader(f)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
if len(marks) < args.steps + 1:
    sys.exit("marker %r seen %d times, need %d" % (args.marker, len(marks), args.steps + 1))
lo, hi = marks[-args.steps - 1], marks[-1]
win = rows[lo + 1:hi + 1]
span = (rows[hi][1] - rows[lo][1]) / 1e6 / args.steps


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"at::native::(\(anonymous namespace\)::)?", "", n)
    return n[:110]


agg = collections.OrderedDict()
for s, e, n in win:
    d = agg.setdefault(short(n), [0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
tot = sum(d[1] for d in agg.values())
lines = ["window: %d steps, %.3f ms/step wall between markers, %.3f ms/step summed kernel time, %d launches/step"
         % (args.steps, span, tot / 1e3 / args.steps, len(win) // args.steps), "",
         "| kernel | launches/step | us/step | us/launch | % |", "|---|---|---|---|---|"]
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.top]:
    lines.append("| `%s` | %.1f | %.1f | %.1f | %.1f |" % (k, c / args.steps, us / args.steps, us / c, 100 * us / tot))
txt = "\n".join(lines)
print(txt)
if args.seq:
    last = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = last[0][0]
    with open(args.seq, "w") as f:
        for s_, e_, n_ in last:
            f.write("%10.1f %8.1f  %s\n" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3,
                                            n_[:args.seq_name_chars] if args.seq_name_chars else short(n_)))
if args.out:
    with open(args.out, "w") as f:
        f.write(txt + "\n")
