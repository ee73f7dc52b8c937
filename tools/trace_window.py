"""Per-step kernel table from a rocprofv3 --kernel-trace CSV: keeps the launches between the (N+1)-th last and
the last occurrence of a once-per-step marker kernel (so warm-up, MIOpen's find-mode trials and the sub-measurements
of bench.py do not pollute it) and aggregates by kernel name.
usage: python tools/trace_window.py <kernel_trace.csv> [--steps 10] [--marker k_roi_head_losses] [--top 70] [--out file.md]"""
import argparse
import collections
import csv
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--marker", default="k_roi_head_losses")
ap.add_argument("--top", type=int, default=70)
ap.add_argument("--out")
ap.add_argument("--seq", help="also write the launches of the LAST step in time order (start us, duration us, name)")
ap.add_argument("--seq-name-chars", type=int, default=0, help="--seq: raw kernel names cut to this many characters")
ap.add_argument("--launches", help="substring of a kernel name: table of EVERY launch of it in the last step -- grid, duration, "
                                   "what ran concurrently (other queues) and for how much of its duration")
ap.add_argument("--launches-out")
args = ap.parse_args()

rows = []
with open(args.trace) as f:
    rd = csv.DictReader(f)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                     r.get("Queue_Id", "")))
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
for s, e, n, *_ in win:
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
        for s_, e_, n_, g_, w_, q_ in last:
            f.write("%10.1f %8.1f  q%-3s %s\n" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, q_,
                                                  n_[:args.seq_name_chars] if args.seq_name_chars else short(n_)))
if args.out:
    with open(args.out, "w") as f:
        f.write(txt + "\n")
if args.launches:
    last = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = last[0][0]
    out = ["launches of `%s` in the last step (time order); overlap = share of the launch's duration during which a kernel of "
           "ANOTHER queue was running" % args.launches, "",
           "| # | start us | us | workgroups | queue | overlap | concurrent kernels (share of this launch) |", "|---|---|---|---|---|---|---|"]
    k = 0
    for s_, e_, n_, g_, w_, q_ in last:
        if args.launches not in n_:
            continue
        k += 1
        conc = collections.OrderedDict()
        covered = []
        for s2, e2, n2, g2, w2, q2 in last:
            if q2 == q_ or e2 <= s_ or s2 >= e_:
                continue
            ov = min(e_, e2) - max(s_, s2)
            conc[short(n2)[:48]] = conc.get(short(n2)[:48], 0) + ov
            covered.append((max(s_, s2), min(e_, e2)))
        covered.sort()
        tot_c, cur = 0, None
        for a_, b_ in covered:                      # union of the overlap intervals
            if cur is None or a_ > cur[1]:
                if cur:
                    tot_c += cur[1] - cur[0]
                cur = [a_, b_]
            else:
                cur[1] = max(cur[1], b_)
        if cur:
            tot_c += cur[1] - cur[0]
        dur = max(e_ - s_, 1)
        out.append("| %d | %.1f | %.1f | %d | %s | %.0f %% | %s |" % (
            k, (s_ - t0) / 1e3, dur / 1e3, g_ // max(w_, 1), q_, 100.0 * tot_c / dur,
            ", ".join("%s %.0f %%" % (n, 100.0 * v / dur) for n, v in sorted(conc.items(), key=lambda kv: -kv[1])[:4])))
    txt2 = "\n".join(out)
    print(txt2)
    if args.launches_out:
        with open(args.launches_out, "w") as f:
            f.write(txt2 + "\n")
