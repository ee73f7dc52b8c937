"""Clusters of consecutive short launches in one step of a kernel trace (the --seq output of tools/trace_window.py
written with --seq-name-chars): where the torch glue between the big kernels sits.
usage: python tools/tiny_runs.py <seq.txt> [--max-us 6.6] [--min-run 3]"""
import argparse
import re

ap = argparse.ArgumentParser()
ap.add_argument("seq")
ap.add_argument("--max-us", type=float, default=6.6)
ap.add_argument("--min-run", type=int, default=3)
args = ap.parse_args()


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("at::native::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"vectorized_elementwise_kernel<\d+, ", "vec<", n)
    n = re.sub(r"elementwise_kernel_manual_unroll<128, 4, gpu_kernel_impl(_nocast)?<", "elt<", n)
    n = n.replace("binary_internal::", "").replace("BinaryFunctor<float, float, float, ", "Bin<")
    m = re.match(r"(vec|elt)<(\w+(<\w+(, \w+)*>)?)", n)
    if m:
        n = m.group(2)
    else:
        n = re.sub(r"\(.*", "", n)
    return n[:40]


rows = []
for i, line in enumerate(open(args.seq)):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+(.*)", line.rstrip("\n"))
    rows.append((i + 1, float(m.group(2)), short(m.group(3))))
runs, cur = [], []
for r in rows + [(0, 1e9, "")]:
    if r[1] < args.max_us:
        cur.append(r)
    else:
        if len(cur) >= args.min_run:
            runs.append(cur)
        cur = []
tiny = [r for r in rows if r[1] < args.max_us]
print("%d launches, %d under %.1f us (%.0f us); runs of >= %d:" % (len(rows), len(tiny), args.max_us,
                                                                  sum(r[1] for r in tiny), args.min_run))
for r in runs:
    print("%d-%d n=%d %.0f us: %s" % (r[0][0], r[-1][0], len(r), sum(x[1] for x in r), ", ".join(x[2] for x in r)))
