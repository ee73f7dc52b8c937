"""Queue occupancy and gaps of one training step from tools/trace_window.py --seq output."""
import collections, sys
rows = []
for l in open(sys.argv[1]):
    p = l.split(None, 3)
    rows.append((float(p[0]), float(p[1]), p[2], p[3].strip() if len(p) > 3 else "(unnamed)"))
qs = collections.Counter(r[2] for r in rows)
end = max(r[0] + r[1] for r in rows)
print("queues", dict(qs), "step span us", round(end))
for q in qs:
    rr = [r for r in rows if r[2] == q]
    print(q, "launches", len(rr), "busy us", round(sum(r[1] for r in rr)), "first", round(rr[0][0]), "last end", round(max(r[0] + r[1] for r in rr)))
iv = sorted((r[0], r[0] + r[1]) for r in rows)
cov, cur, gaps = 0, None, []
for a, b in iv:
    if cur is None:
        cur = [a, b]
    elif a > cur[1]:
        gaps.append((cur[1], a)); cov += cur[1] - cur[0]; cur = [a, b]
    else:
        cur[1] = max(cur[1], b)
cov += cur[1] - cur[0]
print("some kernel running", round(cov), "us; nothing running", round(end - cov), "us in", len(gaps), "gaps; gaps > 3 us:",
      sum(1 for a, b in gaps if b - a > 3), "=", round(sum(b - a for a, b in gaps if b - a > 3)), "us")
small = [r for r in rows if r[1] < 7]
print("kernels under 7 us:", len(small), "sum", round(sum(r[1] for r in small)), "us")
# windows where only small kernels run: sum of (gap + small kernel) chains
by = collections.Counter()
for r in small:
    by[(r[2], r[3][:60])] += 1
for k, v in by.most_common(25):
    print("  ", v, k)
print("largest gaps (nothing running): start us, length, kernel before -> kernel after")
ends = sorted(rows, key=lambda r: r[0] + r[1])
for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:40]:
    before = max((r for r in rows if r[0] + r[1] <= a + 0.05), key=lambda r: r[0] + r[1])
    after = min((r for r in rows if r[0] >= b - 0.05), key=lambda r: r[0])
    print("%9.1f %6.1f  %s %s -> %s %s" % (a, b - a, before[2], before[3][:44], after[2], after[3][:44]))
