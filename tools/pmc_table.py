"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV: python tools/pmc_table.py counters.csv <kernel substring>"""
import csv
import sys
from collections import defaultdict

if __name__ == "__main__":
    path, sub = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    tot, cnt = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if sub in r["Kernel_Name"]:
                key = (r["Kernel_Name"][:70], r["Counter_Name"])
                tot[key] += float(r["Counter_Value"])
                cnt[key].add(r["Dispatch_Id"])
    for (k, c), v in sorted(tot.items()):
        n = len(cnt[(k, c)])
        print("%-72s %-24s launches %5d  per launch %16.1f" % (k, c, n, v / n))
