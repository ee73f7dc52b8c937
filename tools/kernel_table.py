"""Per-kernel table of a rocprofv3 --kernel-trace CSV: python tools/kernel_table.py trace.csv [top]"""
import csv
import sys
from collections import defaultdict

if __name__ == "__main__":
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            tot[r["Kernel_Name"]] += d
            cnt[r["Kernel_Name"]] += 1
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    total = sum(tot.values())
    print("total kernel time %.1f ms, %d launches" % (total / 1e3, sum(cnt.values())))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
        print("%-110s calls %6d  total %9.2f ms  avg %9.1f us  %5.1f %%" % (k[:110], cnt[k], v / 1e3, v / cnt[k], 100 * v / total))
