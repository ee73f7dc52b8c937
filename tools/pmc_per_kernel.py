"""Average of one rocprofv3 PMC counter per sparse-conv kernel instantiation (JSON on stdout).
usage: python tools/pmc_per_kernel.py <counter_collection.csv> <COUNTER_NAME>"""
import collections
import csv
import json
import sys


def short(name):
    for k in ("k_sconv_gemm", "k_sconv_mfma"):
        i = name.find(k + "<")
        if i >= 0:
            a = [x.strip() for x in name[i + len(k) + 1:name.find(">", i)].split(",")]
            return "%s<%s,%s>" % (k, a[0], a[1])
    return None


agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and short(r["Kernel_Name"]):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
print(json.dumps({k: sum(v) / len(v) for k, v in sorted(agg.items())}, indent=1))
