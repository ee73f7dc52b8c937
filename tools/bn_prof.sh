cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -o t -- python3 $GRAFT_REPO_ROOT/tools/bn_time.py > /dev/null 2>&1; python3 - <<PY
import csv, collections, statistics
rows=list(csv.DictReader(open("/tmp/pb/t_kernel_trace.csv")))
seq=[(int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:34]) for r in rows if "k_bn" in r["Kernel_Name"]]
seq.sort(); cnt=0; out=collections.OrderedDict()
for t,d,n in seq:
    if "stats<false>" in n: cnt+=1
    out.setdefault(((cnt-1)//101,n),[]).append(d)
for (c,n),v in out.items(): print(c, n, "%.1f"%statistics.median(v))
PY
