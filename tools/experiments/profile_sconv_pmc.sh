#!/bin/bash
# tools/profile_sconv_pmc.sh <tag> [ENV=VALUE ...]: SQ / LDS counters of the sparse-conv block kernels in bench.py --roofline-only
# (one rocprofv3 --pmc pass per counter group, kernel trace in the same pass) -> gpurun_out/<tag>/summary.txt: per kernel the
# average per launch of every counter and the derived shares.  The extra arguments are exported (GLX_SCONV_ARITH, GLX_SCONV_VARIANT).
R=$GRAFT_REPO_ROOT; TAG=${1:-sconv_pmc}; shift; OUT=$R/gpurun_out/$TAG
for e in "$@"; do export "$e"; done
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
B="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
# (a fourth group of TCP_* counters did not terminate on this pool: the pass ran until the call's limit)
i=0
for G in "$A" "$B" "$C"; do
  i=$((i+1))
  rm -rf /tmp/p_s$i
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/p_s$i -o s -- python3 $R/bench.py --roofline-only > $OUT/pass$i.log 2>&1
  cp $(find /tmp/p_s$i -name "*counter_collection.csv" | head -1) $OUT/sconv_counters_$i.csv 2>/dev/null
  cp $(find /tmp/p_s$i -name "*kernel_trace.csv" | head -1) $OUT/sconv_trace_$i.csv 2>/dev/null
done
python3 - $OUT > $OUT/summary.txt 2>&1 <<'PY'
import collections, csv, glob, os, sys
d = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for cc in sorted(glob.glob(os.path.join(d, "sconv_counters_*.csv"))):
    i = cc.rsplit("_", 1)[1].split(".")[0]
    dur = {}
    kt = os.path.join(d, "sconv_trace_%s.csv" % i)
    if os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    seen = set()
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "k_sconv_gemm" not in k:
            continue
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur and (r["Dispatch_Id"], i) not in seen:
            seen.add((r["Dispatch_Id"], i))
            out[k]["ns(pass %s)" % i].append(dur[r["Dispatch_Id"]])
for k, c in sorted(out.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print("==", k, "launches", max(len(v) for v in c.values()))
    for n in sorted(m):
        print("   %-34s %.4g" % (n, m[n]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        print("   shares of the wave-cycles: waiting %.2f | stalled at issue %.2f (LDS %.3f) | issuing %.3f (VALU %.3f LDS %.3f VMEM %.3f scalar %.3f)"
              % (m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc,
                 m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                 m.get("SQ_ACTIVE_INST_VMEM", 0) / wc, m.get("SQ_ACTIVE_INST_SCA", 0) / wc))
    ns = m.get("ns(pass 1)")
    if ns and "SQ_LDS_IDX_ACTIVE" in m:
        print("   LDS index unit active per CU: %.0f cycles of the launch's %.0f ns; bank-conflict cycles %.0f; MFMA busy per SIMD %.0f cycles"
              % (m["SQ_LDS_IDX_ACTIVE"] / 256, ns, m.get("SQ_LDS_BANK_CONFLICT", 0) / 256, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024))
PY
find $OUT -name "*.csv" -size +8M -delete
cat $OUT/summary.txt
