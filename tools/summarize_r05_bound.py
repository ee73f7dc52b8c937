"""profiles-ready summaries of tools/profile_r05_bound.sh: per kernel the average of every collected counter per launch, the
average duration (kernel trace of the same pass), and the derived fractions.  argv[1] = the pass directory.
Writes <dir>/r05_sconv_bound.md, <dir>/r05_bev_mfma.md, <dir>/r05_bev_pmc.json."""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
SIMDS, CUS = 1024, 256


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return None
    targs = (m.group(2) or "")
    targs = ",".join(a.strip() for a in targs.strip("<>").split(",")) if targs else ""
    return m.group(1) + ("<" + targs + ">" if targs else "")


def collect(prefix):
    """{kernel: {counter: mean per launch, 'ns': mean duration, 'n': launches}} over all passes of one program."""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for cc in sorted(glob.glob(os.path.join(d, prefix + "_counters_*.csv"))):
        i = cc.rsplit("_", 1)[1].split(".")[0]
        kt = os.path.join(d, "%s_trace_%s.csv" % (prefix, i))
        dur = {}
        if os.path.exists(kt):
            for r in csv.DictReader(open(kt)):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        seen = set()
        for r in csv.DictReader(open(cc)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            did = r["Dispatch_Id"]
            if did in dur and (did, i) not in seen:
                seen.add((did, i))
                out[k]["ns@" + r["Counter_Name"]].append(dur[did])
    res = {}
    for k, c in out.items():
        res[k] = {n: sum(v) / len(v) for n, v in c.items()}
        res[k]["launches"] = max(len(v) for v in c.values())
    return res


def table(res, want, title, notes):
    md = ["# " + title, ""] + notes + [""]
    cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_INSTS_MFMA", "SQ_WAVES", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT",
            "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
            "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INSTS_SALU", "SQ_INST_LEVEL_LDS",
            "SQ_INST_LEVEL_VMEM", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"]
    for k in sorted(res, key=lambda k: -res[k].get("ns@SQ_WAVE_CYCLES", res[k].get("ns@GRBM_GUI_ACTIVE", 0))):
        if not any(w in k for w in want):
            continue
        c = res[k]
        ns = c.get("ns@SQ_WAVE_CYCLES") or c.get("ns@GRBM_GUI_ACTIVE") or 0
        md += ["## `%s`  (%d launches per pass, %.1f us average under the profiler)" % (k, c["launches"], ns / 1e3), "",
               "| counter | per launch |", "|---|---|"]
        for n in cols:
            if n in c:
                md.append("| %s | %.4g |" % (n, c[n]))
        md.append("")
        wc, busy = c.get("SQ_WAVE_CYCLES"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        d_ = []
        gui = c.get("GRBM_GUI_ACTIVE")
        ns_g = c.get("ns@GRBM_GUI_ACTIVE")
        ghz = gui / 8.0 / ns_g if gui and ns_g else None
        if ghz:
            d_.append("effective clock GRBM_GUI_ACTIVE / 8 / wall = **%.2f GHz** (reads high on dispatches shorter than 0.3 ms: "
                      "MI355X_MICROARCH.md 'DVFS give-back')" % ghz)
        if busy and ns:
            d_.append("matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (wall x 1024 SIMDs x f): **%.3f** at the nominal 2.4 GHz"
                      % (busy / (ns * 2.4 * SIMDS)) + (", **%.3f** at the effective clock" % (busy / (ns * ghz * SIMDS)) if ghz else ""))
        if busy and c.get("SQ_INSTS_MFMA"):
            d_.append("busy cycles per MFMA instruction: %.1f" % (busy / c["SQ_INSTS_MFMA"]))
        if wc:
            for n, lab in (("SQ_WAIT_ANY", "waves parked (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue stalls"),
                           ("SQ_ACTIVE_INST_ANY", "issuing")):
                if n in c:
                    d_.append("%s = %.3f of the wave-cycles (%s)" % (n, c[n] / wc, lab))
        if c.get("SQ_LDS_IDX_ACTIVE") and ns:
            d_.append("LDS array active = SQ_LDS_IDX_ACTIVE / (wall x 2.4 GHz x 256 CUs) = %.3f; bank conflicts %.3f of the active "
                      "cycles" % (c["SQ_LDS_IDX_ACTIVE"] / (c.get("ns@SQ_LDS_IDX_ACTIVE", ns) * 2.4 * CUS),
                                  c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
        if c.get("SQ_WAIT_INST_LDS") is not None and wc:
            pass
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            f, w = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
            d_.append("HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 counts a 128-byte read request as "
                      "64 bytes) = %.1f MB" % ((2 * f + w) / 1e6))
        md += ["* " + x for x in d_] + [""]
    return md


s = collect("sconv")
b = collect("bev")
notes_s = ["`bash tools/profile_r05_bound.sh`: rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python3 bench.py "
           "--roofline-only (the event-bracketed eager pass of configs[1]'s launches), averages per launch; durations are the "
           "kernel trace of the same pass (profiled runs hold a lower clock than un-profiled ones).",
           "The in-kernel clock and the per-phase cycle shares of the loop (TRACE build) are in `r05_sconv_tiles.txt`."]
open(os.path.join(d, "r05_sconv_bound.md"), "w").write("\n".join(table(s, ["k_sconv_gemm<64,64", "k_sconv_gemm<128,64", "k_sconv_mfma<32,32"],
                                                                   "What bounds k_sconv_gemm<64,64>: counters (round 5)", notes_s)) + "\n")
notes_b = ["`bash tools/profile_r05_bound.sh`: rocprofv3 --kernel-trace --pmc <group> -- python3 tools/bev_micro.py 10 -- the own 3x3 "
           "kernels alone on the block layers' shapes (4 x 64 x 200 x 176 and 4 x 128 x 100 x 88, random data), averages per launch.",
           "Un-profiled event-timed averages of the same program: `r05_bev_micro.txt`."]
open(os.path.join(d, "r05_bev_mfma.md"), "w").write("\n".join(table(b, ["k_conv3x3"], "BEV 3x3 convolutions: matrix pipe, LDS, waits (round 5)",
                                                                  notes_b)) + "\n")
pmc = {k: {n: v[n] for n in ("FETCH_SIZE", "WRITE_SIZE", "launches") if n in v} for k, v in b.items() if "k_conv3x3" in k}
for k, v in pmc.items():
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
json.dump(pmc, open(os.path.join(d, "r05_bev_pmc.json"), "w"), indent=1)
print("wrote", os.listdir(d))
