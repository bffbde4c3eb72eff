#!/usr/bin/env python3
"""Folds rocprofv3 --pmc passes of SQ / GRBM counters into per-kernel matrix-pipe and issue statistics.

usage: pmc_sq_to_json.py <dir with pmc_mfma*/ and pmc_sq*/ sub-directories> <out.json> <label>

Units (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units" and "rocprofv3 PMC slots"): SQ_VALU_MFMA_BUSY_CYCLES counts shader
cycles during which a SIMD's matrix pipe is busy, summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
kernel's duration in shader cycles is GRBM_GUI_ACTIVE / 8; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count
quad-cycles.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)  (rocprofiler's own MfmaUtil
formula, derived_counters.xml, with CU_NUM = 256); tools/mfma_probe-style calibration: profiles/*_pmc_sq_calibration.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CUS, SIMDS = 256, 4


def fold(dirs):
    """-> {kernel: {counter: mean per launch}}, {kernel: launches}"""
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            names = {}
            for row in csv.DictReader(open(f)):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (disp, ctr), v in per.items():
                acc[names[disp]][ctr].append(v)
    out, n = {}, {}
    for k, d in acc.items():
        out[k] = {c: sum(v) / len(v) for c, v in d.items()}
        n[k] = max(len(v) for v in d.values())
    return out, n


def short(name):
    # "void (anonymous namespace)::kernel<..>(args)" -> "kernel<..>": the namespace prefix must go BEFORE the cut at the argument list
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


def main():
    root, outp, label = sys.argv[1], sys.argv[2], sys.argv[3]
    dirs = sorted(glob.glob(os.path.join(root, "pmc_mfma*")) + glob.glob(os.path.join(root, "pmc_sq*")))
    vals, n = fold(dirs)
    res = {}
    for k in sorted(vals):
        v = vals[k]
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui < 2000:          # < ~1 us: fills, copies
            continue
        e = {"launches_averaged": n[k], "kernel_cycles(GRBM_GUI_ACTIVE/8)": gui}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            e["SQ_VALU_MFMA_BUSY_CYCLES"] = v["SQ_VALU_MFMA_BUSY_CYCLES"]
            e["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * CUS * SIMDS)
        if "SQ_INSTS_VALU_MFMA_MOPS_F16" in v:
            e["SQ_INSTS_VALU_MFMA_MOPS_F16"] = v["SQ_INSTS_VALU_MFMA_MOPS_F16"]
        for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in v:
                e[c] = v[c]
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            e["SQ_WAVE_CYCLES(quad)"] = wc
            for c, nm in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac"),
                          ("SQ_ACTIVE_INST_VALU", "active_inst_valu_frac"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds_frac"),
                          ("SQ_ACTIVE_INST_VMEM", "active_inst_vmem_frac"), ("SQ_ACTIVE_INST_MISC", "active_inst_misc_frac"),
                          ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac")):
                if c in v:
                    e[nm] = v[c] / wc
        if "SQ_BUSY_CYCLES" in v:
            e["SQ_BUSY_CYCLES(quad)"] = v["SQ_BUSY_CYCLES"]
        if "SQ_WAVES" in v:
            e["SQ_WAVES"] = v["SQ_WAVES"]
        res[short(k)] = e
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["_meta"] = {"label": label, "csrc_sha": bench.csrc_digest(),
                    "normalisation": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs); *_frac of SQ rows = counter / SQ_WAVE_CYCLES (both quad-cycles, summed over waves)",
                    "passes": [os.path.basename(d) for d in dirs]}
    json.dump(res, open(outp, "w"), indent=1)
    for k, e in res.items():
        if k == "_meta":
            continue
        print("%-58s n=%5d cyc %9.0f mfma_busy %6.3f wait_any %5.2f wait_inst %5.2f active %5.2f valu %5.2f" % (
            k[:58], e["launches_averaged"], e["kernel_cycles(GRBM_GUI_ACTIVE/8)"], e.get("mfma_busy_frac", float("nan")),
            e.get("wait_any_frac", float("nan")), e.get("wait_inst_any_frac", float("nan")), e.get("active_inst_any_frac", float("nan")),
            e.get("active_inst_valu_frac", float("nan"))))


if __name__ == "__main__":
    main()
