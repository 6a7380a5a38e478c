"""Summarise a tools/pmc.sh run: per-launch FETCH_SIZE / WRITE_SIZE of ccal::k_lm_trial (sweeping launches) and the
calibration factors (bytes per counter unit) measured with tools/pmc_calib.hip.   pmc_summary.py [workload] [out dir]"""
import csv
import glob
import json
import os
import re
import statistics
import sys

WL = sys.argv[1] if len(sys.argv) > 1 else "C3"
OUT = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/pmc"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for tag in ("", "calib_"):
        files = glob.glob(os.path.join(ROOT, OUT, tag + ctr, "**", "*counter_collection.csv"),
                          recursive=True)
        vals = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != ctr:
                    continue
                # "void ccal::k_lm_trial<false>(ccal::TrialArgs)" -> "ccal::k_lm_trial" (single-GPU instantiation)
                name = re.sub(r"<.*>", "", row["Kernel_Name"].split("(")[0]).replace("void ", "").strip()
                if name == "ccal::k_lm_trial" and int(row["Grid_Size"]) <= 256:
                    name += "[prologue-only]"       # the terminating launch of a solve: one workgroup, no sweep
                vals.setdefault(name, []).append(float(row["Counter_Value"]))
        out[tag + ctr] = {k: dict(n=len(v), mean=statistics.mean(v), median=statistics.median(v)) for k, v in vals.items()}
calib_bytes = float(1 << 31)
summary = {"workload": WL}
f_unit = out.get("calib_FETCH_SIZE", {}).get("k_read8", {}).get("mean")
w_unit = out.get("calib_WRITE_SIZE", {}).get("k_write8", {}).get("mean")
summary["calibration"] = {"bytes": calib_bytes, "FETCH_SIZE_units_for_read": f_unit, "WRITE_SIZE_units_for_write": w_unit,
                          "bytes_per_FETCH_unit": calib_bytes / f_unit if f_unit else None,
                          "bytes_per_WRITE_unit": calib_bytes / w_unit if w_unit else None}
for k in ("ccal::k_lm_trial", "ccal::k_lm_row"):
    fs = out.get("FETCH_SIZE", {}).get(k)
    ws = out.get("WRITE_SIZE", {}).get(k)
    if fs and ws:  # noqa
        bf = summary["calibration"]["bytes_per_FETCH_unit"] or 1024.0
        bw = summary["calibration"]["bytes_per_WRITE_unit"] or 1024.0
        summary[k] = {"launches": fs["n"], "FETCH_SIZE_mean": fs["mean"], "WRITE_SIZE_mean": ws["mean"],
                      "hbm_read_bytes_per_launch": fs["mean"] * bf, "hbm_write_bytes_per_launch": ws["mean"] * bw,
                      "hbm_bytes_per_launch": fs["mean"] * bf + ws["mean"] * bw}
os.makedirs(os.path.join(ROOT, OUT), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, OUT, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
