"""Copies what tools/profiles.sh <round> left under gpurun_out/ into profiles/ (tracked) and writes the per-instantiation
kernel-trace summary.  usage: collect_profiles.py [round-prefix]   (default r05)"""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def template_args(name):
    """'void ccal::k_lm_trial<0, false, 1, 1, false, false>(ccal::TrialArgs)' -> ['0', 'false', '1', '1', 'false', 'false']"""
    return [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")] if "<" in name else []


def trace_summary(path):
    by, meta = {}, {}
    with open(path) as f:
        for row in csv.DictReader(f):
            n = row["Kernel_Name"]
            if "k_lm_trial" not in n:
                continue
            by.setdefault(n, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
            a = template_args(n)
            if len(a) >= 3 and a[2] == "1":                    # a later trial: the instantiation the roofline is quoted on
                meta = {"grid_threads": [int(row["Workgroup_Size_X"]), int(row["Grid_Size_X"])], "vgpr": row["VGPR_Count"],
                        "sgpr": row["SGPR_Count"], "lds_bytes": row["LDS_Block_Size"], "scratch": row["Scratch_Size"]}
    out = {"by_kernel": {}}
    for n, v in sorted(by.items()):
        # A graph is sized for the trials the previous solve needed; a solve that ends earlier leaves SURPLUS launches that find
        # `done` and return at once (a few us).  They are not sweeps: launches shorter than 60 % of the instantiation's median
        # are counted apart and excluded from mean and median.
        med0 = statistics.median(v)
        full = [x for x in v if x >= 0.6 * med0]
        out["by_kernel"][n] = {"launches": len(full), "mean_us": statistics.fmean(full), "median_us": statistics.median(full),
                               "surplus_early_return_launches_excluded": len(v) - len(full),
                               "mean_us_all_launches": statistics.fmean(v)}
    out.update(meta)
    return out


runs = {"C3 1000x54 (bench.py --steps 50, as the driver runs it)": "c3", "C5 10000x150 on one GPU": "c5",
        "C5 per-GPU share at 8 GPUs (1250x150)": "c5shard"}
summary = {"what": "rocprofv3 --kernel-trace of the commands in tools/profiles.sh: duration of ccal::k_lm_trial by instantiation "
                   "<FLOW, LOOP, KIND, WPI, SOLO, MULTI> (KIND 0 = first launch of a solve, 1 = later trial, 2 = terminating prologue-only launch); surplus early-return launches excluded",
           "runs": {k: trace_summary(os.path.join(G, R + "_prof", d, "prof_kernel_trace.csv")) for k, d in runs.items()}}
json.dump(summary, open(os.path.join(P, R + "_kernel_trace_summary.json"), "w"), indent=1)
for src, dst in (("_prof/c3/prof_kernel_stats.csv", "_bench_kernel_stats.csv"), ("_prof/c5/prof_kernel_stats.csv", "_c5_kernel_stats.csv"),
                 ("_prof/c5shard/prof_kernel_stats.csv", "_c5_shard1250_kernel_stats.csv"), ("_prof/maps/prof_kernel_stats.csv", "_maps_kernel_stats.csv"),
                 ("_prof/stamps.log", "_sweep_wave_stamps.txt")):
    if os.path.exists(os.path.join(G, R + src)):
        shutil.copy(os.path.join(G, R + src), os.path.join(P, R + dst))
for src, dst in (("pmc/summary.json", "_pmc_traffic_summary.json"), ("pmc_c5/summary.json", "_c5_pmc_traffic_summary.json"),
                 ("pmc_util/summary.json", "_pmc_utilisation_summary.json"),
                 ("pmc_util_c5/summary.json", "_c5_pmc_utilisation_summary.json"), ("wave_stamps_block7.json", "_wave_stamps.json")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, R + dst))
for k, v in summary["runs"].items():
    print(k, {n.split("k_lm_trial")[1].split("(")[0]: (round(d["mean_us"], 2), round(d["median_us"], 2)) for n, d in v["by_kernel"].items()})
