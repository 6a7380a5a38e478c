#!/bin/bash
# Utilisation counters of ccal::k_lm_trial (MFMA busy, VALU/LDS instruction counts, waits, instruction-cache misses);
# SQ / SQC counters only, <= 8 per pass, kernel-trace only.  Run via gpurun from the repo root.
#   tools/pmc_util.sh [workload=C3] [out=gpurun_out/pmc_util]
WL=${1:-C3}; OUT=${2:-gpurun_out/pmc_util}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
run() { # name, counters...
  local name=$1; shift
  rm -rf $OUT/$name
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python3 bench.py --workload $WL --steps 20 --warmup 2 --no-cpu-baseline --timed-only > $OUT/bench_$name.log 2>&1 || echo "pass $name failed"
}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY
run c GRBM_GUI_ACTIVE
run d SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
WL=$WL OUT=$OUT python3 - <<'PY'
import csv, glob, json, os, statistics
root = os.environ.get("GRAFT_REPO_ROOT", ".")
out_dir = os.path.join(root, os.environ["OUT"])
out = {}
for f in glob.glob(os.path.join(out_dir, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "ccal::k_lm_trial" not in row["Kernel_Name"] or int(row["Grid_Size"]) <= 256:
            continue                      # sweeping launches only (the terminating one has a single workgroup)
        out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {"workload": os.environ["WL"], "kernel": "ccal::k_lm_trial, sweeping launches of the timed region",
           "counters": {k: {"launches": len(v), "mean": statistics.mean(v)} for k, v in sorted(out.items())}}
json.dump(summary, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
