#!/bin/bash
# HBM-side traffic of the sweep kernel from PMC counters, one counter per pass (the TCC block
# cannot hold FETCH_SIZE and WRITE_SIZE together; --pmc runs carry --kernel-trace only), plus a calibration of both
# counters on a known byte count with this code's 8-B-per-lane access width.  Run via gpurun from the repo root.
#   tools/pmc.sh [workload=C3] [out=gpurun_out/pmc]
set -e
WL=${1:-C3}; OUT=${2:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib tools/pmc_calib.hip
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$ctr $OUT/calib_$ctr
  timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$ctr -o pmc -- python3 bench.py --workload $WL --steps 20 --warmup 2 --no-cpu-baseline --timed-only > $OUT/bench_$ctr.log 2>&1
  timeout -k 10 100 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/calib_$ctr -o pmc -- /tmp/pmc_calib > $OUT/calib_$ctr.log 2>&1
done
python3 tools/pmc_summary.py $WL $OUT
