#!/bin/bash
# round-2 profile set: rocprofv3 kernel stats (C3 bench as the driver runs it, C5 whole and per-GPU shard), PMC traffic, PMC utilisation, wave stamps
O=gpurun_out/r02_prof; mkdir -p $O gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/c3.log 2>&1; echo "c3 stats exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o prof -- python3 bench.py --workload C5 --steps 20 --warmup 3 --no-cpu-baseline --timed-only > $O/c5.log 2>&1; echo "c5 stats exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5shard -o prof -- python3 bench.py --workload C5 --images 1250 --steps 30 --warmup 3 --no-cpu-baseline --timed-only > $O/c5shard.log 2>&1; echo "c5 shard stats exit $?"
cut -c1-150 $O/c3/prof_kernel_stats.csv | head -8; cut -c1-150 $O/c5/prof_kernel_stats.csv | head -5; cut -c1-150 $O/c5shard/prof_kernel_stats.csv | head -4
bash tools/pmc.sh > $O/pmc_traffic.log 2>&1; tail -25 $O/pmc_traffic.log
bash tools/pmc_util.sh C3 gpurun_out/pmc_util > $O/pmc_util_c3.log 2>&1; tail -40 $O/pmc_util_c3.log | head -60
bash tools/pmc_util.sh C5 gpurun_out/pmc_util_c5 > $O/pmc_util_c5.log 2>&1; tail -5 $O/pmc_util_c5.log
timeout -k 10 200 python tools/stamps.py > $O/stamps.log 2>&1; tail -20 $O/stamps.log | head -16
