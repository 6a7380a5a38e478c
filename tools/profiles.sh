#!/bin/bash
# The committed profile set of a round (run through gpurun from the repository root, then tools/collect_profiles.py <round>
# here): rocprofv3 kernel stats of the C3 bench as the driver runs it, of C5 whole and of its per-GPU share at 8 GPUs, of the
# map / metric entry points; PMC traffic (C3 and C5) and utilisation passes (their own runs: --pmc with --kernel-trace only);
# wave stamps.          tools/profiles.sh <round, e.g. r03>
R=${1:-r05}; O=gpurun_out/${R}_prof; mkdir -p $O gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o prof -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $O/c3.log 2>&1; echo "c3 stats exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o prof -- python3 bench.py --workload C5 --steps 20 --warmup 3 --no-cpu-baseline --timed-only > $O/c5.log 2>&1; echo "c5 stats exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5shard -o prof -- python3 bench.py --workload C5 --images 1250 --steps 30 --warmup 3 --no-cpu-baseline --timed-only > $O/c5shard.log 2>&1; echo "c5 shard stats exit $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/maps -o prof -- python3 tools/maps_variance.py > $O/maps.log 2>&1; echo "maps stats exit $?"
for d in c3 c5 c5shard maps; do cut -c1-150 $O/$d/prof_kernel_stats.csv | head -7; done
bash tools/pmc.sh C3 gpurun_out/pmc > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log
bash tools/pmc.sh C5 gpurun_out/pmc_c5 > $O/pmc_traffic_c5.log 2>&1; tail -12 $O/pmc_traffic_c5.log
bash tools/pmc_util.sh C3 gpurun_out/pmc_util > $O/pmc_util_c3.log 2>&1; tail -5 $O/pmc_util_c3.log
bash tools/pmc_util.sh C5 gpurun_out/pmc_util_c5 > $O/pmc_util_c5.log 2>&1; tail -5 $O/pmc_util_c5.log
timeout -k 10 200 python tools/stamps.py > $O/stamps.log 2>&1; tail -20 $O/stamps.log | head -16
