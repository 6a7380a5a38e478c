#!/bin/bash
# Round 5: waves per image for boards of ONE 64-corner pass (P = 54: BASELINE.json configs[1] / [2]): the image's corners split
# over 2 / 4 waves of the workgroup (CCAL_WPI, gate lifted with CCAL_WPI_MINP=8), shipped library, one box, alternating.
#   tools/r05_wpi_small_probe.sh <out-name> [lib]
O=gpurun_out/$1; LIB=$2; mkdir -p $(dirname $O); rm -f $O
for rep in 1 2; do for n in 6 25 100 250 500 1000; do for w in 1 2 4; do
  wl=C1; [ $n -ge 500 ] && wl=C3
  CCAL_LIB=$LIB CCAL_WPI_MINP=8 CCAL_WPI=$w CCAL_WPI_CAP=1024 timeout -k 10 200 python tools/probe.py --workload $wl --images $n --steps 200 --tag "wpi$w" 2>&1 | grep probe >> $O || exit 1
done; done; done
cat $O
