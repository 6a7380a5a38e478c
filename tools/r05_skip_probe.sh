O=gpurun_out/r05a/skip.txt; rm -f $O
for rep in 1 2; do for v in default skipA skipC skipAC; do
  if [ "$v" = default ]; then LIB=; else LIB=$PWD/build_variants/$v.so; fi
  for n in "" 1250; do
  CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C5 ${n:+--images $n} --b2b-only --tag $v 2>&1 | grep probe >> $O || exit 1
  done
  CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C3 --b2b-only --tag $v 2>&1 | grep probe >> $O || exit 1
done; done; cat $O
