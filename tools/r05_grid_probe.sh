O=gpurun_out/r05c/grid.txt; mkdir -p gpurun_out/r05c; rm -f $O
for rep in 1 2; do
for cfg in "default:512" "default:500" "sg:500" "sg3:768" "sg3:625" "sg4:1024" "sg4:834" "sg4:625" "sg4:500"; do
  v=${cfg%%:*}; g=${cfg##*:}
  if [ "$v" = default ]; then LIB=; else LIB=$PWD/build_variants/$v.so; fi
  CCAL_MAX_GRID=$g CCAL_LIB=$LIB timeout -k 10 200 python tools/probe.py --workload C5 --steps 40 --tag "$v/grid$g" 2>&1 | grep probe >> $O || exit 1
done; done; cat $O
