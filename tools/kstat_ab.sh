# serialized kernel stats for old/new library builds (one call): tools/kstat_ab.sh <old.so> <kernel substring>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for x in old new; do
  rm -rf $R/gpurun_out/ks_$x
  if [ $x = old ]; then export ECGMM_LIB=$R/$1; else unset ECGMM_LIB; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$x -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --serialize > $R/gpurun_out/ks_$x.log 2>&1 || exit 1
  echo "== $x"; python3 $R/tools/kstats.py $R/gpurun_out/ks_$x 13 80 | grep -E "total|$2"
done
