# serialized kernel stats under several values of one environment variable: tools/env_kstat.sh VAR "v1 v2 ..." <kernel substring> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $2; do
  rm -rf $R/gpurun_out/ks_e
  export $1=$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_e -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --serialize $4 > $R/gpurun_out/ks_e.log 2>&1 || exit 1
  echo "== $1=$v"; python3 $R/tools/kstats.py $R/gpurun_out/ks_e 13 80 | grep -E "total|$3"
done
