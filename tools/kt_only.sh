cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O; rm -rf $O/kt
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $O/kt.log 2>&1 || exit 1
python3 $R/tools/kstats.py $O/kt 13 60 > $O/kernel_stats.txt
head -5 $O/kernel_stats.txt
