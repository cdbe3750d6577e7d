cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt5 $R/gpurun_out/kt5s
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt5 -o kt --output-format csv -- python3 $R/bench.py --workload signal12 --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $R/gpurun_out/kt5.log 2>&1 && python3 $R/tools/kstats.py $R/gpurun_out/kt5 13 40 > $R/gpurun_out/kt5_summary.txt 2>&1; tail -1 $R/gpurun_out/kt5.log
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt5s -o kt --output-format csv -- python3 $R/bench.py --workload signal12 --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-prof --serialize > $R/gpurun_out/kt5s.log 2>&1 && python3 $R/tools/kstats.py $R/gpurun_out/kt5s 13 40 > $R/gpurun_out/kt5s_summary.txt 2>&1; tail -1 $R/gpurun_out/kt5s.log
