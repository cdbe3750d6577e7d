# per-kernel averages of a short image-only bench under rocprofv3 --kernel-trace (stem / pool / BN kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 $R/bench.py --workload image_only --batch 256 --steps 4 --warmup 1 --no-cpu-baseline --no-prof --serialize > $R/gpurun_out/kt.log 2>&1 && python3 $R/tools/kstats.py $R/gpurun_out/kt 5 40 > $R/gpurun_out/kt_summary.txt 2>&1; tail -1 $R/gpurun_out/kt.log
