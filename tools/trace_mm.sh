cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktmm
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktmm -o kt --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof $EXTRA > $R/gpurun_out/ktmm.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/ktmm 8 50 > $R/gpurun_out/ktmm_summary.txt
python3 $R/tools/gaps.py $R/gpurun_out/ktmm > $R/gpurun_out/ktmm_gaps.txt
tail -1 $R/gpurun_out/ktmm.log | cut -c1-200
