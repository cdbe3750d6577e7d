cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktd
export ECGMM_FORCE_DDP=1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktd -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $R/gpurun_out/ktd.log 2>&1 || exit 1
python3 $R/tools/timeline.py $R/gpurun_out/ktd --list > $R/gpurun_out/ktd_timeline.txt
