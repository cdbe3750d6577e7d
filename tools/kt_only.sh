# one overlapped kernel trace of the default bench (no PMC): gpurun_out/kt1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt1 -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof $1 > $R/gpurun_out/kt1.log 2>&1 || exit 1
python3 $R/tools/timeline.py $R/gpurun_out/kt1 --list > $R/gpurun_out/kt1_timeline.txt
