# HBM traffic of the implicit-GEMM kernel class: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only),
# on the one-stream bench (--serialize) so a kernel's counters are not shared with concurrent launches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 2 --warmup 1 --no-prof --no-cpu-baseline --serialize"
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- $B > $R/gpurun_out/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- $B > $R/gpurun_out/pmc_write.log 2>&1 &&
python3 $R/tools/roofline_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/igemm_traffic.json
