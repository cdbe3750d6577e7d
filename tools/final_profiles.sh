# Evidence for profiles/: the bench line, rocprofv3 --kernel-trace --stats of the same command (headline = concurrent
# streams; plus the one-stream --serialize variant, where a kernel's duration is its own), and the PMC traffic passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
# (--no-prof: without bench.py's own event bracketing and its extra one-stream pass, so every launch in the trace is a launch of
# the overlapped step)
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $O/kt.log 2>&1 || exit 1
python3 $R/tools/kstats.py $O/kt 13 60 > $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/kts -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --serialize > $O/kts.log 2>&1 || exit 1
python3 $R/tools/kstats.py $O/kts 13 60 > $O/kernel_stats_serialized.txt
echo traces-done
# HBM traffic of the conv kernel class: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (kernel-trace only)
B="python3 $R/bench.py --steps 2 --warmup 1 --no-prof --no-cpu-baseline --serialize"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- $B > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- $B > $O/pmc_write.log 2>&1 || exit 1
python3 $R/tools/roofline_traffic.py $O/pmc_fetch $O/pmc_write $O/igemm_traffic.json || exit 1
# (the file is pulled back with gpurun_out/final/ and committed from there: tools/pull_profiles.sh rNN vK copies it to
# profiles/rNN_igemm_traffic.json, which bench.py reads as roofline.traffic -- the lease-side profiles/ does not come back)
python3 $R/tools/pmc_kernels.py $O/pmc_fetch > $O/pmc_by_kernel.txt 2>&1; python3 $R/tools/pmc_kernels.py $O/pmc_write >> $O/pmc_by_kernel.txt 2>&1
echo pmc-done
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
python3 $R/bench.py --workload image_only --batch 128 --no-cpu-baseline > $O/bench_cfg2.json 2>/dev/null || exit 1
python3 $R/bench.py --workload signal12 --batch 512 --no-cpu-baseline > $O/bench_cfg5.json 2>/dev/null || exit 1
python3 $R/bench.py --image-hw 250x2500 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f2.json 2>/dev/null || exit 1
echo final-done-main
# SURVEY 8d extras: the train.py-faithful frozen-encoder step, and exact-fp32 runs of cfg 3 / cfg 5
python3 $R/bench.py --freeze-encoders --no-cpu-baseline > $O/bench_cfg3_frozen.json 2>/dev/null || exit 1
python3 $R/bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_fp32.json 2>/dev/null || exit 1
python3 $R/bench.py --workload signal12 --batch 512 --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg5_fp32.json 2>/dev/null || exit 1
echo extras-done
