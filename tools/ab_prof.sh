# A/B two builds of the library under rocprofv3 --kernel-trace on the serialized image-only (or $WORKLOAD) step.
# usage: bash tools/ab_prof.sh <libA.so> <libB.so>   -> gpurun_out/ab_A.txt, gpurun_out/ab_B.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${WORKLOAD:-image_only}
for tag in A B; do
  if [ $tag = A ]; then export ECGMM_LIB=$R/$1; else export ECGMM_LIB=$R/$2; fi
  rm -rf $R/gpurun_out/kt$tag
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt$tag -o kt --output-format csv -- python3 $R/bench.py --workload $W --batch 256 --steps 4 --warmup 1 --no-cpu-baseline --no-prof --serialize > $R/gpurun_out/kt$tag.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $R/gpurun_out/kt$tag 5 45 > $R/gpurun_out/ab_$tag.txt 2>&1
done
echo ab-done
