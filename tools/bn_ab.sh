cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in A B; do
  rm -rf $R/gpurun_out/kt_$v
  if [ $v = A ]; then export ECGMM_LIB=$R/ecg-multimodal-model_amd/libecgmm_hip_A.so; else unset ECGMM_LIB; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$v -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --serialize > $R/gpurun_out/kt_$v.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $R/gpurun_out/kt_$v 13 60 > $R/gpurun_out/kt_${v}_summary.txt
  echo "== $v"; grep -i "bn_act\|total kernel" $R/gpurun_out/kt_${v}_summary.txt | cut -c1-120
done
