cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.txt gpurun_out/wg.txt
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "conv" > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for g in old new old new; do echo "== $g" >> gpurun_out/wg.txt; if [ $g = old ]; then LIBARG="--lib $GRAFT_REPO_ROOT/ab_old.so"; else LIBARG=""; fi; timeout -k 10 200 python3 tools/conv_bench.py --only wgrad --layers l1.3x3,l2.3x3s2,l2.3x3,l3.3x3,l4.3x3,l2.ds,s1.k3,s2.k3,s3.k3 $LIBARG 2>&1 | grep wgrad >> gpurun_out/wg.txt || exit 1; done
cat gpurun_out/wg.txt
bash tools/lib_ab.sh ab_old.so && cat gpurun_out/ab.txt
