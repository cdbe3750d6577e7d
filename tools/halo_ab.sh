cd $GRAFT_REPO_ROOT
rm -f gpurun_out/ab.txt gpurun_out/hb.txt
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or dgrad" > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for g in old new old new; do echo "== $g" >> gpurun_out/hb.txt; if [ $g = old ]; then LIBARG="--lib $GRAFT_REPO_ROOT/ab_old.so"; else LIBARG=""; fi; timeout -k 10 200 python3 tools/conv_bench.py --only fwd,dgrad --layers l1.3x3,l2.3x3,l3.3x3,l4.3x3,s3.k3 --halo 2 $LIBARG 2>&1 | grep -E "fwd|dgrad" >> gpurun_out/hb.txt || exit 1; done
cat gpurun_out/hb.txt
bash tools/lib_ab.sh ab_old.so && cat gpurun_out/ab.txt
