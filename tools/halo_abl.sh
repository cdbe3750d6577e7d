cd $GRAFT_REPO_ROOT
rm -f gpurun_out/habl.txt
for n in ${ABLS:-0 1 2 3 4 5 6 7}; do
  if [ $n = 0 ]; then LIBARG=""; else LIBARG="--lib $GRAFT_REPO_ROOT/ecg-multimodal-model_amd/libecgmm_hip_habl$n.so"; fi
  echo "== abl $n (0 = shipped; 1 no MFMA, 2 no LDS fragment reads, 3 no in-loop fills, 4 no epilogue, 5 no per-slice barrier, 6 no output stores, 7 no statistics)" >> gpurun_out/habl.txt
  timeout -k 10 120 python3 tools/conv_bench.py --only fwd,dgrad --layers l1.3x3,l2.3x3,l3.3x3,l4.3x3 --halo 2 $LIBARG 2>&1 | grep -E "fwd|dgrad" >> gpurun_out/habl.txt || exit 1
done
cat gpurun_out/habl.txt
