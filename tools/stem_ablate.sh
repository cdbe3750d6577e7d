# run on the GPU box: times the stem kernels of the default build and of the ablation builds made by `make stem_abl`
cd $GRAFT_REPO_ROOT
python3 tools/stem_bench.py || exit 1
for n in 1 2 3 4; do
  f=$GRAFT_REPO_ROOT/ecg-multimodal-model_amd/libecgmm_hip_abl$n.so
  [ -f $f ] && { python3 tools/stem_bench.py --lib $f || exit 1; }
done
echo ablate-done
