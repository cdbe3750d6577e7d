cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -k "stem or multimodal or resnet18 or image" > gpurun_out/stem_test.log 2>&1 || { tail -30 gpurun_out/stem_test.log; exit 1; }
tail -1 gpurun_out/stem_test.log
python3 tools/stem_bench.py || exit 1
bash tools/stem_prof.sh
grep -i "stem" gpurun_out/kt_summary.txt | cut -c1-120
