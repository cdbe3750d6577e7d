# tests + bench lines for the headline and the 12-lead configuration (no prof, no cpu baseline)
cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; mkdir -p $O; rm -f $O/*.json
python3 -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_parallel_gpu.py -x -q > $O/test.log 2>&1 || { tail -30 $O/test.log; exit 1; }
tail -2 $O/test.log
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-prof --steps 40 --warmup 10 > $O/plain_$i.json 2>/dev/null || exit 1
python3 bench.py --workload signal12 --batch 512 --no-cpu-baseline --no-prof --steps 40 --warmup 10 > $O/sig12_$i.json 2>/dev/null || exit 1
ECGMM_SIDE_WGRAD=0 python3 bench.py --workload signal12 --batch 512 --no-cpu-baseline --no-prof --steps 40 --warmup 10 > $O/sig12_noside_$i.json 2>/dev/null || exit 1
done
for f in $O/*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f', d['ms_per_step'])"; done
