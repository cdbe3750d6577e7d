# same-call comparison of several environment settings: tools/ab_env.sh "A=1 B=0" "A=0" ...   (each run twice, interleaved)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for e in "$@"; do
    echo "== $e" >> gpurun_out/ab.txt
    env $e python3 bench.py --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms_hipevent']['median'])" >> gpurun_out/ab.txt || exit 1
  done
done
cat gpurun_out/ab.txt
