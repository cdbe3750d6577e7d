# same-call A/B of one environment switch: tools/quick_ab.sh VAR "a b" [bench args]  (bench lines with VAR=a / VAR=b, twice each)
cd $GRAFT_REPO_ROOT
V=${1:-ECGMM_STEM_FUSE}
VALS=${2:-"0 1"}
A=${3:-}
for i in 1 2; do
  for x in $VALS; do
    echo "== $V=$x" >> gpurun_out/ab.txt
    env $V=$x python3 bench.py --no-cpu-baseline --no-prof $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms_hipevent'])" >> gpurun_out/ab.txt || exit 1
  done
done
