# same-call A/B of two builds of the library: tools/lib_ab.sh <old.so> [bench args]   (the in-tree build is "new")
cd $GRAFT_REPO_ROOT
OLD=$1; A=${2:-}
for i in 1 2; do
  for x in old new; do
    echo "== $x" >> gpurun_out/ab.txt
    if [ $x = old ]; then export ECGMM_LIB=$GRAFT_REPO_ROOT/$OLD; else unset ECGMM_LIB; fi
    python3 bench.py --no-cpu-baseline --no-prof $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms_hipevent'])" >> gpurun_out/ab.txt || exit 1
  done
done
