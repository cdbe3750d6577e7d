cd $GRAFT_REPO_ROOT
run() { env "$@" python3 bench.py --no-cpu-baseline --no-prof $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms_hipevent']['median'])"; }
for i in 1 2 3; do
A=""
echo "== plain"; run ECGMM_NOP=1
echo "== force ddp"; run ECGMM_FORCE_DDP=1
done
