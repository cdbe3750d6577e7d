# robustness: odd per-GPU batches through the whole step (ragged tiles, image slots that do not divide the batch)
cd $GRAFT_REPO_ROOT
for b in 2 7 100 255; do
  python3 bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch', d['config']['per_gpu_batch'], 'ms', d['ms_per_step'], 'loss', d['config']['final_loss'])" || exit 1
done
python3 bench.py --workload signal12 --batch 33 --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sig12 batch 33 ms', d['ms_per_step'], 'loss', d['config']['final_loss'])" || exit 1
python3 bench.py --image-hw 250x2500 --batch 3 --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fullres batch 3 ms', d['ms_per_step'], 'loss', d['config']['final_loss'])" || exit 1
echo odd-done
