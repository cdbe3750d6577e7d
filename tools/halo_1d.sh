cd $GRAFT_REPO_ROOT
for h in 0 2 0 2; do echo "== halo $h"; timeout -k 10 200 python3 tools/conv_bench.py --only fwd,dgrad --layers s1.k3,s2.k3,s3.k3 --halo $h 2>&1 | grep -E "fwd|dgrad"; done
echo "== B=512"
for h in 0 2; do echo "== halo $h"; timeout -k 10 200 python3 tools/conv_bench.py --batch 512 --only fwd,dgrad --layers s1.k3,s2.k3,s3.k3 --halo $h 2>&1 | grep -E "fwd|dgrad"; done
