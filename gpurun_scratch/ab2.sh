cd $GRAFT_REPO_ROOT
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do
for v in 0 1; do
  echo "== bench 150k stats_over_slices=$v"; USC3D_STATS_OVER_SLICES=$v timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
done
