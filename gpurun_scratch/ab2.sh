cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for at in end sync; do
  echo "== bench 150k prefetch_at=$at"; USC3D_PREFETCH_AT=$at timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
done
for at in end sync; do
  echo "== bench 20k prefetch_at=$at"; USC3D_PREFETCH_AT=$at timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --voxels 20000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
timeout 600 python -m pytest tests/test_gpu_multirank.py -m gpu -x -q 2>&1 | tail -3
