cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for v in 20000 150000; do
  echo "== bench $v"; timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --voxels $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
