cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for r in 1 2; do
  echo "== bench 150k run $r"; timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
echo "== bench 20k"; timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --voxels 20000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
timeout 600 python tools/op_census.py > gpurun_out/op_census.txt 2>gpurun_out/op_census.err; tail -3 gpurun_out/op_census.err
