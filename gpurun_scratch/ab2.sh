cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --mode backbone --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700
timeout 600 python bench.py --mode backbone --steps 3 --warmup 1 --cpu-sample-voxels 4000 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cpu_baseline'])" | cut -c1-600
