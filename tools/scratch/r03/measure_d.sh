cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('$1', 'scenes/s', round(r['value'],2), 'ms/step', round(r['ms_per_step'],2))"; }
USC3D_FORK_WGRAD=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line fork_wgrad
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line base
USC3D_FORK_WGRAD=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --voxels 40000 2>/dev/null | line fork_wgrad_40k
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --voxels 40000 2>/dev/null | line base_40k
for k in 16 24; do python bench.py --mode ncut --scenes $k --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('ncut K', r['scenes_in_flight'], round(r['value'],2), 'scenes/s', round(r['ms_per_step'],1), 'ms/scene', r['config']['masks'])"; done
