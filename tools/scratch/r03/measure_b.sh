set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_step_parity.py -q 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -20
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('$1', 'scenes/s', round(r['value'],2), 'ms/step', round(r['ms_per_step'],2), 'B', c.get('scenes_per_gpu'), 'voxels', c['voxels_per_scene'], {k:round(v,2) for k,v in c.items() if k.startswith('step_ms')})"; }
for B in 1 2 4 8; do python bench.py --no-cpu-baseline --steps 12 --warmup 4 --scenes-per-gpu $B 2>$O/b$B.err | tee $O/bench_B$B.json | line B=$B; done
python bench.py --no-cpu-baseline --steps 32 --warmup 8 --rotate 8 2>$O/rot.err | tee $O/bench_rotate8.json | line rotate8
USC3D_WGRAD_ONE_SLICE_FROM=128 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | line wgradS1
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tee $O/bench.json | line base
for k in 1 2 3 4 6; do python bench.py --mode ncut --scenes $k --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('ncut K', r['scenes_in_flight'], round(r['value'],2), 'scenes/s', round(r['ms_per_step'],1), 'ms/scene', r['config']['masks'])"; done
