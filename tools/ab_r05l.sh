cd $GRAFT_REPO_ROOT
V=${1:-80000}
run() { name=$1; shift; env "$@" timeout 300 python bench.py --voxels $V --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(r['ms_per_step'],2), r['config']['voxels_per_scene'])"; }
run default X=1
run no_lane USC3D_WGRAD_LANE_MAX_ROWS=0
run no_keys USC3D_KV_SIDE_STREAM=0
run no_bn_tile USC3D_BN_TILE_ROWS=0
run no_lane_no_keys USC3D_WGRAD_LANE_MAX_ROWS=0 USC3D_KV_SIDE_STREAM=0
run_d1() { timeout 300 python bench.py --voxels $V --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --prefetch-depth 1 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(\"depth1\", round(r[\"ms_per_step\"],2))"; }; run_d1
timeout 300 python bench.py --voxels $V --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --rotate 0 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(\"rotate0\", round(r[\"ms_per_step\"],2))"
