cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --no-zorder --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); c=r['config']; print('$name', round(r['ms_per_step'],2), c['streams'])"; }
run default X=1



run no_probe USC3D_STREAM_PROBE=0
run no_keys_stream USC3D_KV_SIDE_STREAM=0
run no_lane USC3D_WGRAD_LANE_MAX_ROWS=0
run default_again X=1
run lane_high_priority USC3D_LANE_PRIORITY=-1
