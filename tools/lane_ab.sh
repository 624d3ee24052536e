cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-order 2>gpurun_out/lane_err.txt | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['ms_per_step'], r['config']['loss'])" || tail -n 3 gpurun_out/lane_err.txt; }
L="USC3D_WGRAD_LANE_MAX_ROWS=100000000 USC3D_WGRAD_LANE_MIN_ROWS=24576"
run base ""
run lane_big "$L"
run lane_big_cu55 "$L USC3D_LANE_CU_PATTERN=55555555"
run lane_big_cu0f "$L USC3D_LANE_CU_PATTERN=0f0f0f0f"
run lane_big_cu3333 "$L USC3D_LANE_CU_PATTERN=33333333"
run base ""
run lane_all148k "USC3D_WGRAD_LANE_MAX_ROWS=100000000 USC3D_WGRAD_LANE_MIN_ROWS=100000"
run lane_big "$L"
