python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "wgrad or large_map or full" 2>&1 | tail -2
for t in 2048 1024 4096 6912; do echo "== tile target $t"; USC3D_WGRAD_TILE_TARGET=$t USC3D_PROF_SHAPES=1 python tools/conv_report.py 2>/dev/null | grep -E "wgrad_full_kernel<3, 3> \[n=4011228|wgrad_kernel<3, true>|wgrad_kernel<1, false>|total conv"; done
for i in 1 2; do python bench.py --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['ms_per_step'],3))"; done
