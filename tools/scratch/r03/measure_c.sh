cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
for gk in "64 8" "64 12" "48 6" "48 8" "48 10" "32 8" "32 12" "32 16"; do set -- $gk; USC3D_TRI_G=$1 python bench.py --mode ncut --scenes $2 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('ncut G=$1 K', r['scenes_in_flight'], round(r['value'],2), 'scenes/s', round(r['ms_per_step'],1), 'ms/scene', r['config']['masks'], 'eig ms', round(r['roofline']['avg_call_ms'],2))"; done
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
