cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
USC3D_SORTED_TUNE=1 timeout 1200 python tools/sorted_plan_sweep.py > gpurun_out/sorted_plan_sweep.txt 2>&1
cut -c1-330 gpurun_out/sorted_plan_sweep.txt | tail -12
for ks in 0 3000; do echo "== bench ks_rows=$ks"; USC3D_SORTED_KS_ROWS=$ks timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; done
