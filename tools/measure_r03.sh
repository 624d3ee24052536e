# round-3 measurement set (run on the GPU box from the repo root): bash tools/measure_r03.sh <tag>
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r03}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('$1', 'scenes/s', round(r['value'],2), 'ms/step', round(r['ms_per_step'],2), 'B', c.get('scenes_per_gpu'), 'voxels', c['voxels_per_scene'], {k:round(v,2) for k,v in c.items() if k.startswith('step_ms')})"; }
# 1. default bench line (with cpu baseline)
python bench.py > $O/bench.json 2> $O/bench.err
# 2. step time vs scene size, scenes per GPU, rotation
for v in 20000 80000 150000 300000; do python bench.py --no-cpu-baseline --voxels $v --steps 20 --warmup 5 2>/dev/null | line voxels=$v; done > $O/step_vs_scene_size.txt
for B in 1 2 4 8; do python bench.py --no-cpu-baseline --steps 12 --warmup 4 --scenes-per-gpu $B 2>/dev/null | tee $O/bench_B$B.json | line B=$B; done > $O/scenes_per_gpu.txt
python bench.py --no-cpu-baseline --steps 32 --warmup 8 --rotate 8 2>/dev/null | tee $O/bench_rotate8.json | line rotate8 >> $O/scenes_per_gpu.txt
# 3. kernel stats of the bench command
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/prof_summary.py $O/prof 90 > $O/kernel_stats_summary.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
# 4. PMC traffic passes (separate runs, counters only with kernel-trace)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
for k in gather_gemm wgrad; do python tools/pmc_summary.py $O/pmc_fetch $k; python tools/pmc_summary.py $O/pmc_write $k; done > $O/pmc_traffic_raw.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/prof
# 5. HBM report, ncut bench, conv per shape
python tools/hbm_report.py > $O/hbm_bound_kernels.txt 2>/dev/null
python bench.py --mode ncut > $O/bench_ncut.json 2>/dev/null
for k in 1 2 4 8 16 24; do python bench.py --mode ncut --scenes $k --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('ncut scenes in flight', r['scenes_in_flight'], round(r['value'],2), 'scenes/s', round(r['ms_per_step'],1), 'ms/scene', r['config']['masks'], 'masks')"; done > $O/ncut_scenes_in_flight.txt
USC3D_PROF_SHAPES=1 python tools/conv_report.py > $O/conv_per_shape.txt 2>/dev/null
ls -la $O
