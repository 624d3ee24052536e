# round-6 measurement set (run on the GPU box from the repo root): bash tools/measure_r06.sh <tag> [quick]
# bench.py defaults since round 5: value = the reference's row order, 8 rotated scenes, second loop in z-order cell rows.
set -x
cd $GRAFT_REPO_ROOT
T=${1:-r06}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('$1', 'scenes/s', round(r['value'],2), 'ms/step', round(r['ms_per_step'],2), 'z-order', round(r.get('value_zorder') or 0,2), 'B', c.get('scenes_per_gpu'), 'voxels', c['voxels_per_scene'], {k:round(v,2) for k,v in c.items() if k.startswith('step_ms')})"; }
# 1. default bench line (with cpu baseline incl. the unscaled full-size pass)
python bench.py > $O/bench.json 2> $O/bench.err
# 2. step time vs scene size, scenes per GPU, rotation spread
for v in 20000 80000 150000 300000; do python bench.py --no-cpu-baseline --voxels $v --steps 20 --warmup 5 2>/dev/null | line voxels=$v; done > $O/step_vs_scene_size.txt
for B in 1 2 4 8; do python bench.py --no-cpu-baseline --no-zorder --steps 12 --warmup 4 --scenes-per-gpu $B 2>/dev/null | tee $O/bench_B$B.json | line B=$B; done > $O/scenes_per_gpu.txt
python bench.py --no-cpu-baseline --no-zorder --steps 32 --warmup 8 --rotate 8 --rotate-spread 0.2 2>/dev/null | tee $O/bench_rotate8_spread20.json | line rotate8_spread0.2 >> $O/scenes_per_gpu.txt
python bench.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 --rotate 0 2>/dev/null | line rotate0_one_scene >> $O/scenes_per_gpu.txt
# 3. sections of the step on the compute stream, key-preparation stream on / off; host vs device
for s in 1 0; do echo "USC3D_KV_SIDE_STREAM=$s"; USC3D_KV_SIDE_STREAM=$s python tools/decoder_spans.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | tail -n 8; done > $O/step_sections.txt
python tools/stream_queue_probe.py 2>/dev/null | tail -n 9 > $O/stream_queue_probe.txt
for t in 1 0; do echo -n "prefetch_thread=$t: "; USC3D_PREFETCH_THREAD=$t python tools/host_threads.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | tail -n 1; done > $O/host_threads.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | grep '^{' | tail -1 > $O/bench_steps20_warmup5.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-zorder --force-dist 2>/dev/null | grep '^{' | tail -1 > $O/bench_world1_rccl.json
python tools/soak.py --steps 400 2>/dev/null | tail -1 > $O/soak.json
for v in 150000 20000; do for t in 1 0; do echo -n "voxels=$v prefetch_thread=$t: "; USC3D_PREFETCH_THREAD=$t python tools/host_vs_device.py --no-cpu-baseline --voxels $v 2>/dev/null | tail -n 1; done; done > $O/host_vs_device.txt
# 4. two ranks (gloo, sharing this box's one device: the code path, not a scaling number) drawing scenes through the sampler
python bench.py --gpus 2 --dist-backend gloo --rotate 4 --steps 8 --warmup 2 --no-cpu-baseline --no-zorder > $O/bench_2rank_gloo_rotate.json 2> $O/bench_2rank.err
python bench.py --gpus 2 --dist-backend gloo --dry-collectives --no-cpu-baseline > $O/bench_2rank_dry_collectives.json 2>> $O/bench_2rank.err
[ "$2" = quick ] && exit 0
# 5. kernel stats of the bench command (+ the per-queue overlap report of the same trace)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-zorder > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
python tools/prof_summary.py $O/prof 90 > $O/kernel_stats_summary.txt
python tools/stream_overlap.py $O/prof 14 > $O/stream_overlap.txt
python tools/stream_timeline.py $O/prof > $O/timeline_last_step.txt
python tools/timeline_sections.py $O/timeline_last_step.txt > $O/timeline_sections.txt
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
# 6. PMC traffic passes (separate runs, counters only with kernel-trace), in the row order of `value`
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-zorder > /dev/null 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-zorder > /dev/null 2>&1)
for k in gather_gemm wgrad bn_apply bn_tile_apply adamw; do python tools/pmc_summary.py $O/pmc_fetch $k; python tools/pmc_summary.py $O/pmc_write $k; done > $O/pmc_traffic_raw.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/prof
# 7. HBM report, ncut bench, conv per shape, tile-form BN
python tools/hbm_report.py > $O/hbm_bound_kernels.txt 2>/dev/null
python bench.py --mode ncut > $O/bench_ncut.json 2>/dev/null
USC3D_PROF_SHAPES=1 python tools/conv_report.py > $O/conv_per_shape.txt 2>/dev/null
python tools/bn_tile_bench.py > $O/bn_tile_bench.txt 2>/dev/null
ls -la $O
