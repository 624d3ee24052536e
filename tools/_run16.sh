cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python bench.py --scenes-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | python -c "import sys,json,torch; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('$name', round(r['value'],2), 'scenes/s', round(r['ms_per_step'],1), 'ms p50', round(c['step_ms_p50'],1), 'p90', round(c['step_ms_p90'],1), 'max', round(c['step_ms_max'],1), 'at', c.get('step_ms_max_at'))"; }
for rep in 1 2; do
run default X=1
run roundup4 PYTORCH_HIP_ALLOC_CONF=roundup_power2_divisions:4 PYTORCH_CUDA_ALLOC_CONF=roundup_power2_divisions:4
run nosplit PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:64 PYTORCH_CUDA_ALLOC_CONF=max_split_size_mb:64
run laneoff USC3D_WGRAD_LANE_MAX_ROWS=0
run gc06 PYTORCH_HIP_ALLOC_CONF=garbage_collection_threshold:0.99 PYTORCH_CUDA_ALLOC_CONF=garbage_collection_threshold:0.99
done
