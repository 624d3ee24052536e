cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout 300 python bench.py --gpus 2 --dist-backend gloo --rotate 4 --steps 4 --warmup 1 --no-cpu-baseline --no-zorder 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(r['ms_per_step'],1), r['config'].get('streams'))"; }
run hwq3 GPU_MAX_HW_QUEUES=3
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8 GPU_MAX_HW_QUEUES=8
run no_prefetch_thread USC3D_PREFETCH_THREAD=0
