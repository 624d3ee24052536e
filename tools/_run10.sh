cd $GRAFT_REPO_ROOT
bash tools/ab.sh -r 3 plain: bn10k:USC3D_BN_TILE_ROWS=10000
bash tools/ab.sh -r 3 -x "--force-dist" fd: fd_r05join:USC3D_LANE_ORDERED_COLLECTIVES=0
for conf in "" "expandable_segments:True"; do echo "== alloc conf '$conf'"; PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_CUDA_ALLOC_CONF=$conf python bench.py --scenes-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print(round(r['value'],2), round(r['ms_per_step'],1), {k:round(c[k],1) for k in c if k.startswith('step_ms')})"; done
