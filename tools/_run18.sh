cd $GRAFT_REPO_ROOT
for rep in 1 2; do for s in 0 1; do
USC3D_STEADY=$s python tools/soak.py --steps 400 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steady$s', {k:round(r[k],2) for k in ('ms_per_step_p50','ms_per_step_p99','ms_per_step_max')}, r['memory'][-1])"
done; done
python -m pytest tests/test_gpu_memory.py -q -p no:cacheprovider 2>&1 | tail -5
