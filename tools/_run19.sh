cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in 0 2 1; do for s in 1; do
USC3D_STEPS_IN_FLIGHT=$f USC3D_STEADY=$s python tools/soak.py --steps 400 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight$f steady$s', {k:round(r[k],2) for k in ('ms_per_step_p50','ms_per_step_p99','ms_per_step_max')}, r['memory'][-1], r['peak_allocated_MB'])"
done; done; done
USC3D_STEPS_IN_FLIGHT=2 USC3D_STEADY=0 python tools/soak.py --steps 400 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight2 steady0', {k:round(r[k],2) for k in ('ms_per_step_p50','ms_per_step_p99','ms_per_step_max')}, r['memory'][-1], r['peak_allocated_MB'])"
