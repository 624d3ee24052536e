cd $GRAFT_REPO_ROOT
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=r['config']; print('$1', round(r['value'],2), 'scenes/s', round(r['ms_per_step'],2), 'ms p50', round(c['step_ms_p50'],2), 'p90', round(c['step_ms_p90'],2), 'max', round(c['step_ms_max'],2), 'at', c.get('step_ms_max_at'), (c.get('steady_state') or {}).get('reserved_after'))"; }
for rep in 1 2 3; do
for s in 1 0; do
USC3D_STEADY=$s python bench.py --no-cpu-baseline --no-zorder 2>/dev/null | line "b1_default_steady$s"
USC3D_STEADY=$s python bench.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5 2>/dev/null | line "b1_20_5_steady$s"
done; done
for rep in 1 2; do for s in 1 0; do
USC3D_STEADY=$s python bench.py --scenes-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline --no-zorder 2>/dev/null | line "b8_steady$s"
done; done
python tools/soak.py --steps 400 > gpurun_out/r06_soak.json 2> gpurun_out/r06_soak.err; tail -2 gpurun_out/r06_soak.err; python -c "
import json; r=json.loads(open('gpurun_out/r06_soak.json').read().strip().splitlines()[-1]); print({k:r[k] for k in r if k!='memory'}); print(r['memory'][0], r['memory'][-1])"
