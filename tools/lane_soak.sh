# the weight-gradient lane must be deterministic: final loss after N steps, three runs with the lane (all rows), one bounded at 50 000 rows, one without
cd $GRAFT_REPO_ROOT
N=${1:-150}
O=gpurun_out/lane_soak; mkdir -p $O
loss() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'loss', repr(r['config']['loss']), 'ms/step', round(r['ms_per_step'],3))"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-zorder --steps $N --warmup 0"
{
for rep in 1 2 3; do $B 2>$O/on.err | loss lane_all_$rep; done
USC3D_WGRAD_LANE_MAX_ROWS=50000 $B 2>$O/b.err | loss lane_50000
USC3D_WGRAD_LANE_MAX_ROWS=0 $B 2>$O/off.err | loss lane_off
$B --no-graphs 2>$O/e.err | loss eager_lane_all_1
$B --no-graphs 2>$O/e.err | loss eager_lane_all_2
} | tee $O/soak.txt
