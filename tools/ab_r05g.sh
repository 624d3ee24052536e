# the weight-gradient lane on a stream MEASURED to run beside the compute stream (streams.py), row bounds 0 / 3000 / 10000 / 50000
cd $GRAFT_REPO_ROOT
T=${1:-ab_lane}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/step', round(r['ms_per_step'],3), 'loss', r['config']['loss'])"; }
B="timeout 200 python bench.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5"
{
for rep in 1 2; do
for rows in 50000 200000; do
USC3D_WGRAD_LANE_MAX_ROWS=$rows $B 2>$O/l$rows.err | line lane_rows${rows}_$rep
done
done
} | tee $O/ab.txt
tail -2 $O/l10000.err
