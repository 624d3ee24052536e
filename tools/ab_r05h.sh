# the lane's weight-gradient grids bounded (USC3D_WGRAD_GRID_LIMIT workgroups along x) so that the chain's small kernels find free CUs
cd $GRAFT_REPO_ROOT
T=${1:-ab_grid}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/step', round(r['ms_per_step'],3), 'loss', r['config']['loss'])"; }
B="timeout 200 python bench.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5"
{
for rep in 1 2; do
for g in ${LIMITS:-0 96 160 224}; do
USC3D_WGRAD_GRID_LIMIT=$g $B 2>$O/g$g.err | line grid_limit${g}_$rep
done
done
} | tee $O/ab.txt
