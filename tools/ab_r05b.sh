cd $GRAFT_REPO_ROOT
T=${1:-ab}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/step', round(r['ms_per_step'],3))"; }
B="python bench.py --no-cpu-baseline --no-reference-order --steps 30 --warmup 6"
run() { name=$1; shift; env "$@" $B 2>$O/$name.err | line $name; }
{
for rep in 1 2 3; do
run base_$rep     USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=0
run t600_$rep     USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=600
run t2500_$rep    USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=2500
run t12288_$rep   USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=12288
run wgbig_$rep    USC3D_WGRAD_BIG=1 USC3D_BN_TILE_ROWS=0
done
for v in 0 4096; do echo -n "tile=$v host_vs_device: "; USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=$v python tools/host_vs_device.py --no-cpu-baseline 2>/dev/null | tail -n 1; done
} | tee $O/ab.txt
