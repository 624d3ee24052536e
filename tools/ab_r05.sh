# round-5 A/B of the kernel switches on one box: bash tools/ab_r05.sh <tag>
cd $GRAFT_REPO_ROOT
T=${1:-ab}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/step', round(r['ms_per_step'],3), 'scenes/s', round(r['value'],2), 'frac', r['roofline']['frac'] if r.get('roofline') else None)"; }
B="python bench.py --no-cpu-baseline --no-reference-order --steps 20 --warmup 5"
run() { name=$1; shift; env "$@" $B 2>$O/$name.err | line $name; }
{
run base     USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=0
run ch64     USC3D_SORTED_CH=64 USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=0
run wgbig    USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=1 USC3D_BN_TILE_ROWS=0
run tile     USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=12288
run tile4k   USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=4096
run all      USC3D_SORTED_CH=64 USC3D_WGRAD_BIG=1 USC3D_BN_TILE_ROWS=12288
run base2    USC3D_SORTED_CH=32 USC3D_WGRAD_BIG=0 USC3D_BN_TILE_ROWS=0
run all2     USC3D_SORTED_CH=64 USC3D_WGRAD_BIG=1 USC3D_BN_TILE_ROWS=12288
} | tee $O/ab.txt
