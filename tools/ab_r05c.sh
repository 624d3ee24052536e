cd $GRAFT_REPO_ROOT
T=${1:-ab}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/step', round(r['ms_per_step'],3))"; }
B="python bench.py --no-cpu-baseline --no-reference-order --steps 20 --warmup 5"
{
for rep in 1 2 3; do
$B 2>$O/n.err | line normal_$rep
$B --prefetch-ahead 40 2>$O/a.err | line ahead_$rep
USC3D_PREFETCH_THREAD=0 $B 2>$O/t.err | line nothread_$rep
done
} | tee $O/ab.txt
tail -3 $O/a.err
