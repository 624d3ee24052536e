# A/B of the decoder key-preparation side stream (USC3D_KV_SIDE_STREAM) on the default bench line
cd $GRAFT_REPO_ROOT
T=${1:-ab_kv}
O=gpurun_out/$T; mkdir -p $O
line() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/step', round(r['ms_per_step'],3), 'loss', r.get('final_loss', r.get('loss')))"; }
B="timeout 200 python bench.py --no-cpu-baseline --no-zorder --steps 20 --warmup 5"
{
for rep in 1 2 3; do
$B 2>$O/on.err | line side_on_$rep
USC3D_KV_SIDE_STREAM=0 $B 2>$O/off.err | line side_off_$rep
done
} | tee $O/ab.txt
tail -5 $O/on.err
