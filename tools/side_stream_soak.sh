# the key-preparation stream must not change a bit: final loss after N steps, stream on (3 runs) vs off (1 run)
cd $GRAFT_REPO_ROOT
N=${1:-150}
O=gpurun_out/side_soak; mkdir -p $O
loss() { python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'loss', repr(r['config']['loss']), 'ms/step', round(r['ms_per_step'],3))"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-zorder --steps $N --warmup 0"
{
USC3D_KV_SIDE_STREAM=0 $B 2>$O/off.err | loss off
for rep in 1 2 3; do $B 2>$O/on.err | loss on_$rep; done
USC3D_KV_SIDE_STREAM=0 $B --no-graphs 2>$O/off.err | loss eager_off
$B --no-graphs 2>$O/on.err | loss eager_on
} | tee $O/soak.txt
python - <<'PY'
import re
v = dict(re.match(r"(\S+) loss (\S+)", l).groups() for l in open("gpurun_out/side_soak/soak.txt") if " loss " in l)
ok = all(v[k] == v["off"] for k in ("on_1", "on_2", "on_3")) and v["eager_on"] == v["eager_off"]
print("BIT-IDENTICAL" if ok else "MISMATCH", v)
PY
