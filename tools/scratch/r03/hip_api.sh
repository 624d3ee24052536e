cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/api -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 3 > $GRAFT_REPO_ROOT/$O/api.log 2>&1)
ls $O/api/*/ | head
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r03h/api/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Function"] for r in rows]
c = collections.Counter(names)
for k, v in c.most_common(40): print(v, k)
# context of memcpy-like calls in the last quarter of the trace
n = len(rows)
ctx = collections.Counter()
for i in range(n * 3 // 4, n):
    if "emcpy" in names[i]:
        ctx[(names[i], names[i-2], names[i-1], names[i+1] if i+1<n else "-")] += 1
for k, v in ctx.most_common(40): print(v, k)
PY
rm -rf $O/api
