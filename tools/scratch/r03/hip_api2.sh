cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/api -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 3 > $GRAFT_REPO_ROOT/$O/api.log 2>&1)
python - <<'PY'
import csv, glob, collections
fa = glob.glob("gpurun_out/r03h/api/**/*hip_api_trace.csv", recursive=True)[0]
fk = glob.glob("gpurun_out/r03h/api/**/*kernel_trace.csv", recursive=True)[0]
api = list(csv.DictReader(open(fa)))
ker = list(csv.DictReader(open(fk)))
print(ker[0].keys())
by_corr = {r["Correlation_Id"]: r for r in api}
api.sort(key=lambda r: int(r["Start_Timestamp"]))
idx_of = {r["Correlation_Id"]: i for i, r in enumerate(api)}
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(ker) if "adamw_kernel" in r["Kernel_Name"]]
step = ker[ad[-2] + 1: ad[-1] + 1]
c = collections.Counter()
ctx = collections.Counter()
prevk = None
for r in step:
    if "copyBuffer" in r["Kernel_Name"]:
        a = by_corr.get(r["Correlation_Id"])
        name = a["Function"] if a else "?"
        tid = a["Thread_Id"] if a else "?"
        c[(name, tid)] += 1
        if prevk and "group_reduce" in prevk and a:
            i = idx_of[r["Correlation_Id"]]
            same = [x["Function"] for x in api[max(0, i - 40): i + 12] if x["Thread_Id"] == tid and x["Function"] not in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration")]
            ctx[tuple(same[-8:])] += 1
    prevk = r["Kernel_Name"]
for k, v in c.most_common(): print(v, k)
# thread / stream of the kernels around a copy that follows group_reduce
shown = 0
for j, r in enumerate(step):
    if "copyBuffer" in r["Kernel_Name"] and j and "group_reduce" in step[j-1]["Kernel_Name"] and shown < 3:
        shown += 1
        for x in step[j-2:j+3]:
            a = by_corr.get(x["Correlation_Id"])
            print("   ", x["Kernel_Name"][:40], "queue", x["Queue_Id"], "stream", x["Stream_Id"], "thread", x["Thread_Id"], "api", a["Function"] if a else "?", a["Thread_Id"] if a else "?")
tids = collections.Counter((r["Thread_Id"], r["Stream_Id"], r["Queue_Id"]) for r in step)
print(tids)
for k, v in ctx.most_common(6): print(v, k)
PY
rm -rf $O/api
