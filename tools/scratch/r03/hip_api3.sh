cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-trace --output-format json -d $GRAFT_REPO_ROOT/$O/api -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 2 > $GRAFT_REPO_ROOT/$O/api.log 2>&1)
ls -la $O/api/*/
python - <<'PY'
import json, glob, collections
f = glob.glob("gpurun_out/r03h/api/**/*results.json", recursive=True)[0]
d = json.loads(open(f, errors="replace").read())
r = d["rocprofiler-sdk-tool"][0]
print(r.keys())
strings = r.get("strings", {})
print(strings.keys() if isinstance(strings, dict) else type(strings))
bufs = r["buffer_records"]
print(bufs.keys())
hip = bufs["hip_api"]
print(len(hip), hip[0])
cb = r.get("callback_records", {})
print(cb.keys() if isinstance(cb, dict) else type(cb))
h = cb.get("hip_api_traces", []) if isinstance(cb, dict) else []
print(len(h), h[:1])
cands = [x for x in (h or hip) if "emcpy" in json.dumps(x)[:400]]
print(len(cands)); 
cnt = collections.Counter()
for x in cands:
    a = {y["name"]: y["value"] for y in x["args"]}
    cnt[(x["thread_id"], a.get("kind"), a.get("sizeBytes"), x["stream_id"]["handle"])] += 1
for k, v in cnt.most_common(30): print(v, k)
PY

rm -rf $O/api
