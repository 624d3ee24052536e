import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(overlap, hold, port, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", USC3D_OVERLAP_ALLREDUCE=overlap,
               USC3D_REDUCER_HOLD_REPLAYED=hold)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "gpurun_scratch", "mr_dump2.py"), out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if not os.path.exists(out):
        print(r.stderr[-1500:]); raise SystemExit(1)
    return json.load(open(out))
a = run("1", "0", 29700, "/tmp/mr_a.json")
b = run("1", "1", 29701, "/tmp/mr_b.json")
a2 = run("1", "0", 29702, "/tmp/mr_a2.json")
print("early-all loss", a["loss"], a2["loss"], " hold-replayed loss", b["loss"], "steps", len(a["steps"]), len(b["steps"]))
for k, (ga, gb) in enumerate(zip(a["steps"], b["steps"])):
    diff = [(n, ga[n], gb[n]) for n in gb if ga.get(n) != gb[n]]
    print("step", k, "params with different gradient sums:", len(diff), "of", len(gb))
    for n, x, y in diff[:40]:
        print("    ", n, x, y)
    if diff:
        break
