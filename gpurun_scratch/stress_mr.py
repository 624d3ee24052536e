import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(overlap, port, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", USC3D_OVERLAP_ALLREDUCE=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "gpurun_scratch", "mr_dump2.py"), out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if not os.path.exists(out):
        print(r.stderr[-1500:]); raise SystemExit(1)
    return json.load(open(out))
def bg():
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "1", "--voxels", "80000",
                             "--no-cpu-baseline"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
ref = run("0", 29700, "/tmp/mr_ref.json")
print("single loss", ref["loss"], flush=True)
for i in range(10):
    b = bg()
    time.sleep(0.5 * (i % 4))
    try:
        got = run("1", 29701 + i, f"/tmp/mr_{i}.json")
    finally:
        b.kill(); b.wait()
    first = None
    for k, (ga, gb) in enumerate(zip(got["steps"], ref["steps"])):
        diff = [(n, ga[n], gb[n]) for n in gb if ga.get(n) != gb[n]]
        if diff:
            first = (k, diff); break
    print(i, "overlap loss", got["loss"], "first differing step:", None if first is None else (first[0], len(first[1])), flush=True)
    if first:
        for n, a, c in first[1][:25]:
            print("    ", n, a, c)
