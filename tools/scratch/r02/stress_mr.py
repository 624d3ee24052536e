import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(overlap, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", USC3D_OVERLAP_ALLREDUCE=overlap)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--voxels", "40000", "--dist-backend", "gloo", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return out.stderr[-300:], ""
    r = json.loads(lines[0])
    return r["config"]["loss"], r["config"]["grad_allreduce"]
def bg():
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "400", "--warmup", "1", "--voxels", "80000",
                             "--no-cpu-baseline"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
ref, _ = run("0", 29700)
print("single", repr(ref), flush=True)
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 36
for i in range(N):
    b = bg()
    time.sleep(0.7 * (i % 5))
    try:
        got, note = run("1", 29701 + i)
    finally:
        b.kill(); b.wait()
    if got != ref:
        bad += 1
        print(i, "DEVIATION", repr(got), flush=True)
    if i == 0:
        print(note, flush=True)
print("deviating overlapped runs:", bad, "of", N)
