import json, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def run(extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--voxels", "40000", "--no-cpu-baseline"] + list(extra)
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1])["config"]["loss"] if lines else out.stderr[-300:]
bg = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3000", "--warmup", "1", "--voxels", "80000",
                       "--no-cpu-baseline"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
try:
    print("single-GPU losses:", [repr(run()) for _ in range(8)], flush=True)
    print("no-graphs losses:", [repr(run(["--no-graphs"])) for _ in range(4)], flush=True)
    # SDPA backward determinism
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 8, 100, 16, device="cuda", requires_grad=True) for _ in range(3))
    go = torch.randn(1, 8, 100, 16, device="cuda")
    ref = None; bad = 0
    for it in range(300):
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        g = torch.autograd.grad(o, (q, k, v), go)
        if ref is None: ref = [t.clone() for t in g]
        elif not all(torch.equal(a, b) for a, b in zip(g, ref)): bad += 1
    print("sdpa backward runs differing from the first:", bad, "of 299")
finally:
    bg.kill()
