"""Developer aid: per-shape timing of every sparse-conv launch of one bench step.
USC3D_PROF_SHAPES=1 python tools/conv_report.py [--mode mask3d|backbone]"""
import os, sys, json, subprocess
os.environ["USC3D_PROF_SHAPES"] = "1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True, env=os.environ)
line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("ms/step", d["ms_per_step"])
rows = sorted(d["roofline"]["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print(f"total conv ms {tot:.2f}")
for k, v in rows:
    print(f"{v['ms']:8.3f} ms {v['launches']:4d}x {1e3*v['ms']/v['launches']:8.1f} us  {v['tflops'] or 0:6.1f} TF  {k}")
