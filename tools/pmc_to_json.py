#!/usr/bin/env python
"""profiles/pmc_traffic.json from the text tools/pmc_summary.py wrote for the FETCH_SIZE and WRITE_SIZE passes
(usage: python tools/pmc_to_json.py <raw summary txt> <out json> "<comment>")."""
import hashlib
import json
import os
import re
import sys

raw, out_path, comment = sys.argv[1], sys.argv[2], sys.argv[3]
vals, name = {}, None
for line in open(raw).read().splitlines():
    if line.startswith("   "):
        parts = line.split()
        vals.setdefault(name, {})[parts[0]] = (float(parts[2]), int(line.split("n=")[1]))
    else:
        name = line.strip()
out = {"_comment": comment}
for n, d in vals.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        k = re.sub(r"\(.*$", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))
        f, nf = d["FETCH_SIZE"]
        w, _ = d["WRITE_SIZE"]
        out[k] = {"fetch_kb": f, "write_kb": w, "launches": nf, "bytes_per_launch": int((2 * f + w) * 1024)}
if "usc::gather_gemm_sorted_kernel<4, 4>" in out:      # bench.py labels the sorted kernel without template arguments
    out["usc::gather_gemm_sorted_kernel"] = dict(out["usc::gather_gemm_sorted_kernel<4, 4>"], note="NB=4 instantiation")
# sha256 of every kernel source the figures were measured on: bench.py reports `traffic: null` for a kernel whose source
# file no longer matches (a stale committed constant must not pass for a measurement)
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unscene3d_amd", "csrc")
out["_source_sha256"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()
                         for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))}
json.dump(out, open(out_path, "w"), indent=1)
