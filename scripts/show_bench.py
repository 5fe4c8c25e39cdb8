"""print the kernel-family table of a bench.py JSON line: python scripts/show_bench.py file.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["ms_per_step"], "ms/step |", d["dtype"])
r = d["roofline"]
print("roofline:", r["kernel"][:40], r["ms_per_step"], "ms", r["achieved"], r["unit"], "frac", r["frac"])
for k in ("roofline_encoder", "roofline_encoder_fp8"):
    if k in d:
        print(k, d[k]["ms_per_step"], "ms", d[k]["achieved"], "frac", d[k]["frac"])
for k, v in sorted(d["kernel_families"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print(f"{k:24s} {v['ms_per_step']:7.3f} ms  {v['achieved']:8.1f} {v['unit']} frac {v['frac']}  ({v['launches_per_step']} launches)")
print("mfma share", d.get("mfma_kernels_share_of_step"), "| parity", {k: v for k, v in d.get("parity_at_dtype", {}).items() if k.startswith("l")})
