"""print the kernel-family table of bench.py JSON lines: python scripts/show_bench.py file.json [file2.json ...]
(the LAST line of each file is the bench line; closing the pipe early - `| head`, `| tail` of a dead reader - ends the script quietly)"""
import json
import os
import sys


def show(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(f"== {path}")
    print(d["value"], d["unit"], d["ms_per_step"], "ms/step |", d["dtype"], "|", d.get("config", {}).get("workload", ""))
    r = d["roofline"]
    print("roofline:", r["kernel"][:40], r["ms_per_step"], "ms", r["achieved"], r["unit"], "frac", r["frac"])
    for k in ("roofline_encoder", "roofline_encoder_fp8"):
        if k in d:
            print(k, d[k]["ms_per_step"], "ms", d[k]["achieved"], "frac", d[k]["frac"])
    for k, v in sorted(d["kernel_families"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        print(f"{k:24s} {v['ms_per_step']:7.3f} ms  {v['achieved']:8.1f} {v['unit']} frac {v['frac']}  ({v['launches_per_step']} launches)")
    print("mfma share", d.get("mfma_kernels_share_of_step"), "| parity", {k: v for k, v in d.get("parity_at_dtype", {}).items() if k.startswith("l")})


if __name__ == "__main__":
    try:
        for f in sys.argv[1:]:
            show(f)
        sys.stdout.flush()
    except BrokenPipeError:
        # the reader went away: point stdout at /dev/null so that the interpreter's exit flush does not raise again, and leave with success
        os.dup2(os.open(os.devnull, os.O_WRONLY), sys.stdout.fileno())
        sys.exit(0)
