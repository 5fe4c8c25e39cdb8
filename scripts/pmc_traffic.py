"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel-family HBM traffic per launch.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE counts 128-byte fabric requests at 64 bytes for wide
coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Both are reported in KiB.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import re
import sys


def family(name):
    n = re.sub(r"\(.*", "", name).replace("void cvcs::", "").replace("cvcs::", "")
    if n.startswith("conv3x3_halo_kernel"):
        return "conv3x3_halo_kernel"
    if n.startswith("conv_igemm_kernel"):
        return "conv_igemm_kernel"
    if n.startswith("wgrad") or n.startswith("void wgrad"):
        return "wgrad"          # every weight-gradient kernel, split-K / bias reduces included (bench.py's `wgrad` family = one entry per layer)
    if n.startswith("dwconv"):
        return "dwconv"         # depthwise forward / data gradient / weight gradient launches (bench.py's `hbm_dwconv` family)
    return re.sub(r"<.*", "", n)


def is_reduce(name):
    return "reduce" in name


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        agg[f][0] += 0 if (f == "wgrad" and is_reduce(r["Kernel_Name"])) else 1      # a layer's reduce launches ride on its main launch
        agg[f][1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for f in sorted(set(fetch) | set(write)):
    n = max(fetch[f][0], write[f][0])
    rd = 2.0 * fetch[f][1] * 1024 / max(1, fetch[f][0])
    wr = write[f][1] * 1024 / max(1, write[f][0])
    out[f] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for f, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{f:28s} n={v['launches']:4d} read {v['read_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB per launch")
