"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count, mean duration."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cvcs::", "").replace("cvcs::", "")
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v)/tot*100:5.1f}%  n={len(v):4d} avg={sum(v)/len(v):9.1f}us  {k}")
