"""Per-kernel SQ counters from a rocprofv3 --pmc pass (counter_collection.csv)."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cvcs::", "").replace("cvcs::", "")
    key = (n, r["Grid_Size"])
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[key] += 1
for key, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    w = c.get("SQ_WAVE_CYCLES", 1)
    print(f"{key[0][:52]:52s} grid={key[1]:>9s} n={cnt[key]:3d} " + " ".join(
        f"{k.replace('SQ_', '')}={v / w * 100:5.1f}%" for k, v in sorted(c.items()) if k != "SQ_WAVE_CYCLES") + f" wave_cycles={w/cnt[key]:.3g}")
