"""Per-dispatch table of the LAST training step of a profiled bench.py run: kernel, grid, duration (rocprofv3 --kernel-trace) joined by
dispatch order with the FETCH_SIZE / WRITE_SIZE passes (separate runs of the same command: the launch sequence of a step is deterministic).
usage: dispatch_table.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> <steps in the trace run> <steps in the pmc runs>
gfx950 corrections as scripts/pmc_traffic.py (FETCH_SIZE doubled, both in KiB)."""
import csv
import re
import sys


def short(name):
    n = re.sub(r"\(.*", "", name).replace("void cvcs::", "").replace("cvcs::", "").replace("void ", "")
    return n.replace("unsigned short", "bf16")


def last_step(rows, key, steps):
    """the dispatches of the last step: the tail of the list, split at the last `steps` occurrences of the step's first kernel (sgd_kernel ends a step)"""
    ends = [i for i, r in enumerate(rows) if "sgd_kernel" in r[key]]
    return rows[ends[-2] + 1: ends[-1] + 1] if len(ends) >= 2 else rows


tr = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Dispatch_Id"]))
tr = last_step(tr, "Kernel_Name", 0)
pm = {}
for path, ctr in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    rows = sorted((r for r in csv.DictReader(open(path)) if r["Counter_Name"] == ctr), key=lambda r: int(r["Dispatch_Id"]))
    pm[ctr] = last_step(rows, "Kernel_Name", 0)
n = len(tr)
ok = all(len(v) == n for v in pm.values())
print(f"# {n} dispatches in the last step; pmc passes aligned: {ok}")
tot_t = tot_r = tot_w = 0.0
for i, r in enumerate(tr):
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = f'{int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}'
    rd = wr = float("nan")
    if ok:
        assert short(pm["FETCH_SIZE"][i]["Kernel_Name"]) == short(r["Kernel_Name"]), (i, r["Kernel_Name"], pm["FETCH_SIZE"][i]["Kernel_Name"])
        rd = 2.0 * float(pm["FETCH_SIZE"][i]["Counter_Value"]) * 1024 / 1e6
        wr = float(pm["WRITE_SIZE"][i]["Counter_Value"]) * 1024 / 1e6
        tot_r += rd; tot_w += wr
    tot_t += dur
    print(f"{i:4d} {short(r['Kernel_Name'])[:52]:52s} {grid:>14s} {dur:8.1f} us  rd {rd:8.1f} MB  wr {wr:8.1f} MB  {(rd + wr) / dur if dur > 0 else 0:6.2f} TB/s")
print(f"# total {tot_t / 1e3:.2f} ms  read {tot_r / 1e3:.2f} GB  write {tot_w / 1e3:.2f} GB")
