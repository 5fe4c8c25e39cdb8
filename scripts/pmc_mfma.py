"""MFMA utilisation per kernel family from a rocprofv3 --pmc pass that collected SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES,
GRBM_GUI_ACTIVE (and optionally SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY):
  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs)   (the gfx94x derived-metric formula -
  ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md "PMC slots" - with one calibration: rocprofv3 reports GRBM_GUI_ACTIVE
  SUMMED over the 8 XCDs; a launch of known duration reads 8x its shader cycles: 3.77 M for a 205 us conv3x3_halo launch)
usage: pmc_mfma.py <counter_collection.csv> [<out.txt>]"""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void cvcs::", "").replace("cvcs::", "")
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[n].add(r["Dispatch_Id"])
lines = [f"{'kernel':64s} {'launches':>8s} {'MfmaUtil %':>10s} {'GUI_ACTIVE/launch':>18s} {'wait_any %':>10s} {'active_inst %':>13s}"]
for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    util = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    wa = 100.0 * c.get("SQ_WAIT_ANY", 0.0) / wc if wc else float("nan")
    ai = 100.0 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else float("nan")
    lines.append(f"{n[:64]:64s} {len(disp[n]):8d} {util:10.1f} {gui / 8 / len(disp[n]):18.0f} {wa:10.1f} {ai:13.1f}")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
