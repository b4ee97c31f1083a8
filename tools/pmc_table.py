"""Per-kernel table from a tools/profile_cmd.sh output directory: average duration (kernel trace) and, from the PMC pass,
MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) and the clock GRBM_GUI_ACTIVE / 8 /
duration.  usage: python tools/pmc_table.py gpurun_out/prof_<tag> [name filter]"""
import collections, csv, glob, os, re, sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""


def short(n):
    n = re.sub(r"plship::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:110]


rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if flt in k:
            rows[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, d in sorted(rows.items()):
    big = max(x[1] for v in d.values() for x in v)
    ent = {}
    for c, v in d.items():
        sel = [x for x in v if x[1] >= 0.5 * big] or v
        ent[c] = sum(x[0] for x in sel) / len(sel)
        ent["dur_us"] = sum(x[1] for x in sel) / len(sel) / 1e3
        ent["n"] = len(sel)
    line = f"{k:110s} n={ent['n']:5d} dur {ent['dur_us']:9.1f} us"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and "GRBM_GUI_ACTIVE" in ent:
        per_xcd = ent["GRBM_GUI_ACTIVE"] / 8.0
        line += f"  MFMA-pipe {ent['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / per_xcd:6.3f}  clock {per_xcd / (ent['dur_us'] * 1e3):5.2f} GHz"
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"):
        if c in ent:
            line += f"  {c} {ent[c]:.3g}"
    print(line)
