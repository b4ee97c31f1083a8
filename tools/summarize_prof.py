"""Summarise a tools/profile_bench.sh output directory into profiles/<tag>_*.{csv,json}.
usage: python tools/summarize_prof.py gpurun_out/prof_<tag> <tag>"""
import collections, csv, glob, json, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(out_dir, f"{tag}_bench_kernel_stats.csv"))


def short(n):
    m = re.search(r"gemm_tn_f64_kernel<(\d+), (\d+).*?plship::(\w+)", n)
    if m:
        return f"gemm_tn_f64<{m.group(1)}x{m.group(2)}>::{m.group(3)}"
    m = re.search(r"gemm_tn_f64_kg_tri_kernel<.*?plship::(\w+)", n)  # (round 4 folded every k-split launch into one name)
    if m:
        return f"gemm_tn_f64_kg_tri::{m.group(1)}"
    m = re.search(r"gemm_tn_f64_kg_kernel<(\d+).*?plship::(\w+)", n)
    if m:
        return f"gemm_tn_f64_kg<{m.group(1)}>::{m.group(2)}"
    m = re.search(r"gemm_tn_f64_rows_kernel<.*?plship::(\w+)", n)
    if m:
        return f"gemm_tn_f64_rows::{m.group(1)}"
    m = re.search(r"small_rank_step_kernel<(\d+), (-?\d+), (-?\d+), (true|false|0|1)>", n)
    if m:
        return f"small_rank_step_kernel<KB={m.group(1)},cost={m.group(2)},link={m.group(3)},energies={m.group(4)}>"
    m = re.search(r"small_rank_kernel<(\d+), (\d+), (-?\d+), (-?\d+)>", n)
    if m:
        mode = {"0": "drift", "1": "value", "2": "drift+value"}[m.group(2)]
        return f"small_rank_kernel<KB={m.group(1)},{mode},cost={m.group(3)},link={m.group(4)}>"
    m = re.search(r"plship::(\w+)", n)
    return m.group(1) if m else None


rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc*", "bench_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        rows[k][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
summary = {}
for k, d in rows.items():
    # keep the step-sized launches only: the longest ones of that kernel (setup GEMMs and the M x M x J products of
    # the extras run the same kernel on much smaller shapes)
    big = max(x[1] for v in d.values() for x in v)
    ent = {}
    for c, v in d.items():
        sel = [x for x in v if x[1] >= 0.5 * big] or v
        ent[c] = sum(x[0] for x in sel) / len(sel)
        ent["_dur_ms"] = sum(x[1] for x in sel) / len(sel) / 1e6
        ent["_launches"] = len(sel)
    if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half the bytes
        # of a wide coalesced stream -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores
        ent["hbm_read_bytes_per_launch"] = 2.0 * ent["FETCH_SIZE"] * 1024
        ent["hbm_write_bytes_per_launch"] = ent["WRITE_SIZE"] * 1024
        ent["hbm_bytes_per_launch"] = ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and "GRBM_GUI_ACTIVE" in ent:
        per_simd = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0   # 256 CUs x 4 SIMDs
        per_xcd = ent["GRBM_GUI_ACTIVE"] / 8.0
        ent["mfma_pipe_utilisation"] = per_simd / per_xcd
        ent["clock_ghz"] = per_xcd / (ent["_dur_ms"] * 1e6)
    if "SQ_ACTIVE_INST_VALU" in ent and "GRBM_GUI_ACTIVE" in ent:
        # quad-cycles the vector ALU (MFMA included: the same issue port for fp64) is busy, per SIMD, against the launch's cycles
        ent["valu_busy_frac"] = ent["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (ent["GRBM_GUI_ACTIVE"] / 8.0)
    if "TCC_HIT_sum" in ent:
        ent["l2_hit_rate"] = ent["TCC_HIT_sum"] / (ent["TCC_HIT_sum"] + ent["TCC_MISS_sum"])
    summary[k] = ent
sys.path.insert(0, ROOT)
from tools.source_hash import kernel_source_hash  # noqa: E402

summary["_kernel_source_hash"] = kernel_source_hash()  # bench.py quotes roofline.traffic only from a matching summary
json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, e in summary.items():
    if not isinstance(e, dict):
        continue
    print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in e.items() if kk.startswith(("_", "hbm", "mfma", "clock", "l2"))})
