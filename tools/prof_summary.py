#!/usr/bin/env python3
"""Summarise a tools/prof_r02.sh output directory: per kernel, average duration (kernel trace) and average counter values
per dispatch (only the dispatches with the kernel's largest grid), plus derived figures: clock under load, VALU busy
fraction, VALU instructions per wave, HBM-side bytes (FETCH_SIZE doubled per the guide's gfx950 correction)."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

out = sys.argv[1]
WANT = ("ntt_pass_kernel", "ntt_col2048_kernel", "ntt_block_kernel", "ntt_lat", "tip5_", "merkle_", "fill_random")


def short(name):
    return name.replace("void tfk::", "").replace("(tfk::NttPassArgs)", "")


def files(pat):
    return glob.glob(os.path.join(out, pat), recursive=True)


summary = {"kernels": {}}
dur = collections.defaultdict(list)
for f in files("stats/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(w in k for w in WANT):
            dur[k].append((int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in dur.items():
    g = max(x[0] for x in v)
    d = [x[1] for x in v if x[0] == g]
    summary["kernels"].setdefault(short(k), {})["duration"] = {"dispatches": len(d), "grid": g, "avg_us": sum(d) / len(d) / 1e3,
                                                                "median_us": statistics.median(d) / 1e3, "total_ms_all_grids": sum(x[1] for x in v) / 1e6,
                                                                "dispatches_all_grids": len(v)}
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files(sub + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(w in k for w in WANT):
                vals[k][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"]), r.get("Dispatch_Id")))
    for k, cs in vals.items():
        e = summary["kernels"].setdefault(short(k), {}).setdefault("counters", {})
        for c, v in cs.items():
            g = max(x[0] for x in v)
            per_disp = collections.defaultdict(float)  # a counter may be reported per XCD/instance: sum within a dispatch
            for x in v:
                if x[0] == g:
                    per_disp[x[2]] += x[1]
            e[c] = sum(per_disp.values()) / len(per_disp)
for k, e in summary["kernels"].items():
    c, d = e.get("counters", {}), e.get("duration")
    der = {}
    if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
        der["valu_instr_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_INSTS_VALU"):
        der["quad_cycles_per_valu_instr"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]
    if c.get("GRBM_GUI_ACTIVE") and d:
        der["clock_under_load_mhz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / d["avg_us"]  # counter summed over the 8 XCDs
        if c.get("SQ_INSTS_VALU"):
            # VALU issue: one wave-instruction occupies a SIMD for 4 cycles; 1024 SIMDs
            der["valu_busy_frac_at_4_cycles"] = c["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
            der["g_wave_instr_per_s"] = c["SQ_INSTS_VALU"] / (d["avg_us"] * 1e-6) / 1e9
    if c.get("SQ_WAVE_CYCLES"):
        for w in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if w in c:
                der[w.lower() + "_frac_of_wave_cycles"] = c[w] / c["SQ_WAVE_CYCLES"]
        if c.get("GRBM_GUI_ACTIVE"):
            der["resident_waves_per_simd"] = 4.0 * c["SQ_WAVE_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)
    if "FETCH_SIZE" in c:
        der["fetch_bytes_x2_gfx950"] = 2.0 * 1024.0 * c["FETCH_SIZE"]
    if "WRITE_SIZE" in c:
        der["write_bytes"] = 1024.0 * c["WRITE_SIZE"]
    if "fetch_bytes_x2_gfx950" in der and "write_bytes" in der:
        der["hbm_side_bytes_per_dispatch"] = der["fetch_bytes_x2_gfx950"] + der["write_bytes"]
        if d:
            der["hbm_side_tb_per_s"] = der["hbm_side_bytes_per_dispatch"] / (d["avg_us"] * 1e-6) / 1e12
    e["derived"] = der
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, e in sorted(summary["kernels"].items(), key=lambda kv: -kv[1].get("duration", {}).get("total_ms_all_grids", 0)):
    d = e.get("duration", {})
    print(f"{k}\n   dispatches {d.get('dispatches')} grid {d.get('grid')} avg {d.get('avg_us', 0):.1f} us median {d.get('median_us', 0):.1f} us")
    for kk, vv in e.get("derived", {}).items():
        print(f"   {kk:40s} {vv:.6g}")
