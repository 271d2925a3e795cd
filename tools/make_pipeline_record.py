#!/usr/bin/env python3
"""profiles/pipeline_counters.json from a tools/prof_r02.sh directory of `tools/prof_target.py --ntt 0 --merkle 0 --coset 0 --pipeline N`
(bench.py's commit_pipeline shapes): HBM-side bytes of the low-degree extension's kernels per call, vector / matrix instructions of the
row-hashing kernel and the tree per call, and a per-kernel table.   usage: make_pipeline_record.py gpurun_out/prof_<tag> <tag> <N>"""
import collections, csv, glob, json, os, sys

src, tag, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
try:
    library = json.load(open(os.path.join(src, "library.json")))
except Exception:
    library = None


def short(name):
    return name.replace("void tfk::", "").replace("(tfk::NttPassArgs)", "")


def sums(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(src, sub + "/**/*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "stats/**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
sq, mf, fe, wr = sums("pmc_sq"), sums("pmc_mfma"), sums("pmc_fetch"), sums("pmc_write")
is_lde = lambda k: k.startswith("ntt_") or "ntt_" in k.split("<")[0]
is_hash = lambda k: "tip5_" in k or "merkle_" in k
rec = {"library": library, "reps": reps,
       "source": f"rocprofv3 --kernel-trace (+ --pmc passes, each its own run) on tools/prof_target.py --pipeline {reps} (tools/prof_r02.sh {tag}); sums over ALL dispatches of a call, divided by the repetitions; "
                 "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md"}
lde_k, rows_k = {}, {}
lde_bytes = rows_valu = rows_mfma = 0.0
for k in sorted(dur, key=lambda x: -sum(dur[x])):
    e = {"us_per_call": round(sum(dur[k]) / reps / 1e3, 1), "launches_per_call": round(len(dur[k]) / reps, 2)}
    if k in sq and sq[k].get("SQ_WAVES"):
        e["valu_instr_per_wave"] = round(sq[k]["SQ_INSTS_VALU"] / sq[k]["SQ_WAVES"], 1)
        if sq[k].get("GRBM_GUI_ACTIVE"):
            e["valu_busy_frac_at_4_cycles"] = round(sq[k]["SQ_INSTS_VALU"] * 4.0 / 1024.0 / (sq[k]["GRBM_GUI_ACTIVE"] / 8.0), 3)
        if sq[k].get("SQ_WAVE_CYCLES"):
            e["sq_wait_any_frac"] = round(sq[k].get("SQ_WAIT_ANY", 0.0) / sq[k]["SQ_WAVE_CYCLES"], 3)
    if k in fe or k in wr:
        hb = 2048.0 * fe[k].get("FETCH_SIZE", 0.0) + 1024.0 * wr[k].get("WRITE_SIZE", 0.0)
        e["hbm_side_gb_per_call"] = round(hb / reps / 1e9, 4)
    if is_lde(k):
        lde_k[k] = e
        lde_bytes += (2048.0 * fe[k].get("FETCH_SIZE", 0.0) + 1024.0 * wr[k].get("WRITE_SIZE", 0.0)) / reps
    elif is_hash(k):
        if k in mf:
            e["mfma_per_wave"] = round(mf[k].get("SQ_INSTS_MFMA", 0.0) / max(mf[k].get("SQ_WAVES", 1.0), 1.0), 1)
        rows_k[k] = e
        rows_valu += sq[k].get("SQ_INSTS_VALU", 0.0) / reps if k in sq else 0.0
        rows_mfma += mf[k].get("SQ_INSTS_MFMA", 0.0) / reps if k in mf else 0.0
rec.update({"lde_hbm_bytes_per_call": lde_bytes, "lde_kernels": lde_k, "rows_valu_wave_instr_per_call": rows_valu, "rows_mfma_wave_instr_per_call": rows_mfma,
            "rows_kernels": rows_k})
json.dump(rec, open(os.path.join(root, "profiles", "pipeline_counters.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
