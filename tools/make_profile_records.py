#!/usr/bin/env python3
"""Turn a tools/prof_r02.sh output directory into the two records bench.py reads: profiles/valu_counts.json (dynamic VALU
wave-instructions per transform / per tree, clock under load) and profiles/hbm_traffic_ntt.json (HBM-side bytes per launch).
usage: make_profile_records.py gpurun_out/prof_<tag> <tag>"""
import collections, csv, glob, json, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
summ = json.load(open(os.path.join(src, "summary.json")))
try:
    library = json.load(open(os.path.join(src, "library.json")))  # the build these counters were taken on (tf_source_hash)
except Exception:
    library = None
ntt, clocks, per, tot, n = 0.0, [], {}, 0.0, 0
for k, e in summ["kernels"].items():
    if "ntt_pass_kernel" in k and e.get("duration", {}).get("grid") == 8388608:  # the 256 x 2^20 BFE dispatches
        c, d = e["counters"], e["derived"]
        ntt += c["SQ_INSTS_VALU"]
        clocks.append(d["clock_under_load_mhz"])
        per["void tfk::" + k + "(tfk::NttPassArgs)"] = {"FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "hbm_bytes": d["hbm_side_bytes_per_dispatch"],
                                                        "dispatches": e["duration"]["dispatches"], "avg_us_under_pmc_profiler": e["duration"]["avg_us"],
                                                        "valu_instr_per_wave": d["valu_instr_per_wave"], "valu_busy_frac_at_4_cycles": d["valu_busy_frac_at_4_cycles"]}
        tot += d["hbm_side_bytes_per_dispatch"]
        n += 1
# Merkle: every dispatch of the Tip5 / Merkle kernels of the trees built by tools/prof_target.py
mer, trees = collections.defaultdict(float), 0
for f in glob.glob(os.path.join(src, "pmc_sq/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU" and ("tip5" in r["Kernel_Name"] or "merkle" in r["Kernel_Name"]):
            mer[r["Kernel_Name"]] += float(r["Counter_Value"])
            if "merkle_top_kernel" in r["Kernel_Name"]:
                trees += 1
mfma, mfma_busy, mtrees = 0.0, 0.0, 0
for f in glob.glob(os.path.join(src, "pmc_mfma/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "tip5" in r["Kernel_Name"] or "merkle" in r["Kernel_Name"]:
            if r["Counter_Name"] == "SQ_INSTS_MFMA":
                mfma += float(r["Counter_Value"])
                if "merkle_top_kernel" in r["Kernel_Name"]:
                    mtrees += 1
            elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                mfma_busy += float(r["Counter_Value"])
# configs[3]: the two PRE2 pass kernels of the 64 x 2^22 XFE coset evaluation (SCALE 1 column pass, LAST1024 pass)
coset_busy = [e["derived"]["valu_busy_frac_at_4_cycles"] for k, e in summ["kernels"].items()
              if "ntt_pass_kernel" in k and "valu_busy_frac_at_4_cycles" in e.get("derived", {})
              and (k.startswith("ntt_pass_kernel<false, 1, 0, false, true, false, true") or k.startswith("ntt_pass_kernel<false, 0, 0, true, false, false, true"))]
# (round 6: the first pass of configs[3] is ntt_col2048_kernel<false, 1>, bound by the memory pipe at ~0.68 busy; the last pass stays the PRE2 kernel)
coset_busy += [e["derived"]["valu_busy_frac_at_4_cycles"] for k, e in summ["kernels"].items()
               if k.startswith("ntt_col2048_kernel<false, 1>") and "valu_busy_frac_at_4_cycles" in e.get("derived", {})]
rec = {"library": library, "coset_eval_valu_busy_frac_at_4_cycles": coset_busy or None, "source": f"rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU ... (tools/prof_r02.sh {tag} -> profiles/{tag}_rocprof_summary.json): SQ_INSTS_VALU summed over the dispatches of one step / one tree",
       "ntt_valu_wave_instr_per_transform_2p20": ntt / 256, "ntt_valu_instr_per_element": ntt * 64 / 2 ** 28,
       "ntt_clock_under_load_mhz": round(sum(clocks) / len(clocks), 1) if clocks else None,
       "valu_peak_note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 614.4 G wave-instr/s: every instruction of these kernels is of the 4-cycle class (carry adds, v_mad_u64_u32, VOP3; profiles/r03_instr_rates.txt).  SQ_ACTIVE_INST_VALU counts quad-cycles, so its ratio to SQ_INSTS_VALU is the counter's granularity, not a cost"}
old = {}
try:
    old = json.load(open(os.path.join(root, "profiles", "valu_counts.json")))
except Exception:
    pass
if trees:
    m = sum(mer.values()) / trees
    rec["merkle_valu_wave_instr_per_tree_2p24"] = m
    rec["merkle_valu_instr_per_hash_pair"] = m * 64 / (2 ** 24 - 1)
    if mtrees:
        rec["merkle_mfma_wave_instr_per_tree_2p24"] = mfma / mtrees
        rec["merkle_mfma_busy_cycles_per_tree_2p24"] = mfma_busy / mtrees
        rec["merkle_note"] = ("SQ_INSTS_VALU counts the v_mfma_i32_16x16x64_i8 of the matrix-pipe Tip5 kernels as VALU instructions; per 16 hash_pairs "
                              "(one wave): %.1f VALU instructions of which %.1f MFMA" % (m * 16 / (2 ** 24 - 1), mfma / mtrees * 16 / (2 ** 24 - 1)))
else:  # NTT-only profile: keep the Merkle figures of the last record
    for k in ("merkle_valu_wave_instr_per_tree_2p24", "merkle_valu_instr_per_hash_pair", "merkle_mfma_wave_instr_per_tree_2p24", "merkle_mfma_busy_cycles_per_tree_2p24", "merkle_note"):
        if k in old:
            rec[k] = old[k]
json.dump(rec, open(os.path.join(root, "profiles", "valu_counts.json"), "w"), indent=1)
json.dump({"library": library, "log_n": 20, "batch": 256, "launches_per_step": 2,
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/prof_r02.sh {tag} -> profiles/{tag}_rocprof_summary.json), tools/prof_target.py: every dispatch is a full-size one",
           "per_kernel": per,
           "correction": "FETCH_SIZE doubled: on gfx950 it tallies the 128-byte requests of a fully coalesced stream at 64 B (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is (it equals the 2^31 bytes each pass must write)",
           "hbm_bytes_per_launch": tot / n, "algorithmic_bytes_per_launch": 2147483648,
           "note": "each pass reads and writes every element exactly once (2 x the algorithmic 16 B/element per transform is the two-pass design); the inter-pass table stays in L2 since the data stream is non-temporal. These counters sit on the L2's memory side: reads served by the Infinity Cache are counted as well, so Infinity-Cache residency cannot be read off them."},
          open(os.path.join(root, "profiles", "hbm_traffic_ntt.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
