"""bench.py's CPU legs and record-matching logic, exercised without a GPU at small sizes: the oracle timings the bench reports as
`cpu_baseline` (kind "port") must run and return what the parity checks compare with, and a stored profile must only be quoted for
the library build it was taken on."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_legs_run_and_return_oracle_words(oracle):
    b = _bench()
    info, _ = b.cpu_baseline_ntt(10, 8, b.SEED_C2)
    assert info["kind"] == "port" and info["unit"] == "GFelts/s" and info["value"] > 0 and info["cores"] >= 1 and "build_flags" in info
    info, root, nodes = b.cpu_baseline_merkle(1 << 10, b.SEED_C3)
    want = oracle.merkle_build(oracle.fill_random(5 << 10, b.SEED_C3))
    assert np.array_equal(nodes, want) and np.array_equal(root, want[5:10]) and info["unit"] == "leaves/s"
    off = oracle.bfe_new(7)
    info, want0 = b.cpu_baseline_coset(1 << 10, 4, b.SEED_C4, off)
    assert np.array_equal(want0, oracle.coset_evaluate(oracle.fill_random(3 << 10, b.SEED_C4), off, 1 << 10, width=3))
    assert info["unit"] == "G points/s" and info["cores"] >= 1 and info["single_thread_value"] > 0


def test_stored_profile_is_quoted_only_for_the_build_it_was_taken_on():
    b = _bench()
    ident = {"tf_version": 1001, "source_hash": "0123456789abcdef"}
    assert b.profile_matches({"library": dict(ident)}, ident)
    assert not b.profile_matches({"library": {"tf_version": 1001, "source_hash": "fedcba9876543210"}}, ident)
    assert not b.profile_matches({"library": {"tf_version": 1000, "source_hash": "0123456789abcdef"}}, ident)
    assert not b.profile_matches({"library": None}, ident) and not b.profile_matches({}, ident) and not b.profile_matches(None, ident)


def test_library_reports_the_hash_of_its_sources(tf):
    """tf_source_hash() (no device needed) is the first 16 hex digits of the SHA-256 of the csrc sources in the Makefile's order."""
    import hashlib
    import re

    csrc = os.path.join(ROOT, "twenty-first_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    common = re.search(r"^COMMON := (.*)$", mk, re.M).group(1).split()
    sources = re.search(r"^SOURCES := (.*)$", mk, re.M).group(1).replace("$(COMMON)", " ".join(common)).split()
    h = hashlib.sha256()
    for name in sorted(sources):
        h.update(open(os.path.join(csrc, name), "rb").read())
    got = tf.lib().tf_source_hash().decode()
    assert got in (h.hexdigest()[:16], h.hexdigest()[:16] + "-ab"), "libtf_hip.so is older than its sources: rebuild (make -C twenty-first_amd/csrc)"


def test_every_leg_survives_in_the_tail_of_the_bench_line():
    """The driver's record keeps only the tail of a long line (round 5's lost `merkle.value` that way): bench.py emits a compact `summary`
    as the LAST key -- every leg's value, ms_per_step, roofline.frac, cpu_baseline.value + cores, parity -- in at most 1.5 KB.  Here: a
    captured full record (profiles/r05_bench_final.json), the summary rebuilt by bench.summary_block, the line cut to its last 6 KB as a
    reader of the driver's record sees it, and all legs found again by bench.parse_summary_from_tail; the headline keys still come first."""
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = json.load(open(os.path.join(root, "profiles", "r05_bench_final.json")))
    rec.pop("summary", None)
    rec["summary"] = bench.summary_block(rec)
    line = json.dumps(rec)
    assert len(line) > 8192, "the point of the test: the line is longer than what the driver keeps"
    assert len(json.dumps(rec["summary"])) <= 1536
    assert list(rec.keys())[:5] == ["metric", "value", "unit", "n_gpus", "steps"] and list(rec.keys())[-1] == "summary"
    got = bench.parse_summary_from_tail(line[-6144:])
    assert got == rec["summary"]
    for leg in ("headline", "merkle", "coset_eval", "config5", "commit_pipeline"):
        assert got[leg]["value"] and got[leg]["ms"] and got[leg]["cpu"] and got[leg]["cores"] and got[leg]["parity"].startswith("green")
    assert got["merkle"]["value"] == rec["merkle"]["value"] and got["merkle"]["frac"] == rec["merkle"]["roofline"]["frac"]
    assert got["coset_eval"]["frac"] == rec["coset_eval"]["roofline"]["frac"] and got["headline"]["frac"] == rec["roofline"]["frac"]
    assert got["config5"]["trees_leaves_per_s"] == rec["config5"]["merkle"]["value"]
    # a failed leg and an unchecked one stay readable
    rec2 = dict(rec, merkle={"error": "RuntimeError('x')"}, coset_eval=dict(rec["coset_eval"], parity="not checked (--no-cpu-baseline)", cpu_baseline=None))
    sm = bench.summary_block(rec2)
    assert "error" in sm["merkle"] and sm["coset_eval"]["parity"] == "not checked" and "cpu" not in sm["coset_eval"]


def test_stored_counter_records_belong_to_this_build(tf):
    """bench.py quotes `roofline.traffic`, `valu_bound` and the Tip5 / composite fractions from stored rocprofv3 records only when their
    `library.source_hash` is tf_source_hash() of the library it times (a comment-only edit of a source once left the three records one
    build behind: the line then drops those keys instead of going stale, but the committed state should never be in that position)."""
    import json

    want = tf.lib().tf_source_hash().decode().replace("-ab", "")
    for name in ("hbm_traffic_ntt.json", "valu_counts.json", "pipeline_counters.json"):
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert rec["library"]["source_hash"] == want, f"profiles/{name} was recorded on build {rec['library']['source_hash']}, the sources are {want}: re-run tools/r06_final_session.sh a"
