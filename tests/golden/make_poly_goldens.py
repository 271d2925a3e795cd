#!/usr/bin/env python3
"""Generate goldens for the polynomial callers built in the second half of round 2 from the property-pinned oracle.

The reference pins none of these with literal vectors (its tests are property tests and doc examples, math/polynomial.rs:3440-3800,
:4593-4638); the oracle's restatements are pinned by those properties and doc examples (tests/test_oracle_kat.py).  This script
records inputs (as SplitMix64 seeds: tfo.fill_random) and canonical-value outputs for small cases so that later changes to the
oracle or to the HIP path are caught word for word:
  zerofier (:1462-1475), lagrange_interpolate (:1565-1606), clean_divide (:2358-2411), barycentric_evaluate (:2609-2637),
  fast_coset_evaluate with an XFieldElement offset (:1374-1399), Tip5::trace (tip5/mod.rs:538-548).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tfo  # noqa: E402


def vals(raw):
    return [int(v) for v in tfo.to_values(np.asarray(raw, dtype=np.uint64).reshape(-1))]


def main():
    out = {"generator": "tests/golden/make_poly_goldens.py (oracle/tf_oracle.c); inputs are tfo.fill_random(count, seed), outputs canonical values",
           "cases": []}
    for width in (1, 3):
        n = 6
        roots = tfo.fill_random(n * width, 9001 + width)
        out["cases"].append({"op": "zerofier", "width": width, "n": n, "seed": 9001 + width, "out": vals(tfo.zerofier(roots, width))})
        d, v = tfo.fill_random(n * width, 9011 + width), tfo.fill_random(n * width, 9021 + width)
        out["cases"].append({"op": "interpolate", "width": width, "n": n, "domain_seed": 9011 + width, "values_seed": 9021 + width,
                             "out": vals(tfo.lagrange_interpolate(d, v, width))})
        cw = tfo.fill_random(8 * width, 9031 + width)
        x = tfo.fill_random(3, 9041)
        out["cases"].append({"op": "barycentric_evaluate", "width": width, "n": 8, "codeword_seed": 9031 + width, "indeterminate_seed": 9041,
                             "out": vals(tfo.barycentric_evaluate(cw, x, width))})
    q, b = tfo.fill_random(5, 9051), tfo.fill_random(4, 9052)
    a = tfo.poly_mul(q, b)
    out["cases"].append({"op": "clean_divide", "quotient_seed": 9051, "nq": 5, "divisor_seed": 9052, "nb": 4, "dividend": vals(a),
                         "out": vals(tfo.clean_divide(a, b, 0))})
    c, off = tfo.fill_random(3 * 5, 9061), tfo.fill_random(3, 9062)
    out["cases"].append({"op": "coset_evaluate_xfe_offset", "n_coeffs": 5, "order": 8, "coeffs_seed": 9061, "offset_seed": 9062,
                         "out": vals(tfo.coset_evaluate_xfe_offset(c, off, 8))})
    s = tfo.fill_random(16, 9071)
    trace, _ = tfo.tip5_trace(s)
    out["cases"].append({"op": "tip5_trace", "state_seed": 9071, "out": vals(trace)})
    json.dump(out, open(os.path.join(HERE, "poly_goldens.json"), "w"), indent=1)
    print("wrote poly_goldens.json")


if __name__ == "__main__":
    main()
