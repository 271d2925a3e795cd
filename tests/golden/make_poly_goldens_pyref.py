#!/usr/bin/env python3
"""Generate tests/golden/poly_goldens_pyref.json from tests/pyref.py ALONE -- pure-Python integers, schoolbook definitions, no C
oracle anywhere (tests/golden/make_poly_goldens.py records what the oracle computes; this file is its independent counterpart).

Inputs AND outputs are stored explicitly as canonical values, so neither the oracle's nor the device's random generator is part of
the fixture.  Both the oracle (tests/test_oracle_kat.py) and the HIP path (tests/test_gpu_next_rows.py, -m gpu) must reproduce every
case: zerofier (polynomial.rs:1462-1475), interpolate (:1565-1606), clean_divide (:2358-2411), barycentric_evaluate (:2609-2637),
fast_coset_evaluate with a BFieldElement and with an XFieldElement offset (:1374-1399), ntt / intt (ntt.rs:67-125) and
fast_multiply (:900-932) at small sizes.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pyref  # noqa: E402


def main():
    cases = []
    seed = 0x7F21A000
    for width in (1, 3):
        F = pyref.Field(width)
        for n in (1, 2, 5, 9):
            seed += 1
            roots = F.group(pyref.splitmix_values(n * width, seed))
            cases.append({"op": "zerofier", "width": width, "n": n, "roots": F.flat(roots), "out": F.flat(pyref.zerofier(F, roots))})
        for n in (1, 3, 7):
            seed += 1
            dom = F.group(pyref.splitmix_values(n * width, seed))
            val = F.group(pyref.splitmix_values(n * width, seed ^ 0x5555))
            cases.append({"op": "interpolate", "width": width, "n": n, "domain": F.flat(dom), "values": F.flat(val),
                          "out": F.flat(pyref.lagrange_interpolate(F, dom, val))})
        for n in (2, 8, 16):
            seed += 1
            cw = F.group(pyref.splitmix_values(n * width, seed))
            x = tuple(pyref.splitmix_values(3, seed ^ 0x3333))
            cases.append({"op": "barycentric_evaluate", "width": width, "n": n, "codeword": F.flat(cw), "indeterminate": list(x),
                          "out": list(pyref.barycentric_evaluate(F, cw, x))})
        for n_coeffs, order in ((1, 1), (3, 4), (5, 8), (16, 16), (7, 32)):
            seed += 1
            c = F.group(pyref.splitmix_values(n_coeffs * width, seed))
            off = pyref.splitmix_values(1, seed ^ 0x7777)[0]
            cases.append({"op": "coset_evaluate", "width": width, "n_coeffs": n_coeffs, "order": order, "coeffs": F.flat(c), "offset": off,
                          "out": F.flat(pyref.coset_evaluate(F, c, F.lift(off), order))})
        for n in (1, 2, 8, 32):
            seed += 1
            x = F.group(pyref.splitmix_values(n * width, seed))
            limbs = [pyref.dft(x)] if width == 1 else [pyref.dft([e[k] for e in x]) for k in range(3)]
            out = limbs[0] if width == 1 else [limbs[k][i] for i in range(n) for k in range(3)]
            cases.append({"op": "ntt", "width": width, "n": n, "in": F.flat(x), "out": out})
        for na, nb in ((1, 1), (3, 5), (9, 8)):
            seed += 1
            a = F.group(pyref.splitmix_values(na * width, seed))
            b = F.group(pyref.splitmix_values(nb * width, seed ^ 0x1111))
            cases.append({"op": "multiply", "width": width, "na": na, "nb": nb, "a": F.flat(a), "b": F.flat(b), "out": F.flat(pyref.poly_mul(F, a, b))})
    F1, F3 = pyref.Field(1), pyref.Field(3)
    for nq, nb in ((1, 1), (5, 4), (12, 9)):
        seed += 1
        q = pyref.splitmix_values(nq, seed)
        b = pyref.splitmix_values(nb, seed ^ 0x2222)
        a = pyref.poly_mul(F1, q, b)
        quo, rem = pyref.long_divide(a, b)
        assert quo == q and not any(rem)
        cases.append({"op": "clean_divide", "dividend": a, "divisor": b, "out": quo})
    for n_coeffs, order in ((5, 8), (4, 16)):
        seed += 1
        c = F3.group(pyref.splitmix_values(3 * n_coeffs, seed))
        off = tuple(pyref.splitmix_values(3, seed ^ 0x4444))
        cases.append({"op": "coset_evaluate_xfe_offset", "n_coeffs": n_coeffs, "order": order, "coeffs": F3.flat(c), "offset": list(off),
                      "out": F3.flat(pyref.coset_evaluate(F3, c, off, order))})
    out = {"generator": "tests/golden/make_poly_goldens_pyref.py (tests/pyref.py: pure-Python schoolbook definitions, independent of oracle/tf_oracle.c); "
                        "every number is a canonical value (BFieldElement::value)",
           "cases": cases}
    json.dump(out, open(os.path.join(HERE, "poly_goldens_pyref.json"), "w"), indent=None, separators=(",", ":"))
    print(f"wrote poly_goldens_pyref.json: {len(cases)} cases")


if __name__ == "__main__":
    main()
