#!/usr/bin/env python3
"""Extract the *data tables* of the reference (constants, not code) into a JSON fixture.

Runs only in the build container (needs /root/reference); the output
tests/golden/reference_tables.json is committed and is what the tests read.
Tables: PRIMITIVE_ROOTS (math/b_field_element.rs:43-78), LOOKUP_TABLE (tip5/mod.rs:50-64),
ROUND_CONSTANTS (tip5/mod.rs:68-149), MDS_MATRIX_FIRST_COLUMN (tip5/mod.rs:154-157).
"""
import json
import os
import re

REF = "/root/reference/twenty-first/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def between(text, start, end):
    i = text.index(start)
    j = text.index(end, i)
    return text[i:j]


def main():
    bfe = open(os.path.join(REF, "math/b_field_element.rs")).read()
    tip5 = open(os.path.join(REF, "tip5/mod.rs")).read()

    roots_src = between(bfe, "const PRIMITIVE_ROOTS", "};")
    roots = {int(k): int(v) for k, v in re.findall(r"(\d+)u64 => (\d+)", roots_src)}

    lut_src = between(tip5, "pub const LOOKUP_TABLE", "];")
    lut = [int(x) for x in re.findall(r"\b(\d+)\b", lut_src.split("=", 1)[1])]
    assert len(lut) == 256

    rc_src = between(tip5, "pub const ROUND_CONSTANTS", "];")
    rcs = [int(x) for x in re.findall(r"BFieldElement::new\((\d+)\)", rc_src)]
    assert len(rcs) == 80

    mds_src = between(tip5, "pub const MDS_MATRIX_FIRST_COLUMN", "];")
    mds = [int(x) for x in re.findall(r"\b(\d+)\b", mds_src.split("=", 1)[1])]
    assert len(mds) == 16

    out = {
        "source": "Neptune-Crypto/twenty-first v2.0.2 (constant tables only)",
        "primitive_roots": {str(k): str(v) for k, v in sorted(roots.items())},
        "lookup_table": lut,
        "round_constants": [str(x) for x in rcs],
        "mds_matrix_first_column": mds,
    }
    with open(os.path.join(HERE, "reference_tables.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote reference_tables.json")


if __name__ == "__main__":
    main()
