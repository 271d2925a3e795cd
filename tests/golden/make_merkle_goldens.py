#!/usr/bin/env python3
"""Generate Merkle-root goldens from the KAT-pinned oracle.

The reference pins no literal Merkle root; roots are pinned transitively
(hash_pair == hash_10 KATs + structural tests).  This script records the roots of
MerkleTree::test_tree_of_height(h) (util_types/merkle_tree.rs:980-987: leaf i =
hash_varlen([i])) for h = 0..10 so that later changes to the oracle or the HIP path are caught.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tfo  # noqa: E402


def main():
    roots = {}
    for h in range(0, 11):
        leaves = np.concatenate([tfo.hash_varlen(tfo.to_raw([i])) for i in range(1 << h)])
        nodes = tfo.merkle_build(leaves).reshape(-1, 5)
        roots[str(h)] = tfo.digest_hex(nodes[1])
    out = {"generator": "tests/golden/make_merkle_goldens.py (oracle/tf_oracle.c)", "test_tree_of_height_roots": roots}
    json.dump(out, open(os.path.join(HERE, "merkle_roots.json"), "w"), indent=1)
    print("wrote merkle_roots.json")


if __name__ == "__main__":
    main()
