"""Generates tests/golden/merkle_script_roots.json by RUNNING the reference's own
Python helper scripts/merkle_tree.py (hashlib SHA-256 tree, 8 leaves, each leaf
hashed as str(value), pairs hashed upwards) on several leaf sets.

The reference script is imported from /root/reference at generation time only;
the committed JSON holds inputs + expected outputs (data, no reference source).
In src/merkle.rs terms the script is MerkleTree::new with leafs_per_node=1,
inner_children=2 and Python's str(int) as Display (zero prints "0").

    python tests/golden/gen_merkle_script_roots.py
"""
import contextlib
import importlib.util
import io
import json
import os
import random

REF = "/root/reference/scripts/merkle_tree.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "merkle_script_roots.json")

spec = importlib.util.spec_from_file_location("ref_merkle_tree", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

GL_P = 2**64 - 2**32 + 1
BB_P = 2013265921
rng = random.Random(0x5EED)
cases = [
    ("script_default_0_to_7", list(range(8))),
    ("goldilocks_random", [rng.randrange(1, GL_P) for _ in range(8)]),
    ("goldilocks_edge", [GL_P - 1, 1, 10**19, 10**19 - 1, 2**32, 2**32 - 1, 9, 10]),
    ("babybear_random", [rng.randrange(1, BB_P) for _ in range(8)]),
]
vectors = []
for name, leafs in cases:
    with contextlib.redirect_stdout(io.StringIO()):
        root = mod.calculate_tree_root(leafs)
        level0 = [mod.hash_leaf(str(v)).hex() for v in leafs]
    vectors.append(dict(name=name, leafs=[str(v) for v in leafs], leaf_digests=level0, root=root.hex()))

json.dump(dict(source="alv-around/mini-stark scripts/merkle_tree.py (run, not copied)",
               tree=dict(leafs_per_node=1, inner_children=2, zero_as_empty=0), vectors=vectors),
          open(OUT, "w"), indent=1)
print("wrote", OUT)
