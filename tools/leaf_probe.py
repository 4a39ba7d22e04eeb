#!/usr/bin/env python3
"""Leaf hashing of wide rows in isolation (ms_merkle_commit, lpn = 128): for rocprofv3 --pmc SQ_INSTS_VALU runs of the two-block (LAZY) and one-block leaf kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mini_stark_amd as ms
P = 2**64 - 2**32 + 1
lpn = int(sys.argv[1]) if len(sys.argv) > 1 else 128
groups = 1 << 16
rs = np.random.RandomState(3)
leafs = (rs.randint(0, 2**62, size=groups * lpn, dtype=np.int64).astype(np.uint64) * np.uint64(4) + rs.randint(0, 4, size=groups * lpn).astype(np.uint64)) % np.uint64(P)
ctx = ms.Context(0)
for _ in range(2):
    rc, nodes, root = ctx.merkle_commit(leafs, 1, lpn, 2)
    assert rc == 0
print("ok", root.hex() if hasattr(root, "hex") else root)
