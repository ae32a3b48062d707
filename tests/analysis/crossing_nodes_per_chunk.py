"""How many nodes of a 256-leaf chunk cross its boundary -- the length of the list k_chunk leaves for k_cross (csrc/lbvh.hip:
CROSS_CAP slots per chunk; a chunk with more crossing nodes than CROSS_CAP - 1 is marked dense and k_cross goes through all its
256 nodes in CROSS_CAP threads).  CPU analysis on the oracle's tree (test infrastructure): BASELINE config 2 and config 3.

    python tests/analysis/crossing_nodes_per_chunk.py [n]

A node belongs to the chunk of its own index (Karras: one end of its range); it crosses if the other end lies in another chunk.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench      # noqa: E402
import oracle     # noqa: E402

C = 256
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
oracle.build()
for name, (coords, radii) in (("config 2 (uniform)", bench.uniform_scene(n)), ("config 3 (clustered)", bench.clustered_scene(n, 0.0152))):
    ref = oracle.collide(coords, radii, padded=-(-n // 512) * 512, capacity=0)
    nodes = ref["nodes"][:n - 1]                      # internal nodes; node i covers [min(i, other), max(i, other)]
    idx = np.arange(n - 1, dtype=np.int64)
    right = nodes["right_edge"].astype(np.int64)      # the range's last leaf (collision.cl:104-120)
    # a node's own index is one end of its range: the first leaf if right_edge != i, else the last
    # (children: data[0] = left child whose range starts at the node's first leaf)
    first = np.where(right == idx, -1, idx)
    # for nodes that END at their index, the first leaf = first leaf of the left child chain: recover from the left child's range
    left = nodes["data"][:, 0].astype(np.int64)
    # iterate: first[i] for backward nodes = first of left child (internal: < n - 1) or the leaf position (left - (n - 1))
    todo = np.flatnonzero(first < 0)
    while len(todo):
        l = left[todo]
        leaf = l >= n - 1
        val = np.where(leaf, l - (n - 1), np.where(first[np.minimum(l, n - 2)] >= 0, first[np.minimum(l, n - 2)], -1))
        first[todo] = val
        todo = todo[val < 0]
    assert (first <= idx).all() and (right >= idx).all()
    crossing = (first // C != idx // C) | (right // C != idx // C)
    per_chunk = np.bincount((idx // C)[crossing], minlength=-(-n // C))
    qs = [50, 90, 99, 99.9]
    print("%-22s n = %d: crossing nodes %.2f %% of all; per chunk mean %.1f, percentiles %s = %s, max %d; chunks with > 15: %.2f %%, > 31: %.2f %%, > 63: %.3f %%"
          % (name, n, 100.0 * crossing.mean(), per_chunk.mean(), qs, [int(np.percentile(per_chunk, q)) for q in qs], per_chunk.max(),
             100.0 * (per_chunk > 15).mean(), 100.0 * (per_chunk > 31).mean(), 100.0 * (per_chunk > 63).mean()))
